#include "engine.h"

#include <cstdio>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <sstream>
#include <stdexcept>

#include "dev_switch.h"
#include "fp8.h"
#include "graphics.h"
#include "log.h"

namespace ju {

namespace {
thread_local int g_DryLaunches = 0;
}
bool launchesAreDry() { return g_DryLaunches > 0; }
DryLaunchScope::DryLaunchScope() { ++g_DryLaunches; }
DryLaunchScope::~DryLaunchScope() { --g_DryLaunches; }

Engine::Tensor &Engine::addTensor(
    const std::string &name, std::size_t count, bool f32, bool state) {
	Tensor t;
	t.buf = DeviceBuffer(count * (f32 ? 4 : 2));
	t.count = count;
	t.isF32 = f32;
	t.isState = state;
	auto res = m_Tensors.emplace(name, std::move(t));
	if (!res.second) throw std::logic_error("duplicate tensor " + name);
	return res.first->second;
}

Engine::Tensor &Engine::addTowerTensor(const std::string &name, int H, int W, int C) {
	Tensor &t = addTensor(name, towerPixels(H, W) * C);
	t.towerH = H;
	t.towerW = W;
	t.towerC = C;
	return t;
}

Engine::Operand Engine::operand(const std::string &name) {
	Tensor &t = m_Tensors.at(name);
	Operand o;
	o.ptr = t.buf.get();
	if (t.towerC) {
		o.pitch = towerPitch(t.towerW);
		o.ptr = t.buf.as<unsigned char>() + towerOrigin(t.towerW) * t.towerC * 2;
	}
	return o;
}

namespace {
int padTo16(int c) { return (c + 15) / 16 * 16; }

// Runtimes of this process per device, and the completion event of the frame submitted
// last by one that uses the resident tower (Engine::chainBegin / chainEnd).
struct DeviceChain {
	// Held by an engine from its wait on `last` until its own completion event is recorded,
	// so that wait, launches and record are ONE step per device: two threads driving two
	// resident runtimes cannot both pass the wait against the same previous frame.
	std::mutex mutex;
	int engines = 0;
	const void *lastOwner = nullptr;
	hipEvent_t last = nullptr;
};
std::mutex g_ChainMapMutex;
std::map<int, DeviceChain> g_Chains;  // (map nodes never move: references stay valid)
DeviceChain &chainOf(int device) {
	std::lock_guard<std::mutex> lock(g_ChainMapMutex);
	return g_Chains[device];
}
}  // namespace

std::unique_lock<std::mutex> Engine::chainBegin() {
	DeviceChain &c = chainOf(m_Device);
	std::unique_lock<std::mutex> lock(c.mutex);
	if (c.engines > 1 && m_Resident && c.last != nullptr && c.lastOwner != this) {
		JU_HIP(hipStreamWaitEvent(m_Stream, c.last, 0));
	}
	return lock;
}

void Engine::chainEnd(std::unique_lock<std::mutex> &lock) {
	DeviceChain &c = chainOf(m_Device);
	if (c.engines > 1 && m_Resident) {
		m_FrameDone.record(m_Stream);
		c.last = m_FrameDone.get();
		c.lastOwner = this;
	}
	lock.unlock();
}

// Which units of the flow auto-encoder (models.py:334-481) run as one launch each.
void Engine::planFlowUnits() {
	m_FlowUnits.clear();
	const ModelConfig &c = m_Config;
	if (c.flowArch != 0) return;
	const int nb = static_cast<int>(c.flowFilters.size()) / 2;
	const bool hasHeadConv = c.flowFilters.size() % 2 != 0;
	m_FlowUnits.resize(2 * nb + 1);
	if (!m_FlowFused) return;
	int cin = 3 * c.numFlowInputs;
	int h = c.paddedHeight(), w = c.paddedWidth();
	bool prevUps = false;  // the previous unit ends with a bilinear x2
	for (int i = 0; i < 2 * nb; ++i) {
		const int f = c.flowFilters[i];
		const bool pool = i < nb;
		if (prevUps) {
			h *= 2;
			w *= 2;
		}
		FlowUnit &u = m_FlowUnits[i];
		const bool even = h % 2 == 0 && w % 2 == 0;
		if ((!pool || (m_FusedPool && even)) && (!prevUps || even)) {
			if (prevUps && flowBlockSupported(padTo16(cin), f, true, pool, false, h, w)) {
				u.fused = u.upsIn = true;
			} else if (flowBlockSupported(padTo16(cin), f, false, pool, false, h, w)) {
				u.fused = true;
			}
		}
		if (pool) {
			h /= 2;
			w /= 2;
		}
		prevUps = !pool;
		cin = f;
	}
	if (hasHeadConv) {  // flow/conv_1 (BN, act) + flow/conv_2 (bias, 32 channels, f16 flow head)
		if (prevUps) {
			h *= 2;
			w *= 2;
		}
		const int f = c.flowFilters.back();
		FlowUnit &u = m_FlowUnits[2 * nb];
		const bool even = h % 2 == 0 && w % 2 == 0;
		if (f == 32) {
			if (prevUps && even && flowBlockSupported(padTo16(cin), f, true, false, true)) {
				u.fused = u.upsIn = true;
			} else if (flowBlockSupported(padTo16(cin), f, false, false, true)) {
				u.fused = true;
			}
		}
	}
}

bool Engine::flowConvIsFused(const std::string &name) const {
	if (m_Config.flowArch != 0 || m_FlowUnits.empty()) return false;
	const int nb = static_cast<int>(m_Config.flowFilters.size()) / 2;
	if (name == "flow/conv_1" || name == "flow/conv_2") return m_FlowUnits[2 * nb].fused;
	if (name.rfind("flow/block_", 0) == 0) {
		const int k = std::atoi(name.c_str() + 11);
		return k >= 1 && k <= 2 * nb && m_FlowUnits[k - 1].fused;
	}
	return false;
}

Engine::ConvWeights &Engine::addConv(const std::string &name, const FoldedConv &f,
    const std::vector<int> &cinMap, int H, int W) {
	ConvWeights cw;
	// generator conv_1 (64 padded input channels) runs as layer 0 of the resident tower
	const bool towerLayer = name.rfind("generator/block_", 0) == 0 || name == "generator/conv_1";
	// flow-resnet with 64 filters: its blocks are packed like the generator's tower
	const bool flowTower = m_Config.flowArch == 1 && m_Config.flowResFilters == 64;
	const bool flowBlock = flowTower && name.rfind("flow/block_", 0) == 0;
	if ((towerLayer && f.cout == 64) || flowBlock || name == "generator/conv_trans_1") {
		cw.nb = 2;  // the tower kernels and the fused tail read the 64-cout-block layout
		cw.rw = 2;
	} else if (flowConvIsFused(name)) {
		cw.nb = 1;  // flow_block_kernel: one 32-cout block of fragments per wave
		cw.rw = 1;
	} else if (m_FlowFused && name.rfind("flow/", 0) == 0 && f.taps == 9 && (cinMap.size() == 128 || cinMap.size() == 256) &&
	           f.cout % 32 == 0 && H * W <= 32768) {
		cw.nb = 1;  // conv_splitk_kernel (the coarse levels of the flow net): 32-cout blocks of fragments
		cw.rw = 1;
		cw.splitK = true;
	} else {
		convTiling(H, W, f.cout, &cw.nb, &cw.rw);
	}
	const auto packed = packConvWeights(f, cinMap, cw.nb, m_DType);
	cw.w = DeviceBuffer(packed.size() * 2);
	cw.w.upload(packed.data(), packed.size() * 2);
	cw.bias = DeviceBuffer(f.bias.size() * 4);
	cw.bias.upload(f.bias.data(), f.bias.size() * 4);
	if (f.cin == 64 && f.cout == 64 && f.taps == 9 &&
	    (name.rfind("generator/block_", 0) == 0 || flowBlock)) {
		const auto blk = packConvWeights(f, cinMap, 1, m_DType);
		cw.wBlock = DeviceBuffer(blk.size() * 2);
		cw.wBlock.upload(blk.data(), blk.size() * 2);
	}
	cw.cinP = static_cast<int>(cinMap.size());
	cw.cout = f.cout;
	cw.taps = f.taps;
	cw.cinReal = f.cin;
	if (m_Fp8Tower && name.rfind("generator/block_", 0) == 0 && f.cout == 64 && f.cin == 64) {
		const Fp8ConvWeights q = packFp8TowerWeights(f);
		m_Fp8Folded.emplace(name, f);
		Fp8Conv dev;
		dev.w = DeviceBuffer(q.w.size());
		dev.w.upload(q.w.data(), q.w.size());
		dev.scaleA = DeviceBuffer(q.scaleA.size() * 4);
		dev.scaleA.upload(q.scaleA.data(), q.scaleA.size() * 4);
		m_Fp8Convs.emplace(name, std::move(dev));
	}
	// the resident tower's own copy of its layers' weights, in the fragment order of ITS MFMA shape (the per-block and
	// per-convolution kernels keep reading cw.w / cw.wBlock)
	auto residentPack = [&](const std::vector<int> &map64) {
		return residentTowerM16() ? packTowerWeightsM16(f, map64, m_DType) : packConvWeights(f, map64, 2, m_DType);
	};
	if (towerLayer && f.cout == 64 && cinMap.size() == 64) {  // tower layers, in execution order
		const auto w = residentPack(cinMap);
		m_TowerHostW.insert(m_TowerHostW.end(), w.begin(), w.end());
		m_TowerHostB.insert(m_TowerHostB.end(), f.bias.begin(), f.bias.end());
	} else if (towerLayer) {  // (other generator widths never run the resident kernel; kept for the layer count)
		m_TowerHostW.insert(m_TowerHostW.end(), packed.begin(), packed.end());
		m_TowerHostB.insert(m_TowerHostB.end(), f.bias.begin(), f.bias.end());
	}
	if (flowBlock) {
		const auto w = residentPack(cinMap);
		m_FlowTowerHostW.insert(m_FlowTowerHostW.end(), w.begin(), w.end());
		m_FlowTowerHostB.insert(m_FlowTowerHostB.end(), f.bias.begin(), f.bias.end());
	} else if (flowTower && name == "flow/conv_1") {
		// layer 0 of the flow tower reads 64-channel records: same kernel, input
		// channels 12.. are zero (the per-layer path keeps its own 16-channel packing)
		std::vector<int> map64(64, -1);
		for (int k = 0; k < f.cin && k < 64; ++k) map64[k] = k;
		const auto head = residentPack(map64);
		m_FlowTowerHostW.insert(m_FlowTowerHostW.end(), head.begin(), head.end());
		m_FlowTowerHostB.insert(m_FlowTowerHostB.end(), f.bias.begin(), f.bias.end());
	}
	auto res = m_Convs.emplace(name, std::move(cw));
	if (!res.second) throw std::logic_error("duplicate conv " + name);
	return res.first->second;
}

void Engine::buildWeights(const ModelFile &model) {
	// fold + check on the host (model.cpp), then tile, pack and upload layer by layer
	for (const ConvSpec &s : foldModel(model)) addConv(s.name, s.conv, s.cinMap, s.H, s.W);
	{
		const TensorView &k2 = model.tensor("generator/conv_trans_2/kernel", {2, 2, 3, 32});
		const TensorView &b2 = model.tensor("generator/conv_trans_2/bias", {3});
		m_TailW2 = DeviceBuffer(k2.count * 4);
		m_TailW2.upload(k2.data, k2.count * 4);
		m_TailB2 = DeviceBuffer(48);  // convT2 bias (3 f32) + from byte 16: the frame's channel sums (3 x 64 bits)
		m_TailB2.upload(b2.data, 12);
		const auto frag = packTailWeights(k2.data, m_DType);
		m_TailW2Frag = DeviceBuffer(frag.size() * 2);
		m_TailW2Frag.upload(frag.data(), frag.size() * 2);
	}
}

void Engine::addConvStep(std::vector<Step> *prog, const std::string &tag,
    const std::string &wname, Operand in, Operand res, Operand out, int H, int W,
    bool relu, bool outHead, bool tower, bool pool, bool upsample, ItemStride item) {
	auto it = m_Convs.find(wname);
	if (it == m_Convs.end()) throw std::logic_error("missing conv weights " + wname);
	const ConvWeights &cw = it->second;
	ConvParams p{};
	p.in = in.ptr;
	p.wgt = cw.w.get();
	p.bias = cw.bias.as<float>();
	p.res = res.ptr;
	p.out = out.ptr;
	p.inPitch = in.pitch;
	p.resPitch = res.pitch;
	p.outPitch = out.pitch;
	p.H = H;
	p.W = W;
	p.cin = cw.cinP;
	p.cout = cw.cout;
	p.taps = cw.taps;
	{  // the layer's activation (models.py:24-27): the flow net and the generator each have their own
		const bool flowLayer = wname.rfind("flow/", 0) == 0;
		const int act = flowLayer ? m_Config.flowActivation : m_Config.genActivation;
		p.relu = relu ? (act == 1 ? 2 : 1) : 0;
		p.slope = flowLayer ? m_Config.flowNegativeSlope : m_Config.genNegativeSlope;
	}
	p.outHead = outHead ? 1 : 0;
	p.nb = cw.nb;
	p.rw = cw.rw;
	if (pool) {  // the pooled epilogue pairs the two rows of a wave
		p.pool = 1;
		p.rw = 2;
	}
	p.upsample = upsample ? 1 : 0;
	if (upsample) {
		// + the low-resolution patch: with rw = 2 the workgroup needs 96 KB of LDS and
		// only one fits a CU (measured 21 us against 16 us); 4-row tiles (74 KB) keep two
		p.rw = 1;
	}
	const DType dt = m_DType;
	Step s;
	s.tag = tag;
	s.flops = 2.0 * H * W * cw.taps * cw.cinReal * cw.cout * std::max(item.items, 1);
	if (item.items > 1) {  // a look-ahead launch: the layer of `items` frames (split-K convolutions only)
		p.items = item.items;
		p.inItemBytes = item.in;
		p.outItemBytes = item.out;
		// (split-K and generic convolutions have an item dimension; the tower kernel and residual inputs do not)
		if (tower || p.res != nullptr) throw std::logic_error("look-ahead: no batched form of " + wname);
	}
	if (tower) {
		s.run = [dt, p](hipStream_t st) { launchConvTower(dt, p, st); };
	} else if (cw.splitK && convSplitKSupported(p)) {
		if (!m_Zeros.get()) m_Zeros = DeviceBuffer(256);  // (DeviceBuffer memory starts zeroed)
		const void *zeros = m_Zeros.get();
		s.run = [dt, p, zeros](hipStream_t st) { launchConvSplitK(dt, p, zeros, st); };
	} else {
		s.run = [dt, p](hipStream_t st) { launchConv(dt, p, st); };
	}
	prog->push_back(std::move(s));
}

// The flow auto-encoder (models.py:334-481) as launches: `items` = 1 for the per-frame program of binding set `set`,
// > 1 for a look-ahead pass -- the same launches over `items` consecutive frames (grid.z), on the batch tensors, the
// first block reading the frames of m_BatchIO (submitBatch).  A look-ahead pass exists only where every launch of
// the plan has an item dimension (the one-launch blocks and the split-K convolutions): std::logic_error otherwise.
void Engine::addFlowAutoencoder(std::vector<Step> *progOut, int set, int items) {
	std::vector<Step> &prog = *progOut;
	const ModelConfig &c = m_Config;
	const DType dt = m_DType;
	const bool batch = items > 1;
	const int H = c.frameHeight, W = c.frameWidth;
	const int PH = c.paddedHeight(), PW = c.paddedWidth();
	const int padTop = (PH - H) / 2, padLeft = (PW - W) / 2;  // models.py:783-787
	const FrameIO *io = &m_IO;
	const FrameIO *batchIO = m_BatchIO;
	const void *packedIn = m_Packed[set].get();
	void *packedOut = m_Packed[set ^ 1].get();
	const int nIn = c.numFlowInputs;
	const unsigned *sums = c.normalizeBrightness ? m_TailB2.as<unsigned>() + 4 : nullptr;
	auto T = [&](const std::string &n) -> void * { return (batch ? m_BatchTensors : m_Tensors).at(n).buf.get(); };
	auto Op = [&](const std::string &n) { return batch ? Operand{T(n), 0} : operand(n); };
	// bytes of ONE frame's tensor: the item stride of a look-ahead launch
	auto itemBytes = [&](const std::string &n) -> long {
		const Tensor &t = m_Tensors.at(n);
		return static_cast<long>(t.count) * (t.isF32 ? 4 : 2);
	};
	auto noBatch = [&](const char *what) {
		if (batch) throw std::logic_error(std::string("look-ahead: no batched form of ") + what);
	};
	const Operand none{};
	const bool packInBlock = flowPacksInBlock();
	if (batch && (!packInBlock || sums != nullptr)) noBatch("the input packing");
	Operand cur{packedOut, 0};
	long curItem = 0;  // item stride of `cur`
	int h = PH, w = PW;
	bool flowHeadDone = false;  // the fused head block wrote the flow tensor
	const int nb = static_cast<int>(c.flowFilters.size()) / 2;
	// the decoder's bilinear x2 is folded into the staging of the conv that follows
	// when that conv stages its input once (cout = 32: every cout-group workgroup
	// would repeat the expansion; measured: a loss already with two groups) and
	// reads 64-channel chunks
	auto fusesUpsample = [&](const std::string &next, int cin) {
		auto it = m_Convs.find(next);
		return m_FusedUpsample && it != m_Convs.end() && cin % 64 == 0 && it->second.nb == 1 &&
		       it->second.cout <= 32;
	};
	// one launch for a whole block (both convs, pool, preceding upsample) where planned
	const int flowAct = c.flowActivation == 1 ? 2 : 1;
	auto addBlockStep = [&](const std::string &convA, const std::string &convB, const void *in, long inItem,
	                        const std::string &outName, int bh, int bw, bool ups, bool pool, bool outHead, int act2) {
		const ConvWeights &wa = m_Convs.at(convA), &wb = m_Convs.at(convB);
		FlowBlockLaunch fb{};
		fb.in = in;
		fb.w1 = wa.w.get();
		fb.b1 = wa.bias.as<float>();
		fb.w2 = wb.w.get();
		fb.b2 = wb.bias.as<float>();
		fb.out = T(outName);
		fb.H = bh;
		fb.W = bw;
		fb.cin = wa.cinP;
		fb.cmid = wa.cout;
		fb.upsample = ups;
		fb.pool = pool;
		fb.outHead = outHead;
		fb.act1 = flowAct;
		fb.act2 = act2;
		fb.slope = c.flowNegativeSlope;
		fb.items = items;
		fb.inItemBytes = inItem;
		fb.outItemBytes = itemBytes(outName);
		const double flops =
		    items * 2.0 * bh * bw * 9.0 * (double(wa.cinReal) * wa.cout + double(wb.cinReal) * wb.cout);
		if (packInBlock && convA == "flow/block_1/conv_1") {
			fb.packPrev = packedIn;
			fb.packOut = packedOut;
			fb.frameH = H;
			fb.frameW = W;
			fb.padTop = padTop;
			fb.padLeft = padLeft;
			fb.numInputs = nIn;
			fb.sums = sums;
			prog.push_back({"flow", flops, [dt, fb, io, batchIO, items](hipStream_t s) {
				                FlowBlockLaunch f = fb;  // the caller's frames are known at launch time only
				                f.packFrame = io->in;
				                f.packFrameStride = io->inStride;
				                for (int i = 0; i < items && items > 1; ++i) {
					                f.packFrames[i] = batchIO[i].in;
					                f.packFrameStrides[i] = batchIO[i].inStride;
				                }
				                launchFlowBlock(dt, f, s);
			                }});
			return;
		}
		prog.push_back({"flow", flops, [dt, fb](hipStream_t s) { launchFlowBlock(dt, fb, s); }});
	};
	auto unitUpsIn = [&](int k) { return k < static_cast<int>(m_FlowUnits.size()) && m_FlowUnits[k].fused && m_FlowUnits[k].upsIn; };
	bool upsampleNext = false;  // `cur` is half resolution: the next conv upsamples it
	for (int i = 0; i < 2 * nb; ++i) {
		const std::string n = "flow/block_" + std::to_string(i + 1);
		const int f = c.flowFilters[i];
		if (m_FlowUnits[i].fused) {
			const bool pool = i < nb;
			if (upsampleNext != m_FlowUnits[i].upsIn) throw std::logic_error("flow plan out of step");
			addBlockStep(n + "/conv_1", n + "/conv_2", cur.ptr, curItem, pool ? n + "/resample" : n + "/a_2", h, w,
			    upsampleNext, pool, false, flowAct);
			upsampleNext = false;
			if (pool) {
				h /= 2;
				w /= 2;
				cur = Operand{T(n + "/resample"), 0};
				curItem = itemBytes(n + "/resample");
			} else {
				const void *src = T(n + "/a_2");
				void *dst = T(n + "/resample");
				if (unitUpsIn(i + 1)) {  // the next unit expands it while staging
					upsampleNext = true;
					cur = Op(n + "/a_2");
					curItem = itemBytes(n + "/a_2");
				} else {
					const std::string next = i + 1 < 2 * nb ? "flow/block_" + std::to_string(i + 2) + "/conv_1"
					                         : (c.flowFilters.size() % 2 ? "flow/conv_1" : "");
					if (fusesUpsample(next, f) && !(i + 1 < static_cast<int>(m_FlowUnits.size()) && m_FlowUnits[i + 1].fused)) {
						upsampleNext = true;
						cur = Op(n + "/a_2");
						curItem = itemBytes(n + "/a_2");
					} else {
						prog.push_back({"flow", 0.0,
						    [=](hipStream_t s) { launchUpsample2(dt, src, dst, h, w, f, s, items); }});
						cur = Operand{dst, 0};
						curItem = itemBytes(n + "/resample");
					}
				}
				h *= 2;
				w *= 2;
			}
			continue;
		}
		const ItemStride st1{items, curItem, itemBytes(n + "/a_1")};
		addConvStep(&prog, "flow", n + "/conv_1", cur, none, Op(n + "/a_1"), h, w, true, false,
		    false, false, upsampleNext, st1);
		upsampleNext = false;
		const bool fusePool = i < nb && m_FusedPool;
		const std::string out2 = fusePool ? n + "/resample" : n + "/a_2";
		const ItemStride st2{items, itemBytes(n + "/a_1"), itemBytes(out2)};
		addConvStep(&prog, "flow", n + "/conv_2", Op(n + "/a_1"), none, Op(out2), h, w, true, false, false, fusePool,
		    false, st2);
		const void *src = T(n + "/a_2");  // dense tensors
		void *dst = T(n + "/resample");
		if (i < nb) {
			if (!fusePool) {
				noBatch("the pooling launch");
				prog.push_back({"flow", 0.0,
				    [=](hipStream_t s) { launchMaxPool2(dt, src, dst, h, w, f, s); }});
			}
			h /= 2;
			w /= 2;
			cur = Operand{dst, 0};
			curItem = itemBytes(n + "/resample");
		} else {
			const std::string next = i + 1 < 2 * nb ? "flow/block_" + std::to_string(i + 2) + "/conv_1"
			                         : (c.flowFilters.size() % 2 ? "flow/conv_1" : "");  // not the head
			const bool nextFused = i + 1 < static_cast<int>(m_FlowUnits.size()) && m_FlowUnits[i + 1].fused;
			if (unitUpsIn(i + 1) || (!nextFused && fusesUpsample(next, f))) {
				upsampleNext = true;
				cur = Op(n + "/a_2");
				curItem = itemBytes(n + "/a_2");
			} else {
				prog.push_back({"flow", 0.0,
				    [=](hipStream_t s) { launchUpsample2(dt, src, dst, h, w, f, s, items); }});
				cur = Operand{dst, 0};
				curItem = itemBytes(n + "/resample");
			}
			h *= 2;
			w *= 2;
		}
	}
	if (c.flowFilters.size() % 2) {
		if (m_FlowUnits[2 * nb].fused) {  // flow/conv_1 + flow/conv_2 -> the f16 flow head, one launch
			if (upsampleNext != m_FlowUnits[2 * nb].upsIn) throw std::logic_error("flow plan out of step");
			addBlockStep("flow/conv_1", "flow/conv_2", cur.ptr, curItem, "flow", h, w, upsampleNext, false, true, 0);
			upsampleNext = false;
			flowHeadDone = true;
		} else {
			addConvStep(&prog, "flow", "flow/conv_1", cur, none, Op("flow/a_1"), h, w, true, false,
			    false, false, upsampleNext, ItemStride{items, curItem, itemBytes("flow/a_1")});
			upsampleNext = false;
			cur = Op("flow/a_1");
			curItem = itemBytes("flow/a_1");
		}
	}
	if (upsampleNext) throw std::logic_error("flow head cannot take a half-resolution input");
	if (!flowHeadDone) {
		addConvStep(&prog, "flow", "flow/conv_2", cur, none, Op("flow"), h, w, false, true, false, false, false,
		    ItemStride{items, curItem, itemBytes("flow")});
	}
}

// the flow net's first block builds the packed tensor itself when it runs as one launch
bool Engine::flowPacksInBlock() const {
	const ModelConfig &c = m_Config;
	return m_PackInBlock && c.flowArch == 0 && !m_FlowUnits.empty() && m_FlowUnits[0].fused &&
	       3 * c.numFlowInputs <= 16 && !c.flowFilters.empty() && c.flowFilters[0] == 32;
}

void Engine::buildProgram(int set) {
	const ModelConfig &c = m_Config;
	const DType dt = m_DType;
	std::vector<Step> &prog = m_Program[set];
	prog.clear();
	const int H = c.frameHeight, W = c.frameWidth;
	const int PH = c.paddedHeight(), PW = c.paddedWidth();
	const int padTop = (PH - H) / 2, padLeft = (PW - W) / 2;  // models.py:783-787
	const FrameIO *io = &m_IO;  // read at launch time: staging buffers or the caller's
	const void *packedIn = m_Packed[set].get();
	void *packedOut = m_Packed[set ^ 1].get();
	// the recurrent state this program reads / writes: m_State[set] -> m_State[set ^ 1], read at launch (capture)
	// time -- a look-ahead pass binds its frames' programs to its own chain of state buffers (submitBatch)
	m_StateBind[set].in = m_State[set].get();
	m_StateBind[set].out = m_State[set ^ 1].get();
	const StateBind *sb = &m_StateBind[set];
	const int nIn = c.numFlowInputs;
	auto T = [&](const std::string &n) -> void * { return m_Tensors.at(n).buf.get(); };
	auto Op = [&](const std::string &n) { return operand(n); };
	const Operand none{};

	// normalize_brightness: three integer channel sums per frame, consumed by the pack,
	// warp and tail kernels (models.py:772-779, 802-803, 809-810)
	const unsigned *sums = c.normalizeBrightness ? m_TailB2.as<unsigned>() + 4 : nullptr;
	if (sums) {
		unsigned *sumsOut = m_TailB2.as<unsigned>() + 4;
		prog.push_back({"pack", 0.0,
		    [=](hipStream_t s) { launchFrameSums(io->in, io->inStride, H, W, sumsOut, s); }});
	}
	// (the flow net's first block builds the packed tensor itself when it runs as one launch)
	const bool packInBlock = flowPacksInBlock();
	if (!packInBlock) {
		prog.push_back({"pack", 0.0, [=](hipStream_t s) {
			                launchPackFrames(dt, io->in, io->inStride, packedIn, packedOut, H, W, PH, PW,
			                    padTop, padLeft, nIn, sums, s);
		                }});
	}
	// one launch for a 64-filter residual block outside the resident tower
	auto addResBlockStep = [&](const std::string &tag, const std::string &block, Operand in, Operand out,
	                           int bh, int bw, int act, float slope) {
		const ConvWeights &wa = m_Convs.at(block + "/conv_1"), &wb = m_Convs.at(block + "/conv_2");
		FlowBlockLaunch fb{};
		fb.in = in.ptr;
		fb.inPitch = in.pitch;
		fb.w1 = wa.wBlock.get();
		fb.b1 = wa.bias.as<float>();
		fb.w2 = wb.wBlock.get();
		fb.b2 = wb.bias.as<float>();
		fb.out = out.ptr;
		fb.outPitch = out.pitch;
		fb.H = bh;
		fb.W = bw;
		fb.cin = fb.cmid = 64;
		fb.residual = true;
		fb.act1 = fb.act2 = act;
		fb.slope = slope;
		prog.push_back({tag, 2.0 * bh * bw * 9.0 * 64 * 64 * 2,
		    [dt, fb](hipStream_t s) { launchFlowBlock(dt, fb, s); }});
	};
	// ---- flow net ----
	Operand cur{packedOut, 0};
	int h = PH, w = PW;
	bool flowHeadDone = false;  // the fused head block wrote the flow tensor
	if (c.flowArch == 0) {
		addFlowAutoencoder(&prog, set, 1);
		flowHeadDone = true;  // (or its own head launch: addFlowAutoencoder)
	} else if (m_ResidentFlow) {
		// flow-resnet body = conv_1 + residual blocks, 64 filters: ONE launch of the
		// resident tower kernel (own mailbox and publish counters)
		void *in64 = T("flow/in64");
		prog.push_back({"flow", 0.0,
		    [=](hipStream_t s) { launchExpandChannels(dt, packedOut, in64, PH * PW, s); }});
		ResidentTowerParams rp{};
		rp.in = in64;
		rp.inPitch = PW;
		rp.hasHead = 1;
		rp.out = Op("flow/trunk").ptr;
		rp.weights = m_FlowTowerW.get();
		rp.bias = m_FlowTowerB.as<float>();
		rp.mailbox = m_FlowMail.get();
		rp.counters = m_FlowFlags.as<unsigned>();
		rp.error = m_ResErrorDev;
		rp.debug = m_Tensors.at("tower_profile").buf.get();
		rp.H = PH;
		rp.W = PW;
		rp.GX = m_FlowGX;
		rp.GY = m_FlowGY;
		rp.RH = m_FlowRH;
		rp.nLayers = 1 + 2 * c.flowResBlocks;
		rp.leaky = c.flowActivation == 1 ? 1 : 0;
		rp.slope = c.flowNegativeSlope;
		prog.push_back({"flow",
		    2.0 * PH * PW * 9.0 * (3.0 * c.numFlowInputs * 64 + 64.0 * 64 * 2 * c.flowResBlocks),
		    [=](hipStream_t s) { launchResidentTower(dt, rp, s); }});
		cur = Op("flow/trunk");
	} else {
		addConvStep(&prog, "flow", "flow/conv_1", cur, none, Op("flow/x0"), h, w, true, false);
		const char *xs[2] = {"flow/x0", "flow/x1"};
		int a = 0;
		for (int i = 0; i < c.flowResBlocks; ++i) {
			const std::string n = "flow/block_" + std::to_string(i + 1);
			if (m_BlockFused && c.flowResFilters == 64) {
				addResBlockStep("flow", n, Op(xs[a]), Op(xs[a ^ 1]), h, w, c.flowActivation == 1 ? 2 : 1,
				    c.flowNegativeSlope);
				a ^= 1;
				continue;
			}
			addConvStep(&prog, "flow", n + "/conv_1", Op(xs[a]), none, Op("flow/t"), h, w, true, false);
			addConvStep(&prog, "flow", n + "/conv_2", Op("flow/t"), Op(xs[a]), Op(xs[a ^ 1]), h, w,
			    true, false);
			a ^= 1;
		}
		cur = Op(xs[a]);
	}
	if (!flowHeadDone) addConvStep(&prog, "flow", "flow/conv_2", cur, none, Op("flow"), h, w, false, true);
	// ---- warp + space-to-depth + concat ----
	{
		m_FlowCur = T("flow");
		const void *const *flowSlot = &m_FlowCur;  // (a look-ahead pass points it at its frame's field)
		// the generator input lives in the tower layout (zero border = conv_1's padding), so that conv_1 can run
		// on the tower's per-layer kernel wherever it is a launch of its own (8-bit towers, per-block towers)
		const Operand genInOp = Op("gen_in");
		void *genIn = genInOp.ptr;
		const int genPitch = genInOp.pitch;
		void *preWarp = c.temporalStrength > 0.0f ? T("pre_warp") : nullptr;
		prog.push_back({"warp", 0.0, [=](hipStream_t s) {
			                launchWarpPack(dt, sb->in, *flowSlot, io->in, io->inStride, genIn, genPitch, H, W, PW,
			                    padTop, padLeft, sums, preWarp, s);
		                }});
	}
	// ---- generator ----
	// calibration mode (JU_CALIBRATE=1, tools/calibrate.py): one launch per convolution, and
	// after each the largest |output| of the layer folded into tower_profile[layer] -- works
	// for every geometry and activation, unlike the resident kernel's in-kernel maxima
	auto addCalib = [&](int layer, const std::string &tensor) {
		if (!m_Calibrate) return;
		const Tensor &t = m_Tensors.at(tensor);
		const void *src = t.buf.get();
		const std::size_t n = t.count;
		unsigned *dst = m_Tensors.at("tower_profile").buf.as<unsigned>() + layer;
		prog.push_back({"calib", 0.0, [=](hipStream_t s) { launchAbsMax(dt, src, n, dst, s); }});
	};
	if (m_Calibrate) {
		void *profile = m_Tensors.at("tower_profile").buf.get();
		const std::size_t bytes = (1 + 2 * static_cast<std::size_t>(c.genBlocks)) * 4;
		prog.push_back({"calib", 0.0, [=](hipStream_t s) { JU_HIP(hipMemsetAsync(profile, 0, bytes, s)); }});
	}
	if (!m_Resident || m_Fp8Tower) {  // (the 16-bit resident tower runs conv_1 as its layer 0)
		// (64-filter generators: conv_tower_kernel -- 38 instead of 51 us at 640x448; other widths: the generic kernel)
		addConvStep(&prog, "gen_head", "generator/conv_1", Op("gen_in"), none, Op("trunk_a"), H, W,
		    true, false, c.genFilters == 64 && !m_Calibrate);
		addCalib(0, "trunk_a");
	}
	const char *xs[2] = {"trunk_a", "trunk_b"};
	int a = 0;
	bool tailInTower = false;
	if (m_Resident && m_Fp8Tower) {
		// the 8-bit tower in one launch: trunk_a (conv_1's output) -> trunk_b
		ResidentTower8Params rp{};
		rp.in = m_Tensors.at("trunk_a").buf.get();
		rp.out = m_Tensors.at("trunk_b").buf.get();
		rp.weights = m_Fp8TowerW.get();
		rp.scaleA = m_Fp8TowerScaleA.as<int>();
		rp.bias = m_Fp8TowerBias.as<float>();
		rp.scaleB = m_Fp8TowerScaleB.as<int>();
		rp.outMul = m_Fp8TowerMul.as<float>();
		rp.mailbox = m_ResMail.get();
		rp.counters = m_ResFlags.as<unsigned>();
		rp.error = m_ResErrorDev;
		rp.H = H;
		rp.W = W;
		rp.GX = m_ResGX;
		rp.GY = m_ResGY;
		rp.RH = m_ResRH;
		rp.nLayers = 2 * c.genBlocks;
		rp.leaky = c.genActivation == 1 ? 1 : 0;
		rp.slope = c.genNegativeSlope;
		rp.debug = m_Tensors.at("tower_profile").buf.get();
		prog.push_back({"tower", 2.0 * H * W * 9.0 * 64.0 * 64 * 2 * c.genBlocks,
		    [=](hipStream_t s) { launchResidentTower8(dt, rp, s); }});
		a = 1;
	} else if (m_Resident) {
		// one launch for the whole tower
		ResidentTowerParams rp{};
		rp.in = Op("gen_in").ptr;  // tower layout [..][pitch][64] at image pixel (0, 0); conv_1 is layer 0 of the launch
		rp.inPitch = Op("gen_in").pitch;
		rp.hasHead = 1;
		rp.out = Op("trunk_b").ptr;
		rp.weights = m_TowerW.get();
		rp.bias = m_TowerB.as<float>();
		rp.mailbox = m_ResMail.get();
		rp.counters = m_ResFlags.as<unsigned>();
		rp.error = m_ResErrorDev;
		rp.debug = m_Tensors.at("tower_profile").buf.get();
		rp.H = H;
		rp.W = W;
		rp.GX = m_ResGX;
		rp.GY = m_ResGY;
		rp.RH = m_ResRH;
		rp.nLayers = 1 + 2 * c.genBlocks;
		rp.leaky = c.genActivation == 1 ? 1 : 0;
		rp.slope = c.genNegativeSlope;
		tailInTower = m_FusedTail && m_TailInTower && c.genFilters == 64 && c.genActivation == 0;
		if (tailInTower) {  // the tail runs on the tower's LDS-resident last layer
			rp.tailW1 = m_Convs.at("generator/conv_trans_1").w.get();
			rp.tailB1 = m_Convs.at("generator/conv_trans_1").bias.as<float>();
			rp.tailW2 = m_TailW2Frag.get();
			rp.tailB2 = m_TailB2.as<float>();
			rp.state = nullptr;  // (sb->out, at launch time)
			rp.sums = sums;
		}
		TailFusedLaunch tf{};  // (debug variants of the tower kernel have no fused-tail form: separate launch)
		if (tailInTower) {
			const ConvWeights &cw = m_Convs.at("generator/conv_trans_1");
			const Operand xin = Op("trunk_b");
			tf.x = xin.ptr;
			tf.xPitch = xin.pitch;
			tf.w1 = cw.w.get();
			tf.b1 = cw.bias.as<float>();
			tf.w2 = m_TailW2Frag.get();
			tf.b2 = m_TailB2.as<float>();
			tf.state = nullptr;  // (sb->out, at launch time)
			tf.sums = sums;
			tf.H = H;
			tf.W = W;
			tf.slope = -1.0f;
		}
		prog.push_back({"tower",
		    2.0 * H * W * 9.0 * (51.0 * 64 + 64.0 * 64 * 2 * c.genBlocks) +
		        (tailInTower ? 2.0 * H * W * (64.0 * 128 + 4 * 4 * 32 * 3) : 0.0),
		    [=](hipStream_t s) {
			    ResidentTowerParams r = rp;
			    if (r.tailW1 != nullptr && towerVariant() != 0) {
				    // a diagnostic variant of the kernel (plain schedule, calibration, phase profile,
				    // ablations): it writes the trunk, and the same tail code runs as its own launch
				    r.tailW1 = nullptr;
				    launchResidentTower(dt, r, s);
				    TailFusedLaunch t = tf;
				    t.state = sb->out;
				    t.frame = io->in;
				    t.frameStride = io->inStride;
				    t.outU8 = io->out;
				    t.outStride = io->outStride;
				    launchTailFused(dt, t, s);
				    return;
			    }
			    if (r.tailW1 != nullptr) {  // caller's frames are known at launch time only
				    r.state = sb->out;
				    r.frame = io->in;
				    r.frameStride = io->inStride;
				    r.outU8 = io->out;
				    r.outStride = io->outStride;
			    }
			    launchResidentTower(dt, r, s);
		    }});
		a = 1;
	} else if (m_Fp8Tower) {
		// stream (fp16, trunk_a, updated in place) + e4m3 copies x8 / t8 of the conv inputs
		void *streamBuf = m_Tensors.at("trunk_a").buf.get();
		void *x8 = m_Fp8X.get(), *t8 = m_Fp8T.get();
		const int e0 = m_Fp8Exp[0];
		const bool leaky8 = c.genActivation == 1;
		prog.push_back({"tower", 0.0,
		    [=](hipStream_t s) { launchQuantizeTower(dt, streamBuf, x8, H, W, e0, leaky8, s); }});
		for (int i = 0; i < c.genBlocks && m_BlockFused; ++i) {
			// one launch per block: the e4m3 stream copy ping-pongs between the two tensors
			const std::string n = "generator/block_" + std::to_string(i + 1);
			const Fp8Conv &q1 = m_Fp8Convs.at(n + "/conv_1"), &q2 = m_Fp8Convs.at(n + "/conv_2");
			Fp8BlockLaunch fb{};
			fb.in8 = (i & 1) ? t8 : x8;
			fb.out8 = (i & 1) ? x8 : t8;
			fb.stream = streamBuf;
			fb.w1 = q1.w.get();
			fb.w2 = q2.w.get();
			fb.scaleA1 = q1.scaleA.as<int>();
			fb.scaleA2 = q2.scaleA.as<int>();
			fb.b1 = m_Convs.at(n + "/conv_1").bias.as<float>();
			fb.b2 = m_Convs.at(n + "/conv_2").bias.as<float>();
			fb.inExp = m_Fp8Exp[2 * i];
			fb.midExp = m_Fp8Exp[2 * i + 1];
			fb.outExp = (i + 1 < c.genBlocks) ? m_Fp8Exp[2 * i + 2] : 0;  // (the last copy has no reader)
			fb.H = H;
			fb.W = W;
			fb.leaky = leaky8 ? 1 : 0;
			fb.slope = c.genNegativeSlope;
			prog.push_back({"tower", 2.0 * H * W * 9.0 * 64 * 64 * 2,
			    [=](hipStream_t s) { launchResBlockFp8(dt, fb, s); }});
		}
		for (int i = 0; i < c.genBlocks && !m_BlockFused; ++i) {
			const std::string n = "generator/block_" + std::to_string(i + 1);
			for (int j = 0; j < 2; ++j) {
				const std::string name = n + (j ? "/conv_2" : "/conv_1");
				const Fp8Conv &q = m_Fp8Convs.at(name);
				Fp8TowerParams fp{};
				fp.in8 = j ? t8 : x8;
				fp.weights = q.w.get();
				fp.scaleA = q.scaleA.as<int>();
				fp.bias = m_Convs.at(name).bias.as<float>();
				fp.stream = j ? streamBuf : nullptr;
				fp.out8 = j ? x8 : t8;
				fp.inExp = m_Fp8Exp[2 * i + j];
				// the last block's e4m3 copy has no reader; any exponent will do
				fp.outExp = (2 * i + j + 1 < 2 * c.genBlocks) ? m_Fp8Exp[2 * i + j + 1] : 0;
				fp.H = H;
				fp.W = W;
				fp.leaky = leaky8 ? 1 : 0;
				fp.slope = c.genNegativeSlope;
				prog.push_back({"tower", 2.0 * H * W * 9.0 * 64 * 64,
				    [=](hipStream_t s) { launchConvTowerFp8(dt, fp, s); }});
			}
		}
	} else {
		for (int i = 0; i < c.genBlocks; ++i) {
			const std::string n = "generator/block_" + std::to_string(i + 1);
			if (m_BlockFused && c.genFilters == 64) {
				addResBlockStep("tower", n, Op(xs[a]), Op(xs[a ^ 1]), H, W, c.genActivation == 1 ? 2 : 1,
				    c.genNegativeSlope);
				a ^= 1;
				continue;
			}
			addConvStep(&prog, "tower", n + "/conv_1", Op(xs[a]), none, Op("trunk_t"), H, W, true,
			    false, true);
			addCalib(2 * i + 1, "trunk_t");
			addConvStep(&prog, "tower", n + "/conv_2", Op("trunk_t"), Op(xs[a]), Op(xs[a ^ 1]), H, W,
			    true, false, true);
			addCalib(2 * i + 2, xs[a ^ 1]);
			a ^= 1;
		}
	}
	m_TrunkOut = xs[a];
	if (tailInTower) {
		// (nothing: the resident tower wrote the HR state and the frame)
	} else if (m_FusedTail && c.genFilters == 64) {
		const ConvWeights &cw = m_Convs.at("generator/conv_trans_1");
		TailFusedLaunch tf{};
		const Operand xin = Op(xs[a]);
		tf.x = xin.ptr;
		tf.xPitch = xin.pitch;
		tf.w1 = cw.w.get();
		tf.b1 = cw.bias.as<float>();
		tf.w2 = m_TailW2Frag.get();
		tf.b2 = m_TailB2.as<float>();
		tf.state = nullptr;  // (sb->out, at launch time)
		tf.sums = sums;
		tf.H = H;
		tf.W = W;
		tf.slope = c.genActivation == 1 ? c.genNegativeSlope : -1.0f;
		prog.push_back({"tail", 2.0 * H * W * (64.0 * 128 + 4 * 4 * 32 * 3), [=](hipStream_t s) {
			                TailFusedLaunch t = tf;
			                t.state = sb->out;
			                t.frame = io->in;
			                t.frameStride = io->inStride;
			                t.outU8 = io->out;
			                t.outStride = io->outStride;
			                launchTailFused(dt, t, s);
		                }});
	} else {
		addConvStep(&prog, "tail", "generator/conv_trans_1", Op(xs[a]), none, Op("tail_y"), H, W,
		    true, false);
		const void *y = T("tail_y");
		const float *w2 = m_TailW2.as<float>();
		const float *b2 = m_TailB2.as<float>();
		prog.push_back({"tail", 2.0 * (2 * H) * (2 * W) * 4 * 32 * 3, [=](hipStream_t s) {
			                launchTail(dt, y, w2, b2, io->in, io->inStride, sb->out, io->out,
			                    io->outStride, H, W, sums, s);
		                }});
	}
	// ---- optional output filter (frame_moving_avg.py): replaces the clip output for
	// both of its consumers, the u8 frame and the fed-back state ----
	if (c.temporalStrength > 0.0f) {
		const void *preWarp = T("pre_warp");
		unsigned long long *acc = m_TemporalAcc.as<unsigned long long>();
		const TemporalParams tp{c.temporalStrength, c.temporalThreshold, c.temporalGain, c.temporalWindow,
		    c.temporalL2 ? 1 : 0, c.temporalLimit ? 1 : 0, c.temporalLuma ? 1 : 0};
		prog.push_back({"temporal", 0.0, [=](hipStream_t s) {
			                launchTemporalFilter(sb->out, preWarp, io->out, io->outStride, H, W, sums,
			                    acc, tp, s);
		                }});
	}
}

Engine::Engine(int device, const void *blob, std::size_t size, int dtypeOverride)
    : m_Device(device) {
	// Like TensorRTBackend (tensorrt_backend.cc:118), the engine is built on the
	// CURRENT device; the caller (c_api.cpp) selects it for the call.
	int current = -1;
	JU_HIP(hipGetDevice(&current));
	if (current != device) throw std::logic_error("Engine must be created with its device current");
	ModelFile model(blob, size);
	m_Config = model.config();
	const ModelConfig &c = m_Config;
	int dt = dtypeOverride >= 0 ? dtypeOverride : c.computeDtype;
	if (dt == 2) {
		// JU_DTYPE_FP8: the 64->64 block convolutions run on e4m3 operands (fp8.h); the
		// residual stream and every other layer stay fp16
		if (c.genFilters != 64 || c.genBlocks < 1) {
			throw std::invalid_argument("fp8 tower needs a 64-filter generator with at least one residual block");
		}
		m_Fp8Tower = true;
		dt = kF16;
	}
	if (dt != kF16 && dt != kBF16) throw std::invalid_argument("Unsupported compute dtype");
	m_DType = static_cast<DType>(dt);
	const int H = c.frameHeight, W = c.frameWidth;
	const int PH = c.paddedHeight(), PW = c.paddedWidth();

	const char *tailMode = devSwitch(Dev::Tail);
	m_FusedTail = !(tailMode && std::string(tailMode) == "split");
	// The fused tail runs INSIDE the resident tower launch wherever that kernel is used with a
	// ReLU generator (its last layer is in LDS: the trunk is never written or re-read, one launch
	// less).  Bit-identical to the separate launch (same row code); round 2 measured +0.5-1 % and
	// kept it opt-in, under round 3's steady-state timing it is +0.9 % at 480x270 and +1.9 % for
	// psp-fast (tools/submit_overhead.py, three A/B rounds), so it is the default now.
	// JU_TAIL=fused: the separate tail_fused_kernel launch; JU_TAIL=split: the two-kernel tail.
	m_TailInTower = !(tailMode && (std::string(tailMode) == "fused" || std::string(tailMode) == "split"));
	const char *packMode = devSwitch(Dev::Pack);
	m_PackInBlock = !(packMode && std::string(packMode) == "split");  // JU_PACK=split: pack_frames_kernel as its own launch
	const char *poolMode = devSwitch(Dev::Pool);
	m_FusedPool = !(poolMode && std::string(poolMode) == "split");
	const char *upMode = devSwitch(Dev::Upsample);
	m_FusedUpsample = !(upMode && std::string(upMode) == "split");
	const char *flowConv = devSwitch(Dev::FlowConv);
	m_FlowFused = !(flowConv && std::string(flowConv) == "generic") && m_FusedUpsample;
	planFlowUnits();

	buildWeights(model);

	// ---- resident tower: needs 64 filters and one co-resident workgroup per region ----
	{
		const char *mode = devSwitch(Dev::Tower);
		int cus = 0;
		JU_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device));
		bool wanted = !(mode && (std::string(mode) == "layers" || std::string(mode) == "convs"));
		m_BlockFused = !(mode && std::string(mode) == "convs");
		const char *calib = devSwitch(Dev::Calibrate);
		m_Calibrate = calib && calib[0] == '1';
		if (m_Calibrate) {
			if (m_Fp8Tower) throw std::invalid_argument("calibration mode: calibrate the fp16 / bf16 engine, not the 8-bit one");
			wanted = false;
			m_BlockFused = false;
		}
		if (wanted && c.genFilters == 64 && c.genBlocks >= 1 &&
		    residentTowerGeometry(c.frameHeight, c.frameWidth, cus, &m_ResGX, &m_ResGY, &m_ResRH)) {
			m_Resident = m_ResidentCapable = true;
			m_TowerW = DeviceBuffer(m_TowerHostW.size() * 2);
			m_TowerW.upload(m_TowerHostW.data(), m_TowerHostW.size() * 2);
			m_TowerB = DeviceBuffer(m_TowerHostB.size() * 4);
			m_TowerB.upload(m_TowerHostB.data(), m_TowerHostB.size() * 4);
			// (`activation: lrelu`: twice the slots -- the epoch travels beside the values)
			m_ResMail = DeviceBuffer(m_Fp8Tower ? residentMailboxBytes8(m_ResGX, m_ResGY, c.genActivation == 1)
			                                    : residentMailboxBytes(m_ResGX, m_ResGY, c.genActivation == 1));
			m_ResFlags = DeviceBuffer(residentCounterBytes(m_ResGX, m_ResGY));  // publish counts per region
			m_ResError = PinnedWords(64);
			m_ResErrorDev = m_ResError.device();
		}
		m_TowerHostW.clear();
		m_TowerHostW.shrink_to_fit();
		const char *flowMode = devSwitch(Dev::Flow);
		if (flowMode && std::string(flowMode) == "convs") m_BlockFused = false;
		if (m_Resident && !(flowMode && (std::string(flowMode) == "layers" || std::string(flowMode) == "convs")) &&
		    c.flowArch == 1 &&
		    c.flowResFilters == 64 && c.flowResBlocks >= 1 && 3 * c.numFlowInputs <= 64 &&
		    residentTowerGeometry(PH, PW, cus, &m_FlowGX, &m_FlowGY, &m_FlowRH)) {
			m_ResidentFlow = m_ResidentFlowCapable = true;
			m_FlowTowerW = DeviceBuffer(m_FlowTowerHostW.size() * 2);
			m_FlowTowerW.upload(m_FlowTowerHostW.data(), m_FlowTowerHostW.size() * 2);
			m_FlowTowerB = DeviceBuffer(m_FlowTowerHostB.size() * 4);
			m_FlowTowerB.upload(m_FlowTowerHostB.data(), m_FlowTowerHostB.size() * 4);
			m_FlowMail = DeviceBuffer(residentMailboxBytes(m_FlowGX, m_FlowGY, c.flowActivation == 1));
			m_FlowFlags = DeviceBuffer(residentCounterBytes(m_FlowGX, m_FlowGY));
		}
		m_FlowTowerHostW.clear();
		m_FlowTowerHostW.shrink_to_fit();
	}

	// ---- buffers (all zero-initialised) ----
	const std::size_t lr = static_cast<std::size_t>(H) * W;
	const std::size_t plr = static_cast<std::size_t>(PH) * PW;
	m_InStage = DeviceBuffer(lr * 4);
	m_OutStage = DeviceBuffer(lr * 16 * 4);
	m_RawStage = DeviceBuffer(lr * 16 * 4);
	for (int i = 0; i < 2; ++i) {
		m_State[i] = DeviceBuffer(lr * 16 * 4 * 2);  // f16 [4H][4W][4]
		m_Packed[i] = DeviceBuffer(plr * 16 * 2);     // [PH][PW][16]
	}
	if (c.flowArch == 0) {
		const int nb = static_cast<int>(c.flowFilters.size()) / 2;
		std::size_t px = plr;
		for (int i = 0; i < 2 * nb; ++i) {
			const std::string n = "flow/block_" + std::to_string(i + 1);
			const std::size_t f = c.flowFilters[i];
			addTensor(n + "/a_1", px * f);
			addTensor(n + "/a_2", px * f);
			px = i < nb ? px / 4 : px * 4;
			addTensor(n + "/resample", px * f);
		}
		if (c.flowFilters.size() % 2) addTensor("flow/a_1", px * c.flowFilters.back());
	} else {
		if (m_ResidentFlow) {
			addTensor("flow/in64", plr * 64);  // [PH][PW][64]: ch 0..15 = the packed flow input, rest zero
			addTowerTensor("flow/trunk", PH, PW, 64);
		}
		addTensor("flow/x0", plr * c.flowResFilters);
		addTensor("flow/x1", plr * c.flowResFilters);
		addTensor("flow/t", plr * c.flowResFilters);
	}
	addTensor("flow", plr * 32, false, true);  // the flow head: f16 whatever the compute type
	addTowerTensor("gen_in", H, W, 64);
	if (c.temporalStrength > 0.0f) {
		addTensor("pre_warp", lr * 16 * 4);  // f16 [4H][4W][4]
		m_TemporalAcc = DeviceBuffer(8 * temporalAccWords(H, W, c.temporalWindow));
	}
	addTowerTensor("trunk_a", H, W, c.genFilters);
	addTowerTensor("trunk_b", H, W, c.genFilters);
	addTowerTensor("trunk_t", H, W, c.genFilters);
	if (m_Fp8Tower) {
		m_Fp8X = DeviceBuffer(towerPixels(H, W) * 64);
		m_Fp8T = DeviceBuffer(towerPixels(H, W) * 64);
		// activation ranges: optional calibration tensor, else one fixed scale (fp8.h)
		m_Fp8Exp.assign(2 * static_cast<std::size_t>(c.genBlocks), fp8ActivationExponent(kFp8DefaultAmax));
		if (model.has("generator/fp8_amax")) {
			const TensorView &am = model.tensor("generator/fp8_amax", {2 * c.genBlocks});
			for (int i = 0; i < 2 * c.genBlocks; ++i) m_Fp8Exp[i] = fp8ActivationExponent(am.data[i]);
		}
		if (m_ResidentCapable) {
			// operands of the one-launch 8-bit tower, convolution by convolution
			const int L = 2 * c.genBlocks;
			std::vector<unsigned char> w;
			std::vector<int> scaleA, scaleB(L);
			std::vector<float> bias, mul(L + 1);
			for (int i = 0; i < L; ++i) {
				const std::string name = "generator/block_" + std::to_string(i / 2 + 1) + (i % 2 ? "/conv_2" : "/conv_1");
				const Fp8ConvWeights q = packFp8TowerWeights(m_Fp8Folded.at(name));
				w.insert(w.end(), q.w.begin(), q.w.end());
				scaleA.insert(scaleA.end(), q.scaleA.begin(), q.scaleA.end());
				const std::vector<float> &b = m_Fp8Folded.at(name).bias;
				bias.insert(bias.end(), b.begin(), b.end());
				scaleB[i] = 127 - m_Fp8Exp[i];
				mul[i] = std::ldexp(1.0f, i + 1 < L ? m_Fp8Exp[i + 1] : 0);  // (the last copy has no reader)
			}
			mul[L] = std::ldexp(1.0f, m_Fp8Exp[0]);
			auto up = [](DeviceBuffer *d, const void *src, std::size_t n) {
				*d = DeviceBuffer(n);
				d->upload(src, n);
			};
			up(&m_Fp8TowerW, w.data(), w.size());
			up(&m_Fp8TowerScaleA, scaleA.data(), scaleA.size() * 4);
			up(&m_Fp8TowerBias, bias.data(), bias.size() * 4);
			up(&m_Fp8TowerScaleB, scaleB.data(), scaleB.size() * 4);
			up(&m_Fp8TowerMul, mul.data(), mul.size() * 4);
		}
		m_Fp8Folded.clear();
	}
	addTensor("tail_y", lr * 128);
	addTensor("tower_profile", 256 * 4 * 8 * 2, true);  // u64 cycle sums of the diagnostic tower variant

	buildProgram(0);
	buildProgram(1);

	m_IO.in = m_InStage.as<std::uint8_t>();
	m_IO.inStride = static_cast<std::ptrdiff_t>(W) * 4;
	m_IO.out = m_OutStage.as<std::uint8_t>();
	m_IO.outStride = static_cast<std::ptrdiff_t>(W) * 16;
	const char *direct = devSwitch(Dev::Direct);
	m_PreferDirect = !(direct && direct[0] == '0');
	const char *noGraph = std::getenv("JU_NO_GRAPH");
	const bool useGraph = !(noGraph && noGraph[0] == '1');
	if (const char *retry = std::getenv("JU_RESIDENT_RETRY")) m_RetryBase = static_cast<unsigned>(std::atoi(retry));
	const char *directGraph = devSwitch(Dev::DirectGraph);
	m_DirectGraph = !(directGraph && directGraph[0] == '0');
	if (const char *spin = devSwitch(Dev::SyncSpinUs)) m_SpinUs = static_cast<unsigned>(std::atoi(spin));
	// frames per look-ahead pass of processBatch (1 = frame by frame)
	if (const char *la = std::getenv("JU_LOOKAHEAD")) m_BatchMax = std::min(std::max(std::atoi(la), 1), kFlowBatchMax);

	// From here on this engine launches kernels.  It joins its device's chain first and holds
	// the chain's lock until it is ready: the eager passes below run the resident tower, which
	// must not overlap a frame of another resident runtime on this device (each wants a
	// workgroup on every CU).  Frames those runtimes have in flight are waited for once, here
	// (they record completion events only while more than one engine exists).
	DeviceChain &chain = chainOf(m_Device);
	std::lock_guard<std::mutex> chainLock(chain.mutex);
	++chain.engines;
	try {
		if (chain.engines > 1) JU_HIP(hipDeviceSynchronize());
		// One eager pass per binding set: sets the kernels' dynamic-LDS attributes
		// and surfaces launch errors before anything is captured.
		m_UseGraph = false;
		// JU_TRACE_STEPS=<file> (developer switch): every launch of this eager pass is appended to the file BEFORE it
		// runs and waited for after -- the last line names the kernel behind a GPU memory fault, which the runtime
		// otherwise reports asynchronously and without a name
		const char *trace = devSwitch(Dev::TraceSteps);
		const bool traceSync = !devSwitch(Dev::TraceNoSync);  // (JU_TRACE_NOSYNC=1: the list only)
		for (int s = 0; s < 2; ++s) {
			int k = 0;
			for (const Step &st : m_Program[s]) {
				if (trace) {
					if (std::FILE *f = std::fopen(trace, "a")) {
						std::fprintf(f, "%dx%d dtype %d%s set %d step %d %s\n", W, H, static_cast<int>(m_DType), m_Fp8Tower ? " fp8" : "", s, k,
						    st.tag.c_str());
						std::fclose(f);
					}
				}
				st.run(m_Stream);
				if (trace && traceSync) m_Stream.synchronize();
				++k;
			}
		}
		m_Stream.synchronize();
		if (const unsigned code = takeResidentError()) {
			fallbackToLayers(code);
			for (int s = 0; s < 2; ++s) {
				for (const Step &st : m_Program[s]) st.run(m_Stream);
			}
			m_Stream.synchronize();
		}
		reset();
		m_UseGraph = useGraph;
		if (m_UseGraph) {
			for (int s = 0; s < 2; ++s) {
				m_Graph[s] = GraphExec::capture(m_Stream, [&] {
				for (const Step &st : m_Program[s]) st.run(m_Stream);
			});
			}
			m_Stream.synchronize();
		}
		std::ostringstream ss;
		ss << "engine ready: " << W << "x" << H << " -> " << 4 * W << "x" << 4 * H << ", "
		   << (m_DType == kF16 ? "fp16" : "bf16") << ", " << m_Program[0].size()
		   << " launches/frame, tower=" << (m_Resident ? "resident" : "per-layer")
		   << ", graph=" << (m_UseGraph ? "on" : "off");
		logMessage(LogLevel::Info, "Engine", ss.str());
	} catch (...) {
		--chain.engines;  // (no destructor runs for a half-built engine)
		throw;
	}
}

Engine::~Engine() {
	try {
		DeviceGuard g(m_Device);
		(void)hipStreamSynchronize(m_Stream);
		DeviceChain &c = chainOf(m_Device);
		std::lock_guard<std::mutex> lock(c.mutex);
		if (c.lastOwner == this) {  // (its event dies with this engine; the work behind it is done)
			c.last = nullptr;
			c.lastOwner = nullptr;
		}
		if (c.engines > 0) --c.engines;
	} catch (...) {
	}
}

unsigned Engine::takeResidentError() {
	volatile unsigned *word = m_ResError.host();
	if (word == nullptr || *word == 0) return 0;
	const unsigned code = *word;
	*word = 0;
	return code;
}

// The resident tower needs every workgroup co-resident (one per CU).  When a bounded
// neighbour wait expires -- CUs masked off or taken by another process -- the engine
// does not stay broken: it switches to the per-layer tower kernels for good.
void Engine::fallbackToLayers(unsigned code) {
	std::ostringstream ss;
	ss << "resident tower kernel: a bounded wait on a neighbouring workgroup expired (code 0x"
	   << std::hex << code << std::dec << "); " << m_ResGX * m_ResGY
	   << " workgroups are not all co-resident on this device. Switching to the per-layer tower "
	      "kernels (slower)";
	++m_Fallbacks;
	m_CleanFrames = 0;
	// retry after 512, 2048, 8192, ... clean frames
	m_RetryAfter = m_RetryBase ? std::min<std::uint64_t>(std::uint64_t(m_RetryBase) << (2 * std::min(m_Fallbacks - 1, 8u)), 1u << 30) : 0;
	if (m_RetryAfter) ss << "; the resident kernel will be tried again after " << m_RetryAfter << " frames";
	logMessage(LogLevel::Warning, "Engine", ss.str());
	m_Resident = false;
	m_ResidentFlow = false;
	m_DirectGraphs.clear();  // they replay the resident program
	dropBatchGraphs();
	// the aborted launch left the slot epochs of the regions out of step: start them over
	m_ResMail.zeroAsync(m_Stream);
	m_ResFlags.zeroAsync(m_Stream);
	m_FlowMail.zeroAsync(m_Stream);
	m_FlowFlags.zeroAsync(m_Stream);
	m_Stream.synchronize();
	for (int s = 0; s < 2; ++s) {
		m_Graph[s] = GraphExec();
		buildProgram(s);
	}
	if (m_UseGraph) {
		// One eager pass FIRST: the per-layer tower kernels have never been launched on
		// this device, and their first launch sets the dynamic-LDS attribute
		// (hipFuncSetAttribute) -- not something to do inside a stream capture; launch
		// errors also surface here.  Only the CURRENT binding set's program, on the staging
		// buffers: it writes scratch tensors and the OUTPUT half of the state ping-pong,
		// which the caller's re-run overwrites; the other set's program (same kernels,
		// other pointers) would overwrite the INPUT half the re-run still needs.
		const FrameIO keep = m_IO;
		m_IO.in = m_InStage.as<std::uint8_t>();
		m_IO.inStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 4;
		m_IO.out = m_OutStage.as<std::uint8_t>();
		m_IO.outStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 16;
		for (const Step &st : m_Program[m_Idx]) st.run(m_Stream);
		m_Stream.synchronize();
		for (int s = 0; s < 2; ++s) {
			m_Graph[s] = GraphExec::capture(m_Stream, [&] {
				for (const Step &st : m_Program[s]) st.run(m_Stream);
			});
		}
		m_Stream.synchronize();
		m_IO = keep;
	}
}

// Back to the one-launch tower after enough clean frames on the fallback path.
void Engine::restoreResident() {
	m_Stream.synchronize();
	m_Resident = true;
	m_ResidentFlow = m_ResidentFlowCapable;
	m_CleanFrames = 0;
	m_DirectGraphs.clear();
	dropBatchGraphs();
	m_ResMail.zeroAsync(m_Stream);
	m_ResFlags.zeroAsync(m_Stream);
	m_FlowMail.zeroAsync(m_Stream);
	m_FlowFlags.zeroAsync(m_Stream);
	for (int s = 0; s < 2; ++s) {
		m_Graph[s] = GraphExec();
		buildProgram(s);
	}
	if (m_UseGraph) {  // (the resident kernels ran before: nothing sets an attribute inside the capture)
		const FrameIO keep = m_IO;
		m_IO.in = m_InStage.as<std::uint8_t>();
		m_IO.inStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 4;
		m_IO.out = m_OutStage.as<std::uint8_t>();
		m_IO.outStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 16;
		for (int s = 0; s < 2; ++s) {
			m_Graph[s] = GraphExec::capture(m_Stream, [&] {
				for (const Step &st : m_Program[s]) st.run(m_Stream);
			});
		}
		m_IO = keep;
	}
	m_Stream.synchronize();
	logMessage(LogLevel::Info, "Engine", "trying the resident tower kernel again");
}

void Engine::maybeRestoreResident() {
	if (!m_Resident && m_ResidentCapable && m_RetryAfter != 0 && ++m_CleanFrames >= m_RetryAfter) {
		restoreResident();
	}
}

void Engine::reset() {
	DeviceGuard g(m_Device);
	for (int i = 0; i < 2; ++i) {
		m_State[i].zeroAsync(m_Stream);
		m_Packed[i].zeroAsync(m_Stream);
	}
	m_Stream.synchronize();
	m_Idx = 0;
}

FrameSize Engine::frameSize() const {
	const auto w = static_cast<std::size_t>(m_Config.frameWidth);
	const auto h = static_cast<std::size_t>(m_Config.frameHeight);
	return {w, h, w * 4, h * 4};
}

namespace {
// Scoped map of a graphics resource on the engine's stream (cuda.h:310-349 GraphicsResource):
// unmapped again when the copy has been enqueued, also when that throws.
struct MappedResource {
	GraphicsHandle *h;
	hipStream_t stream;
	GraphicsArray array;
	MappedResource(void *handle, hipStream_t s) : h(static_cast<GraphicsHandle *>(handle)), stream(s) {
		if (h == nullptr || h->backend == nullptr) throw std::invalid_argument("processImage: NULL graphics resource");
		array = h->backend->map(h->resource, stream);
	}
	~MappedResource() { h->backend->unmap(h->resource, stream); }
	MappedResource(const MappedResource &) = delete;
	MappedResource &operator=(const MappedResource &) = delete;
};
}  // namespace

void Engine::stageIn(const Frame &in) {
	const FrameSize fs = frameSize();
	if (in.location == Location::GraphicsResource) {
		// map -> texture array -> staging buffer -> unmap (cuda_convert.h:57-77,
		// cuda_convert.cc.cu:380-397); the array's own extent is what counts
		MappedResource m(in.ptr, m_Stream);
		if (!m.array.fourBytes || m.array.width != fs.inputWidth || m.array.height != fs.inputHeight ||
		    in.width != fs.inputWidth || in.height != fs.inputHeight) {
			throw std::invalid_argument("processImage: input texture must be " + std::to_string(fs.inputWidth) + "x" +
			                            std::to_string(fs.inputHeight) + " with four 8-bit channels");
		}
		m.h->backend->copyFromArray(m_InStage.get(), fs.inputWidth * 4, m.array, fs.inputWidth * 4, fs.inputHeight,
		    m_Stream);
		return;
	}
	if (in.ptr == nullptr || in.width != fs.inputWidth || in.height != fs.inputHeight) {
		throw std::invalid_argument("processImage: input image must be exactly " +
		                            std::to_string(fs.inputWidth) + "x" +
		                            std::to_string(fs.inputHeight));
	}
	const std::size_t rowBytes = fs.inputWidth * 4;
	const std::size_t rows = fs.inputHeight;
	const auto plain = static_cast<std::ptrdiff_t>(rowBytes);
	auto *dst = m_InStage.as<std::uint8_t>();
	auto *src = static_cast<std::uint8_t *>(in.ptr);
	if (in.stride > -plain && in.stride < plain) {
		throw std::invalid_argument("processImage: |stride| smaller than a row");
	}
	switch (in.location) {
	case Location::Host:
		if (in.stride == plain) {
			JU_HIP(hipMemcpyAsync(dst, src, rowBytes * rows, hipMemcpyHostToDevice, m_Stream));
		} else if (in.stride > 0) {
			JU_HIP(hipMemcpy2DAsync(dst, rowBytes, src, static_cast<std::size_t>(in.stride),
			    rowBytes, rows, hipMemcpyHostToDevice, m_Stream));
		} else {
			// bottom-up frame (AviSynth RGB32, avisynth_plugin/src/main.cc:125-142): upload
			// the rows in memory order, then flip on the device.
			auto *raw = m_RawStage.as<std::uint8_t>();
			const std::uint8_t *lowest = src + static_cast<std::ptrdiff_t>(rows - 1) * in.stride;
			JU_HIP(hipMemcpy2DAsync(raw, rowBytes, lowest, static_cast<std::size_t>(-in.stride),
			    rowBytes, rows, hipMemcpyHostToDevice, m_Stream));
			launchCopyRows(raw + (rows - 1) * rowBytes, -plain, dst, plain, rowBytes, rows, m_Stream);
		}
		break;
	case Location::Device:
		if (in.stride == plain) {
			JU_HIP(hipMemcpyAsync(dst, src, rowBytes * rows, hipMemcpyDeviceToDevice, m_Stream));
		} else {
			launchCopyRows(src, in.stride, dst, plain, rowBytes, rows, m_Stream);
		}
		break;
	default:
		throw std::invalid_argument(
		    "processImage: GRAPHICS_RESOURCE images are not supported by this runtime");
	}
}

void Engine::stageOut(const Frame &out) {
	const FrameSize fs = frameSize();
	if (out.location == Location::GraphicsResource) {  // cuda_convert.cc.cu:419-436
		MappedResource m(out.ptr, m_Stream);
		if (!m.array.fourBytes || m.array.width != fs.outputWidth || m.array.height != fs.outputHeight ||
		    out.width != fs.outputWidth || out.height != fs.outputHeight) {
			throw std::invalid_argument("processImage: output texture must be " + std::to_string(fs.outputWidth) + "x" +
			                            std::to_string(fs.outputHeight) + " with four 8-bit channels");
		}
		m.h->backend->copyToArray(m.array, m_OutStage.get(), fs.outputWidth * 4, fs.outputWidth * 4, fs.outputHeight,
		    m_Stream);
		return;
	}
	if (out.ptr == nullptr || out.width != fs.outputWidth || out.height != fs.outputHeight) {
		throw std::invalid_argument("processImage: output image must be exactly " +
		                            std::to_string(fs.outputWidth) + "x" +
		                            std::to_string(fs.outputHeight));
	}
	const std::size_t rowBytes = fs.outputWidth * 4;
	const std::size_t rows = fs.outputHeight;
	const auto plain = static_cast<std::ptrdiff_t>(rowBytes);
	const auto *src = m_OutStage.as<std::uint8_t>();
	auto *dst = static_cast<std::uint8_t *>(out.ptr);
	if (out.stride > -plain && out.stride < plain) {
		throw std::invalid_argument("processImage: |stride| smaller than a row");
	}
	switch (out.location) {
	case Location::Host:
		if (out.stride == plain) {
			JU_HIP(hipMemcpyAsync(dst, src, rowBytes * rows, hipMemcpyDeviceToHost, m_Stream));
		} else if (out.stride > 0) {
			JU_HIP(hipMemcpy2DAsync(dst, static_cast<std::size_t>(out.stride), src, rowBytes,
			    rowBytes, rows, hipMemcpyDeviceToHost, m_Stream));
		} else {
			auto *raw = m_RawStage.as<std::uint8_t>();
			launchCopyRows(src, plain, raw + (rows - 1) * rowBytes, -plain, rowBytes, rows, m_Stream);
			std::uint8_t *lowest = dst + static_cast<std::ptrdiff_t>(rows - 1) * out.stride;
			JU_HIP(hipMemcpy2DAsync(lowest, static_cast<std::size_t>(-out.stride), raw, rowBytes,
			    rowBytes, rows, hipMemcpyDeviceToHost, m_Stream));
		}
		break;
	case Location::Device:
		if (out.stride == plain) {
			JU_HIP(hipMemcpyAsync(dst, src, rowBytes * rows, hipMemcpyDeviceToDevice, m_Stream));
		} else {
			launchCopyRows(src, plain, dst, out.stride, rowBytes, rows, m_Stream);
		}
		break;
	default:
		throw std::invalid_argument(
		    "processImage: GRAPHICS_RESOURCE images are not supported by this runtime");
	}
}

bool Engine::directEligible(const Frame &in, const Frame &out) const {
	const FrameSize fs = frameSize();
	const auto inRow = static_cast<std::ptrdiff_t>(fs.inputWidth * 4);
	const auto outRow = static_cast<std::ptrdiff_t>(fs.outputWidth * 4);
	return m_PreferDirect && in.location == Location::Device && out.location == Location::Device &&
	       in.ptr != nullptr && out.ptr != nullptr && in.width == fs.inputWidth && in.height == fs.inputHeight &&
	       out.width == fs.outputWidth && out.height == fs.outputHeight &&
	       (in.stride >= inRow || -in.stride >= inRow) && (out.stride >= outRow || -out.stride >= outRow) &&
	       (reinterpret_cast<std::uintptr_t>(in.ptr) % 4 == 0) && in.stride % 4 == 0 &&
	       (reinterpret_cast<std::uintptr_t>(out.ptr) % 8 == 0) && out.stride % 8 == 0;
}

Engine::DirectEntry &Engine::directEntry(const DirectKey &key) {
	auto it = m_DirectGraphs.find(key);
	if (it == m_DirectGraphs.end()) {
		// evict the least recently used tuple nobody registered
		if (m_DirectGraphs.size() >= kMaxDirectGraphs + 2 * m_RegisteredPairs.size()) {
			auto victim = m_DirectGraphs.end();
			for (auto j = m_DirectGraphs.begin(); j != m_DirectGraphs.end(); ++j) {
				DirectKey pair = j->first;
				pair.idx = 0;
				if (m_RegisteredPairs.count(pair)) continue;
				if (victim == m_DirectGraphs.end() || j->second.lastUse < victim->second.lastUse) victim = j;
			}
			if (victim != m_DirectGraphs.end()) m_DirectGraphs.erase(victim);
		}
		it = m_DirectGraphs.emplace(key, DirectEntry{}).first;
	}
	return it->second;
}

// Records binding set idx's launches against m_IO (nothing executes).
void Engine::captureDirect(DirectEntry *e, int idx) {
	// ONE graph per frame: replaying the first launches as a graph of their own, so that the GPU
	// starts while the CPU still submits the rest, was measured and lost 0.5-0.8 % (1943-1947 against
	// 1926-1933 frames/s, tools/submit_overhead.py): hipGraphLaunch already streams its packets, and
	// the second graph costs a boundary
	e->graph = GraphExec::capture(m_Stream, [&] {
		for (const Step &st : m_Program[idx]) st.run(m_Stream);
	});
}

int Engine::prepareFrames(const Frame &in, const Frame &out) {
	DeviceGuard g(m_Device);
	const FrameSize fs = frameSize();
	if (in.location != Location::GraphicsResource &&
	    (in.ptr == nullptr || in.width != fs.inputWidth || in.height != fs.inputHeight)) {
		throw std::invalid_argument("prepareFrames: input image must be exactly " + std::to_string(fs.inputWidth) + "x" +
		                            std::to_string(fs.inputHeight));
	}
	if (out.location != Location::GraphicsResource &&
	    (out.ptr == nullptr || out.width != fs.outputWidth || out.height != fs.outputHeight)) {
		throw std::invalid_argument("prepareFrames: output image must be exactly " + std::to_string(fs.outputWidth) +
		                            "x" + std::to_string(fs.outputHeight));
	}
	if (!directEligible(in, out) || !m_UseGraph || !m_DirectGraph) return 0;  // staged frames: graphs exist already
	const DirectKey pair{in.ptr, in.stride, out.ptr, out.stride, 0};
	if (!m_RegisteredPairs.count(pair)) {
		if (m_RegisteredPairs.size() >= kMaxRegisteredPairs) {
			// a caller that keeps registering new buffers (and never the old ones again): forget
			// the pair whose graphs were used least recently, with its graphs
			auto victim = m_RegisteredPairs.begin();
			std::uint64_t oldest = ~std::uint64_t(0);
			for (auto it = m_RegisteredPairs.begin(); it != m_RegisteredPairs.end(); ++it) {
				std::uint64_t used = 0;
				for (int idx = 0; idx < 2; ++idx) {
					DirectKey key = *it;
					key.idx = idx;
					auto g = m_DirectGraphs.find(key);
					if (g != m_DirectGraphs.end()) used = std::max(used, g->second.lastUse);
				}
				if (used < oldest) {
					oldest = used;
					victim = it;
				}
			}
			for (int idx = 0; idx < 2; ++idx) {
				DirectKey key = *victim;
				key.idx = idx;
				m_DirectGraphs.erase(key);
			}
			m_RegisteredPairs.erase(victim);
		}
		m_RegisteredPairs.insert(pair);
	}
	std::unique_lock<std::mutex> chain = chainBegin();  // (no capture while another engine's constructor drains the device)
	const FrameIO keep = m_IO;
	m_IO.in = static_cast<const std::uint8_t *>(in.ptr);
	m_IO.inStride = in.stride;
	m_IO.out = static_cast<std::uint8_t *>(out.ptr);
	m_IO.outStride = out.stride;
	int captured = 0;
	try {
		for (int idx = 0; idx < 2; ++idx) {
			DirectKey key = pair;
			key.idx = idx;
			DirectEntry &e = directEntry(key);
			e.lastUse = ++m_DirectClock;
			if (!e.graph.valid()) {
				captureDirect(&e, idx);
				++captured;
				++m_PreparedCaptures;
			}
		}
	} catch (...) {
		m_IO = keep;
		throw;
	}
	m_IO = keep;
	return captured;
}

// ---------------------------------------------------------------------------------------------------------------
// Frame look-ahead.  The flow net reads LR frames only -- never the HR state (models.py:790, 823: its input is the
// packed history of the last num_flow_inputs frames) -- so the flow fields of n consecutive frames can be computed
// before the first of them is upscaled: ONE pass of the flow net's eight launches over n frames instead of n passes.
// At 480x270 each of those launches is ONE round of 136-240 workgroups on 256 CUs -- a 7-24 us latency chain (weights,
// staging, conv A, conv B, stores) with most SIMDs idle most of the time (0.3-0.6 waves per SIMD,
// profiles/r05_pmc_stall_flow.txt): over 8 frames the same launches take 62 instead of 105 us per frame
// (profiles/r05_flow_layers_pass.txt; priced beforehand by tools/probes/flow_batch_estimate.py), and a pass pays one
// synchronisation instead of eight: 2163 -> 2461 frames/s (profiles/r05_lookahead_bench_box_a.txt).  The recurrent part -- warp, tower,
// tail -- stays strictly frame by frame, and every frame's bytes are those of process(): the same kernels add the
// same terms in the same order whatever the launch's size.
//
// State.  Frame i of a pass reads the state frame i - 1 wrote; the pass owns the n - 1 buffers in between, reads
// m_State[set] and leaves the last frame's state in m_State[set ^ 1] and the last history in m_Packed[set ^ 1], as
// ONE process() call would: the pass flips the binding set once, and -- since nothing it wrote is read before the
// pass -- a pass that failed (resident tower: bounded wait expired) can be run again frame by frame.
// ---------------------------------------------------------------------------------------------------------------
bool Engine::batchPlanned(int items) {
	if (m_BatchUnsupported || m_Calibrate || m_Config.flowArch != 0 || !flowPacksInBlock() || m_Config.normalizeBrightness) {
		return false;
	}
	if (items <= m_BatchCap && m_BatchFlow.count({items, 0})) return true;
	try {
		if (items > m_BatchCap) {
			// (the tensors of every pass so far are too small: start over)
			m_Stream.synchronize();
			m_BatchGraphs.clear();
			m_BatchFlow.clear();
			m_BatchTensors.clear();
			// (the whole cap at once: growing later reallocates the tensors every captured pass is bound to -- also the
			// registered ones -- and round 6's bench lost a registered short pass that way.  ~210 MB at 480x270 for 8 frames.)
			const int cap = std::max(items, m_BatchMax);
			for (const auto &kv : m_Tensors) {
				const bool flowTensor = kv.first == "flow" || kv.first.rfind("flow/", 0) == 0;
				if (!flowTensor) continue;
				Tensor t;
				t.count = kv.second.count * cap;
				t.isF32 = kv.second.isF32;
				t.isState = kv.second.isState;
				t.buf = DeviceBuffer(t.count * (t.isF32 ? 4 : 2));
				m_BatchTensors.emplace(kv.first, std::move(t));
			}
			for (int i = 0; i + 1 < cap; ++i) {
				if (!m_BatchState[i].get()) m_BatchState[i] = DeviceBuffer(m_State[0].bytes());
			}
			m_BatchCap = cap;
		}
		for (int set = 0; set < 2; ++set) {
			std::vector<Step> prog;
			addFlowAutoencoder(&prog, set, items);
			m_BatchFlow[{items, set}] = std::move(prog);
		}
		return true;
	} catch (const std::exception &e) {
		// (std::logic_error: a launch of this model's flow plan has no item dimension; anything else -- the pass's
		// tensors did not fit the device -- equally means "frame by frame from now on", not a failed call)
		logMessage(dynamic_cast<const std::logic_error *>(&e) ? LogLevel::Info : LogLevel::Warning, "Engine",
		    std::string("frame look-ahead is off for this runtime: ") + e.what());
		m_BatchUnsupported = true;
		m_BatchFlow.clear();
		m_BatchTensors.clear();
		m_BatchCap = 0;
		return false;
	}
}

void Engine::setLookahead(int frames) {
	DeviceGuard g(m_Device);
	const int cap = std::min(std::max(frames, 1), kFlowBatchMax);
	if (cap < m_BatchMax) {
		// graphs of longer passes can no longer be asked for: drop them (their launches may still be in flight)
		m_Stream.synchronize();
		for (auto it = m_BatchGraphs.begin(); it != m_BatchGraphs.end();) {
			it = static_cast<int>(it->first.size()) > cap ? m_BatchGraphs.erase(it) : std::next(it);
		}
	}
	m_BatchMax = cap;
}

void Engine::dropBatchGraphs() {
	m_BatchGraphs.clear();
}

// The launches of one look-ahead pass over the n frames of m_BatchIO, in stream order (recorded when m_Stream is
// capturing): the flow net over all frames, then frame by frame the rest of binding set `set`'s per-frame program,
// bound to the frame's buffers, its flow field and its link of the state chain.
void Engine::runBatch(int set, int n, const std::function<void(const Step &, bool)> *around) {
	auto run = [&](const Step &st) {
		if (around) (*around)(st, false);
		st.run(m_Stream);
		if (around) (*around)(st, true);
	};
	const std::vector<Step> &flow = m_BatchFlow.at({n, set});
	const long flowItem = static_cast<long>(m_Tensors.at("flow").count) * 2;
	const unsigned char *flowBase = m_BatchTensors.at("flow").buf.as<unsigned char>();
	const FrameIO keepIO = m_IO;
	const StateBind keepBind = m_StateBind[set];
	const void *keepFlow = m_FlowCur;
	struct Restore {
		std::function<void()> f;
		~Restore() { f(); }
	} restore{[&] {
		m_IO = keepIO;
		m_StateBind[set] = keepBind;
		m_FlowCur = keepFlow;
	}};
	for (const Step &st : flow) run(st);
	for (int i = 0; i < n; ++i) {
		m_IO = m_BatchIO[i];
		m_FlowCur = flowBase + i * flowItem;
		m_StateBind[set].in = i == 0 ? keepBind.in : m_BatchState[i - 1].get();
		m_StateBind[set].out = i + 1 == n ? keepBind.out : m_BatchState[i].get();
		for (const Step &st : m_Program[set]) {
			if (st.tag != "flow" && st.tag != "pack") run(st);
		}
		// (a host frame: its bytes are complete in m_PassOut[i] -- tell the thread that copies them out)
		if (m_BatchHost[i].hostOut) launchSignalHost(m_PassSignal.device(), m_Stream);
	}
}

// A frame may go into a pass when each of its two images is either a device-resident one the kernels can read / write in
// place (directEligible's conditions) or a host image of the right size (staged through the pass's own device buffers).
bool Engine::passEligible(const Frame &in, const Frame &out) const {
	const FrameSize fs = frameSize();
	const auto inRow = static_cast<std::ptrdiff_t>(fs.inputWidth * 4);
	const auto outRow = static_cast<std::ptrdiff_t>(fs.outputWidth * 4);
	auto sized = [](const Frame &f, std::size_t w, std::size_t h, std::ptrdiff_t row) {
		return f.ptr != nullptr && f.width == w && f.height == h && (f.stride >= row || -f.stride >= row);
	};
	if (!sized(in, fs.inputWidth, fs.inputHeight, inRow) || !sized(out, fs.outputWidth, fs.outputHeight, outRow)) return false;
	auto side = [&](const Frame &f, unsigned align) {
		if (f.location == Location::Host) return true;
		return f.location == Location::Device && m_PreferDirect && reinterpret_cast<std::uintptr_t>(f.ptr) % align == 0 &&
		       f.stride % static_cast<std::ptrdiff_t>(align) == 0;
	};
	return side(in, 4) && side(out, 8);
}

// Binds the n frames of a pass: m_BatchIO[i] = what frame i's kernels read and write -- the caller's device memory, or
// for a host image the pass's device buffer i, addressed with the SIGN of the caller's stride (a bottom-up host frame is
// uploaded / downloaded in memory order and read / written bottom-up by the kernels: no flip pass).  The key of the
// pass's graph is made of those bindings, so all-host passes of one length and orientation share one graph.
std::vector<Engine::DirectKey> Engine::bindBatch(const Frame *in, const Frame *out, int n, int set) {
	const FrameSize fs = frameSize();
	const auto inRow = static_cast<std::ptrdiff_t>(fs.inputWidth * 4);
	const auto outRow = static_cast<std::ptrdiff_t>(fs.outputWidth * 4);
	std::vector<DirectKey> key(static_cast<std::size_t>(n));
	for (int i = 0; i < n; ++i) {
		FrameIO &io = m_BatchIO[i];
		m_BatchHost[i].hostIn = in[i].location == Location::Host;
		m_BatchHost[i].hostOut = out[i].location == Location::Host;
		if (m_BatchHost[i].hostIn) {
			if (!m_PassIn[i].get()) m_PassIn[i] = DeviceBuffer(fs.inputHeight * static_cast<std::size_t>(inRow));
			auto *base = m_PassIn[i].as<std::uint8_t>();
			io.in = in[i].stride >= 0 ? base : base + static_cast<std::ptrdiff_t>(fs.inputHeight - 1) * inRow;
			io.inStride = in[i].stride >= 0 ? inRow : -inRow;
		} else {
			io.in = static_cast<const std::uint8_t *>(in[i].ptr);
			io.inStride = in[i].stride;
		}
		if (m_BatchHost[i].hostOut) {
			if (!m_PassOut[i].get()) m_PassOut[i] = DeviceBuffer(fs.outputHeight * static_cast<std::size_t>(outRow));
			if (!m_PassSignal.host()) m_PassSignal = PinnedWords(64);
			if (!m_CopyStream) m_CopyStream = std::make_unique<Stream>();
			auto *base = m_PassOut[i].as<std::uint8_t>();
			io.out = out[i].stride >= 0 ? base : base + static_cast<std::ptrdiff_t>(fs.outputHeight - 1) * outRow;
			io.outStride = out[i].stride >= 0 ? outRow : -outRow;
		} else {
			io.out = static_cast<std::uint8_t *>(out[i].ptr);
			io.outStride = out[i].stride;
		}
		key[i] = DirectKey{io.in, io.inStride, io.out, io.outStride, set};
	}
	return key;
}

// Every host input of the pass into its device buffer, rows in MEMORY order (the binding carries the orientation), on the
// engine's stream in front of the pass's launches.  Pageable memory: the runtime stages or page-locks per call, as in
// stageIn; 0.52 MB per frame.
void Engine::uploadPassInputs(const Frame *in, int n) {
	const FrameSize fs = frameSize();
	const std::size_t rowBytes = fs.inputWidth * 4, rows = fs.inputHeight;
	for (int i = 0; i < n; ++i) {
		if (!m_BatchHost[i].hostIn) continue;
		const auto *p0 = static_cast<const std::uint8_t *>(in[i].ptr);
		const std::uint8_t *lowest = in[i].stride >= 0 ? p0 : p0 + static_cast<std::ptrdiff_t>(rows - 1) * in[i].stride;
		const std::size_t pitch = static_cast<std::size_t>(in[i].stride >= 0 ? in[i].stride : -in[i].stride);
		if (pitch == rowBytes) {
			JU_HIP(hipMemcpyAsync(m_PassIn[i].get(), lowest, rowBytes * rows, hipMemcpyHostToDevice, m_Stream));
		} else {
			JU_HIP(hipMemcpy2DAsync(m_PassIn[i].get(), rowBytes, lowest, pitch, rowBytes, rows, hipMemcpyHostToDevice, m_Stream));
		}
	}
}

// The thread blocked in processBatch: wait for frame i's completion count, copy frame i out on the copy stream while the
// GPU runs frame i + 1, in order.  The wait is bounded by the pass itself: once the engine's stream has drained, a count
// that has not arrived never will.
void Engine::drainPassOutputs(const Frame *out, int n) {
	const FrameSize fs = frameSize();
	const std::size_t rowBytes = fs.outputWidth * 4, rows = fs.outputHeight;
	volatile unsigned *word = m_PassSignal.host();
	unsigned due = 0;
	bool any = false;
	for (int i = 0; i < n; ++i) {
		if (!m_BatchHost[i].hostOut) continue;
		++due;
		auto arrived = [&] { return static_cast<int>(*word - m_PassSignalBase) >= static_cast<int>(due); };
		for (unsigned spins = 1; !arrived(); ++spins) {
			if ((spins & 255u) == 0) {
				const hipError_t st = hipStreamQuery(m_Stream);
				if (st == hipSuccess) {
					if (arrived()) break;
					throw std::runtime_error("look-ahead pass: the completion count of a host frame did not arrive");
				}
				if (st != hipErrorNotReady) JU_HIP(st);
			} else {
				__builtin_ia32_pause();
			}
		}
		auto *p0 = static_cast<std::uint8_t *>(out[i].ptr);
		std::uint8_t *lowest = out[i].stride >= 0 ? p0 : p0 + static_cast<std::ptrdiff_t>(rows - 1) * out[i].stride;
		const std::size_t pitch = static_cast<std::size_t>(out[i].stride >= 0 ? out[i].stride : -out[i].stride);
		if (pitch == rowBytes) {
			JU_HIP(hipMemcpyAsync(lowest, m_PassOut[i].get(), rowBytes * rows, hipMemcpyDeviceToHost, *m_CopyStream));
		} else {
			JU_HIP(hipMemcpy2DAsync(lowest, pitch, m_PassOut[i].get(), rowBytes, rowBytes, rows, hipMemcpyDeviceToHost, *m_CopyStream));
		}
		any = true;
	}
	if (any) JU_HIP(hipStreamSynchronize(*m_CopyStream));
}

Engine::DirectEntry &Engine::batchEntry(const std::vector<DirectKey> &key) {
	auto it = m_BatchGraphs.find(key);
	if (it == m_BatchGraphs.end()) {
		// least recently used out -- among the tuples nobody registered: a tuple handed to prepareBatch keeps its graphs
		// (the header promises that process calls on it never capture), as registered pairs do; only a caller that keeps
		// registering new tuples (more than kMaxRegisteredBatches) loses the registered one it used least recently
		std::size_t registered = 0;
		for (const auto &kv : m_BatchGraphs) registered += kv.second.registered ? 1 : 0;
		if (m_BatchGraphs.size() - registered >= kMaxBatchGraphs || registered >= kMaxRegisteredBatches) {
			const bool fromRegistered = m_BatchGraphs.size() - registered < kMaxBatchGraphs;
			auto victim = m_BatchGraphs.end();
			for (auto j = m_BatchGraphs.begin(); j != m_BatchGraphs.end(); ++j) {
				if (j->second.registered != fromRegistered) continue;
				if (victim == m_BatchGraphs.end() || j->second.lastUse < victim->second.lastUse) victim = j;
			}
			if (victim != m_BatchGraphs.end()) m_BatchGraphs.erase(victim);
		}
		it = m_BatchGraphs.emplace(key, DirectEntry{}).first;
	}
	it->second.lastUse = ++m_DirectClock;
	return it->second;
}

// ju_prepare_batch: the graphs of a tuple of frame buffers a caller is going to hand to processBatch, one per
// binding set, captured NOW (as prepareFrames does for one pair): nothing executes, no buffer is touched.  Returns
// the graphs captured; 0 for a tuple that will not go as one pass.
int Engine::prepareBatch(const Frame *in, const Frame *out, int n) {
	if (n < 0 || (n > 0 && (in == nullptr || out == nullptr))) throw std::invalid_argument("prepareBatch: bad arguments");
	DeviceGuard g(m_Device);
	if (n < 2 || n > m_BatchMax || !m_UseGraph || !m_DirectGraph) return 0;
	for (int i = 0; i < n; ++i) {
		if (!passEligible(in[i], out[i])) return 0;
	}
	if (!batchPlanned(n)) return 0;
	std::unique_lock<std::mutex> chain = chainBegin();  // (no capture while another engine's constructor drains the device)
	int captured = 0;
	for (int set = 0; set < 2; ++set) {
		DirectEntry &e = batchEntry(bindBatch(in, out, n, set));
		e.registered = true;
		if (e.graph.valid()) continue;
		{
			DryLaunchScope dry;  // the attributes of the tile heights this pass's launch sizes choose
			for (const Step &st : m_BatchFlow.at({n, set})) st.run(m_Stream);
		}
		e.graph = GraphExec::capture(m_Stream, [&] { runBatch(set, n); });
		e.seen = 2;
		++captured;
		++m_PreparedCaptures;
	}
	return captured;
}

// One look-ahead pass over frames [0, n): enqueue only.  On return the binding set is flipped ONCE (see above).
void Engine::submitBatch(const Frame *in, const Frame *out, int n) {
	const int set = m_Idx;
	const std::vector<DirectKey> key = bindBatch(in, out, n, set);
	uploadPassInputs(in, n);  // (outside the chain lock: a pageable upload blocks its caller)
	m_PassSignalBase = m_PassSignal.host() ? *m_PassSignal.host() : 0u;
	for (int i = 0; i < n; ++i) m_BatchHostFrames += (m_BatchHost[i].hostIn || m_BatchHost[i].hostOut) ? 1 : 0;
	{
		std::unique_lock<std::mutex> chain = chainBegin();
		bool replayed = false;
		if (m_UseGraph && m_DirectGraph) {
			DirectEntry &e = batchEntry(key);
			// (first sighting: eager -- it also sets the dynamic-LDS attribute of a tile height this pass's launch
			// sizes choose for the first time, which must not happen inside a capture; second: capture and replay)
			if (!e.graph.valid() && ++e.seen >= 2) {
				e.graph = GraphExec::capture(m_Stream, [&] { runBatch(set, n); });
				++m_InlineCaptures;
			}
			if (e.graph.valid()) {
				e.graph.launch(m_Stream);
				++m_GraphReplays;
				replayed = true;
			}
		}
		if (!replayed) {
			runBatch(set, n);
			++m_EagerRuns;
		}
		chainEnd(chain);
	}
	m_Idx = set ^ 1;
	m_BatchFrames += static_cast<std::uint64_t>(n);
}

void Engine::processBatch(const Frame *in, const Frame *out, int count) {
	if (count < 0 || (count > 0 && (in == nullptr || out == nullptr))) throw std::invalid_argument("processBatch: bad arguments");
	DeviceGuard g(m_Device);
	int i = 0;
	while (i < count) {
		// the longest run of frames from i that can go as one pass: device-resident, and none of them READING what an
		// earlier frame of the pass writes (frame by frame such an input would be read after that write; the pass's flow
		// sweep reads every input first)
		auto range = [](const Frame &f) {
			const auto rows = static_cast<std::ptrdiff_t>(f.height);
			const auto *p0 = static_cast<const std::uint8_t *>(f.ptr);
			const std::uint8_t *lo = f.stride >= 0 ? p0 : p0 + (rows - 1) * f.stride;
			const std::size_t bytes = static_cast<std::size_t>(rows - 1) * static_cast<std::size_t>(f.stride >= 0 ? f.stride : -f.stride) + f.width * 4;
			return std::make_pair(lo, lo + bytes);
		};
		// ... nor WRITING what an earlier frame of the pass reads: on the normal path that write comes after the read
		// (frame k's tail after frame j's, j < k), but a pass whose resident tower timed out is run again frame by frame
		// from its inputs, which must then still be what they were (advisor, round 5)
		auto overlap = [](const std::pair<const std::uint8_t *, const std::uint8_t *> &a,
		                  const std::pair<const std::uint8_t *, const std::uint8_t *> &b) {
			return a.first < b.second && b.first < a.second;
		};
		int n = 0;
		while (i + n < count && n < m_BatchMax && passEligible(in[i + n], out[i + n])) {
			const auto r = range(in[i + n]), w = range(out[i + n]);
			bool clash = false;
			for (int k = 0; k < n && !clash; ++k) {  // (host and device addresses are different spaces)
				clash = (in[i + n].location == out[i + k].location && overlap(r, range(out[i + k]))) ||
				        (out[i + n].location == in[i + k].location && overlap(w, range(in[i + k])));
			}
			if (clash) break;
			++n;
		}
		if (n < 2 || !batchPlanned(n)) {
			process(in[i], out[i]);
			++i;
			continue;
		}
		const int set = m_Idx;
		submitBatch(in + i, out + i, n);
		drainPassOutputs(out + i, n);  // host frames: each copied out while the next one runs
		m_Stream.synchronizeSpin(m_SpinUs);
		if (const unsigned code = takeResidentError()) {
			// nothing the pass wrote was one of its inputs -- neither the state (see above) nor a frame buffer (the pass
			// splitter): the same frames again, one by one, on the per-block kernels
			m_Idx = set;
			m_BatchFrames -= static_cast<std::uint64_t>(n);
			for (int k = 0; k < n; ++k) m_BatchHostFrames -= (m_BatchHost[k].hostIn || m_BatchHost[k].hostOut) ? 1 : 0;
			fallbackToLayers(code);
			for (int k = 0; k < n; ++k) {
				submit(in[i + k], out[i + k]);
				m_Stream.synchronize();
			}
		} else {
			for (int k = 0; k < n; ++k) maybeRestoreResident();
		}
		i += n;
	}
}

void Engine::runProgram() {
	if (m_UseGraph && !m_DirectIO && m_Graph[m_Idx].valid()) {
		m_Graph[m_Idx].launch(m_Stream);
		++m_GraphReplays;
		return;
	}
	if (m_UseGraph && m_DirectIO && m_DirectGraph) {
		const DirectKey key{m_IO.in, m_IO.inStride, m_IO.out, m_IO.outStride, m_Idx};
		DirectEntry &e = directEntry(key);
		e.lastUse = ++m_DirectClock;
		if (!e.graph.valid()) {
			DirectKey pair = key;
			pair.idx = 0;
			// a registered pair whose graph was dropped (fallback / return to the resident
			// kernel) is captured again at once; an unknown tuple at its second sighting --
			// a caller that hands over a fresh pointer every frame never pays a capture
			if (++e.seen >= 2 || m_RegisteredPairs.count(pair)) {
				captureDirect(&e, m_Idx);  // records the launches (nothing executes here) ...
				++m_InlineCaptures;
			}
		}
		if (e.graph.valid()) {  // ... and replay
			e.graph.launch(m_Stream);
			++m_GraphReplays;
			return;
		}
	}
	for (const Step &st : m_Program[m_Idx]) st.run(m_Stream);
	++m_EagerRuns;
}

void Engine::submit(const Frame &in, const Frame &out) {
	const FrameSize fs = frameSize();
	// Device-resident frames: the kernels read the caller's input and write the caller's
	// output directly (any signed stride), no staging copies.
	m_DirectIO = directEligible(in, out);
	// The per-device chain lock covers the wait on the previous resident frame, the program's launches
	// (and an inline graph capture) and the completion record -- NOT the staging copies: a pageable
	// host-to-device copy blocks its caller, and would block every other runtime of the device with it
	// (advisor, round 3).  They are ordered by the stream; a copy kernel beside another runtime's
	// resident tower merely waits for a free CU or delays a workgroup's start by microseconds.
	if (m_DirectIO) {
		m_IO.in = static_cast<const std::uint8_t *>(in.ptr);
		m_IO.inStride = in.stride;
		m_IO.out = static_cast<std::uint8_t *>(out.ptr);
		m_IO.outStride = out.stride;
	} else {
		m_IO.in = m_InStage.as<std::uint8_t>();
		m_IO.inStride = static_cast<std::ptrdiff_t>(fs.inputWidth) * 4;
		m_IO.out = m_OutStage.as<std::uint8_t>();
		m_IO.outStride = static_cast<std::ptrdiff_t>(fs.outputWidth) * 4;
		stageIn(in);
	}
	{
		std::unique_lock<std::mutex> chain = chainBegin();
		runProgram();
		chainEnd(chain);
	}
	if (!m_DirectIO) stageOut(out);
	m_Idx ^= 1;  // state ping-pong (tensorrt_backend.cc:277)
}

void Engine::enqueue(const Frame &in, const Frame &out) {
	DeviceGuard g(m_Device);
	submit(in, out);
}

void Engine::process(const Frame &in, const Frame &out) {
	DeviceGuard g(m_Device);
	submit(in, out);
	m_Stream.synchronizeSpin(m_SpinUs);
	if (const unsigned code = takeResidentError()) {
		// the frame's inputs (previous state, frame history) are intact: the step only
		// wrote the other half of the ping-pong -- run it again on the per-layer path
		// (submit() flipped m_Idx; the re-run needs the same binding set again.  If the
		// fallback or the re-run throws, the failed frame must not count as a step either.)
		m_Idx ^= 1;
		fallbackToLayers(code);
		submit(in, out);
		m_Stream.synchronize();
	} else {
		maybeRestoreResident();
	}
}

void Engine::synchronize() {
	DeviceGuard g(m_Device);
	m_Stream.synchronize();
	if (const unsigned code = takeResidentError()) {
		fallbackToLayers(code);
		throw std::runtime_error(
		    "resident tower kernel failed (workgroups not co-resident); the frames enqueued since "
		    "the last ju_synchronize are invalid. The runtime now uses the per-layer path: "
		    "ju_reset and continue");
	}
}

std::vector<std::string> Engine::tensorNames() const {
	std::vector<std::string> names = {"state", "flow_in", "trunk"};
	for (const auto &kv : m_Tensors) names.push_back(kv.first);
	return names;
}

std::size_t Engine::readTensor(const std::string &name, float *dst, std::size_t capacity) {
	DeviceGuard g(m_Device);
	const void *src = nullptr;
	std::size_t count = 0;
	bool f32 = false;
	const Tensor *tw = nullptr;
	DType dt = m_DType;
	const std::size_t lr = static_cast<std::size_t>(m_Config.frameHeight) * m_Config.frameWidth;
	if (name == "state") {  // the state the NEXT frame will read = last output_raw
		src = m_State[m_Idx].get();
		count = lr * 16 * 4;
		dt = kF16;
	} else if (name == "flow_in") {
		src = m_Packed[m_Idx].get();
		count = static_cast<std::size_t>(m_Config.paddedHeight()) * m_Config.paddedWidth() * 16;
	} else {
		const std::string key = name == "trunk" ? m_TrunkOut : name;
		auto it = m_Tensors.find(key);
		if (it == m_Tensors.end()) throw std::invalid_argument("unknown tensor " + name);
		src = it->second.buf.get();
		count = it->second.count;
		f32 = it->second.isF32;
		if (it->second.isState) dt = kF16;
		if (it->second.towerC) tw = &it->second;
	}
	const std::size_t outCount =
	    tw ? static_cast<std::size_t>(tw->towerH) * tw->towerW * tw->towerC : count;
	if (dst == nullptr) return outCount;
	if (capacity < outCount) throw std::invalid_argument("readTensor: buffer too small");
	m_Stream.synchronize();
	if (f32) {
		JU_HIP(hipMemcpy(dst, src, count * 4, hipMemcpyDeviceToHost));
	} else {
		DeviceBuffer tmp(count * 4);
		launchToFloat(dt, src, tmp.as<float>(), count, m_Stream);
		m_Stream.synchronize();
		if (tw) {  // strip the zero border of the tower layout: dense [H][W][C] out
			const std::size_t rowElems = static_cast<std::size_t>(tw->towerW) * tw->towerC;
			JU_HIP(hipMemcpy2D(dst, rowElems * 4,
			    tmp.as<float>() + towerOrigin(tw->towerW) * tw->towerC,
			    static_cast<std::size_t>(towerPitch(tw->towerW)) * tw->towerC * 4, rowElems * 4,
			    tw->towerH, hipMemcpyDeviceToHost));
		} else {
			JU_HIP(hipMemcpy(dst, tmp.get(), count * 4, hipMemcpyDeviceToHost));
		}
	}
	return outCount;
}

double Engine::flopsOf(const std::string &tagSpecIn) const {
	std::string tagSpec = tagSpecIn;
	std::size_t at = tagSpec.find("@frame");
	bool inPass = false;
	if (at == std::string::npos && (at = tagSpec.find("@pass")) != std::string::npos) inPass = true;
	if (at != std::string::npos) tagSpec = tagSpec.substr(0, at);
	std::string tag = tagSpec;
	int only = -1;
	const std::size_t hash = tagSpec.find('#');
	if (hash != std::string::npos) {
		tag = tagSpec.substr(0, hash);
		only = std::atoi(tagSpec.c_str() + hash + 1);
	}
	double f = 0.0;
	int k = 0;
	// ("flow@pass": the pass's flow launches, each over all its frames -- once timeSteps has planned them)
	const auto passFlow = m_BatchFlow.find({m_BatchMax, 0});
	for (const Step &s : (inPass && tag == "flow" && passFlow != m_BatchFlow.end()) ? passFlow->second : m_Program[0]) {
		if (tag.empty() || s.tag == tag) {
			if (only < 0 || k == only) f += s.flops;
			++k;
		}
	}
	return f;
}

double Engine::timeSteps(const std::string &tagSpecIn, int iters, int *launches) {
	DeviceGuard g(m_Device);
	// "flow#3" = only the 4th step tagged "flow" (per-layer timing); "tower@frame" = the steps
	// tagged "tower" timed INSIDE whole frames (every step of the frame runs, events bracket the
	// tagged launches): the kernel in the clock / cache context of the real workload, which is
	// what a kernel trace of the benchmark averages -- back-to-back launches of the tower alone
	// draw more power and read 4-5 % slower on the same box
	// "tower@pass" = the same inside look-ahead passes of JU_LOOKAHEAD frames (processBatch): there the towers of
	// consecutive frames follow one another with only the warp in between; "flow@pass": the pass's flow launches
	// (each covers all frames of the pass)
	std::string tagSpec = tagSpecIn;
	bool inFrame = false, inPass = false;
	std::size_t at = tagSpec.find("@frame");
	if (at != std::string::npos) {
		inFrame = true;
		tagSpec = tagSpec.substr(0, at);
	} else if ((at = tagSpec.find("@pass")) != std::string::npos) {
		inPass = true;
		tagSpec = tagSpec.substr(0, at);
		if (m_BatchMax < 2 || !batchPlanned(m_BatchMax)) throw std::invalid_argument("timeSteps: this model has no look-ahead passes");
	}
	std::string tag = tagSpec;
	int only = -1;
	const std::size_t hash = tagSpec.find('#');
	if (hash != std::string::npos) {
		tag = tagSpec.substr(0, hash);
		only = std::atoi(tagSpec.c_str() + hash + 1);
	}
	std::vector<const Step *> steps;
	int k = 0;
	const bool passFlow = inPass && tag == "flow";
	for (const Step &s : passFlow ? m_BatchFlow.at({m_BatchMax, m_Idx}) : m_Program[m_Idx]) {
		if (tag.empty() || s.tag == tag) {
			if (only < 0 || k == only) steps.push_back(&s);
			++k;
		}
	}
	if (launches) *launches = static_cast<int>(steps.size());
	if (steps.empty() || iters <= 0) return 0.0;
	m_IO.in = m_InStage.as<std::uint8_t>();
	m_IO.inStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 4;
	m_IO.out = m_OutStage.as<std::uint8_t>();
	m_IO.outStride = static_cast<std::ptrdiff_t>(m_Config.frameWidth) * 16;
	double ms = 0.0;
	std::unique_lock<std::mutex> chain = chainBegin();  // (the timed launches may be resident towers)
	if (inPass) {
		const int n = m_BatchMax, set = m_Idx;
		for (int i = 0; i < n; ++i) {  // (every frame of the pass on the staging buffers)
			m_BatchIO[i] = m_IO;
			m_BatchHost[i] = PassFrame{};
		}
		std::vector<std::unique_ptr<Event>> ev;
		const std::function<void(const Step &, bool)> around = [&](const Step &s, bool) {
			if (std::find(steps.begin(), steps.end(), &s) == steps.end()) return;
			ev.emplace_back(new Event());
			ev.back()->record(m_Stream);
		};
		runBatch(set, n);  // warm
		for (int i = 0; i < iters; ++i) runBatch(set, n, &around);
		chainEnd(chain);
		m_Stream.synchronize();
		double sum = 0.0;
		for (std::size_t i = 0; i + 1 < ev.size(); i += 2) sum += static_cast<double>(Event::elapsedMs(*ev[i], *ev[i + 1]));
		ms = ev.empty() ? 0.0 : sum / (static_cast<double>(ev.size()) / 2);
	} else if (inFrame) {
		auto isTimed = [&](const Step *s) { return std::find(steps.begin(), steps.end(), s) != steps.end(); };
		for (const Step &s : m_Program[m_Idx]) s.run(m_Stream);  // warm
		std::vector<std::unique_ptr<Event>> ev;
		for (int i = 0; i < iters; ++i) {
			for (const Step &s : m_Program[m_Idx]) {
				const bool timed = isTimed(&s);
				if (timed) {
					ev.emplace_back(new Event());
					ev.back()->record(m_Stream);
				}
				s.run(m_Stream);
				if (timed) {
					ev.emplace_back(new Event());
					ev.back()->record(m_Stream);
				}
			}
		}
		chainEnd(chain);
		m_Stream.synchronize();
		double sum = 0.0;
		for (std::size_t i = 0; i + 1 < ev.size(); i += 2) sum += static_cast<double>(Event::elapsedMs(*ev[i], *ev[i + 1]));
		ms = sum / (static_cast<double>(iters) * steps.size());
	} else {
		for (const Step *s : steps) s->run(m_Stream);  // warm
		Event t0, t1;
		t0.record(m_Stream);
		for (int i = 0; i < iters; ++i) {
			for (const Step *s : steps) s->run(m_Stream);
		}
		t1.record(m_Stream);
		chainEnd(chain);
		t1.synchronize();
		ms = static_cast<double>(Event::elapsedMs(t0, t1)) / (static_cast<double>(iters) * steps.size());
	}
	// The timed launches ran outside the frame sequence: scratch tensors, the output
	// staging buffer and (with JU_TAIL=tower) the recurrent state were overwritten.  Start
	// the stream from a clean state again, and do not leave a bounded-wait failure of the
	// resident tower behind for the next process() to trip over.
	const unsigned code = takeResidentError();
	if (code) fallbackToLayers(code);  // (also zeroes the mailboxes: the aborted launch left the slot epochs out of step)
	reset();
	if (code) {
		std::ostringstream ss;
		ss << "timeSteps: the resident tower kernel reported a bounded-wait timeout (code 0x" << std::hex
		   << code << "); the timing is invalid";
		throw std::runtime_error(ss.str());
	}
	return ms;
}

double Engine::stat(const std::string &key) const {
	if (key == "graph_replays") return static_cast<double>(m_GraphReplays);
	if (key == "eager_runs") return static_cast<double>(m_EagerRuns);
	if (key == "graph_captures") return static_cast<double>(m_InlineCaptures);
	if (key == "prepared_captures") return static_cast<double>(m_PreparedCaptures);
	if (key == "registered_pairs") return static_cast<double>(m_RegisteredPairs.size());
	if (key == "resident_tower") return m_Resident ? 1.0 : 0.0;
	if (key == "resident_flow") return m_ResidentFlow ? 1.0 : 0.0;
	if (key == "tower_fast") {  // the generator's resident tower runs the fast schedule (16-bit models, every region of its shape)
		return (m_Resident && !m_Fp8Tower && residentTowerFast() && !m_Calibrate &&
		        residentTowerFastGeometry(m_Config.frameHeight, m_Config.frameWidth, m_ResGX, m_ResGY, m_ResRH)) ? 1.0 : 0.0;
	}
	if (key == "fallbacks") return static_cast<double>(m_Fallbacks);
	if (key == "lookahead_frames") return static_cast<double>(m_BatchFrames);  // frames that went through look-ahead passes
	if (key == "lookahead_host_frames") return static_cast<double>(m_BatchHostFrames);  // ... of them with a host image
	if (key == "lookahead_max") return static_cast<double>(m_BatchMax);
	if (key == "launches_per_frame") return static_cast<double>(m_Program[0].size());
	if (key == "tower_variant") return static_cast<double>(towerVariant());  // (developer switch, tests)
	if (key == "direct_graphs") {
		double n = 0;
		for (const auto &kv : m_DirectGraphs) n += kv.second.graph.valid() ? 1 : 0;
		return n;
	}
	throw std::invalid_argument("unknown stat " + key);
}

}  // namespace ju
