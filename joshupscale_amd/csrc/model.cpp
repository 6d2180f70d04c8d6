#include "model.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <stdexcept>

namespace ju {

namespace {

constexpr char kMagic[8] = {'J', 'U', 'P', 'W', 'G', 'T', '\0', '\1'};
constexpr std::size_t kHeaderBytes = 128;
constexpr std::size_t kEntryBytes = 128;

template <typename T>
T readLE(const unsigned char *p) {
	T v;
	std::memcpy(&v, p, sizeof(T));  // x86-64 / little endian host
	return v;
}

}  // namespace

ModelFile::ModelFile(const void *blob, std::size_t size) {
	if (blob == nullptr || size < kHeaderBytes) {
		throw std::invalid_argument("Invalid model: too small");
	}
	m_Bytes.assign(static_cast<const unsigned char *>(blob),
	    static_cast<const unsigned char *>(blob) + size);
	const unsigned char *b = m_Bytes.data();
	if (std::memcmp(b, kMagic, 8) != 0) {
		// The reference's .trt engines (core/src/tensorrt_backend.cc:129-143) are
		// rejected here with a clear message rather than mis-parsed.
		throw std::invalid_argument(
		    "Invalid model: not a JoshUpscale-AMD .jupw container (TensorRT engines are not "
		    "supported by this runtime)");
	}
	const auto version = readLE<std::uint32_t>(b + 8);
	const auto headerBytes = readLE<std::uint32_t>(b + 12);
	if (version != 1 || headerBytes < kHeaderBytes || headerBytes > size) {
		throw std::invalid_argument("Invalid model: unsupported container version");
	}
	auto u32 = [&](std::size_t off) { return static_cast<int>(readLE<std::uint32_t>(b + off)); };
	ModelConfig &c = m_Config;
	c.frameHeight = u32(16);
	c.frameWidth = u32(20);
	const int scale = u32(24);
	c.numFlowInputs = u32(28);
	c.flowArch = u32(32);
	c.flowPadFactor = u32(36);
	c.normalizeBrightness = u32(40) != 0;
	c.genFilters = u32(44);
	c.genBlocks = u32(48);
	c.flowResFilters = u32(52);
	c.flowResBlocks = u32(56);
	const int nFlowFilters = u32(60);
	if (nFlowFilters < 0 || nFlowFilters > 8) {
		throw std::invalid_argument("Invalid model: bad flow filter count");
	}
	for (int i = 0; i < nFlowFilters; ++i) c.flowFilters.push_back(u32(64 + 4 * i));
	c.bnEps = readLE<float>(b + 96);
	c.computeDtype = u32(100);
	const int nTensors = u32(104);
	c.temporalStrength = readLE<float>(b + 108);
	c.temporalThreshold = readLE<float>(b + 112);
	{
		const auto acts = readLE<std::uint32_t>(b + 116);
		c.flowActivation = static_cast<int>(acts & 0xffu);
		c.genActivation = static_cast<int>((acts >> 8) & 0xffu);
		if (acts >> 16) throw std::invalid_argument("Invalid model: unknown activation field");
		c.flowNegativeSlope = readLE<float>(b + 120);
		c.genNegativeSlope = readLE<float>(b + 124);
	}
	if (headerBytes >= 160) {  // extended temporal-filter fields (model_file.py)
		c.temporalWindow = u32(128);
		c.temporalGain = readLE<float>(b + 132);
		const auto flags = readLE<std::uint32_t>(b + 136);
		if (flags >> 3) throw std::invalid_argument("Invalid model: unknown temporal filter flags");
		c.temporalL2 = (flags & 1u) != 0;
		c.temporalLimit = (flags & 2u) != 0;
		c.temporalLuma = (flags & 4u) != 0;
	}
	if (scale != 4) throw std::invalid_argument("Invalid model: scale must be 4");
	if (nTensors < 0 || headerBytes + static_cast<std::size_t>(nTensors) * kEntryBytes > size) {
		throw std::invalid_argument("Invalid model: truncated tensor table");
	}
	for (int i = 0; i < nTensors; ++i) {
		const unsigned char *e = b + headerBytes + static_cast<std::size_t>(i) * kEntryBytes;
		char name[93];
		std::memcpy(name, e, 92);
		name[92] = '\0';
		const auto ndim = readLE<std::uint32_t>(e + 92);
		if (ndim > 4) throw std::invalid_argument("Invalid model: tensor rank > 4");
		TensorView t;
		std::size_t count = 1;
		for (std::uint32_t d = 0; d < ndim; ++d) {
			const auto dim = readLE<std::uint32_t>(e + 96 + 4 * d);
			// (bounded per dimension: the product below cannot wrap, dims fit an int)
			if (dim == 0 || dim > 65536) {
				throw std::invalid_argument(std::string("Invalid model: bad dimension in ") + name);
			}
			t.dims.push_back(static_cast<int>(dim));
			count *= dim;
			if (count > size) throw std::invalid_argument(std::string("Invalid model: bad tensor entry ") + name);
		}
		const auto off = readLE<std::uint64_t>(e + 112);
		const auto cnt = readLE<std::uint64_t>(e + 120);
		if (cnt != count || off % 4 != 0 || off > size || cnt > (size - off) / 4) {
			throw std::invalid_argument(std::string("Invalid model: bad tensor entry ") + name);
		}
		t.count = count;
		t.data = reinterpret_cast<const float *>(b + off);
		if (!m_Tensors.emplace(name, std::move(t)).second) {
			throw std::invalid_argument(std::string("Invalid model: duplicate tensor ") + name);
		}
	}
	validateConfig(m_Config);
}

void validateConfig(const ModelConfig &c) {
	auto bad = [](const std::string &what) { throw std::invalid_argument("Invalid model: " + what); };
	if (c.frameHeight < 2 || c.frameWidth < 2 || c.frameHeight > 8192 || c.frameWidth > 8192) {
		bad("unsupported frame size");
	}
	if (c.numFlowInputs < 1 || c.numFlowInputs > 5) bad("1..5 flow inputs supported");
	if (c.flowArch != 0 && c.flowArch != 1) bad("unknown flow architecture");
	if (c.computeDtype != kF16 && c.computeDtype != kBF16 && c.computeDtype != 2) {
		bad("unknown compute dtype");  // 2 = JU_DTYPE_FP8 (e4m3 block convolutions over fp16)
	}
	if (c.flowPadFactor < 0 || c.flowPadFactor > 256) bad("flow_pad_factor must be in 0..256");
	// Widths: the reference constructors take any integer (models.py:257-263, 334-339, 484-491); this
	// engine admits what its GPU parity tests run against the oracle (tests/test_gpu_parity.py
	// test_nondefault_widths_match_oracle): multiples of the 32-channel MFMA block up to the largest
	// width tested.  model_file.py validate_config states the same limits with the same messages.
	if (c.genFilters <= 0 || c.genFilters > 256 || c.genFilters % 32 != 0) {
		bad("gen_filters must be a multiple of 32 (at most 256)");
	}
	if (c.genBlocks < 0 || c.genBlocks > 256) bad("gen_blocks must be in 0..256");
	if (!(c.bnEps > 0.0f) || !std::isfinite(c.bnEps)) bad("bn_eps must be positive and finite");
	if (!(c.temporalStrength >= 0.0f && c.temporalStrength <= 1.0f) ||
	    !(c.temporalThreshold >= 0.0f && c.temporalThreshold <= 1.0f)) {
		bad("temporal filter strength/threshold must be in [0, 1]");
	}
	if (c.temporalWindow < 0 || c.temporalWindow > 4096) bad("temporal filter window must be in 0..4096");
	if (!(c.temporalGain >= 0.0f && c.temporalGain <= 1e6f)) bad("temporal filter gain must be in [0, 1e6]");
	// ACTIVATIONS = {relu, lrelu} (models.py:24-27).  The slope must keep the activation
	// monotonic (the flow encoder pools AFTER it in the reference and before it here).
	auto checkAct = [&](int act, float slope, const char *what) {
		if (act != 0 && act != 1) bad(std::string("unknown ") + what + " activation");
		if (act == 1 && !(slope >= 0.0f && slope <= 1.0f)) {
			bad(std::string(what) + " negative_slope must be in [0, 1]");
		}
		if (act == 0 && slope != 0.0f) bad(std::string(what) + " negative_slope set on a relu model");
	};
	checkAct(c.flowActivation, c.flowNegativeSlope, "flow");
	checkAct(c.genActivation, c.genNegativeSlope, "generator");
	if (c.flowArch == 0) {
		const int nb = static_cast<int>(c.flowFilters.size()) / 2;
		const int PH = c.paddedHeight(), PW = c.paddedWidth();
		if (nb < 1 || PH % (1 << nb) != 0 || PW % (1 << nb) != 0) {
			bad("padded frame size must be divisible by 2^(flow depth)");
		}
		for (int f : c.flowFilters) {
			if (f <= 0 || f > 512 || f % 32 != 0) bad("flow filters must be multiples of 32 (at most 512)");
		}
	} else {
		if (c.flowResFilters <= 0 || c.flowResFilters > 256 || c.flowResFilters % 32 != 0) {
			bad("flow_res_filters must be a multiple of 32 (at most 256)");
		}
		if (c.flowResBlocks < 0 || c.flowResBlocks > 256) bad("flow_res_blocks must be in 0..256");
	}
	// Size: several kernels address a tensor with 32-bit byte offsets (buffer descriptors, scalar + lane offsets), so
	// no activation tensor may reach 4 GiB.  The largest is a full-resolution one of the widest layer in the padded
	// layout ((H rounded up to 8) + 2 rows of (W rounded up to 32) + 2 pixels; the HR state is 128 B per LR pixel like
	// a 64-channel layer).  64-channel models: 33 M pixels (8192 x 4064, 5760 x 5800); tests/test_gpu_presets.py runs the
	// engines at that size.  The reference has no such limit -- TensorRT builds an engine for whatever frame it is given.
	{
		int widest = std::max(64, c.genFilters);
		if (c.flowArch == 0) {
			for (std::size_t i = 0; i < c.flowFilters.size(); ++i) {
				// (level of unit i: 0, 1, .., nb - 1, bottom, .., 0: a level-k tensor has 4^-k of the pixels)
				const int nb = static_cast<int>(c.flowFilters.size()) / 2;
				const int level = static_cast<int>(i) < nb ? static_cast<int>(i) : std::max(0, 2 * nb - 1 - static_cast<int>(i));
				widest = std::max(widest, c.flowFilters[i] >> (2 * std::min(level, 4)));
			}
		} else {
			widest = std::max(widest, c.flowResFilters);
		}
		const unsigned long long rows = static_cast<unsigned long long>((c.paddedHeight() + 7) / 8 * 8 + 2);
		const unsigned long long pitch = static_cast<unsigned long long>((c.paddedWidth() + 31) / 32 * 32 + 2);
		if (rows * pitch * 2ull * static_cast<unsigned long long>(widest) > 0xFFC00000ull) {
			bad("frame too large for this model: an activation tensor would reach 4 GiB (" + std::to_string(widest) +
			    " channels x " + std::to_string(rows * pitch) + " pixels x 2 bytes)");
		}
	}
}

const TensorView &ModelFile::tensor(const std::string &name) const {
	auto it = m_Tensors.find(name);
	if (it == m_Tensors.end()) {
		throw std::invalid_argument("Invalid model: missing tensor " + name);
	}
	return it->second;
}

const TensorView &ModelFile::tensor(const std::string &name, const std::vector<int> &dims) const {
	const TensorView &t = tensor(name);
	if (t.dims != dims) {
		throw std::invalid_argument("Invalid model: unexpected shape for " + name);
	}
	return t;
}

namespace {

// per-channel BN scale/shift in double
void bnScaleShift(const ModelFile &m, const std::string &bn, int c, std::vector<double> *scale,
    std::vector<double> *shift) {
	scale->assign(c, 1.0);
	shift->assign(c, 0.0);
	if (bn.empty()) return;
	const float *g = m.tensor(bn + "/gamma", {c}).data;
	const float *be = m.tensor(bn + "/beta", {c}).data;
	const float *mu = m.tensor(bn + "/moving_mean", {c}).data;
	const float *var = m.tensor(bn + "/moving_variance", {c}).data;
	const double eps = m.config().bnEps;
	for (int i = 0; i < c; ++i) {
		const double s = static_cast<double>(g[i]) / std::sqrt(static_cast<double>(var[i]) + eps);
		(*scale)[i] = s;
		(*shift)[i] = static_cast<double>(be[i]) - static_cast<double>(mu[i]) * s;
	}
}

}  // namespace

FoldedConv foldConv(const ModelFile &m, const std::string &convName,
    const std::string &bnPrefix, bool hasBias) {
	const TensorView &k = m.tensor(convName + "/kernel");
	if (k.dims.size() != 4 || k.dims[0] != k.dims[1] || (k.dims[0] != 3 && k.dims[0] != 1)) {
		throw std::invalid_argument("Invalid model: " + convName + " must be a 3x3 or 1x1 conv");
	}
	FoldedConv f;
	f.taps = k.dims[0] * k.dims[1];
	f.cin = k.dims[2];
	f.cout = k.dims[3];
	std::vector<double> scale, shift;
	bnScaleShift(m, bnPrefix, f.cout, &scale, &shift);
	f.w.resize(static_cast<std::size_t>(f.taps) * f.cin * f.cout);
	for (std::size_t i = 0; i < f.w.size(); ++i) {
		f.w[i] = static_cast<float>(static_cast<double>(k.data[i]) * scale[i % f.cout]);
	}
	f.bias.resize(f.cout);
	const float *b = hasBias ? m.tensor(convName + "/bias", {f.cout}).data : nullptr;
	for (int o = 0; o < f.cout; ++o) {
		f.bias[o] = static_cast<float>((b ? static_cast<double>(b[o]) * scale[o] : 0.0) + shift[o]);
	}
	return f;
}

FoldedConv foldConvTranspose2x2(
    const ModelFile &m, const std::string &convName, const std::string &bnPrefix) {
	const TensorView &k = m.tensor(convName + "/kernel");
	if (k.dims.size() != 4 || k.dims[0] != 2 || k.dims[1] != 2) {
		throw std::invalid_argument("Invalid model: " + convName + " must be a 2x2 conv-transpose");
	}
	const int co = k.dims[2], ci = k.dims[3];
	std::vector<double> scale, shift;
	bnScaleShift(m, bnPrefix, co, &scale, &shift);
	FoldedConv f;
	f.taps = 1;
	f.cin = ci;
	f.cout = 4 * co;
	f.w.resize(static_cast<std::size_t>(ci) * 4 * co);
	f.bias.resize(4 * co);
	for (int ab = 0; ab < 4; ++ab) {
		for (int o = 0; o < co; ++o) {
			for (int c = 0; c < ci; ++c) {
				// y[2h+a,2w+b,o] = sum_c x[h,w,c] K[a,b,o,c]   
				f.w[static_cast<std::size_t>(c) * 4 * co + ab * co + o] = static_cast<float>(
				    static_cast<double>(k.data[(static_cast<std::size_t>(ab) * co + o) * ci + c]) *
				    scale[o]);
			}
			f.bias[ab * co + o] = static_cast<float>(shift[o]);
		}
	}
	return f;
}

std::uint16_t floatToF16(float f) {
	std::uint32_t x;
	std::memcpy(&x, &f, 4);
	const std::uint16_t sign = static_cast<std::uint16_t>((x >> 16) & 0x8000u);
	x &= 0x7fffffffu;
	if (x >= 0x7f800000u) return sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u);
	if (x >= 0x477ff000u) return sign | 0x7c00u;  // rounds to >= 65520 -> inf
	if (x < 0x38800000u) {                         // below 2^-14: half subnormal
		float af;
		std::memcpy(&af, &x, 4);
		const float scaled = af * 16777216.0f;  // * 2^24, exact
		const auto r = static_cast<std::uint32_t>(std::nearbyint(scaled));  // RNE
		return sign | static_cast<std::uint16_t>(r);
	}
	const std::uint32_t mant = x & 0x7fffffu;
	const std::uint32_t exp = (x >> 23) - 112u;
	std::uint32_t h = (exp << 10) | (mant >> 13);
	const std::uint32_t rem = mant & 0x1fffu;
	if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
	return sign | static_cast<std::uint16_t>(h);
}

std::uint16_t floatToBF16(float f) {
	std::uint32_t x;
	std::memcpy(&x, &f, 4);
	if ((x & 0x7fffffffu) > 0x7f800000u) return static_cast<std::uint16_t>((x >> 16) | 0x40u);
	x += 0x7fffu + ((x >> 16) & 1u);
	return static_cast<std::uint16_t>(x >> 16);
}

std::vector<std::uint16_t> packTailWeights(const float *k2, DType dt) {
	std::vector<std::uint16_t> out(2 * 64 * 8);
	for (int ks = 0; ks < 2; ++ks) {
		for (int l = 0; l < 64; ++l) {
			for (int j = 0; j < 8; ++j) {
				const int m = l & 31, k = 16 * ks + 8 * (l >> 5) + j;
				float v = 0.f;
				if (m < 16 && (m & 3) < 3) {
					const int q = m >> 2, c = m & 3;  // q = a'*2 + b'
					v = k2[(q * 3 + c) * 32 + k];     // [a'][b'][c][o]
				}
				out[(ks * 64 + l) * 8 + j] = dt == kF16 ? floatToF16(v) : floatToBF16(v);
			}
		}
	}
	return out;
}

namespace {

std::vector<int> identityMap(int cin, int cinP) {
	std::vector<int> m(cinP, -1);
	for (int i = 0; i < cin; ++i) m[i] = i;
	return m;
}

int roundUp16(int v) { return (v + 15) / 16 * 16; }

// Packed generator-input record (see warp_pack_kernel) -> reference channel order
// [LR frame (3), space_to_depth(pre_warp) (48)]  (models.py:523-530).
std::vector<int> generatorInputMap() {
	std::vector<int> m(64, -1);
	for (int i = 0; i < 4; ++i) {
		for (int j = 0; j < 4; ++j) {
			for (int c = 0; c < 3; ++c) m[i * 16 + j * 3 + c] = 3 + (i * 4 + j) * 3 + c;
		}
	}
	m[12] = 0;
	m[13] = 1;
	m[14] = 2;
	return m;
}

}  // namespace

std::vector<ConvSpec> foldModel(const ModelFile &model) {
	const ModelConfig &c = model.config();
	std::vector<ConvSpec> out;
	// one layer: fold, check it against what the graph feeds it and expects of it
	auto add = [&](const std::string &conv, const std::string &bn, bool bias, int h, int w, int taps,
	               int cin, int cout) {
		ConvSpec s;
		s.name = conv;
		s.conv = foldConv(model, conv, bn, bias);
		if (s.conv.taps != taps || s.conv.cin != cin || s.conv.cout != cout) {
			throw std::invalid_argument("Invalid model: " + conv + " is " +
			    std::to_string(s.conv.cin) + " -> " + std::to_string(s.conv.cout) + " (" +
			    std::to_string(s.conv.taps) + " taps), the graph needs " + std::to_string(cin) +
			    " -> " + std::to_string(cout) + " (" + std::to_string(taps) + " taps)");
		}
		s.cinMap = identityMap(cin, roundUp16(cin));
		s.H = h;
		s.W = w;
		out.push_back(std::move(s));
	};
	const int H = c.frameHeight, W = c.frameWidth;
	// ---- flow (models.py:257-331, 334-481) ----
	int cin = 3 * c.numFlowInputs;
	int h = c.paddedHeight(), w = c.paddedWidth();
	if (c.flowArch == 0) {
		const int nb = static_cast<int>(c.flowFilters.size()) / 2;
		for (int i = 0; i < 2 * nb; ++i) {
			const std::string n = "flow/block_" + std::to_string(i + 1);
			const int f = c.flowFilters[i];
			add(n + "/conv_1", n + "/bn_1", false, h, w, 9, cin, f);
			add(n + "/conv_2", n + "/bn_2", false, h, w, 9, f, f);
			cin = f;
			if (i < nb) {
				h /= 2;
				w /= 2;
			} else {
				h *= 2;
				w *= 2;
			}
		}
		if (c.flowFilters.size() % 2) {
			add("flow/conv_1", "flow/bn_1", false, h, w, 9, cin, c.flowFilters.back());
			cin = c.flowFilters.back();
		}
		add("flow/conv_2", "", true, h, w, 9, cin, 32);
	} else {
		const int n = c.flowResFilters;
		add("flow/conv_1", "flow/bn_1", false, h, w, 9, cin, n);
		for (int i = 0; i < c.flowResBlocks; ++i) {
			const std::string b = "flow/block_" + std::to_string(i + 1);
			add(b + "/conv_1", b + "/bn_1", false, h, w, 9, n, n);
			add(b + "/conv_2", b + "/bn_2", false, h, w, 9, n, n);
		}
		add("flow/conv_2", "", true, h, w, 1, n, 32);
	}
	// ---- generator (models.py:484-595) ----
	add("generator/conv_1", "generator/bn_1", false, H, W, 9, 51, c.genFilters);
	out.back().cinMap = generatorInputMap();
	for (int i = 0; i < c.genBlocks; ++i) {
		const std::string b = "generator/block_" + std::to_string(i + 1);
		add(b + "/conv_1", b + "/bn_1", false, H, W, 9, c.genFilters, c.genFilters);
		add(b + "/conv_2", b + "/bn_2", false, H, W, 9, c.genFilters, c.genFilters);
	}
	{
		ConvSpec s;
		s.name = "generator/conv_trans_1";
		s.conv = foldConvTranspose2x2(model, s.name, "generator/bn_2");
		if (s.conv.cout != 128 || s.conv.cin != c.genFilters) {
			throw std::invalid_argument("Invalid model: conv_trans_1 must be gen_filters -> 32");
		}
		s.cinMap = identityMap(s.conv.cin, roundUp16(s.conv.cin));
		s.H = H;
		s.W = W;
		out.push_back(std::move(s));
	}
	(void)model.tensor("generator/conv_trans_2/kernel", {2, 2, 3, 32});
	(void)model.tensor("generator/conv_trans_2/bias", {3});
	return out;
}

std::vector<std::uint16_t> packConvWeights(
    const FoldedConv &c, const std::vector<int> &cinMap, int nb, DType dt) {
	const int cinP = static_cast<int>(cinMap.size());
	if (cinP % 16 != 0 || (nb != 1 && nb != 2) || c.cout % (32 * nb) != 0) {
		throw std::invalid_argument("packConvWeights: cin must pad to 16, cout to 32*nb");
	}
	const int CK = convCK(cinP);
	const int COG = 32 * nb;
	const int KS = CK / 16;
	const int nCC = cinP / CK;
	const int nCOG = c.cout / COG;
	std::vector<std::uint16_t> out(static_cast<std::size_t>(c.taps) * cinP * c.cout);
	std::size_t idx = 0;
	for (int cog = 0; cog < nCOG; ++cog) {
		for (int cc = 0; cc < nCC; ++cc) {
			for (int tap = 0; tap < c.taps; ++tap) {
				for (int ks = 0; ks < KS; ++ks) {
					for (int h = 0; h < 2; ++h) {
						for (int n = 0; n < COG; ++n) {
							for (int j = 0; j < 8; ++j) {
								const int kp = cc * CK + ks * 16 + h * 8 + j;
								const int src = cinMap[kp];
								float v = 0.f;
								if (src >= 0) {
									if (src >= c.cin) {
										throw std::out_of_range("packConvWeights: cinMap");
									}
									v = c.w[(static_cast<std::size_t>(tap) * c.cin + src) * c.cout +
									        cog * COG + n];
								}
								out[idx++] = dt == kF16 ? floatToF16(v) : floatToBF16(v);
							}
						}
					}
				}
			}
		}
	}
	return out;
}

std::vector<std::uint16_t> packTowerWeightsM16(const FoldedConv &c, const std::vector<int> &cinMap, DType dt) {
	if (cinMap.size() != 64 || c.cout != 64 || c.taps != 9) {
		throw std::invalid_argument("packTowerWeightsM16: a 3x3 convolution of 64 (padded) input and 64 output channels");
	}
	std::vector<std::uint16_t> out(static_cast<std::size_t>(9) * 64 * 64);
	std::size_t idx = 0;
	for (int tap = 0; tap < 9; ++tap) {
		for (int ks32 = 0; ks32 < 2; ++ks32) {
			for (int c16 = 0; c16 < 2; ++c16) {
				for (int ch = 0; ch < 2; ++ch) {
					for (int l = 0; l < 64; ++l) {
						const int co = ch * 32 + c16 * 16 + (l & 15);
						for (int j = 0; j < 8; ++j) {
							const int src = cinMap[ks32 * 32 + 8 * (l >> 4) + j];
							float v = 0.f;
							if (src >= 0) {
								if (src >= c.cin) throw std::out_of_range("packTowerWeightsM16: cinMap");
								v = c.w[(static_cast<std::size_t>(tap) * c.cin + src) * c.cout + co];
							}
							out[idx++] = dt == kF16 ? floatToF16(v) : floatToBF16(v);
						}
					}
				}
			}
		}
	}
	return out;
}

}  // namespace ju
