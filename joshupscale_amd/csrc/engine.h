// The per-frame recurrent super-resolution engine (MI355X / gfx950).
//
// Replaces the reference's TensorRTBackend (reference
// core/include/JoshUpscale/core/tensorrt_backend.h:20-53,
// core/src/tensorrt_backend.cc:117-288): same contract — construct from model
// bytes on the current device, `process(in, out)` = stage-in, one execution of
// the inference graph against binding set `idx`, stage-out, synchronise, flip
// `idx` — but the graph is this engine's own schedule of hand-written HIP
// kernels (*_kernels.hip) captured into two hipGraphs instead of a TensorRT
// execution context.
#pragma once

#include <cstddef>
#include <cstdint>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "hip_util.h"
#include "kernels.h"
#include "model.h"

namespace ju {

// Frame descriptor of the boundary (reference core/public/JoshUpscale/core.h:30-38):
// 4 bytes per pixel B,G,R,X; `stride` in bytes, may be negative (bottom-up);
// `ptr` addresses the first logical row.
enum class Location : std::uint8_t { Host = 0, Device = 1, GraphicsResource = 2 };

struct Frame {
	void *ptr;
	Location location;
	std::ptrdiff_t stride;
	std::size_t width;
	std::size_t height;
};

struct FrameSize {
	std::size_t inputWidth, inputHeight, outputWidth, outputHeight;
};

class Engine {
public:
	// dtypeOverride: -1 = the container's hint, else kF16 / kBF16.
	Engine(int device, const void *blob, std::size_t size, int dtypeOverride);
	~Engine();
	Engine(const Engine &) = delete;
	Engine &operator=(const Engine &) = delete;

	// Synchronous, like Runtime::processImage (tensorrt_backend.cc:270-278).
	void process(const Frame &in, const Frame &out);
	// `count` consecutive frames of the stream, synchronously: the frames process(in[0], out[0]) ...
	// process(in[count - 1], out[count - 1]) would write, byte for byte -- but every input must hold its pixels when
	// the call is made (frame look-ahead): where the model and the frames allow (device-resident frames, the flow
	// auto-encoder's one-launch plan) the flow fields of up to kFlowBatchMax frames are computed in one pass of the
	// flow net's launches (engine.cpp, "Frame look-ahead"); anything else runs frame by frame.
	void processBatch(const Frame *in, const Frame *out, int count);
	// Registers a tuple of device-resident frame buffers the caller is going to hand to processBatch as ONE pass
	// (2 .. JU_LOOKAHEAD frames): its graphs, one per binding set, are captured now -- what prepareFrames is to
	// process.  Nothing executes.  Returns the graphs captured (0: the tuple will not go as one pass).
	int prepareBatch(const Frame *in, const Frame *out, int count);
	// Frames per look-ahead pass, 1 (off) .. kFlowBatchMax, clamped.  Default: kFlowBatchMax, or JU_LOOKAHEAD at creation.
	void setLookahead(int frames);
	// Asynchronous variant for device-resident frames: enqueue only.
	void enqueue(const Frame &in, const Frame &out);
	void synchronize();
	// Registers a pair of device-resident frame buffers the caller is going to hand to
	// process() / enqueue(): the per-frame graphs of the pair (one per binding set) are
	// captured NOW, so that no later call pays a capture -- the reference captures its graphs
	// in the constructor too, never inside process (tensorrt_backend.cc:257-263).  Nothing
	// executes and the buffers are not touched.  Pairs that cannot take the direct path
	// (host frames, graphics resources, odd alignment) need nothing: their frames go through
	// the staging buffers, whose graphs the constructor captured.  Returns the number of
	// graphs captured by this call (0 when the pair was registered before).
	int prepareFrames(const Frame &in, const Frame &out);
	// Zero the recurrent state (what destroying and recreating the runtime does
	// in the reference: obs_plugin/src/filter.cc:146-151).
	void reset();

	FrameSize frameSize() const;
	int device() const { return m_Device; }
	DType dtype() const { return m_DType; }
	// JU_DTYPE_* as reported to the caller: 2 when the block convolutions run in e4m3
	int reportedDtype() const { return m_Fp8Tower ? 2 : static_cast<int>(m_DType); }
	const ModelConfig &config() const { return m_Config; }
	void setUseGraph(bool on) { m_UseGraph = on; }

	// ---- introspection (tests, bench) ----
	// Copies a named internal tensor to host as f32; returns its element count
	// (call with dst == nullptr to query).  Names: see Engine::tensorNames().
	std::size_t readTensor(const std::string &name, float *dst, std::size_t capacity);
	std::vector<std::string> tensorNames() const;
	// Average device time (ms) of one launch of the steps carrying `tag`, measured
	// with HIP events on the engine's own stream over `iters` repetitions of that
	// group; *launches receives the number of kernel launches per repetition.  The timed
	// launches overwrite scratch tensors and possibly the recurrent state: the stream is
	// reset() afterwards (callers time AFTER their frames, as bench.py does).
	double timeSteps(const std::string &tag, int iters, int *launches);
	// FLOPs (2*MAC) of the steps carrying `tag` (one repetition).
	double flopsOf(const std::string &tag) const;
	// "graph_replays", "eager_runs", "direct_graphs", "resident_tower", "resident_flow",
	// "launches_per_frame"
	double stat(const std::string &key) const;

private:
	struct Tensor {
		DeviceBuffer buf;
		std::size_t count = 0;
		bool isF32 = false;
		bool isState = false;  // f16 regardless of the compute dtype
		int towerH = 0, towerW = 0, towerC = 0;  // != 0: zero-bordered tower layout
	};
	// A tensor as a conv operand: pointer to image pixel (0,0) + row pitch in pixels
	// (0 = dense).
	struct Operand {
		void *ptr = nullptr;
		int pitch = 0;
	};
	struct Step {
		std::string tag;
		double flops;
		std::function<void(hipStream_t)> run;
	};
	struct ConvWeights {
		DeviceBuffer w;
		DeviceBuffer wBlock;  // 64 -> 64 block convolutions: a second copy packed with nb = 1 for flow_block_kernel
		DeviceBuffer bias;
		int cinP = 0, cout = 0, taps = 9, cinReal = 0;
		int nb = 2, rw = 2;  // tile shape the weights were packed for
		bool splitK = false;  // a coarse flow layer that runs as conv_splitk_kernel (packed with nb = 1)
	};

	Tensor &addTensor(const std::string &name, std::size_t count, bool f32 = false,
	    bool state = false);
	// H x W: the resolution the layer runs at (decides its tile shape)
	ConvWeights &addConv(const std::string &name, const FoldedConv &f,
	    const std::vector<int> &cinMap, int H, int W);
	struct ItemStride {  // a look-ahead launch: the layer of `items` frames, their tensors `in` / `out` bytes apart
		int items;
		long in, out;
		ItemStride(int n = 1, long i = 0, long o = 0) : items(n), in(i), out(o) {}
	};
	void addConvStep(std::vector<Step> *prog, const std::string &tag, const std::string &wname,
	    Operand in, Operand res, Operand out, int H, int W, bool relu, bool outHead,
	    bool tower = false, bool pool = false, bool upsample = false, ItemStride item = ItemStride());
	void addFlowAutoencoder(std::vector<Step> *prog, int set, int items);
	bool flowPacksInBlock() const;
	Operand operand(const std::string &name);
	Tensor &addTowerTensor(const std::string &name, int H, int W, int C);
	void buildWeights(const ModelFile &model);
	void buildProgram(int set);
	void stageIn(const Frame &in);
	void stageOut(const Frame &out);
	void runProgram();
	// Frame buffers the kernels read / write in THIS call.  Graph replay always uses
	// the internal staging buffers (static pointers); eager launches of device-resident
	// frames use the caller's buffers directly (no staging copies).
	struct FrameIO {
		const std::uint8_t *in = nullptr;
		std::ptrdiff_t inStride = 0;
		std::uint8_t *out = nullptr;
		std::ptrdiff_t outStride = 0;
	};
	FrameIO m_IO;
	// what the per-frame program of a binding set reads as the previous HR state / writes as the new one, and the
	// flow field its warp reads: read at launch (capture) time, like m_IO -- a look-ahead pass rebinds them per frame
	struct StateBind {
		const void *in = nullptr;
		void *out = nullptr;
	};
	StateBind m_StateBind[2];
	const void *m_FlowCur = nullptr;
	bool m_DirectIO = false;  // decided per call
	bool m_PreferDirect = true;  // JU_DIRECT=0: always stage (and replay the graph)
	void submit(const Frame &in, const Frame &out);

	int m_Device;
	ModelConfig m_Config;
	DType m_DType;
	Stream m_Stream;
	bool m_UseGraph = true;
	// process() polls the stream this long before it blocks (JU_SYNC_SPIN_US, 0 = block at once)
	unsigned m_SpinUs = 2000;
	int m_Idx = 0;
	std::string m_TrunkOut = "trunk_a";

	std::map<std::string, Tensor> m_Tensors;
	std::map<std::string, ConvWeights> m_Convs;
	// 8-bit tower (JU_DTYPE_FP8, fp8.h): e4m3 weights of the block convolutions, the two
	// e4m3 activation tensors (block input, first conv's output) and the per-tensor
	// exponents, index 2i = input of block i's conv_1, 2i+1 = input of its conv_2
	struct Fp8Conv {
		DeviceBuffer w, scaleA;
	};
	bool m_Fp8Tower = false;
	std::map<std::string, Fp8Conv> m_Fp8Convs;
	std::map<std::string, FoldedConv> m_Fp8Folded;  // (construction only)
	DeviceBuffer m_Fp8X, m_Fp8T;
	std::vector<int> m_Fp8Exp;
	// resident 8-bit tower (tower8_resident_kernel): every block's operands in one buffer each
	DeviceBuffer m_Fp8TowerW, m_Fp8TowerScaleA, m_Fp8TowerBias, m_Fp8TowerScaleB, m_Fp8TowerMul;
	DeviceBuffer m_TailW2, m_TailB2, m_TailW2Frag;
	DeviceBuffer m_Zeros;  // a zero page: the source of out-of-image pixels for LDS-DMA tile staging
	DeviceBuffer m_TemporalAcc;  // 32.32 fixed-point sum of |gen - pre_warp| (temporal filter)
	// flow auto-encoder: which blocks run as ONE launch (flow_block_kernel: both convs, the
	// pool, and the preceding bilinear x2).  Unit k < 2*nb = block k+1, the last = the head
	// pair flow/conv_1 + flow/conv_2.  JU_FLOW_CONV=generic: none.
	struct FlowUnit {
		bool fused = false;
		bool upsIn = false;  // its input is the half-resolution tensor (upsample folded into staging)
	};
	std::vector<FlowUnit> m_FlowUnits;
	bool m_FlowFused = true;
	// residual blocks outside the resident tower (more regions than CUs, LeakyReLU models,
	// after a fallback): one launch per BLOCK (flow_block_kernel, intermediate tensor in LDS);
	// JU_TOWER=convs keeps one launch per convolution
	bool m_BlockFused = true;
	// JU_CALIBRATE=1 (tools/calibrate.py): per-convolution launches + max |output| of every
	// tower layer per frame into "tower_profile" (as float bit patterns)
	bool m_Calibrate = false;
	void planFlowUnits();
	bool flowConvIsFused(const std::string &name) const;
	bool m_FusedUpsample = true;  // flow decoder: bilinear x2 folded into the next conv's staging
	bool m_FusedPool = true;  // max-pool folded into the flow encoder's conv epilogues
	bool m_PackInBlock = true;  // the flow net's first block builds the packed input itself (JU_PACK=split: own launch)
	bool m_FusedTail = true;  // JU_TAIL=split: convT1 as a conv launch + the VALU tail kernel
	bool m_TailInTower = false;  // JU_TAIL=tower: the fused tail runs inside the resident tower launch
	DeviceBuffer m_InStage, m_OutStage, m_RawStage;
	DeviceBuffer m_State[2], m_Packed[2];
	// resident tower (one launch for all residual-block convolutions)
	bool m_Resident = false;
	int m_ResGX = 0, m_ResGY = 0, m_ResRH = 0;
	std::vector<std::uint16_t> m_TowerHostW;
	std::vector<float> m_TowerHostB;
	DeviceBuffer m_TowerW, m_TowerB, m_ResMail, m_ResFlags;
	// flow-resnet (models.py:257-331) through the same resident kernel: conv_1 + its
	// residual blocks are a 64-filter tower too
	bool m_ResidentFlow = false;
	int m_FlowGX = 0, m_FlowGY = 0, m_FlowRH = 0;
	std::vector<std::uint16_t> m_FlowTowerHostW;
	std::vector<float> m_FlowTowerHostB;
	DeviceBuffer m_FlowTowerW, m_FlowTowerB, m_FlowMail, m_FlowFlags;
	// The resident tower needs every workgroup co-resident.  When a bounded wait expires the
	// engine drops to the per-block kernels (fallbackToLayers) -- not for good: after
	// m_RetryAfter clean frames it tries the resident kernel again (the CUs may have been
	// taken only temporarily, e.g. by another process), backing off x4 after every failure.
	// JU_RESIDENT_RETRY=<frames> sets the first interval (default 512, 0 = never retry).
	bool m_ResidentCapable = false, m_ResidentFlowCapable = false;
	unsigned m_RetryBase = 512, m_RetryAfter = 0, m_CleanFrames = 0, m_Fallbacks = 0;
	void restoreResident();
	void maybeRestoreResident();
	// Two resident towers cannot share the GPU (each wants a workgroup on every CU and both
	// would wait for workgroups that cannot be scheduled).  Runtimes of one process on one
	// device therefore chain their frames through events when more than one of them uses the
	// resident kernel; a single runtime pays nothing.
	Event m_FrameDone;
	std::unique_lock<std::mutex> chainBegin();
	void chainEnd(std::unique_lock<std::mutex> &lock);
	PinnedWords m_ResError;              // pinned, device-visible: word 0 = the tower's error report
	unsigned *m_ResErrorDev = nullptr;
	unsigned takeResidentError();        // 0 = none; clears it
	void fallbackToLayers(unsigned code);
	std::vector<Step> m_Program[2];
	GraphExec m_Graph[2];
	// Device-resident frames (JU_LOC_DEVICE, no staging): the kernels take the caller's
	// pointers, so a captured graph is valid for ONE (input, output, strides, binding set)
	// tuple.  Callers reuse a handful of frame buffers (OBS: one texture pair; bench.py:
	// 16 inputs, 1 output), so the graphs are cached by that tuple: prepareFrames() captures
	// a registered pair's two graphs at once (the reference captures in its constructor,
	// tensorrt_backend.cc:257-263); for an unregistered tuple the second call captures it and
	// every later one replays it.  Tuples seen once run eagerly, so a caller that hands over a
	// fresh pointer every frame never pays a capture.
	struct DirectKey {
		const void *in;
		std::ptrdiff_t inStride;
		void *out;
		std::ptrdiff_t outStride;
		int idx;
		bool operator<(const DirectKey &o) const {
			return std::tie(in, inStride, out, outStride, idx) <
			       std::tie(o.in, o.inStride, o.out, o.outStride, o.idx);
		}
	};
	struct DirectEntry {
		GraphExec graph;
		unsigned seen = 0;
		std::uint64_t lastUse = 0;
		bool registered = false;  // a look-ahead tuple handed to prepareBatch: exempt from the LRU eviction
	};
	std::map<DirectKey, DirectEntry> m_DirectGraphs;
	// pairs registered through prepareFrames (idx = 0 in the key): captured at their FIRST
	// sighting also after the cache was dropped (fallback to / return from the per-block path)
	std::set<DirectKey> m_RegisteredPairs;
	bool directEligible(const Frame &in, const Frame &out) const;
	DirectEntry &directEntry(const DirectKey &key);
	void captureDirect(DirectEntry *e, int idx);
	// frame look-ahead (processBatch; engine.cpp): the frames of the pass being recorded, the flow net's tensors for
	// m_BatchCap frames, the state buffers between the frames of a pass, the flow launches per (frames, binding set)
	// and the graphs per tuple of frame buffers
	FrameIO m_BatchIO[kFlowBatchMax];
	std::map<std::string, Tensor> m_BatchTensors;
	DeviceBuffer m_BatchState[kFlowBatchMax - 1];
	int m_BatchCap = 0, m_BatchMax = kFlowBatchMax;
	bool m_BatchUnsupported = false;
	std::map<std::pair<int, int>, std::vector<Step>> m_BatchFlow;
	std::map<std::vector<DirectKey>, DirectEntry> m_BatchGraphs;
	// Host frames inside look-ahead passes (round 6; the AviSynth caller's frames, avisynth_plugin/src/main.cc:113-144, and
	// the only path the reference's own timer measures, scripts/inference/tensorrt/inference.py:245-251).  Frame by frame
	// the 8.3 MB of an output cross the PCIe link while the GPU idles -- the frame's rows all appear in its last
	// microseconds and the next frame needs this one's state.  Inside a pass that is no longer so: every input of the
	// pass is uploaded up front into the pass's own device buffers (m_PassIn), frame i's kernels write m_PassOut[i], a
	// one-thread kernel behind frame i's last launch counts it done in host-mapped memory (m_PassSignal), and the thread
	// blocked in processBatch copies frame i out on a second stream (SDMA, no CU) while frame i + 1 runs: only the last
	// frame's copy is exposed.  The copies go from / to the caller's pageable rows through the HIP runtime as in
	// stageIn / stageOut (cuda_convert.cc.cu:360-459); nothing of the caller's is page-locked (section 7 of DESIGN.md).
	struct PassFrame {
		bool hostIn = false, hostOut = false;
	};
	PassFrame m_BatchHost[kFlowBatchMax];
	DeviceBuffer m_PassIn[kFlowBatchMax], m_PassOut[kFlowBatchMax];
	std::unique_ptr<Stream> m_CopyStream;
	PinnedWords m_PassSignal;
	unsigned m_PassSignalBase = 0;
	std::uint64_t m_BatchHostFrames = 0;
	bool passEligible(const Frame &in, const Frame &out) const;
	void uploadPassInputs(const Frame *in, int n);
	void drainPassOutputs(const Frame *out, int n);
	static constexpr std::size_t kMaxBatchGraphs = 64;          // unregistered tuples (LRU)
	static constexpr std::size_t kMaxRegisteredBatches = 256;   // tuples registered through prepareBatch
	std::uint64_t m_BatchFrames = 0;
	bool batchPlanned(int items);
	void submitBatch(const Frame *in, const Frame *out, int n);
	void runBatch(int set, int n, const std::function<void(const Step &, bool)> *around = nullptr);
	void dropBatchGraphs();
	std::vector<DirectKey> bindBatch(const Frame *in, const Frame *out, int n, int set);
	DirectEntry &batchEntry(const std::vector<DirectKey> &key);
	std::uint64_t m_DirectClock = 0;
	bool m_DirectGraph = true;  // JU_DIRECT_GRAPH=0: device frames always launch eagerly
	static constexpr std::size_t kMaxDirectGraphs = 64;     // unregistered tuples (LRU)
	static constexpr std::size_t kMaxRegisteredPairs = 256;
	// Host frames (JU_LOC_CPU: the AviSynth caller, avisynth_plugin/src/main.cc:113-144) are copied from / to pageable
	// memory through the HIP runtime's own bounce buffers.  Page-locking recycled caller buffers (hipHostRegister at a
	// buffer's second sighting, JU_PIN_HOST=1) was built in round 5, measured at +2.3 % on that PCIe-bound path
	// (profiles/r05_host_path_ab.txt) and REMOVED: a registration on malloc'ed memory outlives its buffer in the HIP
	// runtime's view of those addresses -- later pageable copies from a recycled range were done as direct GPU reads of
	// an unmapped host page ("Memory access fault by GPU ... on address <host heap>", one run in two of the full GPU
	// suite; before that, hipErrorInvalidValue from torch's own hipMemcpy).  DESIGN.md section 7.
	// introspection ("graph_replays" / "eager_runs" / "graph_captures": captures of device-frame
	// graphs made inside process / enqueue, i.e. not by prepareFrames or the constructor)
	std::uint64_t m_GraphReplays = 0, m_EagerRuns = 0, m_InlineCaptures = 0, m_PreparedCaptures = 0;

public:
	std::uint64_t fallbacks() const { return m_Fallbacks; }
	// counters of how the per-frame program ran so far (tests, bench)
	std::uint64_t graphReplays() const { return m_GraphReplays; }
	std::uint64_t eagerRuns() const { return m_EagerRuns; }
};

}  // namespace ju
