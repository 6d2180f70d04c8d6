// Host-side launch interface of the gfx950 kernels ({conv,tower,frame}_kernels.hip).
//
// Every launcher enqueues on the given stream and returns immediately; none of
// them allocates, synchronises or touches the host heap, so the whole per-frame
// sequence can be captured into a hipGraph (engine.cpp), mirroring the
// reference's one captured graph per binding set
// (reference core/src/tensorrt_backend.cc:257-263).
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace ju {

enum DType : int { kF16 = 0, kBF16 = 1 };

inline std::size_t dtypeSize(DType) { return 2; }

// ---- implicit-GEMM convolution on MFMA -----------------------------------
// in   : NHWC [H][W][cin]   16-bit, cin a multiple of 16
// wgt  : kernel-ready weights from packConvWeights() (model.cpp)
// bias : f32 [cout] (BN-folded bias or the layer's own bias)
// res  : optional NHWC [H][W][cout] 16-bit tensor added before the activation
// out  : NHWC [H][W][cout] in the compute type, or f16 (outHead: the flow head)
struct ConvParams {
	const void *in;
	const void *wgt;
	const float *bias;
	const void *res;
	void *out;
	int H, W;
	int cin, cout;
	int taps;    // 9 (3x3 "same") or 1 (1x1)
	int relu;    // activation: 0 none, 1 max(x, 0), 2 LeakyReLU (x < 0 ? slope * x : x)
	float slope; // negative slope for relu == 2
	int outHead; // store f16 whatever the compute type: the flow head (HR pixel offsets up to a few
	             // pixels: f16 resolves 2^-9 .. 2^-8 px there, bf16 would not)
	// 2x2 max-pool fused into the epilogue: out is [H/2][W/2][cout] (dense or pitched
	// in POOLED pixels).  Needs rw == 2, even H and W, no residual, 16-bit output.
	int pool;
	// TF1 bilinear x2 upsampling fused into the tile staging: `in` is the
	// half-resolution tensor [H/2][W/2][cin] (inPitch in its pixels), H and W are the
	// layer's (full) resolution.  Needs 3x3, cin a multiple of 64, nb == 1, even H, W.
	int upsample;
	// multi-chunk register-prefetch path (set by launchConv): 2 = two LDS stages, 1 = one
	int stages;
	// Row pitches in pixels (0 = dense, i.e. W).  The pointers address image
	// pixel (0,0); a tensor kept in the zero-bordered tower layout (below) is
	// passed as its interior origin plus its pitch.
	int inPitch, outPitch, resPitch;
	// Tile shape chosen per layer (convTiling): couts per workgroup = 32*nb (the
	// weights must be packed for the same nb), rows per wave rw (tile = 4*rw rows).
	int nb, rw;
	// Frame look-ahead (Engine::processBatch): the same layer of `items` consecutive frames in one launch, item i at
	// in + i * inItemBytes / out + i * outItemBytes.  0 or 1 = one frame.  Honoured by launchConv (grid.z; no residual
	// operand then) and launchConvSplitK; the flow block launcher has its own item fields (FlowBlockLaunch).
	int items;
	long inItemBytes, outItemBytes;
};

// Zero-bordered activation layout of the generator trunk ("tower layout"):
// [towerRows(H)][towerPitch(W)][C] with the image at row/column offset 1.  The
// one-pixel border and the rows/columns beyond the image are zero and are never
// written, so the persistent tower kernel stages tiles with no bounds checks and
// gets the convolution's zero padding for free.
inline int towerPitch(int W) { return (W + 31) / 32 * 32 + 2; }
inline int towerRows(int H) { return (H + 7) / 8 * 8 + 2; }
inline std::size_t towerPixels(int H, int W) {
	return static_cast<std::size_t>(towerRows(H)) * towerPitch(W);
}
inline std::size_t towerOrigin(int W) { return static_cast<std::size_t>(towerPitch(W)) + 1; }

// Channel chunk shared by the launcher and the packer.
inline int convCK(int cin) { return cin % 64 == 0 ? 64 : (cin % 32 == 0 ? 32 : 16); }

// Tile shape for an H x W layer: prefer the shape with the most operand reuse
// (nb = 2 cout blocks, rw = 2 rows per wave) but fall back to smaller tiles until
// the launch has enough workgroups to occupy the chip; the coarse levels of the
// flow auto-encoder (34x60 ... 68x120 pixels) otherwise run on 40-80 of 256 CUs.
// >= 2 workgroups per CU: the generic kernel stages synchronously, so a second
// resident workgroup is what overlaps loads with MFMAs.
constexpr long kConvTargetWgs = 512;
inline void convTiling(int H, int W, int cout, int *nb, int *rw) {
	const int tilesX = (W + 31) / 32;
	int bestNb = 1, bestRw = 1;
	long bestWgs = -1;
	const int nbs[2] = {2, 1};
	const int rws[2] = {2, 1};
	for (int a = 0; a < 2; ++a) {
		if (cout % (32 * nbs[a]) != 0) continue;
		for (int b = 0; b < 2; ++b) {
			const long wgs = (long)tilesX * ((H + 4 * rws[b] - 1) / (4 * rws[b])) * (cout / (32 * nbs[a]));
			// the largest tile is kept as soon as it fills the chip once (least weight
			// restaging); smaller tiles must reach two workgroups per CU to pay
			if (wgs >= ((a == 0 && b == 0) ? 256 : kConvTargetWgs)) {
				*nb = nbs[a];
				*rw = rws[b];
				return;
			}
			if (wgs > bestWgs) {
				bestWgs = wgs;
				bestNb = nbs[a];
				bestRw = rws[b];
			}
		}
	}
	*nb = bestNb;
	*rw = bestRw;
}

void launchConv(DType dt, const ConvParams &p, hipStream_t stream);

// ---- a flow auto-encoder block as one launch (flow_kernels.hip) ---------------------
// conv A 3x3 cin -> cmid, activation, conv B 3x3 cmid -> cmid, [activation], [2x2
// max-pool]; `upsample`: `in` is the half-resolution tensor [H/2][W/2][cin] and the TF1
// bilinear x2 is part of the tile staging.  Weights: packConvWeights with nb = 1.
// Attribute-only launches: while a DryLaunchScope lives on the calling thread, launchFlowBlock and launchConvSplitK
// do everything a launch does on the host (shape checks, tile choice, the dynamic-LDS attribute of an instantiation
// used for the first time) but enqueue nothing -- Engine::prepareBatch runs a look-ahead pass's flow launches this
// way before it CAPTURES them (hipFuncSetAttribute is not something to do inside a stream capture).
bool launchesAreDry();
struct DryLaunchScope {
	DryLaunchScope();
	~DryLaunchScope();
	DryLaunchScope(const DryLaunchScope &) = delete;
	DryLaunchScope &operator=(const DryLaunchScope &) = delete;
};

constexpr int kFlowBatchMax = 8;  // frames per look-ahead pass of the flow net (Engine::processBatch)
struct FlowBlockLaunch {
	const void *in;
	const void *w1;
	const float *b1;
	const void *w2;
	const float *b2;
	void *out;      // [H][W][cmid], pooled [H/2][W/2][cmid]; f16 when outHead
	int H, W;       // the block's resolution (after the upsampling)
	int inPitch, outPitch;  // row pitches in pixels of the tensors as stored (0 = dense)
	int cin, cmid;  // cin padded to 16
	bool upsample, pool, outHead;
	// res_block (models.py:193-254): out = act(conv B(act(conv A(x))) + x), cin == cmid;
	// in / out may be tower-layout tensors (interior origin + pitch)
	bool residual;
	int act1, act2;  // ConvParams::relu codes
	float slope;
	// The flow net's first block (cin 16, cmid 32, pool) can build its input itself -- the work
	// of launchPackFrames, one launch less: packOut != nullptr.  `in` is then unused; the block
	// reads the u8 frame and the previous packed tensor, and writes the new packed tensor
	// [H][W][16] (H, W = the padded size) as a side effect.
	const std::uint8_t *packFrame;
	std::ptrdiff_t packFrameStride;
	const void *packPrev;
	void *packOut;
	int frameH, frameW, padTop, padLeft, numInputs;
	const unsigned *sums;
	// Frame look-ahead: the block of `items` (<= kFlowBatchMax) consecutive frames in one launch (grid.z), item i at
	// in + i * inItemBytes / out + i * outItemBytes; 0 or 1 = one frame.  With input packing, item i's current frame
	// is packFrames[i] and its history slot k is packFrames[i - k] where the launch holds that frame, else what
	// packPrev holds for it; only the LAST item writes packOut (the history of the frame after the batch), and
	// `sums` must be null.  packFrame / packFrameStride are ignored when items > 1.
	int items;
	long inItemBytes, outItemBytes;
	const std::uint8_t *packFrames[kFlowBatchMax];
	std::ptrdiff_t packFrameStrides[kFlowBatchMax];
};
// H x W: the block's resolution (the 128-filter blocks run as one launch only where their 2-row tiles are ONE round of the
// chip: flow_kernels.hip); 0 = shape only
bool flowBlockSupported(int cin, int cmid, bool upsample, bool pool, bool outHead, int H = 0, int W = 0);
void launchFlowBlock(DType dt, const FlowBlockLaunch &q, hipStream_t stream);

// One 3x3 convolution of the flow net's coarse levels (cin 128 / 256, a few thousand pixels), the
// input channels split over the eight waves of a workgroup (splitk_kernels.hip, conv_splitk_kernel).
// Weights: packConvWeights with nb = 1.  `zeros`: >= 16 zero bytes of device memory.
bool convSplitKSupported(const ConvParams &p);
void launchConvSplitK(DType dt, const ConvParams &p, const void *zeros, hipStream_t stream);

// Persistent 3x3 64->64 kernel of the generator tower.  in/res/out must be
// tower-layout tensors addressed at their interior origin with
// pitch == towerPitch(W); other shapes fall back to launchConv.
void launchConvTower(DType dt, const ConvParams &p, hipStream_t stream);

// ---- 8-bit (e4m3) residual-block convolution (fp8_kernels.hip, scheme: fp8.h) ---
// All tensors are tower-layout ALLOCATION STARTS (row -1, column -1 of the image).
struct Fp8TowerParams {
	const void *in8;      // e4m3 input, 64 B per pixel
	const void *weights;  // packFp8TowerWeights().w on the device
	const int *scaleA;    // [64] E8M0 codes per output channel
	const float *bias;    // [64]
	void *stream;         // second conv of a block: the 16-bit residual stream, updated in place
	                      // (out = relu(conv + stream)); nullptr: first conv (out = relu(conv))
	void *out8;           // e4m3 copy of the output (the next convolution's input)
	int inExp;            // the input tensor holds e4m3(x * 2^inExp)
	int outExp;           // the output copy holds e4m3(y * 2^outExp)
	int H, W;
	int leaky;            // `activation: lrelu` models: LeakyReLU(slope) instead of ReLU
	float slope;
};
void launchConvTowerFp8(DType dt, const Fp8TowerParams &p, hipStream_t stream);
// One residual block of the 8-bit tower per launch (the intermediate e4m3 tensor stays in
// LDS): in8 -> conv A -> ReLU -> e4m3(2^midExp) -> conv B -> + stream -> ReLU -> stream
// (in place) and out8 = e4m3(stream * 2^outExp).  in8 / out8 must be different tensors.
struct Fp8BlockLaunch {
	const void *in8;
	void *out8;
	void *stream;
	const void *w1, *w2;      // packFp8TowerWeights().w of the two convolutions
	const int *scaleA1, *scaleA2;
	const float *b1, *b2;
	int inExp, midExp, outExp;
	int H, W;
	int leaky;            // `activation: lrelu` models
	float slope;
};
void launchResBlockFp8(DType dt, const Fp8BlockLaunch &q, hipStream_t stream);
// e4m3(max(x, 0) * 2^exponent) of a 16-bit tower tensor (allocation starts); leaky: the tensor
// is a LeakyReLU output, e4m3(clamp(x * 2^exponent, +-448))
void launchQuantizeTower(DType dt, const void *in, void *out8, int H, int W, int exponent, bool leaky,
    hipStream_t stream);

// ---- resident tower: every residual-block convolution in one launch --------
// in: first layer's input addressed at image pixel (0,0) with row pitch inPitch
// (0 = dense W); out: tower-layout tensor addressed at its interior origin.
// weights: nLayers consecutive 64->64 3x3 kernels in packConvWeights (nb = 2)
// order, bias nLayers x 64.  hasHead: layer 0 is a plain conv+ReLU (generator
// conv_1), residual blocks (conv, conv + skip) follow.
// mailbox (residentMailboxBytes) carries the halo exchange between neighbouring
// workgroups as self-validating 16-byte slots; counters (residentCounterBytes) holds
// each region's publish counts, from which the slot epochs follow -- both zeroed
// together, once, and again after a failed launch; *error is written (non-zero) if a
// bounded wait expires.  Needs GX*GY co-resident workgroups.
struct ResidentTowerParams {
	const void *in;
	int inPitch;
	int hasHead;
	void *out;
	const void *weights;
	const float *bias;
	void *mailbox;
	unsigned *counters;
	unsigned *error;
	void *debug;  // optional: GX*GY*4*8 u64 cycle sums (diagnostic variant 4 only)
	int H, W;
	int GX, GY, RH;
	int nLayers;
	// `activation: lrelu` models (models.py:24-27): LeakyReLU(slope) instead of ReLU after every
	// layer; the mailbox must then be residentMailboxBytes(GX, GY, true) (twice the slots: a
	// LeakyReLU output has no free sign bit for the slot epoch)
	int leaky;
	float slope;
	// Optional fused generator tail (tailW1 != nullptr): instead of writing the last
	// layer to `out`, every workgroup runs the tail (launchTailFused's arithmetic) on
	// its LDS-resident region and writes the HR state and the BGRX frame directly.
	const void *tailW1;       // convT1 as 1x1 conv 64->128, packConvWeights order with nb = 2
	const float *tailB1;      // [128]
	const void *tailW2;       // packTailWeights fragments
	const float *tailB2;      // [3]
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	void *state;              // f16 [4H][4W][4]
	std::uint8_t *outU8;
	std::ptrdiff_t outStride;
	const unsigned *sums;     // normalize_brightness channel sums or nullptr
};
bool residentTowerGeometry(int H, int W, int numCUs, int *GX, int *GY, int *RH);
std::size_t residentMailboxBytes(int GX, int GY, bool leaky = false);
// every region of the frame has the shape the fast schedule of the resident tower is built for
bool residentTowerFastGeometry(int H, int W, int GX, int GY, int RH);
void setResBlockPlain(int on);       // tests / JU_RES_BLOCK=plain: res_block_kernel instead of res_block_pipe_kernel
bool resBlockPlain();
void setResidentTowerFast(int on);  // tests / JU_TOWER_FAST=0: the general schedule everywhere
void setFp8BlockForm(int form);     // tests / JU_FP8_BLOCK: res_block_fp8_kernel as 0 = by geometry, 1 = solo, 2 = duo
bool residentTowerFast();
inline std::size_t residentCounterBytes(int GX, int GY) {
	return (static_cast<std::size_t>(GX) * GY * 2 * sizeof(unsigned) + 63) / 64 * 64;
}
void launchResidentTower(DType dt, const ResidentTowerParams &p, hipStream_t stream);

// Timing-only ablation switch of the tower kernel (0 = product kernel).
// the resident tower's K loops run v_mfma_f32_16x16x32 (weights packed by packTowerWeightsM16, model.cpp) or 32x32x16
bool residentTowerM16();
void setTowerVariant(int variant);
int towerVariant();
// Test hook: launch the resident tower `n` workgroups short, so that the bounded
// neighbour waits expire (exercises the engine's fallback to the per-layer path).
void setResidentFault(int n);
int residentFaultForTests();

// ---- resident 8-bit tower (tower8_kernels.hip): every residual block of the e4m3 tower in
// one launch; the 16-bit stream and the two e4m3 tiles stay in LDS.  in / out: 16-bit tower
// tensors (allocation starts): generator conv_1's output in, the last block's stream out.
// Per convolution i of the 2 B: weights (packFp8TowerWeights, 36864 B each), scaleA[i][64],
// bias[i][64], scaleB[i] (E8M0 code of its input tensor's scale), outMul[i] (2^e of the
// e4m3 tensor its output feeds); outMul[2 B] = the scale of the tower's input tensor.
struct ResidentTower8Params {
	const void *in;
	void *out;
	const void *weights;
	const int *scaleA;
	const float *bias;
	const int *scaleB;
	const float *outMul;
	void *mailbox;       // residentMailboxBytes8, zeroed with the counters
	unsigned *counters;  // residentCounterBytes
	unsigned *error;
	int H, W;
	int GX, GY, RH;
	int nLayers;
	int leaky;           // `activation: lrelu` models; the mailbox is then residentMailboxBytes8(GX, GY, true)
	float slope;
	void *debug;         // developer builds (-DJU_T8_PROF, tools/tower8_phases.py): [regions][4 waves][8] cycle sums; unused otherwise
};
std::size_t residentMailboxBytes8(int GX, int GY, bool leaky = false);
void launchResidentTower8(DType dt, const ResidentTower8Params &p, hipStream_t stream);

// ---- flow-net helpers -------------------------------------------------------
// cur frame (u8 BGRX, signed row stride) + previous packed history ->
// packed [PH][PW][16]: ch 0-2 current frame (x/255-0.5, zero in the pad border),
// ch 3-11 = previous ch 0-8, ch 12-15 zero.
void launchPackFrames(DType dt, const std::uint8_t *frame, std::ptrdiff_t frameStride,
    const void *prevPacked, void *curPacked, int H, int W, int PH, int PW, int padTop,
    int padLeft, int numInputs, const unsigned *sums, hipStream_t stream);

// normalize_brightness (reference models.py:772-779): exact integer sums of the B, G, R
// bytes of the frame -> three 64-bit sums as (low, high) word pairs sums[0..5]; the kernels taking
// `sums` derive the scalar brightness from them (brightnessOf; nullptr = feature off).
void launchFrameSums(const std::uint8_t *frame, std::ptrdiff_t frameStride, int H, int W,
    unsigned *sums, hipStream_t stream);

// [nPix][16] 16-bit -> channels 0..15 of [nPix][64] records (the rest is not written).
void launchExpandChannels(DType dt, const void *in16, void *out64, int nPix, hipStream_t stream);
void launchMaxPool2(DType dt, const void *in, void *out, int H, int W, int C,
    hipStream_t stream);  // in [H][W][C] -> out [H/2][W/2][C]

// TF1 asymmetric bilinear, in [H][W][C] -> out [2H][2W][C]; items > 1: the dense tensors of a look-ahead launch's
// frames, one after the other
void launchUpsample2(DType dt, const void *in, void *out, int H, int W, int C, hipStream_t stream, int items = 1);

// ---- warp + space-to-depth + concat ----------------------------------------
// state : previous HR output, f16 [4H][4W][4] (B,G,R,0)
// flow  : f16 [PH][PW][32], channel (i*4+j)*2 + {dy,dx} (depth-to-space is free)
// frame : current LR frame, u8 BGRX
// out   : generator input NHWC [H][W][64] in the packed channel order
//         ch = i*16 + j*3 + c for the warped HR pixel (4h+i, 4w+j, c),
//         ch 12,13,14 = current LR frame B,G,R, other spare slots zero.
// preWarpOut (may be null): the warped previous output itself, f16 [4H][4W][4], for
// the temporal filter below.
void launchWarpPack(DType dt, const void *state, const void *flow,
    const std::uint8_t *frame, std::ptrdiff_t frameStride, void *out, int outPitch, int H, int W, int PW,
    int padTop, int padLeft, const unsigned *sums, void *preWarpOut, hipStream_t stream);  // out at image pixel (0, 0), outPitch in pixels (0: W)

// Temporal moving-average output filter with the scene-cut gate of
// scripts/inference/onnx/frame_moving_avg.py:146-302, every mode of that script: global or
// windowed gate (window in HR pixels), sign or tanh(gain * .) gate, L1 / L2 norm, limited
// pre_warp, luma weighting.  state: the new HR state written by the tail (f16 [4H][4W][4],
// generator output minus brightness), rewritten in place together with the u8 frame;
// acc: temporalAccWords(...) 64-bit words of scratch (one 32.32 fixed-point sum per window).
struct TemporalParams {
	float strength, threshold, gain;
	int window;   // 0 = one global mean
	int l2, limit, luma;
};
std::size_t temporalAccWords(int H, int W, int window);
void launchTemporalFilter(void *state, const void *preWarp, std::uint8_t *outU8,
    std::ptrdiff_t outStride, int H, int W, const unsigned *sums, unsigned long long *acc,
    const TemporalParams &tp, hipStream_t stream);

// ---- generator tail ---------------------------------------------------------
// y     : [H][W][128] 16-bit = relu(BN(convT1)) with channel (a*2+b)*32 + o
// w2    : f32 [2][2][3][32] (convT2 kernel, keras layout), b2 f32 [3]
// frame : current LR frame u8 BGRX (bilinear x4 skip)
// stateOut : f16 [4H][4W][4]; outU8 : BGRX [4H][4W][4], X = 0
void launchTail(DType dt, const void *y, const float *w2, const float *b2,
    const std::uint8_t *frame, std::ptrdiff_t frameStride, void *stateOut,
    std::uint8_t *outU8, std::ptrdiff_t outStride, int H, int W, const unsigned *sums,
    hipStream_t stream);  // (the activation of convT1 is applied by the conv launch that makes y)

// Fused tail: both transposed convs on the matrix cores + tanh + skip + clip + pack.
// x: trunk addressed at pixel (0,0) with pitch xPitch (0 = dense); w1/b1: convT1 as a
// 1x1 conv 64->128 (packConvWeights, nb = 2) and its folded bias; w2: convT2 A fragments
// from packTailWeights(); b2 [3]; needs 64 trunk channels and 32 mid channels.
struct TailFusedLaunch {
	const void *x;
	int xPitch;
	const void *w1;
	const float *b1;
	const void *w2;
	const float *b2;
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	void *state;
	std::uint8_t *outU8;
	std::ptrdiff_t outStride;
	const unsigned *sums;
	int H, W;
	float slope;  // activation after convT1: < 0 = ReLU, else LeakyReLU with this negative slope
};
void launchTailFused(DType dt, const TailFusedLaunch &p, hipStream_t stream);

// ---- staging ----------------------------------------------------------------
// Row-wise device copy with signed strides (bottom-up frames).
void launchCopyRows(const std::uint8_t *src, std::ptrdiff_t srcStride, std::uint8_t *dst,
    std::ptrdiff_t dstStride, std::size_t rowBytes, std::size_t rows, hipStream_t stream);

// *word += 1 (system scope) once everything enqueued before it on `stream` has completed: `word` is the device address of
// host-mapped memory (PinnedWords) that the host polls.  Host frames inside look-ahead passes (Engine::processBatch).
void launchSignalHost(unsigned *word, hipStream_t stream);

// max |x| over n 16-bit elements, atomically folded into *out as the bit pattern of a
// non-negative float (the caller zeroes it per frame).  Calibration mode only.
void launchAbsMax(DType dt, const void *in, std::size_t n, unsigned *out, hipStream_t stream);

// 16-bit tensor -> f32 (debug read-back).
void launchToFloat(DType dt, const void *in, float *out, std::size_t n, hipStream_t stream);

}  // namespace ju
