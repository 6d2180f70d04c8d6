// gfx950 (CDNA4, MI355X): the flow auto-encoder's blocks as ONE launch each
// (reference scripts/training/models.py:334-481: conv-BN-act, conv-BN-act, then
// MaxPool2D / UpscaleLayer; and the head pair flow/conv_1 + flow/conv_2).
//
//  * flow_block_kernel   conv A (3x3, CIN -> CMID) -> activation -> conv B (3x3, CMID ->
//                        CMID) -> [activation] -> [2x2 max-pool] in one workgroup per
//                        30-pixel-wide tile: the intermediate tensor lives in LDS only
//                        (one-pixel recompute ring), optional TF1 bilinear x2 of the
//                        INPUT folded into the tile staging.
//
// Why: as separate launches these layers are latency chains, not arithmetic (5.9-16 us
// per launch for 0.9-4.8 GFLOP, 3-7 % MFMA busy): both MFMA
// operands came from LDS (1-1.5 KB of LDS reads per MFMA), tiles were staged through
// registers with two barriers per channel chunk, and every launch paid ~2-3 us of
// boundary.  Here
//   - weights are the MFMA A operand straight from registers (one cout block x one
//     64-channel chunk = 36 fragments = 144 VGPRs per wave, loaded once per wave with
//     plain 16-byte global loads from L2: no LDS traffic for weights at all),
//   - activations are staged once per tile with LDS-DMA (global_load_lds, no VGPR
//     round trip), pixel-major with the 16-byte chunk index XOR-swizzled by the column,
//     and feed the B operand at 0.67 ds_read_b128 per MFMA (4 input-row fragments per
//     (dx, k-step) macro-step serve 6 MFMAs of a row pair), hand-issued one macro-step
//     ahead with counted lgkmcnt waits (the scheme of tower_resident_kernel),
//   - outputs leave through an LDS transpose as whole 64-byte / 128-byte half records,
//     16 B per lane,
//   - the pair's intermediate activation never touches HBM, and a block is one kernel
//     boundary instead of two or three.
// The arithmetic per output element is that of conv_mfma_kernel (fp32 accumulation from
// the bias, activation in f32, one rounding to the 16-bit type); only the fp32 summation
// order differs.  JU_FLOW_CONV=generic keeps the per-layer kernels (tests compare both).
#include "flow_block_common.h"

namespace ju {

namespace {

struct FlowBlockParams {
	const void *in;   // NHWC [H][W][CIN]; UPS: the half-resolution tensor [H/2][W/2][CIN]
	const void *w1;   // packConvWeights(nb = 1): [CMID/32][CIN/CK][9][CK/16][2][32][8]
	const float *b1;  // [CMID]
	const void *w2;   // [CMID/32][1][9][CMID/16][2][32][8]
	const float *b2;  // [CMID]
	void *out;        // [H][W][CMID] (POOL: [H/2][W/2][CMID]); f16 when OUTK == 1
	int H, W;         // the block's resolution (the upsampled one with UPS)
	int inPitch, outPitch;  // row pitches in pixels (of the tensor as stored: half resolution for a
	                        // UPS input / a pooled output)
	int act1, act2;   // 0 none, 1 ReLU, 2 LeakyReLU(slope)
	float slope;
	int skip;         // timing ablation (JU_FB_SKIP, developer only): 1 staging/expansion, 2 conv A, 4 conv B, 8 stores
	int prio;         // wave priority scheme (kernel_common.h applyWavePriority)
	// PACK instantiation (the flow net's first block): the 16-channel input records are built
	// here from the u8 frame and the previous packed tensor (launchPackFrames' arithmetic) and
	// written to `packOut` by the tile that owns the pixel; `in` is unused
	const void *packPrev;
	void *packOut;
	int frameH, frameW, padTop, padLeft, numInputs;
	const unsigned *sums;
	// frame look-ahead: blockIdx.z = the frame of this launch (FlowBlockLaunch::items); item i's tensors at
	// in + i * inItem / out + i * outItem bytes, its frame (PACK) frames[i]
	long inItem, outItem;
	const std::uint8_t *frames[kFlowBatchMax];
	std::ptrdiff_t frameStrides[kFlowBatchMax];
};

template <int CIN, int CMID, int TH, bool UPS, bool POOL, int OUTK, int NW = 4>
struct FbGeom {
	static constexpr int CK1 = CIN >= 64 ? 64 : CIN;   // channels per staged plane
	static constexpr int NPL = CIN / CK1;               // planes (64-channel chunks) of the input
	static constexpr int PBX = CK1 * 2;                 // bytes per pixel per plane
	static constexpr int KS1 = CK1 / 16;
	static constexpr int XR = TH + 4;                   // input rows
	static constexpr int TR = TH + 2;                   // intermediate rows
	static constexpr int XPLANE = XR * kFbW * PBX;
	static constexpr int PBT = CMID * 2;
	static constexpr int KS2 = CMID / 16;
	static constexpr int NCB = CMID / 32;               // cout blocks of both convs
	static constexpr int TBYTES = TR * kFbW * PBT;
	// low-resolution patch under the tile (UPS): rows (y0-2)/2 .. +XR/2, one more for the
	// lower / right bilinear neighbour
	static constexpr int LR = XR / 2 + 1, LC = kFbW / 2 + 1;
	static constexpr int LBYTES = UPS ? NPL * LR * LC * PBX : 0;
	// XPAIR (the decoder block with four input planes, tiles taller than two rows): the up-sampled input lives in LDS
	// two planes at a time -- plane p + 1 is expanded from the patch while conv A runs plane p -- so that 4- and 6-row
	// tiles fit (all four planes: 139 KB of X at 4 rows).  A 2-row tile computes four conv A rows for two (the look-ahead
	// launches, which are several rounds of tiles whatever the height, were spending half their time there:
	// profiles/r05_fb_ablate_pass.txt); a 6-row tile eight for six.
	static constexpr bool XPAIR = UPS && NPL > 2 && TH > 2;
	static constexpr int NPLX = XPAIR ? 2 : NPL;       // planes of X resident at once
	static constexpr int OFF_X = 0;
	static constexpr int OFF_T = NPLX * XPLANE;
	static constexpr int TREGION = TBYTES > LBYTES ? TBYTES : LBYTES;  // the patch is dead before T is written
	// output staging per wave: 32 couts of a row pair (or of its pooled row)
	static constexpr int ESZ = 2;                       // (OUTK 1: f16 instead of T, same size)
	static constexpr int RBW = 32 * ESZ;                // bytes per pixel per cout block
	static constexpr int STAGE_PX = POOL ? 16 : 64;
	static constexpr int STAGE_WAVE = STAGE_PX * RBW;
	static constexpr bool STAGE_IN_X = NPLX * XPLANE >= NW * STAGE_WAVE;  // X is dead once conv A is done
	static constexpr int OFF_STAGE = STAGE_IN_X ? OFF_X : OFF_T + TREGION;
	static constexpr int LDS = OFF_T + TREGION + (STAGE_IN_X ? 0 : NW * STAGE_WAVE);
	static_assert(CIN == 16 || CIN == 32 || CIN == 64 || CIN == 128 || CIN == 256, "input channels");
	// more than two 64-channel planes of conv A weights do not fit the registers at once: they are fetched plane by
	// plane, the next one behind the current plane's MFMAs (two sets of 36 fragments)
#ifndef JU_FB_RELOAD_MIN
#define JU_FB_RELOAD_MIN 3
#endif
	static constexpr bool RELOAD = NPL >= JU_FB_RELOAD_MIN;
	static_assert(CMID == 32 || CMID == 64 || CMID == 128, "block filters");
	static_assert(NW % NCB == 0, "cout blocks over the waves");
	static_assert(TH % 2 == 0 && TH >= 2, "row pairs");
	static constexpr bool FITS = LDS <= 160 * 1024;
};

// NW waves per workgroup: 8 (two per SIMD) wherever the kernel fits 256 registers -- the
// staging, expansion and epilogue phases are VALU work that one wave per SIMD issues at
// half rate, and a partner wave's epilogue runs beside the other's MFMAs.
template <typename T, int CIN, int CMID, int TH, bool UPS, bool POOL, int OUTK, int NW, bool PACK = false>
__global__ __launch_bounds__(NW * 64, NW / 4) void flow_block_kernel(FlowBlockParams p) {
	static_assert(!PACK || (CIN == 16 && !UPS), "PACK: the 16-channel flow input");
	using G = FbGeom<CIN, CMID, TH, UPS, POOL, OUTK, NW>;
	constexpr int NT = NW * 64;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	if constexpr (NW == 8) applyWavePriority(p.prio, wave, NW);
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int x0 = blockIdx.x * kFbOutW;  // first output column of the tile
	const int y0 = blockIdx.y * TH;       // first output row
	const int item = blockIdx.z;          // frame of a look-ahead launch (0 otherwise)
	const T *__restrict__ in = reinterpret_cast<const T *>(static_cast<const unsigned char *>(p.in) + item * p.inItem);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));

	// wave -> jobs: cout block cb (both convs have CMID couts), row pairs pstart, +pstep, ...
	constexpr int NCB = G::NCB;
	const int cb = wave % NCB;
	const int pstart = wave / NCB;
	constexpr int PSTEP = NW / NCB;
	const float s1 = fbActS(p.act1, p.slope), s2 = fbActS(p.act2, p.slope);

	// ---- conv A weights: A fragments of this wave's cout block, straight to registers ----
	// register sets of conv A fragments: all planes, or (RELOAD) two sets at one wave per SIMD / one set at two (its
	// SIMD partner computes while a wave waits for its next plane)
	// (XPAIR: one set -- the taller tile's accumulators need the registers, and the next plane's fragments, fetched behind
	// this plane's last MFMAs, travel while the plane after is expanded)
	constexpr int NWA = G::RELOAD ? ((NW == 4 && !G::XPAIR) ? 2 : 1) : G::NPL;
	Vec8<T> wa[NWA][9 * G::KS1];
	const unsigned char *waSrc = static_cast<const unsigned char *>(p.w1) +
	                             (size_t)cb * G::NPL * (9 * G::KS1 * 1024) + lane * 16;
	auto loadPlaneA = [&](int set, int pl) __attribute__((always_inline)) {
#pragma unroll
		for (int f = 0; f < 9 * G::KS1; ++f) {
			wa[set][f] = *reinterpret_cast<const Vec8<T> *>(waSrc + (size_t)(pl * 9 * G::KS1 + f) * 1024);
		}
	};
	// (Measured and dropped, round 5: issuing this fill BEHIND the tile's staging loads, the staging wait counting it
	// out and a raw barrier in place of __syncthreads() -- whose release fence waits vmcnt(0) -- so that it travels
	// while the staged patch is expanded.  The ISA showed the intended order; the launches took the same time,
	// 14.8 against 14.7-15.1 us for the head block: profiles/r05_fb_late_w_ab.txt.)
	if constexpr (G::RELOAD) {
		loadPlaneA(0, 0);
	} else {
#pragma unroll
		for (int pl = 0; pl < G::NPL; ++pl) loadPlaneA(pl, pl);
	}

	// UPS: planes [plFirst, plFirst + plCount) of the tile from the low-resolution patch (in LDS behind OFF_T) -- see the
	// staging branch below for the arithmetic
	[[maybe_unused]] constexpr int LPPX = G::PBX / 16;
	[[maybe_unused]] auto expandPlanes = [&](int plFirst, int plCount) __attribute__((always_inline)) {
		if constexpr (UPS) {
			constexpr int LPP = LPPX;
			constexpr int NPIX = G::LR * G::LC;
			const unsigned char *smL = smem + G::OFF_T;
			constexpr int BR = G::XR / 2, BC = kFbW / 2;  // 2x2 blocks of the tile
			const int nel = plCount * BR * BC * LPP;
			for (int e = tid; e < nel; e += NT) {
				const int c = e % LPP;
				const int bq = (e / LPP) % (BR * BC);
				const int pl = plFirst + e / (LPP * BR * BC);
				const int br = bq / BC, bc = bq - br * BC;
				const unsigned char *base = smL + pl * (NPIX * G::PBX) + (br * G::LC + bc) * G::PBX + c * 16;
				const Vec8<T> tl = *reinterpret_cast<const Vec8<T> *>(base);
				const Vec8<T> tr = *reinterpret_cast<const Vec8<T> *>(base + G::PBX);
				const Vec8<T> bl = *reinterpret_cast<const Vec8<T> *>(base + G::LC * G::PBX);
				const Vec8<T> brr = *reinterpret_cast<const Vec8<T> *>(base + (G::LC + 1) * G::PBX);
				Vec8<T> o01, o10, o11;
#pragma unroll
				for (int j = 0; j < 8; ++j) {
					const float a = static_cast<float>(tl[j]), b2 = static_cast<float>(tr[j]);
					const float d = static_cast<float>(bl[j]), e2 = static_cast<float>(brr[j]);
					const float top = a + (b2 - a) * 0.5f;
					const float bot = d + (e2 - d) * 0.5f;
					o01[j] = static_cast<T>(top);
					o10[j] = static_cast<T>(a + (d - a) * 0.5f);
					o11[j] = static_cast<T>(top + (bot - top) * 0.5f);
				}
				const int r = 2 * br, k = 2 * bc;
				const int gy = y0 - 2 + r, gx = x0 - 2 + k;
				const Vec8<T> zero = {static_cast<T>(0.f), static_cast<T>(0.f), static_cast<T>(0.f), static_cast<T>(0.f),
				    static_cast<T>(0.f), static_cast<T>(0.f), static_cast<T>(0.f), static_cast<T>(0.f)};
				// (H, W even: a 2x2 block is inside or outside the image as a whole)
				const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
				unsigned char *xb = smem + G::OFF_X + (G::XPAIR ? (pl & 1) : pl) * G::XPLANE + (r * kFbW + k) * G::PBX;
				const unsigned s0 = (static_cast<unsigned>(c) ^ fbSwz<G::PBX>(k)) << 4;
				const unsigned s1x = (static_cast<unsigned>(c) ^ fbSwz<G::PBX>(k + 1)) << 4;
				*reinterpret_cast<Vec8<T> *>(xb + s0) = inside ? tl : zero;
				*reinterpret_cast<Vec8<T> *>(xb + G::PBX + s1x) = inside ? o01 : zero;
				*reinterpret_cast<Vec8<T> *>(xb + kFbW * G::PBX + s0) = inside ? o10 : zero;
				*reinterpret_cast<Vec8<T> *>(xb + (kFbW + 1) * G::PBX + s1x) = inside ? o11 : zero;
			}
		}
	};
	// ---- stage the input tile ----
	// X pixel (r, k) = image (y0 - 2 + r, x0 - 2 + k); pixels outside the image are the
	// convolution's zero padding.
	const bool border = y0 - 2 < 0 || y0 + TH + 2 > p.H || x0 - 2 < 0 || x0 + 32 > p.W;
	constexpr int LPP = G::PBX / 16;  // lanes (16-byte chunks) per pixel
	if (JU_SKIP(p) & 1) {
	} else if constexpr (PACK) {
		// The flow input is born here (pack_frames_kernel's arithmetic, one launch less): ch 0-2 =
		// the current frame (x / 255 - 0.5 - brightness, exact 0 in the pad border), ch 3 .. = the
		// previous tensor's ch 0 .., the rest zero.  One thread per tile pixel; the tile that OWNS
		// a pixel (its 30 x TH output area) also writes the record to the new packed tensor, the
		// next frame's history.  Pixels outside the padded image are the convolution's zeros.
		// A look-ahead launch (item = blockIdx.z of gridDim.z frames) takes history slot k of item i from frame i - k of
		// the launch where there is one -- the same conversion of the same bytes the earlier frame's own pack did -- and
		// from the previous tensor's slot k - i - 1 otherwise; the last item alone writes the new history.
		const float bright = brightnessOf(p.sums, 1.0f / static_cast<float>(p.frameH * p.frameW));
		const T *__restrict__ prev = static_cast<const T *>(p.packPrev);
		T *__restrict__ cur = item + 1 == static_cast<int>(gridDim.z) ? static_cast<T *>(p.packOut) : nullptr;
		const int nch = 3 * p.numInputs;
		const int fromFrames = min(item + 1, p.numInputs);  // history slots 0 .. fromFrames - 1 come from frames
		constexpr int NPIX = G::XR * kFbW;
		for (int q = tid; q < NPIX; q += NT) {
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 2 + r, gx = x0 - 2 + k;
			const T zero = static_cast<T>(0.f);
			Vec8<T> o0 = {zero, zero, zero, zero, zero, zero, zero, zero}, o1 = o0;
			const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			if (inside) {
				const int y = gy - p.padTop, x = gx - p.padLeft;
				const bool inFrame = y >= 0 && y < p.frameH && x >= 0 && x < p.frameW;
				const size_t idx = (size_t)gy * p.W + gx;
				T pv[16], o[16];
#pragma unroll
				for (int j = 0; j < 16; ++j) o[j] = zero;
#pragma unroll
				for (int h = 0; h < 5; ++h) {  // (3 * numInputs <= 16)
					if (h < fromFrames) {
						float c0 = 0.f, c1 = 0.f, c2 = 0.f;
						if (inFrame) {
							const unsigned v = *reinterpret_cast<const unsigned *>(p.frames[item - h] + y * p.frameStrides[item - h] + x * 4);
							c0 = preprocessU8(v & 0xff) - bright;
							c1 = preprocessU8((v >> 8) & 0xff) - bright;
							c2 = preprocessU8((v >> 16) & 0xff) - bright;
						}
						o[3 * h] = static_cast<T>(c0);
						o[3 * h + 1] = static_cast<T>(c1);
						o[3 * h + 2] = static_cast<T>(c2);
					}
				}
				if (fromFrames < p.numInputs) {
					const Vec8<T> p0 = *reinterpret_cast<const Vec8<T> *>(prev + idx * 16);
					const Vec8<T> p1 = *reinterpret_cast<const Vec8<T> *>(prev + idx * 16 + 8);
#pragma unroll
					for (int i = 0; i < 8; ++i) {
						pv[i] = p0[i];
						pv[8 + i] = p1[i];
					}
					// o[j] = pv[j - 3 * fromFrames]: the shift is uniform over the launch's item, one unrolled copy each
					auto shifted = [&](auto sTag) __attribute__((always_inline)) {
						constexpr int S = decltype(sTag)::value;
#pragma unroll
						for (int j = S; j < 16; ++j) o[j] = (j < nch) ? pv[j - S] : zero;
					};
					if (fromFrames == 1) shifted(std::integral_constant<int, 3>{});
					else if (fromFrames == 2) shifted(std::integral_constant<int, 6>{});
					else if (fromFrames == 3) shifted(std::integral_constant<int, 9>{});
					else shifted(std::integral_constant<int, 12>{});
				}
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					o0[i] = o[i];
					o1[i] = o[8 + i];
				}
				if (cur != nullptr && r >= 2 && r < 2 + TH && k >= 2 && k < 2 + kFbOutW) {
					*reinterpret_cast<Vec8<T> *>(cur + idx * 16) = o0;
					*reinterpret_cast<Vec8<T> *>(cur + idx * 16 + 8) = o1;
				}
			}
			unsigned char *xb = smem + G::OFF_X + (r * kFbW + k) * G::PBX;
			const unsigned sw = fbSwz<G::PBX>(k);
			*reinterpret_cast<Vec8<T> *>(xb + ((0u ^ sw) << 4)) = o0;
			*reinterpret_cast<Vec8<T> *>(xb + ((1u ^ sw) << 4)) = o1;
		}
		__syncthreads();
	} else if constexpr (!UPS) {
		if (border) {  // interior tiles are overwritten completely
			for (int i = tid; i < G::NPL * G::XPLANE / 16; i += NT) {
				reinterpret_cast<uint4 *>(smem + G::OFF_X)[i] = make_uint4(0, 0, 0, 0);
			}
			__syncthreads();
		}
		constexpr int NPXI = 1024 / G::PBX;  // pixels per wave-instruction
		constexpr int NPIX = G::XR * kFbW;
		constexpr int NINSTR = (NPIX + NPXI - 1) / NPXI;
#pragma unroll
		for (int pl = 0; pl < G::NPL; ++pl) {
			for (int i = wave; i < NINSTR; i += NW) {
				const int q = i * NPXI + lane / LPP;
				const int r = q / kFbW, k = q - r * kFbW;
				const int gy = y0 - 2 + r, gx = x0 - 2 + k;
				const unsigned c = static_cast<unsigned>(lane % LPP) ^ fbSwz<G::PBX>(k);
				if (q < NPIX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
					fbGlds16(in + ((size_t)gy * p.inPitch + gx) * CIN + pl * 64 + c * 8,
					    smem + G::OFF_X + pl * G::XPLANE + i * 1024);
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
	} else {
		// low-resolution patch (coordinates clamped into the tensor), then the TF1 bilinear
		// x2 (keras_layers.py:46-52: src = dst / 2, edge clamp) with upsample2_kernel's
		// arithmetic, one parity class per pass
		unsigned char *smL = smem + G::OFF_T;
		const int lh = p.H >> 1, lw = p.W >> 1;
		const int ly0 = (y0 >> 1) - 1, lx0 = (x0 >> 1) - 1;
		constexpr int NPXI = 1024 / G::PBX;
		constexpr int NPIX = G::LR * G::LC;
		constexpr int NINSTR = (NPIX + NPXI - 1) / NPXI;
#pragma unroll
		for (int pl = 0; pl < G::NPL; ++pl) {
			for (int i = wave; i < NINSTR; i += NW) {
				const int q = min(i * NPXI + lane / LPP, NPIX - 1);
				const int r = q / G::LC, k = q - r * G::LC;
				const int cy = min(max(ly0 + r, 0), lh - 1), cx = min(max(lx0 + k, 0), lw - 1);
				if (i * NPXI + lane / LPP < NPIX) {
					fbGlds16(in + ((size_t)cy * p.inPitch + cx) * CIN + pl * 64 + (lane % LPP) * 8,
					    smL + pl * (NPIX * G::PBX) + i * 1024);
				}
			}
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
		// One thread per (low-resolution pixel, 16-byte chunk): its 2x2 block of the tile is
		// a copy, two 2-tap means and one 4-tap mean of (self, right, below, diagonal) --
		// upsample2_kernel's expressions with the zero-weight terms dropped (a + (b - a) * 0
		// == a exactly), so the result is bit-identical to the separate kernel.  The patch was
		// loaded with clamped coordinates, so "right" / "below" at the tensor's edge are the
		// edge pixel itself, which is what min(lo + 1, n - 1) selects.
		if constexpr (!G::XPAIR) {
			expandPlanes(0, G::NPL);
		} else {
			expandPlanes(0, 1);  // (the other planes: one ahead of conv A's plane loop, below)
		}
		__syncthreads();  // X complete; the patch (aliasing T) is dead
	}

	// ---- conv A: (TH + 2) rows x 32 columns -> T (LDS), zero outside the image ----
	unsigned colOffA[3], colSwzA[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOffA[dx] = (px + dx) * G::PBX;
		colSwzA[dx] = fbSwz<G::PBX>(px + dx);
	}
	{
		constexpr int NPAIR = G::TR / 2;
		constexpr int MAXP = (NPAIR + PSTEP - 1) / PSTEP;
		// several planes: every pair's accumulators live across the planes; one plane:
		// one pair at a time
		constexpr int HOLD = G::NPL > 1 ? MAXP : 1;
		f32x16 acc[HOLD][2];
		unsigned tOff[4];  // byte offset of this lane's 4 channels of group g inside a T row
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			tOff[g] = px * G::PBT + ((static_cast<unsigned>(cb * 4 + g) ^ fbSwz<G::PBT>(px)) << 4) + hh * 8;
		}
		// (the bias comes in with the weights, once -- not a memory round trip per row pair -- where
		// 16 more registers do not spill: not beside 144 registers of fragments at two waves per SIMD)
		constexpr bool kHoistBiasA = NWA * 9 * G::KS1 * 4 < 144 || NW == 4;
		f32x4 biasA[4];
		if constexpr (kHoistBiasA) {
#pragma unroll
			for (int g = 0; g < 4; ++g) biasA[g] = *reinterpret_cast<const f32x4 *>(p.b1 + cb * 32 + 8 * g + 4 * hh);
		}
		auto initAcc = [&](f32x16(&a)[2]) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				f32x4 bg;
				if constexpr (kHoistBiasA) bg = biasA[g];
				else bg = *reinterpret_cast<const f32x4 *>(p.b1 + cb * 32 + 8 * g + 4 * hh);
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) a[r][4 * g + i] = bg[i];
				}
			}
		};
		auto epilogueA = [&](const f32x16(&a)[2], int pair) {
			// T pixel (tr, px) = image (y0 - 1 + tr, x0 - 1 + px); chunk = cb * 4 + g
			const int gx = x0 - 1 + px;
			const bool colIn = gx >= 0 && gx < p.W;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int tr = 2 * pair + r;
				const int gy = y0 - 1 + tr;
				const float keep = (colIn && gy >= 0 && gy < p.H) ? 1.0f : 0.0f;  // zero padding of conv B
				unsigned char *row = smem + G::OFF_T + tr * (kFbW * G::PBT);
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					*reinterpret_cast<Vec4<T> *>(row + tOff[g]) =
					    pack4<T>(fbAct(a[r][4 * g + 0], s1) * keep, fbAct(a[r][4 * g + 1], s1) * keep,
					        fbAct(a[r][4 * g + 2], s1) * keep, fbAct(a[r][4 * g + 3], s1) * keep);
				}
			}
		};
		if (JU_SKIP(p) & 2) {
		} else if constexpr (G::NPL == 1) {
			for (int pair = pstart; pair < NPAIR; pair += PSTEP) {
				initAcc(acc[0]);
				FbPair<T, G::KS1, G::PBX>::run(ldsBase + G::OFF_X + (2 * pair) * (kFbW * G::PBX), colOffA, colSwzA,
				    hh, wa[0], acc[0]);
				epilogueA(acc[0], pair);
			}
		} else {
#pragma unroll
			for (int s = 0; s < HOLD; ++s) initAcc(acc[s]);
#pragma unroll
			for (int pl = 0; pl < G::NPL; ++pl) {
				// (RELOAD: the next plane's fragments travel behind this plane's MFMAs, into the other register set)
				if constexpr (G::RELOAD && NWA == 2) {
					if (pl + 1 < G::NPL) loadPlaneA((pl + 1) & 1, pl + 1);
				}
				if constexpr (G::XPAIR) {
					// the next plane of the input, into the buffer the plane before this one was read from (every wave is past
					// the barrier that ended that plane)
					if (pl + 1 < G::NPL && !(JU_SKIP(p) & 1)) expandPlanes(pl + 1, 1);
				}
#pragma unroll
				for (int s = 0; s < HOLD; ++s) {
					const int pair = pstart + s * PSTEP;
					if (pair < NPAIR) {
						FbPair<T, G::KS1, G::PBX>::run(
						    ldsBase + G::OFF_X + (G::XPAIR ? (pl & 1) : pl) * G::XPLANE + (2 * pair) * (kFbW * G::PBX), colOffA,
						    colSwzA, hh, wa[G::RELOAD ? (pl & (NWA - 1)) : pl], acc[s]);
					}
				}
				if constexpr (G::RELOAD && NWA == 1) {
					if (pl + 1 < G::NPL) loadPlaneA(0, pl + 1);  // (behind the plane's last MFMAs: the set is free)
				}
				if constexpr (G::XPAIR) __syncthreads();  // this plane's buffer is free, the next plane's complete
			}
#pragma unroll
			for (int s = 0; s < HOLD; ++s) {
				const int pair = pstart + s * PSTEP;
				if (pair < NPAIR) epilogueA(acc[s], pair);
			}
		}
	}

	// ---- conv B weights (registers of conv A's fragments are free now) and its bias: fetched in
	// front of the barrier, so that the wait for the slowest wave's conv A hides the latency (the
	// bias used to be loaded per row pair behind the barrier: a memory round trip in front of every
	// pair's first MFMA) ----
	Vec8<T> wb[9 * G::KS2];
	f32x4 biasB[4];
	{
		const unsigned char *wsrc = static_cast<const unsigned char *>(p.w2) + (size_t)cb * (9 * G::KS2 * 1024) + lane * 16;
#pragma unroll
		for (int f = 0; f < 9 * G::KS2; ++f) {
			// fragment f = (tap, k-step); the packed order is [64-channel plane][tap][4 k-steps] (packConvWeights): one
			// plane for CMID <= 64 (source index = f), two for the 128-filter block
			constexpr int KSP = G::KS2 > 4 ? 4 : G::KS2;
			const int tap = f / G::KS2, ks = f % G::KS2;
			wb[f] = *reinterpret_cast<const Vec8<T> *>(wsrc + (size_t)(((ks / KSP) * 9 + tap) * KSP + ks % KSP) * 1024);
		}
#pragma unroll
		for (int g = 0; g < 4; ++g) biasB[g] = *reinterpret_cast<const f32x4 *>(p.b2 + cb * 32 + 8 * g + 4 * hh);
	}
	__syncthreads();  // T complete, X dead

	// ---- conv B: TH rows x 32 columns (30 valid) -> [activation] -> [pool] -> global ----
	unsigned colOffB[3], colSwzB[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOffB[dx] = (px + dx) * G::PBT;
		colSwzB[dx] = fbSwz<G::PBT>(px + dx);
	}
	unsigned char *stage = smem + G::OFF_STAGE + wave * G::STAGE_WAVE;
	constexpr int NCH = G::RBW / 16;  // 16-byte chunks per staged pixel
	for (int pair = pstart; pair < ((JU_SKIP(p) & 4) ? 0 : TH / 2); pair += PSTEP) {
		f32x16 acc[2];
#pragma unroll
		for (int g = 0; g < 4; ++g) {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
#pragma unroll
				for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = biasB[g][i];
			}
		}
		FbPair<T, G::KS2, G::PBT>::run(ldsBase + G::OFF_T + (2 * pair) * (kFbW * G::PBT), colOffB, colSwzB, hh, wb, acc);
		// residual block (OUTK 2, models.py:248-253): the skip connection is the block's own
		// input at the output pixel; its 8-byte pieces are fetched behind the K loop
		// (the tile was just staged from the same lines: L2 hits; the SIMD's other wave computes
		// meanwhile -- fetched in front of the loop they cost 16 registers the loop has not got)
		Vec4<T> resv[2][4];
		if constexpr (OUTK == 2) {
			static_assert(CIN == CMID && !UPS && !POOL, "residual block: same width in and out");
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int gy = min(y0 + 2 * pair + r, p.H - 1), gx = min(x0 + px, p.W - 1);
				const T *rp = in + ((size_t)gy * p.inPitch + gx) * CIN + cb * 32 + 4 * hh;
#pragma unroll
				for (int g = 0; g < 4; ++g) resv[r][g] = *reinterpret_cast<const Vec4<T> *>(rp + 8 * g);
			}
		}
		if constexpr (POOL) {
			// rows y0 + 2 pair, + 1 (y0 even) and columns px, px ^ 1 (x0 even): vertical max in
			// the lane, horizontal with the neighbouring lane; the activation is monotonic
			// (loader-checked), so pooling the f32 values first is the reference's
			// act-then-pool; both lanes hold the pooled pixel, each stages half its channels
			const int pp = px >> 1;
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				float v[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					float m = fmaxf(acc[0][4 * g + i], acc[1][4 * g + i]);
					m = fmaxf(m, fbSwapPair(m));
					v[i] = fbAct(m, s2);
				}
				if ((g >> 1) == (px & 1)) {
					const unsigned c = static_cast<unsigned>(g) ^ (static_cast<unsigned>(pp) & (NCH - 1));
					*reinterpret_cast<Vec4<T> *>(stage + pp * G::RBW + (c << 4) + hh * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
				}
			}
		} else {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int pi = r * 32 + px;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					float v[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						float x = acc[r][4 * g + i];
						if constexpr (OUTK == 2) x += static_cast<float>(resv[r][g][i]);
						v[i] = fbAct(x, s2);
					}
					const unsigned c = static_cast<unsigned>(g) ^ (static_cast<unsigned>(pi) & (NCH - 1));
					if constexpr (OUTK == 1) {  // the flow head is f16 whatever the compute type
						*reinterpret_cast<Vec4<f16> *>(stage + pi * G::RBW + (c << 4) + hh * 8) = pack4<f16>(v[0], v[1], v[2], v[3]);
					} else {
						*reinterpret_cast<Vec4<T> *>(stage + pi * G::RBW + (c << 4) + hh * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
					}
				}
			}
		}
		// same-wave exchange through LDS: order the writes before the reads
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		{
			unsigned char *outp = static_cast<unsigned char *>(p.out) + item * p.outItem;
			constexpr int PXI = 64 / NCH;  // pixels per wave-instruction
#pragma unroll
			for (int it = 0; it < G::STAGE_PX / PXI; ++it) {
				const int pi = it * PXI + lane / NCH;
				const unsigned slot = static_cast<unsigned>(lane % NCH);
				const unsigned chunk = slot ^ (static_cast<unsigned>(pi) & (NCH - 1));
				const uint4 val = *reinterpret_cast<const uint4 *>(stage + pi * G::RBW + (slot << 4));
				if (JU_SKIP(p) & 8) {
				} else if constexpr (POOL) {
					const int oy = (y0 >> 1) + pair, ox = (x0 >> 1) + pi;
					if (pi < kFbOutW / 2 && oy < (p.H >> 1) && ox < (p.W >> 1)) {
						*reinterpret_cast<uint4 *>(outp + (((size_t)oy * p.outPitch + ox) * CMID + cb * 32) * G::ESZ + chunk * 16) = val;
					}
				} else {
					const int gy = y0 + 2 * pair + (pi >> 5), gx = x0 + (pi & 31);
					if ((pi & 31) < kFbOutW && gy < p.H && gx < p.W) {
						*reinterpret_cast<uint4 *>(outp + (((size_t)gy * p.outPitch + gx) * CMID + cb * 32) * G::ESZ + chunk * 16) = val;
					}
				}
			}
		}
		// the slice is reused by this wave's next pair
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

// 8 waves where the widest instantiation stays within 256 registers (two 64-channel
// planes of conv A weights do not, nor do the 72 fragments of a 128-channel conv B)
template <int CIN, int CMID = 32>
constexpr int fbWaves() {
#if JU_FB_RELOAD_MIN <= 2
	if (CIN == 128 && CMID <= 64) return 8;  // (A/B build: the 128-channel decoder block with plane-by-plane weights at two waves per SIMD)
#endif
	return (CIN > 64 || CMID > 64) ? 4 : 8;
}

template <typename T, int CIN, int CMID, int TH, bool UPS, bool POOL, int OUTK, bool PACK = false>
void launchFlowBlockInst(const FlowBlockParams &p, int items, hipStream_t stream) {
	constexpr int NW = fbWaves<CIN, CMID>();
	using G = FbGeom<CIN, CMID, TH, UPS, POOL, OUTK, NW>;
	static_assert(G::FITS, "tile does not fit LDS");
	auto kern = flow_block_kernel<T, CIN, CMID, TH, UPS, POOL, OUTK, NW, PACK>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), G::LDS, &ldsDone, "flow block");
	if (launchesAreDry()) return;
	dim3 grid((p.W + kFbOutW - 1) / kFbOutW, (p.H + TH - 1) / TH, items);
	hipLaunchKernelGGL(kern, grid, dim3(NW * 64), G::LDS, stream, p);
	hipCheckLaunch("flow_block");
}

// Tile height.  A tile costs its fixed part (weights, staging ramp, the recompute ring's two rows) plus its rows, and a
// launch costs as many of those in series as it has rounds over the CUs (one workgroup per CU by LDS size): the tall
// tile (18 rows: 11 % ring) wins where the launch still fills the chip, the short one (6 rows) on small tensors, and in
// between the height that makes the LAST round full -- 640 x 448: 22 x 25 tiles of 18 rows are three rounds on 256 CUs,
// 22 x 23 of 20 rows are two (measured: profiles/r04_flow_tile_heights.txt).  The bytes do not depend on the choice: an
// output pixel's terms are added in the same order whatever tile it falls into.  JU_FLOW_TILE=<rows> forces a height
// (developer switch, for that measurement).
constexpr int kFbMid = 14;
constexpr int kFbFixedRows = 8;  // the fixed part of a tile in units of one row's work (fitted to the measurement above)

inline long fbLaunchCost(int H, long tilesX, int TH, int numCUs, int items = 1) {
	const long tiles = tilesX * ((H + TH - 1) / TH) * items;  // (a look-ahead launch: the tiles of all its frames)
	return (tiles + numCUs - 1) / numCUs * (TH + kFbFixedRows);
}

inline int fbForcedTile() {
	static const int forced = [] { const char *e = devSwitch(Dev::FlowTile); return e ? std::atoi(e) : 0; }();
	return forced;
}

template <typename T, int CIN, int CMID, bool UPS, bool POOL, int OUTK, bool PACK, int... THS>
void launchFlowBlockBest(const FlowBlockParams &p, int items, int numCUs, hipStream_t stream) {
	const long tilesX = (p.W + kFbOutW - 1) / kFbOutW;
	int best = 0;
	long bestCost = 0;
	auto consider = [&](int TH, bool fits) {
		if (!fits) return;
		const long c = fbForcedTile() == TH ? -1 : fbLaunchCost(p.H, tilesX, TH, numCUs, items);
		if (best == 0 || c < bestCost) {
			best = TH;
			bestCost = c;
		}
	};
	(consider(THS, FbGeom<CIN, CMID, THS, UPS, POOL, OUTK, fbWaves<CIN, CMID>()>::FITS), ...);
	bool done = false;
	auto launch = [&](auto thTag) {
		constexpr int TH = decltype(thTag)::value;
		if constexpr (FbGeom<CIN, CMID, TH, UPS, POOL, OUTK, fbWaves<CIN, CMID>()>::FITS) {
			if (!done && best == TH) {
				launchFlowBlockInst<T, CIN, CMID, TH, UPS, POOL, OUTK, PACK>(p, items, stream);
				done = true;
			}
		}
	};
	(launch(std::integral_constant<int, THS>{}), ...);
	if (!done) throw std::logic_error("flow block: no tile height fits");
}

template <typename T, int CIN, int CMID, bool UPS, bool POOL, int OUTK, bool PACK = false>
void launchFlowBlockT(const FlowBlockParams &p, int items, int numCUs, hipStream_t stream) {
	if constexpr (OUTK == 2) {
		// (64 -> 64 -> 64 residual block, JU_RES_BLOCK=tile only: two 128-byte tiles; 14 rows is what fits)
		launchFlowBlockBest<T, CIN, CMID, UPS, POOL, OUTK, PACK, kFbMid, 6>(p, items, numCUs, stream);
	} else if constexpr (CMID == 128) {
		// (the 128-filter blocks, 68 x 120 at 480 x 270: a few thousand pixels -- short tiles, or most of the chip idles;
		// measured for the encoder / decoder block: 2 rows 11.7 / 23.8 us, 4 rows 16.0 / 32.1, 6 rows 19.9 / 39.1.  One
		// frame is a single round of 2-row tiles; a look-ahead launch of 8 frames is five such rounds or two of 6-row
		// tiles -- the cost rule above picks.  The decoder block's taller tiles keep two of its four input planes in LDS
		// at a time: FbGeom::XPAIR.)
		launchFlowBlockBest<T, CIN, CMID, UPS, POOL, OUTK, PACK, 6, 4, 2>(p, items, numCUs, stream);
	} else {
		launchFlowBlockBest<T, CIN, CMID, UPS, POOL, OUTK, PACK, 20, 18, 10, 6>(p, items, numCUs, stream);
	}
}

template <typename T>
void launchFlowBlockDT(const FlowBlockLaunch &q, hipStream_t stream) {
	FlowBlockParams p{};
	p.in = q.in;
	p.w1 = q.w1;
	p.b1 = q.b1;
	p.w2 = q.w2;
	p.b2 = q.b2;
	p.out = q.out;
	p.H = q.H;
	p.W = q.W;
	p.inPitch = q.inPitch ? q.inPitch : (q.upsample ? q.W / 2 : q.W);
	p.outPitch = q.outPitch ? q.outPitch : (q.pool ? q.W / 2 : q.W);
	p.act1 = q.act1;
	p.act2 = q.act2;
	p.slope = q.slope;
	p.skip = ablationSkipBits();
	p.prio = wavePriorityMode(0);
	p.packPrev = q.packPrev;
	p.packOut = q.packOut;
	p.frameH = q.frameH;
	p.frameW = q.frameW;
	p.padTop = q.padTop;
	p.padLeft = q.padLeft;
	p.numInputs = q.numInputs;
	p.sums = q.sums;
	const int items = q.items > 1 ? q.items : 1;
	if (items > kFlowBatchMax) throw std::invalid_argument("flow block: more look-ahead frames than kFlowBatchMax");
	p.inItem = items > 1 ? q.inItemBytes : 0;
	p.outItem = items > 1 ? q.outItemBytes : 0;
	for (int i = 0; i < kFlowBatchMax; ++i) {
		p.frames[i] = items > 1 ? (i < items ? q.packFrames[i] : nullptr) : (i == 0 ? q.packFrame : nullptr);
		p.frameStrides[i] = items > 1 ? (i < items ? q.packFrameStrides[i] : 0) : (i == 0 ? q.packFrameStride : 0);
	}
	if (items > 1 && q.residual) throw std::invalid_argument("flow block: the residual form has no look-ahead launch");
	const int cus = currentDeviceCUs();
	if (q.packOut != nullptr) {
		if (items > 1 && q.sums != nullptr) throw std::invalid_argument("flow block: look-ahead packing has no brightness sums");
		for (int i = 0; i < items; ++i) {
			if (p.frames[i] == nullptr) throw std::invalid_argument("flow block: input packing needs every frame of the launch");
		}
		if (!(q.cin == 16 && q.cmid == 32 && !q.upsample && q.pool && !q.outHead && !q.residual && q.packPrev)) {
			throw std::invalid_argument("flow block: input packing is built for the first block (16 -> 32 -> 32, pool)");
		}
		return launchFlowBlockT<T, 16, 32, false, true, 0, true>(p, items, cus, stream);
	}
	if (q.upsample && (q.H % 2 || q.W % 2)) throw std::invalid_argument("flow block: fused upsampling needs even H and W");
	if (q.pool && (q.H % 2 || q.W % 2)) throw std::invalid_argument("flow block: fused max-pool needs even H and W");
	// the shapes of the flow auto-encoder's fusable blocks (flowBlockSupported)
	const int outk = q.residual ? 2 : (q.outHead ? 1 : 0);
#define JU_FB_CASE(CIN_, CMID_, UPS_, POOL_, OUTK_)                                              \
	if (q.cin == CIN_ && q.cmid == CMID_ && q.upsample == UPS_ && q.pool == POOL_ && outk == OUTK_) { \
		return launchFlowBlockT<T, CIN_, CMID_, UPS_, POOL_, OUTK_>(p, items, cus, stream);         \
	}
	JU_FB_CASE(16, 32, false, true, 0)   // encoder block 1: 12(16) -> 32 -> 32, pool
	JU_FB_CASE(32, 64, false, true, 0)   // encoder block 2: 32 -> 64 -> 64, pool
	JU_FB_CASE(64, 128, false, true, 0)  // encoder block 3: 64 -> 128 -> 128, pool (round 5: four cout blocks on four waves)
	JU_FB_CASE(256, 128, true, false, 0)  // decoder block 5: up(256) -> 128 -> 128 (round 5: conv A's weights plane by plane)
	JU_FB_CASE(256, 128, false, false, 0)
	JU_FB_CASE(128, 64, true, false, 0)  // last decoder block: up(128) -> 64 -> 64
	JU_FB_CASE(128, 64, false, false, 0)
	JU_FB_CASE(64, 32, true, false, 1)   // head: up(64) -> 32 -> 32 (f16 flow head)
	JU_FB_CASE(64, 32, false, false, 1)
	JU_FB_CASE(64, 64, false, false, 2)  // res_block of a 64-filter tower (generator, flow-resnet)
#undef JU_FB_CASE
	throw std::invalid_argument("flow block: unsupported shape");
}

}  // namespace

bool flowBlockSupported(int cin, int cmid, bool upsample, bool pool, bool outHead, int H, int W) {
	if (cin == 64 && cmid == 64) return false;  // (only as a residual block: FlowBlockLaunch::residual)
	if (cin == 16 && cmid == 32) return !upsample && pool && !outHead;
	if (cin == 32 && cmid == 64) return !upsample && pool && !outHead;
	if (cmid == 128) {
		// Round 5.  These blocks are one launch where their tiles are ONE round of the chip at some tile height they have
		// (2, 4, 6 rows; the decoder block's taller tiles: FbGeom::XPAIR) -- 480 x 270: 4 x 34 = 136 two-row tiles at the
		// 68 x 120 level, 15.1 -> 11.7 us (block 3) and 24.1 -> 23.3 us with three launches fewer (block 5); 640 x 448:
		// 6 x 28 = 168 four-row tiles, 178.8 -> 169.3 us of flow net per frame (with 2-row tiles only -- 336, two rounds --
		// block 5 lost 9 us to the launches of their own there: profiles/r05_flow_layers_ps2.txt).  Larger frames keep the
		// launches per convolution.  JU_FLOW_WIDE=0 keeps them everywhere, 1 fuses the encoder block only, 3 fuses whatever
		// the tile count: A/B runs.
		const char *wideEnv = devSwitch(Dev::FlowWide);  // (read per call: only engine construction asks, and the tests switch it)
		const int wide = wideEnv ? std::atoi(wideEnv) : 2;
		const int rows = (cin == 256 && !upsample) ? 2 : 6;  // the tallest tile of the shape
		const bool oneRound = H <= 0 || W <= 0 || wide >= 3 ||
		                      static_cast<long>((W + kFbOutW - 1) / kFbOutW) * ((H + rows - 1) / rows) <= currentDeviceCUs();
		if (cin == 64) return wide >= 1 && oneRound && !upsample && pool && !outHead;
		if (cin == 256) return wide >= 2 && oneRound && !pool && !outHead;
		return false;
	}
	if (cin == 128 && cmid == 64) return !pool && !outHead;
	if (cin == 64 && cmid == 32) return !pool && outHead;
	return false;
}

void launchFlowBlock(DType dt, const FlowBlockLaunch &q, hipStream_t stream) {
	if (q.residual && q.cin == 64 && q.cmid == 64 && !q.upsample && !q.pool) {
		static const char *mode = devSwitch(Dev::ResBlock);  // "tile": the non-persistent flow_block_kernel form (A/B)
		if (!(mode && std::string(mode) == "tile")) {
			launchResBlockPersistent(dt, q, stream);  // res_block_kernels.hip
			return;
		}
	}
	if (dt == kF16) launchFlowBlockDT<f16>(q, stream);
	else launchFlowBlockDT<bf16>(q, stream);
}

}  // namespace ju
