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
#include "kernel_common.h"

namespace ju {

namespace {

constexpr int kFbW = 34;     // LDS tile width: 32 MFMA columns + 2
constexpr int kFbOutW = 30;  // final output columns per tile (conv B's 32 columns minus its ring)

__device__ __forceinline__ void fbGlds16(const void *g, void *l) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
	    (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// XOR swizzle of the 16-byte chunk index by the tile column, for a pixel record of PB
// bytes (P = PB / 16 chunks): the 16 lanes of a ds_read_b128 group read one logical chunk
// of 16 consecutive columns and must fall on 16 distinct 16-byte slots of a 256-byte
// bank row.
template <int PB>
__device__ __forceinline__ unsigned fbSwz(unsigned col) {
	if constexpr (PB == 128) return (col >> 1) & 7u;
	else if constexpr (PB == 64) return (col >> 2) & 3u;
	else return (col >> 3) & 1u;
}

// One row pair (2 output rows x 32 columns x 32 couts) over one staged channel chunk:
// 9 taps x KS k-steps; per macro-step (dx, ks) the 4 input-row fragments feed 6 MFMAs
// (dy = 0..2 x row 0..1).  rowAddr: LDS byte address of input row 0 of the pair, column 0.
template <typename T, int KS, int PB>
struct FbPair {
	static constexpr int RS = kFbW * PB;  // LDS row stride in bytes
	static constexpr int NMAC = 3 * KS;

	template <int J>
	static __device__ __forceinline__ void rd(Vec8<T> &dst, unsigned a) {
		asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "n"(RS * J));
	}
	// ROT: the four input rows sit in a ring of four row slots starting at slot ROT
	// (conv_splitk_kernel's rolling tile); 0 = rows back to back
	template <int ROT = 0>
	static __device__ __forceinline__ void issue(Vec8<T> (&fb)[4], unsigned rowAddr,
	    const unsigned (&colOff)[3], const unsigned (&colSwz)[3], int hh, int m, int j) {
		const int dx = m / KS, ks = m % KS;
		const unsigned a = rowAddr + colOff[dx] + ((static_cast<unsigned>(ks * 2 + hh) ^ colSwz[dx]) << 4);
		if (j == 0) rd<(0 + ROT) % 4>(fb[0], a);
		else if (j == 1) rd<(1 + ROT) % 4>(fb[1], a);
		else if (j == 2) rd<(2 + ROT) % 4>(fb[2], a);
		else rd<(3 + ROT) % 4>(fb[3], a);
	}
	template <int N>
	static __device__ __forceinline__ void waitLgkm() {
		static_assert(N >= 0 && N <= 4, "lgkmcnt");
		if constexpr (N == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
		else if constexpr (N == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
		else if constexpr (N == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
		else if constexpr (N == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
		else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}

	template <int ROT = 0>
	static __device__ __forceinline__ void run(unsigned rowAddr, const unsigned (&colOff)[3],
	    const unsigned (&colSwz)[3], int hh, const Vec8<T> (&w)[9 * KS], f32x16 (&acc)[2]) {
		Vec8<T> fb[2][4];
		// start from an empty LGKM counter: the counted waits below must see only this
		// loop's own reads
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int j = 0; j < 4; ++j) issue<ROT>(fb[0], rowAddr, colOff, colSwz, hh, 0, j);
#pragma unroll
		for (int m = 0; m < NMAC; ++m) {
			const int set = m & 1;
			const bool more = m + 1 < NMAC;
			const int dx = m / KS, ks = m % KS;
#pragma unroll
			for (int k = 0; k < 6; ++k) {
				// MFMA k = (dy, r) = (k >> 1, k & 1) needs fragment r + dy; fragments are read
				// (and return) in order 0..3
				const int dy = k >> 1, r = k & 1;
				const int need = r + dy;
				const bool fresh = k == 0 || k == 1 || k == 3 || k == 5;
				if (fresh) {
					// outstanding allowed = younger reads of this step + next step's issued so far
					const int allowed = (3 - need) + (more ? (k < 4 ? k : 4) : 0);
					if (allowed >= 4) waitLgkm<4>();
					else if (allowed == 3) waitLgkm<3>();
					else if (allowed == 2) waitLgkm<2>();
					else if (allowed == 1) waitLgkm<1>();
					else waitLgkm<0>();
					__builtin_amdgcn_sched_barrier(0);
				}
				acc[r] = mfma32(w[(dy * 3 + dx) * KS + ks], fb[set][need], acc[r]);
				if (more && k < 4) issue<ROT>(fb[set ^ 1], rowAddr, colOff, colSwz, hh, m + 1, k);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
};

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
	// PACK instantiation (the flow net's first block): the 16-channel input records are built
	// here from the u8 frame and the previous packed tensor (launchPackFrames' arithmetic) and
	// written to `packOut` by the tile that owns the pixel; `in` is unused
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	const void *packPrev;
	void *packOut;
	int frameH, frameW, padTop, padLeft, numInputs;
	const unsigned *sums;
};

// Activation as ONE multiplier: x < 0 ? x * s : x with s = 0 (ReLU), the negative slope
// (LeakyReLU) or 1 (none).  (ReLU of a negative value gives -0.0, which every consumer
// treats as zero.)
__device__ __forceinline__ float fbActS(int act, float slope) {
	return act == 1 ? 0.0f : (act == 2 ? slope : 1.0f);
}
__device__ __forceinline__ float fbAct(float v, float s) {
	return v < 0.0f ? v * s : v;
}

// value of the neighbouring lane (lane ^ 1) without an LDS round trip: DPP quad_perm [1,0,3,2]
__device__ __forceinline__ float fbSwapPair(float v) {
	return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

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
	static constexpr int OFF_X = 0;
	static constexpr int OFF_T = NPL * XPLANE;
	static constexpr int TREGION = TBYTES > LBYTES ? TBYTES : LBYTES;  // the patch is dead before T is written
	// output staging per wave: 32 couts of a row pair (or of its pooled row)
	static constexpr int ESZ = 2;                       // (OUTK 1: f16 instead of T, same size)
	static constexpr int RBW = 32 * ESZ;                // bytes per pixel per cout block
	static constexpr int STAGE_PX = POOL ? 16 : 64;
	static constexpr int STAGE_WAVE = STAGE_PX * RBW;
	static constexpr bool STAGE_IN_X = NPL * XPLANE >= NW * STAGE_WAVE;  // X is dead once conv A is done
	static constexpr int OFF_STAGE = STAGE_IN_X ? OFF_X : OFF_T + TREGION;
	static constexpr int LDS = OFF_T + TREGION + (STAGE_IN_X ? 0 : NW * STAGE_WAVE);
	static_assert(CIN == 16 || CIN == 32 || CIN == 64 || CIN == 128, "input channels");
	static_assert(CMID == 32 || CMID == 64, "block filters");
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
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int x0 = blockIdx.x * kFbOutW;  // first output column of the tile
	const int y0 = blockIdx.y * TH;       // first output row
	const T *__restrict__ in = static_cast<const T *>(p.in);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));

	// wave -> jobs: cout block cb (both convs have CMID couts), row pairs pstart, +pstep, ...
	constexpr int NCB = G::NCB;
	const int cb = NCB == 2 ? (wave & 1) : 0;
	const int pstart = NCB == 2 ? (wave >> 1) : wave;
	constexpr int PSTEP = NCB == 2 ? NW / 2 : NW;
	const float s1 = fbActS(p.act1, p.slope), s2 = fbActS(p.act2, p.slope);

	// ---- conv A weights: A fragments of this wave's cout block, straight to registers ----
	Vec8<T> wa[G::NPL][9 * G::KS1];
	{
		const unsigned char *wsrc = static_cast<const unsigned char *>(p.w1) +
		                            (size_t)cb * G::NPL * (9 * G::KS1 * 1024) + lane * 16;
#pragma unroll
		for (int pl = 0; pl < G::NPL; ++pl) {
#pragma unroll
			for (int f = 0; f < 9 * G::KS1; ++f) {
				wa[pl][f] = *reinterpret_cast<const Vec8<T> *>(wsrc + (size_t)(pl * 9 * G::KS1 + f) * 1024);
			}
		}
	}

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
		const float bright = brightnessOf(p.sums, 1.0f / static_cast<float>(p.frameH * p.frameW));
		const T *__restrict__ prev = static_cast<const T *>(p.packPrev);
		T *__restrict__ cur = static_cast<T *>(p.packOut);
		const int nch = 3 * p.numInputs;
		constexpr int NPIX = G::XR * kFbW;
		for (int q = tid; q < NPIX; q += NT) {
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 2 + r, gx = x0 - 2 + k;
			const T zero = static_cast<T>(0.f);
			Vec8<T> o0 = {zero, zero, zero, zero, zero, zero, zero, zero}, o1 = o0;
			const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			if (inside) {
				const int y = gy - p.padTop, x = gx - p.padLeft;
				float c0 = 0.f, c1 = 0.f, c2 = 0.f;
				if (y >= 0 && y < p.frameH && x >= 0 && x < p.frameW) {
					const unsigned v = *reinterpret_cast<const unsigned *>(p.frame + y * p.frameStride + x * 4);
					c0 = preprocessU8(v & 0xff) - bright;
					c1 = preprocessU8((v >> 8) & 0xff) - bright;
					c2 = preprocessU8((v >> 16) & 0xff) - bright;
				}
				const size_t idx = (size_t)gy * p.W + gx;
				const Vec8<T> p0 = *reinterpret_cast<const Vec8<T> *>(prev + idx * 16);
				const Vec8<T> p1 = *reinterpret_cast<const Vec8<T> *>(prev + idx * 16 + 8);
				T pv[16], o[16];
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					pv[i] = p0[i];
					pv[8 + i] = p1[i];
				}
				o[0] = static_cast<T>(c0);
				o[1] = static_cast<T>(c1);
				o[2] = static_cast<T>(c2);
#pragma unroll
				for (int j = 3; j < 16; ++j) o[j] = (j < nch) ? pv[j - 3] : zero;
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					o0[i] = o[i];
					o1[i] = o[8 + i];
				}
				if (r >= 2 && r < 2 + TH && k >= 2 && k < 2 + kFbOutW) {
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
		constexpr int BR = G::XR / 2, BC = kFbW / 2;  // 2x2 blocks of the tile
		constexpr int NEL = G::NPL * BR * BC * LPP;
		for (int e = tid; e < NEL; e += NT) {
			const int c = e % LPP;
			const int bq = (e / LPP) % (BR * BC);
			const int pl = e / (LPP * BR * BC);
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
			unsigned char *xb = smem + G::OFF_X + pl * G::XPLANE + (r * kFbW + k) * G::PBX;
			const unsigned s0 = (static_cast<unsigned>(c) ^ fbSwz<G::PBX>(k)) << 4;
			const unsigned s1x = (static_cast<unsigned>(c) ^ fbSwz<G::PBX>(k + 1)) << 4;
			*reinterpret_cast<Vec8<T> *>(xb + s0) = inside ? tl : zero;
			*reinterpret_cast<Vec8<T> *>(xb + G::PBX + s1x) = inside ? o01 : zero;
			*reinterpret_cast<Vec8<T> *>(xb + kFbW * G::PBX + s0) = inside ? o10 : zero;
			*reinterpret_cast<Vec8<T> *>(xb + (kFbW + 1) * G::PBX + s1x) = inside ? o11 : zero;
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
		constexpr bool kHoistBiasA = G::NPL * 9 * G::KS1 * 4 < 144 || NW == 4;
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
#pragma unroll
				for (int s = 0; s < HOLD; ++s) {
					const int pair = pstart + s * PSTEP;
					if (pair < NPAIR) {
						FbPair<T, G::KS1, G::PBX>::run(
						    ldsBase + G::OFF_X + pl * G::XPLANE + (2 * pair) * (kFbW * G::PBX), colOffA, colSwzA, hh,
						    wa[pl], acc[s]);
					}
				}
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
		for (int f = 0; f < 9 * G::KS2; ++f) wb[f] = *reinterpret_cast<const Vec8<T> *>(wsrc + (size_t)f * 1024);
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
			unsigned char *outp = static_cast<unsigned char *>(p.out);
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

// ---------------------------------------------------------------------------
// res_block_kernel: one 64-filter residual block (models.py:193-254) per launch, PERSISTENT
// ---------------------------------------------------------------------------
// For towers that cannot use tower_resident_kernel (more 32 x 16 regions than CUs, e.g.
// 640 x 448; after a fallback).  One workgroup per CU loops over
// 14 x 30-pixel tiles:
//   * both convolutions' A fragments stay in registers for the whole launch (2 x 36
//     fragments = 288 VGPRs per wave, one wave per SIMD): no weight traffic per tile;
//   * conv A: X tile (18 x 34 px, LDS) -> activation -> T tile (16 x 34 px, LDS);
//   * while conv B (T -> + skip -> activation -> global) is on the matrix cores, the NEXT
//     tile's X is already in flight into the X buffer by LDS-DMA (X is dead once conv A
//     is done), so tile staging costs no time of its own;
//   * the skip connection is re-read from global (the lines were just staged: L2 hits).
// in / out are addressed at image pixel (0, 0) with a row pitch, so dense and tower-layout
// tensors both work; only image pixels are written (a tower tensor's zero border stays).
constexpr int kRbTH = 14;
constexpr int kRbXR = kRbTH + 4, kRbTR = kRbTH + 2;
constexpr int kRbX = kRbXR * kFbW * 128;           // 78336
constexpr int kRbT = kRbTR * kFbW * 128;           // 69632
constexpr int kRbStageWave = 32 * 64;              // one row of 32 px x 32 couts, 16-bit
constexpr int kRbLds = kRbX + kRbT + 4 * kRbStageWave;
static_assert(kRbLds <= 160 * 1024, "res block tile");

struct ResBlockParams {
	const void *in;
	void *out;
	const void *w1, *w2;   // packConvWeights(nb = 1)
	const float *b1, *b2;
	int H, W, inPitch, outPitch;
	int tilesX, numTiles;
	float s1, s2;          // activation multipliers (fbActS)
	int skip;              // timing ablation (JU_FB_SKIP, developer only)
	unsigned long long *prof;  // developer builds (-DJU_RB_PROF): per-wave cycle sums of workgroup 0
};

template <typename T>
__global__ __launch_bounds__(256, 1) void res_block_kernel(ResBlockParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	const int cb = wave & 1, pl = wave >> 1;  // cout block, pair lane (pairs pl, pl + 2, ...)
	const T *__restrict__ in = static_cast<const T *>(p.in);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));
	unsigned char *smX = smem, *smT = smem + kRbX;
	unsigned char *stage = smem + kRbX + kRbT + wave * kRbStageWave;

	// both convolutions' fragments of this wave's cout block, for the whole launch
	Vec8<T> wa[36], wb[36];
	{
		const unsigned char *a = static_cast<const unsigned char *>(p.w1) + (size_t)cb * (36 * 1024) + lane * 16;
		const unsigned char *b = static_cast<const unsigned char *>(p.w2) + (size_t)cb * (36 * 1024) + lane * 16;
#pragma unroll
		for (int f = 0; f < 36; ++f) {
			wa[f] = *reinterpret_cast<const Vec8<T> *>(a + (size_t)f * 1024);
			wb[f] = *reinterpret_cast<const Vec8<T> *>(b + (size_t)f * 1024);
		}
	}
	f32x4 biasA[4], biasB[4];
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		biasA[g] = *reinterpret_cast<const f32x4 *>(p.b1 + cb * 32 + 8 * g + 4 * hh);
		biasB[g] = *reinterpret_cast<const f32x4 *>(p.b2 + cb * 32 + 8 * g + 4 * hh);
	}
	unsigned colOff[3], colSwz[3], tOff[4];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOff[dx] = (px + dx) * 128;
		colSwz[dx] = fbSwz<128>(px + dx);
	}
#pragma unroll
	for (int g = 0; g < 4; ++g) tOff[g] = px * 128 + ((static_cast<unsigned>(cb * 4 + g) ^ fbSwz<128>(px)) << 4) + hh * 8;

	// X tile of `tile`: image pixels by LDS-DMA, pixels outside the image zeroed by hand
	// (disjoint LDS locations, so the two need no ordering between them)
	auto stageX = [&](int tile) {
		const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
		const int y0 = ty * kRbTH, x0 = tx * kFbOutW;
		constexpr int NPIX = kRbXR * kFbW;
		constexpr int NINSTR = (NPIX + 7) / 8;  // 8 pixels (1 KiB) per wave-instruction
		const bool border = y0 - 2 < 0 || y0 + kRbTH + 2 > p.H || x0 - 2 < 0 || x0 + 32 > p.W;
		for (int i = wave; i < NINSTR; i += 4) {
			const int q = i * 8 + (lane >> 3);
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 2 + r, gx = x0 - 2 + k;
			const unsigned c = static_cast<unsigned>(lane & 7) ^ fbSwz<128>(k);
			const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			if (q < NPIX && inside) {
				fbGlds16(in + ((size_t)gy * p.inPitch + gx) * 64 + c * 8, smX + i * 1024);
			} else if (border && q < NPIX) {
				*reinterpret_cast<uint4 *>(smX + i * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
			}
		}
	};

	int tile = blockIdx.x;
	if (tile < p.numTiles && !(JU_SKIP(p) & 1)) stageX(tile);
	for (; tile < p.numTiles; tile += gridDim.x) {
		const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
		const int y0 = ty * kRbTH, x0 = tx * kFbOutW;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's X has landed (and the last tile's stores left)
		__syncthreads();                                  // ... for every wave; all are done with T too
		// ---- conv A: 16 rows x 32 columns -> T, zero outside the image ----
		for (int pair = pl; pair < kRbTR / 2; pair += 2) {
			f32x16 acc[2];
#pragma unroll
			for (int g = 0; g < 4; ++g) {
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = biasA[g][i];
				}
			}
			if (!(JU_SKIP(p) & 2)) FbPair<T, 4, 128>::run(ldsBase + (2 * pair) * (kFbW * 128), colOff, colSwz, hh, wa, acc);
			if (JU_SKIP(p) & 32) continue;
			const int gx = x0 - 1 + px;
			const bool colIn = gx >= 0 && gx < p.W;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int tr = 2 * pair + r;
				const int gy = y0 - 1 + tr;
				const float keep = (colIn && gy >= 0 && gy < p.H) ? 1.0f : 0.0f;
				unsigned char *row = smT + tr * (kFbW * 128);
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					*reinterpret_cast<Vec4<T> *>(row + tOff[g]) =
					    pack4<T>(fbAct(acc[r][4 * g + 0], p.s1) * keep, fbAct(acc[r][4 * g + 1], p.s1) * keep,
					        fbAct(acc[r][4 * g + 2], p.s1) * keep, fbAct(acc[r][4 * g + 3], p.s1) * keep);
				}
			}
		}
		__syncthreads();  // T complete, X dead
		// ---- the next tile's X travels while conv B computes ----
		if (tile + static_cast<int>(gridDim.x) < p.numTiles && !(JU_SKIP(p) & 1)) stageX(tile + gridDim.x);
		// ---- conv B: 14 rows x 32 columns (30 valid) + skip -> activation -> global ----
		for (int pair = pl; pair < kRbTH / 2; pair += 2) {
			f32x16 acc[2];
#pragma unroll
			for (int g = 0; g < 4; ++g) {
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = biasB[g][i];
				}
			}
			Vec4<T> resv[2][4];
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int gy = min(y0 + 2 * pair + r, p.H - 1), gx = min(x0 + px, p.W - 1);
				const T *rp = in + ((size_t)gy * p.inPitch + gx) * 64 + cb * 32 + 4 * hh;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					if (JU_SKIP(p) & 8) resv[r][g] = Vec4<T>{};
					else resv[r][g] = *reinterpret_cast<const Vec4<T> *>(rp + 8 * g);
				}
			}
			if (!(JU_SKIP(p) & 4)) FbPair<T, 4, 128>::run(ldsBase + kRbX + (2 * pair) * (kFbW * 128), colOff, colSwz, hh, wb, acc);
			if (JU_SKIP(p) & 64) continue;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					float v[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = fbAct(acc[r][4 * g + i] + static_cast<float>(resv[r][g][i]), p.s2);
					const unsigned c = static_cast<unsigned>(g) ^ (static_cast<unsigned>(px) & 3u);
					*reinterpret_cast<Vec4<T> *>(stage + px * 64 + (c << 4) + hh * 8) = pack4<T>(v[0], v[1], v[2], v[3]);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				// one row: 32 px x 64 B = 2 KiB = 2 wave-instructions of 16 B per lane
				unsigned char *outp = static_cast<unsigned char *>(p.out);
				const int gy = y0 + 2 * pair + r;
#pragma unroll
				for (int it = 0; it < 2; ++it) {
					const int pi = it * 16 + (lane >> 2);
					const unsigned slot = static_cast<unsigned>(lane & 3);
					const unsigned chunk = slot ^ (static_cast<unsigned>(pi) & 3u);
					const uint4 val = *reinterpret_cast<const uint4 *>(stage + pi * 64 + (slot << 4));
					const int gx = x0 + pi;
					if (pi < kFbOutW && gy < p.H && gx < p.W && !(JU_SKIP(p) & 16)) {
						*reinterpret_cast<uint4 *>(outp + (((size_t)gy * p.outPitch + gx) * 64 + cb * 32) * 2 + chunk * 16) = val;
					}
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
		}
	}
}


// ---------------------------------------------------------------------------
// res_block_pipe_kernel: res_block_kernel with the epilogues in the MFMAs' shadow
// ---------------------------------------------------------------------------
// ReLU blocks only (every other activation: res_block_kernel).  Same tiles, same LDS tiles, same values as
// res_block_kernel up to the sign of zeros in the intermediate tensors (JU_RES_BLOCK=plain keeps that kernel;
// the tests require equal frames and state), one wave per SIMD with both convolutions' weights in
// accumulation registers -- but a row pair's epilogue no longer runs between two K loops (the plain kernel:
// 23.7 of 58 us per block inside the K loops, the rest mostly epilogues at one wave per SIMD).  A 32x32x16
// MFMA leaves about three VALU issue slots free while it runs (tools/probes/mfma_valu_overlap.hip), so pair
// P's epilogue is spread over the macro-steps of pair P + 1's K loop: two accumulator sets alternate, a group
// of four values (row r, channel group g) per macro-step.
//   conv A: pack -> ReLU on the packed values -> AND mask outside the image -> T tile; only the last pair's
//           epilogue is exposed (T must be complete at the workgroup barrier).
//   conv B: + skip (value i behind MFMA i), pack + ReLU behind MFMA 4, into a per-wave staging tile of two
//           rows behind MFMA 5 (macro-steps 3..10 -- the skip values are fetched behind the pair's OWN last
//           macro-step and need the time), read back transposed behind macro-step 11, and stored as whole
//           16-byte chunks behind macro-step 4 of the K loop after that (behind macro-step 3's wait for the skip
//           values: a wait with loads AND stores pending is vmcnt(0)).  The last pair of a tile finishes
//           inside the next tile's first K loops.
//   X tile: an interior tile is 20 table-driven LDS-DMA instructions per wave, issued behind the MFMAs of the
//           first conv B loop (all four waves issuing them at once stall ~3 k cycles in the CU's vector-memory
//           queue); an edge tile takes the general path of res_block_kernel, and the edge tiles are dealt to the
//           workgroups that have a round less to run.
//   bias:   64 + 64 floats in LDS (by DMA), read into a 16-register vector per phase and used as the C operand
//           of a pair's first MFMAs (no accumulator is initialised by moves).
// In-kernel phase sums: -DJU_RB_PROF + tools/rb_pipe_profile.py.
constexpr int kRpStageRow = kFbOutW * 64;         // one row of 30 px x 32 couts, 16-bit
constexpr int kRpStageWave = 2 * kRpStageRow;
constexpr int kRpLds = kRbX + kRbT + 4 * kRpStageWave + 512;
static_assert(kRpLds <= 160 * 1024, "res block tile (pipelined)");

// FbPair<T, 4, 128>::run with three hooks: `first` (the pair's bias vector: C operand of the first
// MFMA of each row), `atStart()` behind the opening wait, `behind(m, k)` behind MFMA k of macro-step m.
// LDS instructions issued by the hooks only make the counted waits stricter (they count what is
// outstanding, the hooks' instructions are younger than the fragments waited for or complete before them).
// STREAM (measured, not used: see the kernel): the OTHER convolution's fragments replace this one's as they
// die, behind each macro-step the three fragments it used.
template <typename T, bool STREAM, typename FS, typename FB>
__device__ __forceinline__ void rbPipeRun(unsigned rowAddr, const unsigned (&colOff)[3], const unsigned (&colSwz)[3], int hh,
    Vec8<T> (&w)[36], const __amdgpu_buffer_rsrc_t nextW, unsigned nextLane, unsigned nextBase, f32x16 (&acc)[2],
    const f32x16 &first, FS &&atStart, FB &&behind) {
	using P = FbPair<T, 4, 128>;
	typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
	Vec8<T> fb[2][4];
	// (opaque: with every K loop of the tile unrolled, the 12 fragment addresses of each are loop-invariant
	// over the tiles and the compiler keeps all 132 of them in registers -- and spills)
	asm volatile("" : "+s"(rowAddr));
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	__builtin_amdgcn_sched_barrier(0);
	atStart();
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int j = 0; j < 4; ++j) P::template issue<0>(fb[0], rowAddr, colOff, colSwz, hh, 0, j);
#pragma unroll
	for (int m = 0; m < 12; ++m) {
		const int set = m & 1;
		const bool more = m + 1 < 12;
		const int dx = m / 4, ks = m % 4;
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int dy = k >> 1, r = k & 1;
			const int need = r + dy;
			const bool fresh = k == 0 || k == 1 || k == 3 || k == 5;
			if (fresh) {
				const int allowed = (3 - need) + (more ? (k < 4 ? k : 4) : 0);
				if (allowed >= 4) P::template waitLgkm<4>();
				else if (allowed == 3) P::template waitLgkm<3>();
				else if (allowed == 2) P::template waitLgkm<2>();
				else if (allowed == 1) P::template waitLgkm<1>();
				else P::template waitLgkm<0>();
				__builtin_amdgcn_sched_barrier(0);
			}
			acc[r] = mfma32(w[(dy * 3 + dx) * 4 + ks], fb[set][need], (m == 0 && dy == 0) ? first : acc[r]);
			__builtin_amdgcn_sched_barrier(0);  // (the MFMA first: what follows runs in its shadow, not in front of it)
			if (more && k < 4) P::template issue<0>(fb[set ^ 1], rowAddr, colOff, colSwz, hh, m + 1, k);
			behind(m, k);
			__builtin_amdgcn_sched_barrier(0);
		}
		if constexpr (STREAM) {
			// (buffer loads: lane offset in ONE register, the fragment's offset scalar -- flat loads 1 KiB
			// apart are out of immediate range and cost a 64-bit address each)
#pragma unroll
			for (int dy = 0; dy < 3; ++dy) {
				const int f = (dy * 3 + dx) * 4 + ks;
				const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(nextW, nextLane, nextBase + f * 1024, 0);
				w[f] = __builtin_bit_cast(Vec8<T>, v);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

template <typename T>
__global__ __launch_bounds__(256, 1) void res_block_pipe_kernel(ResBlockParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
	typedef unsigned u32x2r __attribute__((ext_vector_type(2)));
	const int tid = threadIdx.x, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (scalar: tile and pair coordinates stay in SGPRs)
	const int cb = wave & 1, pl = wave >> 1;  // cout block, pair lane (pairs pl, pl + 2, ...)
	const T *__restrict__ in = static_cast<const T *>(p.in);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));
	unsigned char *smX = smem, *smT = smem + kRbX;
	unsigned char *stage = smem + kRbX + kRbT + wave * kRpStageWave;
	float *biasLds = reinterpret_cast<float *>(smem + kRbX + kRbT + 4 * kRpStageWave);
	// Global memory through buffer instructions: the lane's part of an address is ONE loop-invariant
	// 32-bit register, the tile / row / fragment part is scalar (a tensor is < 2 GiB).
	const __amdgpu_buffer_rsrc_t rsrcWa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, 2 * 36 * 1024, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcWb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 2 * 36 * 1024, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcIn = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, 0x7ffffff0, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
	const unsigned wLane = static_cast<unsigned>(lane) * 16u;
	const unsigned wBase = static_cast<unsigned>(cb) * (36u * 1024u);

	// Both convolutions' fragments for the whole launch, in accumulation registers (an MFMA reads them there;
	// left alone the compiler keeps part of them in VGPRs and parks addresses in the AGPRs instead, one copy
	// per use).  (One set of 36 with the other convolution's streamed in behind the last K loop of each phase
	// -- the resident tower's way -- was measured: 144 KB of requests per workgroup and phase fill the CU's
	// 64 B/clk vector-memory path for a whole K loop, the streaming loops took 4900-5900 cycles instead of
	// 3450.)
	// (the bias floats go to LDS by DMA: no register, no wait of their own -- the first tile's wait covers them)
	if (wave == 0) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.b1 + lane),
		    (__attribute__((address_space(3))) void *)biasLds, 4, 0, 0);
	} else if (wave == 1) {
		__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.b2 + lane),
		    (__attribute__((address_space(3))) void *)(biasLds + 64), 4, 0, 0);
	}
	// (conv B's set is requested after the first tile's X and pinned in front of the first conv B phase: the
	// first tile starts on conv A's 36 KB instead of waiting for all 72)
	Vec8<T> wa[36], wb[36];
#pragma unroll
	for (int f = 0; f < 36; ++f) {
		wa[f] = __builtin_bit_cast(Vec8<T>, __builtin_amdgcn_raw_buffer_load_b128(rsrcWa, wLane, wBase + f * 1024, 0));
	}
	// this lane's 16 accumulator values of a row: channel cb * 32 + 8 g + 4 hh + i at index 4 g + i
	auto loadBias = [&](int conv) __attribute__((always_inline)) -> f32x16 {
		const float *b = biasLds + conv * 64 + cb * 32 + 4 * hh;
		const f32x4 b0 = *reinterpret_cast<const f32x4 *>(b), b1 = *reinterpret_cast<const f32x4 *>(b + 8),
		            b2 = *reinterpret_cast<const f32x4 *>(b + 16), b3 = *reinterpret_cast<const f32x4 *>(b + 24);
		return f32x16{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3], b2[0], b2[1], b2[2], b2[3], b3[0], b3[1], b3[2], b3[3]};
	};
	unsigned colOff[3], colSwz[3], tOff[4];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOff[dx] = (px + dx) * 128;
		colSwz[dx] = fbSwz<128>(px + dx);
	}
#pragma unroll
	for (int g = 0; g < 4; ++g) tOff[g] = px * 128 + ((static_cast<unsigned>(cb * 4 + g) ^ fbSwz<128>(px)) << 4) + hh * 8;
	const bool stageLane = px < kFbOutW;
	// staging tile: write offset of group g (+ r * kRpStageRow), read-back offset of chunk it (+ r * kRpStageRow)
	unsigned stW[4];
#pragma unroll
	for (int g = 0; g < 4; ++g) stW[g] = px * 64 + ((static_cast<unsigned>(g) ^ (static_cast<unsigned>(px) & 3u)) << 4) + hh * 8;
	const unsigned stR = (lane >> 2) * 64 + ((lane & 3) << 4);
	// output: lane part of chunk `it` (pixel pi = it * 16 + lane / 4 of the row)
	unsigned outLane[2];
	int outPi[2];
#pragma unroll
	for (int it = 0; it < 2; ++it) {
		const int pi = it * 16 + (lane >> 2);
		const unsigned chunk = static_cast<unsigned>(lane & 3) ^ (static_cast<unsigned>(pi) & 3u);
		outLane[it] = static_cast<unsigned>(pi * 64 + cb * 32) * 2u + chunk * 16u;
		outPi[it] = pi < kFbOutW ? pi : 0x40000000;  // (never below a column limit)
	}

	// X tile staging.  An interior tile (no pixel outside the image: all but the frame's edge tiles) is 19-20
	// LDS-DMA instructions per wave whose per-lane source offsets relative to the tile's first pixel do not
	// depend on the tile: computed once (xOff), the tile's origin is the scalar offset -- the per-instruction
	// address arithmetic of the general path (divisions by 34, four bound tests, a 64-bit multiply-add: ~60
	// instructions each) was 5 k cycles per tile, 14 % of the kernel.
	constexpr int kXPix = kRbXR * kFbW;
	constexpr int kXInstr = (kXPix + 7) / 8;        // 77: 8 pixels (1 KiB) per wave-instruction
	constexpr int kXPerWave = (kXInstr + 3) / 4;    // 20
	unsigned xOff[kXPerWave];
	{
		// pixel q = (wave + 4 n) * 8 + lane / 8 of the tile: row r, column k; q advances by 32 per n
		const int q0 = wave * 8 + (lane >> 3);
		int r = q0 / kFbW, k = q0 - r * kFbW;
#pragma unroll
		for (int n = 0; n < kXPerWave; ++n) {
			if (n == kXPerWave - 1) {
				// the 77th instruction has 4 pixels left: every wave's 20th fetches the tile's LAST 8 pixels
				// instead (four of them a second time) -- no lane mask, no branch
				const int ql = kXPix - 8 + (lane >> 3);
				r = ql / kFbW;
				k = ql - r * kFbW;
			}
			const unsigned c = static_cast<unsigned>(lane & 7) ^ fbSwz<128>(k);
			xOff[n] = static_cast<unsigned>(r * p.inPitch + k) * 128u + c * 16u;
			k += 32;
			if (k >= kFbW) {
				k -= kFbW;
				r += 1;
			}
		}
	}
	// Slot v (workgroup b's round r: v = r * grid + b) -> tile.  Edge tiles stage their X through the general
	// path (~5 k cycles where an interior tile pays nothing), and the launch ends with its slowest workgroup:
	// so the edge tiles go to the workgroups that have a round less to run (704 tiles on 256 workgroups: 64 of
	// them run two tiles instead of three), the interior tiles fill the other slots in row-major order.
	const int tRounds = (p.numTiles + static_cast<int>(gridDim.x) - 1) / static_cast<int>(gridDim.x);
	const int tLast = p.numTiles - (tRounds - 1) * static_cast<int>(gridDim.x);  // tiles of the last round
	const int tSlack = static_cast<int>(gridDim.x) - tLast;                       // workgroups with a round less
	const int tilesY = p.numTiles / p.tilesX;
	const int tEdge = 2 * p.tilesX + 2 * (tilesY - 2);
	const bool tRemap = tRounds >= 2 && p.tilesX >= 3 && tilesY >= 3 && tEdge <= (tRounds - 1) * tSlack;
	auto tileXY = [&](int v, int &ty, int &tx) __attribute__((always_inline)) {
		if (!tRemap) {
			ty = v / p.tilesX;
			tx = v - ty * p.tilesX;
			return;
		}
		const int r = v / static_cast<int>(gridDim.x), b = v - r * static_cast<int>(gridDim.x);
		const bool slack = b >= tLast && r < tRounds - 1;
		const int sBefore = r < tRounds - 1 ? r * tSlack + (b > tLast ? b - tLast : 0) : (tRounds - 1) * tSlack;
		if (slack && sBefore < tEdge) {
			const int e = sBefore;  // edge tile number e: top row, bottom row, then the two columns
			if (e < p.tilesX) { ty = 0; tx = e; }
			else if (e < 2 * p.tilesX) { ty = tilesY - 1; tx = e - p.tilesX; }
			else { ty = 1 + ((e - 2 * p.tilesX) >> 1); tx = ((e - 2 * p.tilesX) & 1) ? p.tilesX - 1 : 0; }
		} else {
			const int n = v - (sBefore < tEdge ? sBefore : tEdge);  // interior tile number
			ty = 1 + n / (p.tilesX - 2);
			tx = 1 + n - (ty - 1) * (p.tilesX - 2);
		}
	};
	// instruction n of this wave for the interior tile whose first X pixel is at byte offset `so`
	auto stageXOne = [&](unsigned so, int n) __attribute__((always_inline)) {
		const int i = wave + 4 * n;
		if (n < kXPerWave - 1) {
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcIn, (__attribute__((address_space(3))) void *)(smX + i * 1024), 16, xOff[n], so, 0, 0);
		} else {
			// (every wave, the same bytes to the same place: a branch on the wave puts the instruction into
			// a block of its own, and there the compiler waits vmcnt(0) in front of it)
			__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrcIn, (__attribute__((address_space(3))) void *)(smX + (kXPix - 8) * 128), 16, xOff[n], so, 0, 0);
		}
	};
	auto tileIsBorder = [&](int tile) __attribute__((always_inline)) {
		int ty, tx;
		tileXY(tile, ty, tx);
		const int y0 = ty * kRbTH, x0 = tx * kFbOutW;
		return y0 - 2 < 0 || y0 + kRbTH + 2 > p.H || x0 - 2 < 0 || x0 + 32 > p.W;
	};
	auto stageX = [&](int tile) {
		int ty, tx;
		tileXY(tile, ty, tx);
		const int y0 = ty * kRbTH, x0 = tx * kFbOutW;
		constexpr int NPIX = kRbXR * kFbW;
		constexpr int NINSTR = (NPIX + 7) / 8;  // 8 pixels (1 KiB) per wave-instruction
		const bool border = y0 - 2 < 0 || y0 + kRbTH + 2 > p.H || x0 - 2 < 0 || x0 + 32 > p.W;
		if (!border) {
			const unsigned so = static_cast<unsigned>((y0 - 2) * p.inPitch + (x0 - 2)) * 128u;
#pragma unroll
			for (int n = 0; n < kXPerWave; ++n) stageXOne(so, n);
			return;
		}
		for (int i = wave; i < NINSTR; i += 4) {
			const int q = i * 8 + (lane >> 3);
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 2 + r, gx = x0 - 2 + k;
			const unsigned c = static_cast<unsigned>(lane & 7) ^ fbSwz<128>(k);
			const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			if (q < NPIX && inside) {
				fbGlds16(in + ((size_t)gy * p.inPitch + gx) * 64 + c * 8, smX + i * 1024);
			} else if (border && q < NPIX) {
				*reinterpret_cast<uint4 *>(smX + i * 1024 + lane * 16) = make_uint4(0, 0, 0, 0);
			}
		}
	};

	// ---- pipeline state ----
	f32x16 S0[2], S1[2];       // the two accumulator sets
	u32x2r resv[2][4];         // skip values (4 x 16 bit) of the conv B pair whose epilogue is pending
	u32x4w stg0 = {}, stg1 = {}, stg2 = {}, stg3 = {};  // a finished pair's two rows x two chunks, transposed, on their way out
	int stY = 0, stX = 0;      // ... and where they go: image row of its first row, first column
	bool stOn = false;
	int epY = 0, epX = 0;      // the pending conv B pair's first image row / first column
	bool epOn = false;         // a conv B epilogue is pending (from the previous tile)

	const auto noStart = []() __attribute__((always_inline)) {};
	// stores of the pair in stg0..3 (read back behind the previous K loop's last macro-step)
	auto storeRows = [&]() __attribute__((always_inline)) {
		if (stOn) {
			const int lim = p.W - stX;  // columns of the tile inside the image
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int gy = stY + r;
				const unsigned so = static_cast<unsigned>(gy * p.outPitch + stX) * 128u;
				if (gy < p.H) {
					if (outPi[0] < lim) __builtin_amdgcn_raw_buffer_store_b128(r ? stg2 : stg0, rsrcOut, outLane[0], so, 0);
					if (outPi[1] < lim) __builtin_amdgcn_raw_buffer_store_b128(r ? stg3 : stg1, rsrcOut, outLane[1], so, 0);
				}
			}
			// (MUBUF stores with a scalar offset read their data late when a second wave competes for the
			// vector-memory issue -- DESIGN.md 4b; this kernel runs one wave per SIMD, the wait states are free)
			asm volatile("s_nop 1" ::: "memory");
			stOn = false;
		}
	};
	// skip values of the pair whose first image row is gy0, row r, groups g and g + 1: the lane's part of the
	// address (its column, clamped to the image) once per tile, the row scalar -- nothing per load
	unsigned skipLane = 0;
	auto loadSkip2 = [&](int gy0, int r, int g) __attribute__((always_inline)) {
		const int gy = min(gy0 + r, p.H - 1);
		const unsigned so = static_cast<unsigned>(gy * p.inPitch) * 128u;
		resv[r][g] = __builtin_amdgcn_raw_buffer_load_b64(rsrcIn, skipLane, so + g * 16, 0);
		resv[r][g + 1] = __builtin_amdgcn_raw_buffer_load_b64(rsrcIn, skipLane, so + (g + 1) * 16, 0);
	};
	// The epilogues, a group of four values (row r, channel group g) per macro-step, at most three plain VALU
	// operations behind each MFMA -- what issues in an MFMA's shadow for nothing; more than that, packed-f32
	// operations or wait states between them are paid in full (measured: with the generic x < 0 ? s x : x
	// activation in f32, 19-21 operations per group, the K loops took 3300-3600 cycles instead of 2304 and
	// the kernel gained nothing).  So this kernel serves ReLU blocks only (the host sends every other
	// activation to res_block_kernel): ReLU on the PACKED 16-bit values (a signed 16-bit max with 0, as in
	// the resident tower) and the outside-the-image zeroing as an AND mask.  Negative inputs give +0 where
	// the f32 form gives -0: the tensors differ in the sign of zeros only, every later value is the same.
	float dv[4];
	unsigned dlo = 0, dhi = 0;
	// conv A epilogue of the pair in (a0, a1) (T rows 2 pair, 2 pair + 1): group j behind macro-step j
	auto epiA = [&](const f32x16 &a0, const f32x16 &a1, int pair, unsigned keep0, unsigned keep1, int m, int k) __attribute__((always_inline)) {
		if (m < 8) {
			const int r = m >> 2, g = m & 3;
			if (k == 0) {
				const u32x2r wv = __builtin_bit_cast(u32x2r, reluPacked<T>(pack4<T>((r ? a1 : a0)[4 * g + 0], (r ? a1 : a0)[4 * g + 1],
				    (r ? a1 : a0)[4 * g + 2], (r ? a1 : a0)[4 * g + 3])));
				dlo = wv[0];
				dhi = wv[1];
			} else if (k == 2) {
				dlo &= (r ? keep1 : keep0);
				dhi &= (r ? keep1 : keep0);
			} else if (k == 4) {
				*reinterpret_cast<u32x2r *>(smT + (2 * pair + r) * (kFbW * 128) + tOff[g]) = u32x2r{dlo, dhi};
			}
		}
	};
	// conv B epilogue of the pair in (a0, a1) (skip values in resv): groups behind macro-steps 3..10 (value i
	// = accumulator + skip behind MFMA i, pack + ReLU behind MFMA 4, the staging write behind MFMA 5) and
	// the transposed read-back behind macro-step 11
	auto epiB = [&](const f32x16 &a0, const f32x16 &a1, int m, int k) __attribute__((always_inline)) {
		if (m >= 3 && m < 11) {
			const int j = m - 3, r = j >> 2, g = j & 3;
			if (k < 4) {
				const Vec4<T> rv = __builtin_bit_cast(Vec4<T>, resv[r][g]);
				dv[k] = (r ? a1 : a0)[4 * g + k] + static_cast<float>(rv[k]);
				asm volatile("" : "+v"(dv[k]));  // (keeps the four adds scalar and in their slots)
			} else if (k == 4) {
				const u32x2r wv = __builtin_bit_cast(u32x2r, reluPacked<T>(pack4<T>(dv[0], dv[1], dv[2], dv[3])));
				dlo = wv[0];
				dhi = wv[1];
			} else {
				if (stageLane) *reinterpret_cast<u32x2r *>(stage + r * kRpStageRow + stW[g]) = u32x2r{dlo, dhi};
			}
		} else if (m == 11) {
			// (pixels 30, 31 of the read-back are the neighbouring row's / wave's bytes: never stored)
			if (k == 0) {
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			} else if (k == 1) {
				stg0 = *reinterpret_cast<const u32x4w *>(stage + stR);
			} else if (k == 2) {
				stg1 = *reinterpret_cast<const u32x4w *>(stage + 16 * 64 + stR);
			} else if (k == 3) {
				stg2 = *reinterpret_cast<const u32x4w *>(stage + kRpStageRow + stR);
			} else if (k == 4) {
				stg3 = *reinterpret_cast<const u32x4w *>(stage + kRpStageRow + 16 * 64 + stR);
			}
		}
	};

#ifdef JU_RB_PROF
	unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
		unsigned long long t;
		__builtin_amdgcn_sched_barrier(0);
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
		__builtin_amdgcn_sched_barrier(0);
		return t;
	};
	const unsigned long long tKernel = stamp();
#define RB_STAMP(var) const unsigned long long var = stamp()
#define RB_ADD(slot, a, b) prof[slot] += (b) - (a)
#else
#define RB_STAMP(var)
#define RB_ADD(slot, a, b)
#endif
	int tile = blockIdx.x;
	if (tile < p.numTiles) stageX(tile);
	// (scheduling fences: the first tile's wait below counts on these 36 requests being the YOUNGEST
	// vector-memory operations in flight.  A wrong order would not go unnoticed silently for long -- the
	// frames are compared with the plain kernel's -- but it must not depend on the scheduler's mood.)
#pragma unroll
	for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wa[f]));  // (accumulation registers, see above)
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int f = 0; f < 36; ++f) {
		wb[f] = __builtin_bit_cast(Vec8<T>, __builtin_amdgcn_raw_buffer_load_b128(rsrcWb, wLane, wBase + f * 1024, 0));
	}
	__builtin_amdgcn_sched_barrier(0);
	bool firstTile = true;
	f32x16 bias;
	RB_STAMP(tPro);
	RB_ADD(0, tKernel, tPro);
	for (; tile < p.numTiles; tile += gridDim.x) {
		int ty, tx;
		tileXY(tile, ty, tx);
		const int y0 = ty * kRbTH, x0 = tx * kFbOutW;
		RB_STAMP(t0);
		// this tile's X has landed (and the last tile's stores left); the first tile: everything but the 36
		// requests of conv B's fragments behind it
		if (firstTile) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
		else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		firstTile = false;
		RB_STAMP(t1);
		__syncthreads();                                  // ... for every wave; all are done with T too
		RB_STAMP(t2);
		RB_ADD(2, t0, t2);
		bias = loadBias(0);
		const int gxA = x0 - 1 + px;
		const bool colIn = gxA >= 0 && gxA < p.W;
		auto keepOf = [&](int pair, int r) __attribute__((always_inline)) -> unsigned {
			const int gy = y0 - 1 + 2 * pair + r;
			return (colIn && gy >= 0 && gy < p.H) ? 0xffffffffu : 0u;
		};
		// ---- conv A: pairs pl, pl + 2, pl + 4, pl + 6 of the 16 T rows; sets S0, S1, S0, S1 ----
		{
			// pair 0 of this wave; behind it the previous tile's last conv B epilogue (its accumulators are in S1)
			const bool pend = epOn;
			const f32x16 e0 = S1[0], e1 = S1[1];
			rbPipeRun<T, false>(ldsBase + (2 * pl) * (kFbW * 128), colOff, colSwz, hh, wa, rsrcWb, wLane, wBase, S0, bias, noStart,
			    [&](int m, int k) __attribute__((always_inline)) {
				    if (pend) epiB(e0, e1, m, k);
				    if (m == 4 && k == 0) storeRows();
			    });
			if (pend) {
				stY = epY;
				stX = epX;
				stOn = true;
				epOn = false;
			}
		}
#pragma unroll
		for (int q = 1; q < 4; ++q) {
			const int pair = pl + 2 * q, prev = pair - 2;
			const unsigned k0 = keepOf(prev, 0), k1 = keepOf(prev, 1);
			if (q & 1) {
				const f32x16 e0 = S0[0], e1 = S0[1];
				rbPipeRun<T, false>(ldsBase + (2 * pair) * (kFbW * 128), colOff, colSwz, hh, wa, rsrcWb, wLane, wBase, S1, bias, noStart,
				    [&](int m, int k) __attribute__((always_inline)) {
					    epiA(e0, e1, prev, k0, k1, m, k);
					    if (m == 4 && k == 0) storeRows();
				    });
			} else {
				const f32x16 e0 = S1[0], e1 = S1[1];
				rbPipeRun<T, false>(ldsBase + (2 * pair) * (kFbW * 128), colOff, colSwz, hh, wa, rsrcWb, wLane, wBase, S0, bias, noStart,
				    [&](int m, int k) __attribute__((always_inline)) {
					    epiA(e0, e1, prev, k0, k1, m, k);
					    if (m == 4 && k == 0) storeRows();
				    });
			}
		}
		RB_STAMP(t3);
		RB_ADD(3, t2, t3);
		{
			// the last pair's epilogue (in S1): exposed, T must be complete at the barrier
			const int pair = pl + 6;
			const unsigned k0 = keepOf(pair, 0), k1 = keepOf(pair, 1);
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				unsigned char *row = smT + (2 * pair + r) * (kFbW * 128);
				const unsigned keep = r ? k1 : k0;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					const u32x2r wv = __builtin_bit_cast(u32x2r, reluPacked<T>(pack4<T>(S1[r][4 * g + 0], S1[r][4 * g + 1],
					    S1[r][4 * g + 2], S1[r][4 * g + 3])));
					*reinterpret_cast<u32x2r *>(row + tOff[g]) = u32x2r{wv[0] & keep, wv[1] & keep};
				}
			}
		}
#pragma unroll
		for (int f = 0; f < 36; ++f) asm volatile("" : "+a"(wb[f]));
		bias = loadBias(1);
		// the first conv B pair's skip values (the later pairs fetch theirs behind their own last macro-step)
		skipLane = static_cast<unsigned>(min(x0 + px, p.W - 1) * 64 + cb * 32 + 4 * hh) * 2u;
		loadSkip2(y0 + 2 * pl, 0, 0);
		loadSkip2(y0 + 2 * pl, 0, 2);
		loadSkip2(y0 + 2 * pl, 1, 0);
		loadSkip2(y0 + 2 * pl, 1, 2);
		RB_STAMP(t4);
		RB_ADD(4, t3, t4);
		__syncthreads();  // T complete, X dead
		RB_STAMP(t5);
		RB_ADD(2, t4, t5);
		// The next tile's X: an interior tile's 20 LDS-DMA instructions go out behind the MFMAs of this wave's
		// first conv B K loop (all four waves issuing them at once fill the CU's vector-memory queue and stall
		// ~3 k cycles); an edge tile takes the general path here.
		const int nextTile = tile + static_cast<int>(gridDim.x);
		bool xBehind = false;
		unsigned xSo = 0;
		if (nextTile < p.numTiles) {
			if (tileIsBorder(nextTile)) {
				stageX(nextTile);
			} else {
				int nty, ntx;
				tileXY(nextTile, nty, ntx);
				xSo = static_cast<unsigned>((nty * kRbTH - 2) * p.inPitch + (ntx * kFbOutW - 2)) * 128u;
				xBehind = true;
			}
		}
		RB_STAMP(t6);
		RB_ADD(5, t5, t6);
		// ---- conv B: pairs pl, pl + 2, ... < 7; the LAST pair accumulates in S1 (4 pairs: S0 S1 S0 S1; 3: S1 S0 S1) ----
		// A pair fetches its OWN skip values behind its last macro-step (the previous pair's were consumed by
		// macro-step 10), so one set of skip registers serves the pipeline.
		const int nB = pl == 0 ? 4 : 3;
		auto runB = [&](f32x16 (&acc)[2], const f32x16 (&prevAcc)[2], const int q, const bool hasPrev, auto lastTag) __attribute__((always_inline)) {
			constexpr bool last = decltype(lastTag)::value;  // (profiling only)
			const int pair = pl + 2 * q;
			const f32x16 e0 = prevAcc[0], e1 = prevAcc[1];
			const int gyOwn = y0 + 2 * pair;
			RB_STAMP(tb0);
			rbPipeRun<T, false>(ldsBase + kRbX + (2 * pair) * (kFbW * 128), colOff, colSwz, hh, wb, rsrcWa, wLane, wBase, acc, bias, noStart,
			    [&](int m, int k) __attribute__((always_inline)) {
				    // (the finished pair's stores behind macro-step 4: in front of macro-step 3 they would be in flight
				    // when the skip values are waited for, and a wait with loads AND stores pending is vmcnt(0))
				    if (m == 4 && k == 0) storeRows();
				    if (hasPrev) {
					    epiB(e0, e1, m, k);
					    if (m == 11 && k >= 1 && k < 5) loadSkip2(gyOwn, (k - 1) >> 1, ((k - 1) & 1) * 2);
				    } else if (xBehind) {
					    // (the first conv B pair has no epilogue to run: the next tile's X instead.  Behind the LAST
					    // conv B loop instead -- so that no DMA stands in front of the later waits for skip values --
					    // that loop took 6.2 k cycles and the middle ones were no faster.)
					    if (k == 2) stageXOne(xSo, m);
					    else if (k == 5 && m < kXPerWave - 12) stageXOne(xSo, 12 + m);
				    }
			    });
			RB_STAMP(tb1);
			RB_ADD((q == 0 ? 1 : (last ? 7 : 6)), tb0, tb1);
			if (hasPrev) {  // (the pair before this one is now in stg0..3)
				stY = y0 + 2 * (pair - 2);
				stX = x0;
				stOn = true;
			}
		};
		if (nB == 4) {
			runB(S0, S1, 0, false, std::false_type{});
			runB(S1, S0, 1, true, std::false_type{});
			runB(S0, S1, 2, true, std::false_type{});
			runB(S1, S0, 3, true, std::true_type{});
		} else {
			runB(S1, S0, 0, false, std::false_type{});
			runB(S0, S1, 1, true, std::false_type{});
			runB(S1, S0, 2, true, std::true_type{});
		}
		epY = y0 + 2 * (pl + 2 * (nB - 1));
		epX = x0;
		epOn = true;
	}
	RB_STAMP(tLoop);
	// ---- drain: the last tile's last pair ----
	asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	storeRows();
	if (epOn) {
#pragma unroll
		for (int m = 3; m < 12; ++m) {
#pragma unroll
			for (int k = 0; k < 6; ++k) epiB(S1[0], S1[1], m, k);
		}
		stY = epY;
		stX = epX;
		stOn = true;
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		storeRows();
	}
#ifdef JU_RB_PROF
	{
		const unsigned long long tEnd = stamp();
		if (blockIdx.x == 0 && lane == 0 && p.prof != nullptr) {
			for (int k = 0; k < 8; ++k) p.prof[wave * 8 + k] = prof[k];
			if (wave == 0) p.prof[32] = tEnd - tKernel;
		}
	}
#endif
}

// JU_RES_BLOCK=plain (or the tests' switch): res_block_kernel for every block
static std::atomic<int> g_ResBlockPlain{[] { const char *e = std::getenv("JU_RES_BLOCK"); return (e != nullptr && std::string(e) == "plain") ? 1 : 0; }()};
}  // namespace
void setResBlockPlain(int on) { g_ResBlockPlain = on ? 1 : 0; }
bool resBlockPlain() { return g_ResBlockPlain.load() != 0; }
namespace {

template <typename T>
void launchResBlockT(const FlowBlockLaunch &q, hipStream_t stream) {
	// the pipelined form (epilogues behind the next pair's MFMAs); JU_RES_BLOCK=plain: the plain kernel (tests, A/B).
	// A slope outside [0, 1] (no model the container accepts has one) also takes the plain kernel.
	const bool pipe = !resBlockPlain() && q.act1 == 1 && q.act2 == 1 && ablationSkipBits() == 0;  // (ReLU blocks)
	auto kern = pipe ? res_block_pipe_kernel<T> : res_block_kernel<T>;
	const int ldsBytes = pipe ? kRpLds : kRbLds;
	static std::atomic<std::uint64_t> ldsDone{0}, ldsDonePipe{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), ldsBytes, pipe ? &ldsDonePipe : &ldsDone, "res block");
	const int cus = currentDeviceCUs();
	ResBlockParams p{};
	p.in = q.in;
	p.out = q.out;
	p.w1 = q.w1;
	p.w2 = q.w2;
	p.b1 = q.b1;
	p.b2 = q.b2;
	p.H = q.H;
	p.W = q.W;
	p.inPitch = q.inPitch ? q.inPitch : q.W;
	p.outPitch = q.outPitch ? q.outPitch : q.W;
	p.tilesX = (q.W + kFbOutW - 1) / kFbOutW;
	p.numTiles = p.tilesX * ((q.H + kRbTH - 1) / kRbTH);
	p.s1 = q.act1 == 1 ? 0.0f : (q.act1 == 2 ? q.slope : 1.0f);
	p.s2 = q.act2 == 1 ? 0.0f : (q.act2 == 2 ? q.slope : 1.0f);
	p.skip = ablationSkipBits();
	const int grid = p.numTiles < cus ? p.numTiles : cus;
#ifdef JU_RB_PROF
	static unsigned long long *profBuf = [] { void *b = nullptr; (void)hipMalloc(&b, 64 * 8); return static_cast<unsigned long long *>(b); }();
	static int profLaunches = 0;
	p.prof = profBuf;
#endif
	hipLaunchKernelGGL(kern, dim3(grid), dim3(256), ldsBytes, stream, p);
	hipCheckLaunch("res_block");
#ifdef JU_RB_PROF
	if (pipe && ++profLaunches == 200) {  // (developer build: one dump, of a warm launch)
		unsigned long long h[33];
		(void)hipDeviceSynchronize();
		(void)hipMemcpy(h, profBuf, sizeof(h), hipMemcpyDeviceToHost);
		const char *names[8] = {"prologue", "B first loop (no hooks)", "X wait + barriers", "conv A loops", "A3 epilogue + skip", "stage X issue", "B middle loops", "B last loop (stream)"};
		std::fprintf(stderr, "res_block_pipe workgroup 0: %llu ticks in the kernel\n", h[32]);
		for (int wv = 0; wv < 4; ++wv) {
			std::fprintf(stderr, " wave %d:", wv);
			for (int k = 0; k < 8; ++k) std::fprintf(stderr, " %s %llu,", names[k], h[wv * 8 + k]);
			std::fprintf(stderr, "\n");
		}
	}
#endif
}

// ---------------------------------------------------------------------------
// conv_splitk_kernel: one 3x3 convolution of the COARSE levels of the flow net
// (68x120 and 34x60 at 480x270: 128-256 channels, 2-8 k pixels), K split over the waves
// ---------------------------------------------------------------------------
// These layers are 1-5 GFLOP on a few thousand pixels: as conv_mfma_kernel launches they
// were latency chains (7-14 us each: per 64-channel chunk a global -> register -> LDS
// staging round for BOTH operands, 2-4 chunks in series, 40-140 workgroups).  Here one
// workgroup (8 waves) owns a tile of 32 columns x TH rows x 32*CB output channels and the
// eight waves split the INPUT channels: wave w keeps the A fragments of its CIN/8 channels
// (9 taps x KS k-steps x CB cout blocks) in registers for the whole tile, so every weight
// byte enters the CU once and all of it is in flight at once -- one load round trip instead
// of one per chunk.  The tile is a ring of four input rows per 64-channel plane (LDS-DMA,
// out-of-image pixels fetched from a zero page, so no fill pass): while a row pair's
// partial sums are reduced, the next pair's two rows land on the two rows it no longer
// needs.  Reduction: wave o owns piece o = (row o / 4, cout group o % 4) of the 32 x 32
// partial tile; every wave sends it the matching piece of its accumulators through LDS
// (7 KB per wave), the owner adds the eight pieces in wave order, applies bias and
// activation and puts its 4 values per lane into a staging tile, from which the
// workgroup stores whole 16-byte chunks (or their 2x2 maxima: POOL).
// fp32 accumulation per wave from zero, partial sums added in wave order, bias last: the
// summation order differs from conv_mfma_kernel's, nothing else (JU_FLOW_CONV=generic
// keeps that kernel; tests compare both).
struct SplitKParams {
	const void *in;     // NHWC [H][W][CIN]
	const void *wgt;    // packConvWeights(nb = 1): [COUT/32][CIN/64][9][4][2][32][8]
	const float *bias;  // [COUT]
	void *out;          // [H][W][COUT], or [H/2][W/2][COUT] with POOL
	const void *zeros;  // >= 16 zero bytes in device memory (source of out-of-image pixels)
	int H, W, cout;
	int inPitch, outPitch;
	int tilesX, TH;
	int act;
	float slope;
	int skip;  // timing ablation (JU_FB_SKIP, developer only): 1 weights, 2 staging, 4 K loops, 8 reduction, 16 stores
};

template <int CIN, int CB>
struct SkGeom {
	static constexpr int NPL = CIN / 64;
	static constexpr int KS = CIN / 128;  // k-steps (16 channels) per wave
	static constexpr int XPLANE = 4 * kFbW * 128;
	static constexpr int OFF_P = NPL * XPLANE;
	static constexpr int PBYTES = CB * 8 * 7 * 1024;
	static constexpr int OFF_S = OFF_P + PBYTES;
	static constexpr int SBYTES = CB * 2 * 32 * 64;
	static constexpr int LDS = OFF_S + SBYTES;
	static_assert(CIN == 128 || CIN == 256, "input channels");
	static_assert(LDS <= 160 * 1024, "tile does not fit LDS");
};

template <typename T, int CIN, int CB, bool POOL>
__global__ __launch_bounds__(512, 1) void conv_splitk_kernel(SplitKParams p) {
	using G = SkGeom<CIN, CB>;
	constexpr int KS = G::KS;
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int tx = blockIdx.x % p.tilesX, ty = blockIdx.x / p.tilesX;
	const int x0 = tx * 32, y0 = ty * p.TH;
	const int cog0 = blockIdx.y * CB;
	const int nPairs = (min(p.TH, p.H - y0) + 1) >> 1;
	const T *__restrict__ in = static_cast<const T *>(p.in);
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));
	// this wave's input channels: k-steps wave*KS .. +KS of the CIN/16
	const int plane = (wave * KS) >> 2;
	const int ksBase = (wave * KS) & 3;

	// ---- A fragments: CB cout blocks x 9 taps x KS k-steps, straight to registers ----
	Vec8<T> w[CB][9 * KS];
#pragma unroll
	for (int b = 0; b < CB; ++b) {
		const unsigned char *wsrc = static_cast<const unsigned char *>(p.wgt) +
		    ((size_t)((cog0 + b) * G::NPL + plane) * 9 * 4 + ksBase) * 1024 + lane * 16;
#pragma unroll
		for (int tap = 0; tap < 9; ++tap) {
#pragma unroll
			for (int ks = 0; ks < KS; ++ks) {
				if (!(JU_SKIP(p) & 1)) w[b][tap * KS + ks] = *reinterpret_cast<const Vec8<T> *>(wsrc + (size_t)(tap * 4 + ks) * 1024);
				else w[b][tap * KS + ks] = Vec8<T>{};
			}
		}
	}
	// ---- tile rows [t0, t0 + nr) -> ring slots (t mod 4), all planes ----
	// tile row t = image row y0 - 1 + t, tile column k = image column x0 - 1 + k
	auto stageRows = [&](int t0, int nr) {
		if (JU_SKIP(p) & 2) return;
		const int nPix = nr * kFbW;
		const int nInstr = (nPix + 7) >> 3;  // 8 pixels of one plane per wave-instruction
		const int slot0 = t0 & 3;            // (t0 is even: the rows of a stage never wrap)
		for (int j = wave; j < G::NPL * nInstr; j += 8) {
			const int pl = j / nInstr, i = j - pl * nInstr;
			const int q = i * 8 + (lane >> 3);
			const int r = q / kFbW, k = q - r * kFbW;
			const int gy = y0 - 1 + t0 + r, gx = x0 - 1 + k;
			const unsigned c = static_cast<unsigned>(lane & 7) ^ fbSwz<128>(k);
			if (q < nPix) {
				const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
				const void *src = inside ? static_cast<const void *>(in + ((size_t)gy * p.inPitch + gx) * CIN + pl * 64 + c * 8)
				                         : p.zeros;
				fbGlds16(src, smem + pl * G::XPLANE + slot0 * (kFbW * 128) + i * 1024);
			}
		}
	};
	stageRows(0, 4);

	unsigned colOff[3], colSwz[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		colOff[dx] = (px + dx) * 128;
		colSwz[dx] = fbSwz<128>(px + dx);
	}
	const int hhx = ksBase * 2 + hh;  // chunk index of this wave's first k-step, this lane's half
	// owner role: piece (row orow, cout group og) of every cout block
	const int orow = wave >> 2, og = wave & 3;
	f32x4 biasv[CB];
#pragma unroll
	for (int b = 0; b < CB; ++b) {
		biasv[b] = *reinterpret_cast<const f32x4 *>(p.bias + (cog0 + b) * 32 + 8 * og + 4 * hh);
	}
	const float sAct = fbActS(p.act, p.slope);
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	const unsigned xBase = ldsBase + plane * G::XPLANE;
	for (int pr = 0; pr < nPairs; ++pr) {
		f32x16 acc[CB][2];
#pragma unroll
		for (int b = 0; b < CB; ++b) {
#pragma unroll
			for (int r = 0; r < 2; ++r) {
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[b][r][i] = 0.0f;
			}
		}
#pragma unroll
		for (int b = 0; b < (JU_SKIP(p) & 4 ? 0 : CB); ++b) {
			if (pr & 1) FbPair<T, KS, 128>::template run<2>(xBase, colOff, colSwz, hhx, w[b], acc[b]);
			else FbPair<T, KS, 128>::template run<0>(xBase, colOff, colSwz, hhx, w[b], acc[b]);
		}
		__syncthreads();  // B: rows 2pr, 2pr+1 are dead, the previous pair's pieces and staging tile too
		if (pr + 1 < nPairs) stageRows(2 * pr + 4, 2);
		// ---- pieces to their owners ----
#pragma unroll
		for (int b = 0; b < (JU_SKIP(p) & 8 ? 0 : CB); ++b) {
#pragma unroll
			for (int o = 0; o < 8; ++o) {
				if (o != wave) {
					const int slot = wave - (wave > o ? 1 : 0);
					const f32x4 v = {acc[b][o >> 2][4 * (o & 3)], acc[b][o >> 2][4 * (o & 3) + 1],
					    acc[b][o >> 2][4 * (o & 3) + 2], acc[b][o >> 2][4 * (o & 3) + 3]};
					*reinterpret_cast<f32x4 *>(smem + G::OFF_P + ((b * 8 + o) * 7 + slot) * 1024 + lane * 16) = v;
				}
			}
		}
		__syncthreads();  // C
		// ---- owner: sum in wave order, bias, activation, 16-bit, staging tile ----
#pragma unroll
		for (int b = 0; b < CB; ++b) {
			f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int src = 0; src < (JU_SKIP(p) & 8 ? 0 : 8); ++src) {
				f32x4 v;
				if (src == wave) {
					// (this wave's own piece: select by the uniform owner index)
					f32x4 own = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
					for (int o = 0; o < 8; ++o) {
						if (o == wave) {
							own = f32x4{acc[b][o >> 2][4 * (o & 3)], acc[b][o >> 2][4 * (o & 3) + 1],
							    acc[b][o >> 2][4 * (o & 3) + 2], acc[b][o >> 2][4 * (o & 3) + 3]};
						}
					}
					v = own;
				} else {
					const int slot = src - (src > wave ? 1 : 0);
					v = *reinterpret_cast<const f32x4 *>(smem + G::OFF_P + ((b * 8 + wave) * 7 + slot) * 1024 + lane * 16);
				}
				sum += v;
			}
			const Vec4<T> o16 = pack4<T>(fbAct(sum[0] + biasv[b][0], sAct), fbAct(sum[1] + biasv[b][1], sAct),
			    fbAct(sum[2] + biasv[b][2], sAct), fbAct(sum[3] + biasv[b][3], sAct));
			// staging tile [b][row][px][32 couts], the 16-byte chunk index swizzled by the column
			*reinterpret_cast<Vec4<T> *>(smem + G::OFF_S + ((b * 2 + orow) * 32 + px) * 64 +
			    ((static_cast<unsigned>(og) ^ ((px >> 2) & 3u)) << 4) + hh * 8) = o16;
		}
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the next pair's rows)
		__syncthreads();  // D
		// ---- staging tile -> global, 16 bytes per thread ----
		T *out = static_cast<T *>(p.out);
		if constexpr (!POOL) {
			if (tid < CB * 256) {
				const int q = tid & 3, cx = (tid >> 2) & 31, r = (tid >> 7) & 1, b = tid >> 8;
				const int gy = y0 + 2 * pr + r, gx = x0 + cx;
				if (gy < p.H && gx < p.W && !(JU_SKIP(p) & 16)) {
					const uint4 v = *reinterpret_cast<const uint4 *>(smem + G::OFF_S + ((b * 2 + r) * 32 + cx) * 64 +
					    ((static_cast<unsigned>(q) ^ ((cx >> 2) & 3u)) << 4));
					*reinterpret_cast<uint4 *>(out + ((size_t)gy * p.outPitch + gx) * p.cout + (cog0 + b) * 32 + q * 8) = v;
				}
			}
		} else {
			if (tid < CB * 64) {
				const int q = tid & 3, kx = (tid >> 2) & 15, b = tid >> 6;
				const int gy = (y0 >> 1) + pr, gx = (x0 >> 1) + kx;
				if (gy < (p.H >> 1) && gx < (p.W >> 1)) {
					Vec8<T> m;
#pragma unroll
					for (int r = 0; r < 2; ++r) {
#pragma unroll
						for (int d = 0; d < 2; ++d) {
							const int cx = 2 * kx + d;
							const Vec8<T> v = *reinterpret_cast<const Vec8<T> *>(smem + G::OFF_S + ((b * 2 + r) * 32 + cx) * 64 +
							    ((static_cast<unsigned>(q) ^ ((cx >> 2) & 3u)) << 4));
#pragma unroll
							for (int j = 0; j < 8; ++j) {
								m[j] = (r == 0 && d == 0) ? v[j] : (static_cast<float>(v[j]) > static_cast<float>(m[j]) ? v[j] : m[j]);
							}
						}
					}
					*reinterpret_cast<Vec8<T> *>(out + ((size_t)gy * p.outPitch + gx) * p.cout + (cog0 + b) * 32 + q * 8) = m;
				}
			}
		}
	}
}

template <typename T, int CIN, int CB, bool POOL>
void launchSplitKInst(const SplitKParams &p, int tilesY, hipStream_t stream) {
	using G = SkGeom<CIN, CB>;
	auto kern = conv_splitk_kernel<T, CIN, CB, POOL>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), G::LDS, &ldsDone, "split-K conv");
	hipLaunchKernelGGL(kern, dim3(p.tilesX * tilesY, p.cout / (32 * CB)), dim3(512), G::LDS, stream, p);
	hipCheckLaunch("conv_splitk");
}

// 8 waves where the widest instantiation stays within 256 registers (two 64-channel
// planes of conv A weights do not)
template <int CIN>
constexpr int fbWaves() {
	return CIN > 64 ? 4 : 8;
}

template <typename T, int CIN, int CMID, int TH, bool UPS, bool POOL, int OUTK, bool PACK = false>
void launchFlowBlockInst(const FlowBlockParams &p, hipStream_t stream) {
	constexpr int NW = fbWaves<CIN>();
	using G = FbGeom<CIN, CMID, TH, UPS, POOL, OUTK, NW>;
	static_assert(G::FITS, "tile does not fit LDS");
	auto kern = flow_block_kernel<T, CIN, CMID, TH, UPS, POOL, OUTK, NW, PACK>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), G::LDS, &ldsDone, "flow block");
	dim3 grid((p.W + kFbOutW - 1) / kFbOutW, (p.H + TH - 1) / TH);
	hipLaunchKernelGGL(kern, grid, dim3(NW * 64), G::LDS, stream, p);
	hipCheckLaunch("flow_block");
}

// Tile height: the tall tile (18 rows: 11 % recompute ring) when the launch then still
// fills the chip, the short one (6 rows) for small tensors.
constexpr int kFbTall = 18, kFbMid = 14, kFbShort = 6;

template <typename T, int CIN, int CMID, bool UPS, bool POOL, int OUTK, bool PACK = false>
void launchFlowBlockT(const FlowBlockParams &p, int numCUs, hipStream_t stream) {
	const long tilesX = (p.W + kFbOutW - 1) / kFbOutW;
	if constexpr (PACK) {
		if (tilesX * ((p.H + kFbTall - 1) / kFbTall) * 10 >= 7L * numCUs) {
			return launchFlowBlockInst<T, CIN, CMID, kFbTall, UPS, POOL, OUTK, true>(p, stream);
		}
		return launchFlowBlockInst<T, CIN, CMID, kFbShort, UPS, POOL, OUTK, true>(p, stream);
	} else if constexpr (FbGeom<CIN, CMID, kFbTall, UPS, POOL, OUTK, fbWaves<CIN>()>::FITS) {
		if (tilesX * ((p.H + kFbTall - 1) / kFbTall) * 10 >= 7L * numCUs) {
			return launchFlowBlockInst<T, CIN, CMID, kFbTall, UPS, POOL, OUTK>(p, stream);
		}
	} else if constexpr (FbGeom<CIN, CMID, kFbMid, UPS, POOL, OUTK, fbWaves<CIN>()>::FITS && OUTK == 2) {
		// (64 -> 64 -> 64: two 128-byte tiles; 14 rows is what fits)
		if (tilesX * ((p.H + kFbMid - 1) / kFbMid) * 10 >= 7L * numCUs) {
			return launchFlowBlockInst<T, CIN, CMID, kFbMid, UPS, POOL, OUTK>(p, stream);
		}
	}
	launchFlowBlockInst<T, CIN, CMID, kFbShort, UPS, POOL, OUTK>(p, stream);
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
	p.frame = q.packFrame;
	p.frameStride = q.packFrameStride;
	p.packPrev = q.packPrev;
	p.packOut = q.packOut;
	p.frameH = q.frameH;
	p.frameW = q.frameW;
	p.padTop = q.padTop;
	p.padLeft = q.padLeft;
	p.numInputs = q.numInputs;
	p.sums = q.sums;
	const int cus = currentDeviceCUs();
	if (q.packOut != nullptr) {
		if (!(q.cin == 16 && q.cmid == 32 && !q.upsample && q.pool && !q.outHead && !q.residual && q.packFrame && q.packPrev)) {
			throw std::invalid_argument("flow block: input packing is built for the first block (16 -> 32 -> 32, pool)");
		}
		return launchFlowBlockT<T, 16, 32, false, true, 0, true>(p, cus, stream);
	}
	if (q.upsample && (q.H % 2 || q.W % 2)) throw std::invalid_argument("flow block: fused upsampling needs even H and W");
	if (q.pool && (q.H % 2 || q.W % 2)) throw std::invalid_argument("flow block: fused max-pool needs even H and W");
	// the shapes of the flow auto-encoder's fusable blocks (flowBlockSupported)
	const int outk = q.residual ? 2 : (q.outHead ? 1 : 0);
#define JU_FB_CASE(CIN_, CMID_, UPS_, POOL_, OUTK_)                                              \
	if (q.cin == CIN_ && q.cmid == CMID_ && q.upsample == UPS_ && q.pool == POOL_ && outk == OUTK_) { \
		return launchFlowBlockT<T, CIN_, CMID_, UPS_, POOL_, OUTK_>(p, cus, stream);                \
	}
	JU_FB_CASE(16, 32, false, true, 0)   // encoder block 1: 12(16) -> 32 -> 32, pool
	JU_FB_CASE(32, 64, false, true, 0)   // encoder block 2: 32 -> 64 -> 64, pool
	JU_FB_CASE(128, 64, true, false, 0)  // last decoder block: up(128) -> 64 -> 64
	JU_FB_CASE(128, 64, false, false, 0)
	JU_FB_CASE(64, 32, true, false, 1)   // head: up(64) -> 32 -> 32 (f16 flow head)
	JU_FB_CASE(64, 32, false, false, 1)
	JU_FB_CASE(64, 64, false, false, 2)  // res_block of a 64-filter tower (generator, flow-resnet)
#undef JU_FB_CASE
	throw std::invalid_argument("flow block: unsupported shape");
}

}  // namespace

bool flowBlockSupported(int cin, int cmid, bool upsample, bool pool, bool outHead) {
	if (cin == 64 && cmid == 64) return false;  // (only as a residual block: FlowBlockLaunch::residual)
	if (cin == 16 && cmid == 32) return !upsample && pool && !outHead;
	if (cin == 32 && cmid == 64) return !upsample && pool && !outHead;
	if (cin == 128 && cmid == 64) return !pool && !outHead;
	if (cin == 64 && cmid == 32) return !pool && outHead;
	return false;
}

void launchFlowBlock(DType dt, const FlowBlockLaunch &q, hipStream_t stream) {
	if (q.residual && q.cin == 64 && q.cmid == 64 && !q.upsample && !q.pool) {
		static const char *mode = std::getenv("JU_RES_BLOCK");  // "tile": the non-persistent flow_block_kernel form (A/B)
		if (!(mode && std::string(mode) == "tile")) {
			if (dt == kF16) launchResBlockT<f16>(q, stream);
			else launchResBlockT<bf16>(q, stream);
			return;
		}
	}
	if (dt == kF16) launchFlowBlockDT<f16>(q, stream);
	else launchFlowBlockDT<bf16>(q, stream);
}

bool convSplitKSupported(const ConvParams &p) {
	// (a pooled 256-channel layer does not occur in the flow net: not instantiated, not claimed)
	return p.taps == 9 && (p.cin == 128 || p.cin == 256) && p.cout % 32 == 0 && !p.res && !p.outHead && !p.upsample &&
	       p.nb == 1 && (!p.pool || (p.cin == 128 && p.H % 2 == 0 && p.W % 2 == 0)) && p.H * p.W <= 32768;
}

void launchConvSplitK(DType dt, const ConvParams &q, const void *zeros, hipStream_t stream) {
	if (!convSplitKSupported(q) || !zeros) throw std::invalid_argument("split-K conv: unsupported layer");
	const int cus = currentDeviceCUs();
	SplitKParams p{};
	p.in = q.in;
	p.wgt = q.wgt;
	p.bias = q.bias;
	p.out = q.out;
	p.zeros = zeros;
	p.H = q.H;
	p.W = q.W;
	p.cout = q.cout;
	p.inPitch = q.inPitch ? q.inPitch : q.W;
	p.outPitch = q.outPitch ? q.outPitch : (q.pool ? q.W / 2 : q.W);
	p.act = q.relu;
	p.slope = q.slope;
	p.skip = ablationSkipBits();
	p.tilesX = (q.W + 31) / 32;
	// Tile height and cout blocks per workgroup: every workgroup pulls its cout blocks'
	// whole weights (147 KB per block at 256 channels), so the fewest workgroups that still
	// fill most of the chip in ONE round; two cout blocks per workgroup (128 channels only:
	// LDS) when even the tallest tile leaves more workgroups than CUs.
	const int nCog = q.cout / 32;
	int cb = 1, th = 2;
	for (;;) {
		bool found = false;
		for (th = 2; th <= 16; th += 2) {
			if ((long)p.tilesX * ((q.H + th - 1) / th) * (nCog / cb) <= cus) {
				found = true;
				break;
			}
		}
		if (found || cb == 2 || q.cin != 128 || nCog % 2) break;
		cb = 2;
	}
	if (th > 16) th = 16;
	p.TH = th;
	const int tilesY = (q.H + th - 1) / th;
	const bool f16t = dt == kF16;
#define JU_SK_CASE(CIN_, CB_, POOL_)                                                       \
	if (q.cin == CIN_ && cb == CB_ && (q.pool != 0) == POOL_) {                              \
		if (f16t) launchSplitKInst<f16, CIN_, CB_, POOL_>(p, tilesY, stream);                  \
		else launchSplitKInst<bf16, CIN_, CB_, POOL_>(p, tilesY, stream);                      \
		return;                                                                              \
	}
	JU_SK_CASE(128, 1, false)
	JU_SK_CASE(128, 1, true)
	JU_SK_CASE(128, 2, false)
	JU_SK_CASE(128, 2, true)
	JU_SK_CASE(256, 1, false)
#undef JU_SK_CASE
	throw std::invalid_argument("split-K conv: unsupported shape");
}

}  // namespace ju
