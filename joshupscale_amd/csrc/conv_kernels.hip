// gfx950 (CDNA4, MI355X): the generic implicit-GEMM convolution of the flow net.
//
//  * conv_mfma_kernel   3x3 / 1x1 convolution on v_mfma_f32_32x32x16_{f16,bf16}:
//                       weights are the MFMA A operand (M = cout), activations the
//                       B operand (N = 32 consecutive pixels of one image row), so
//                       each lane's accumulator registers hold 4 consecutive output
//                       channels of ONE pixel and the NHWC store is 8 B/lane.  Input
//                       tile (+halo) and the weight chunk are staged in LDS; the tile
//                       is XOR-swizzled per 16-B chunk so the ds_read_b128 fragment
//                       reads are bank-conflict free.  Optional fused 2x2 max-pool
//                       (epilogue) and bilinear x2 upsampling (staging).
//
// What it computes follows the reference's Keras graph (scripts/training/models.py
// :257-481).
#include "kernel_common.h"

namespace ju {

namespace {

// ---------------------------------------------------------------------------
// implicit-GEMM convolution
// ---------------------------------------------------------------------------
constexpr int kTW = 32;       // tile width in pixels = MFMA N
constexpr int kConvThreads = 256;

// XOR swizzle of the 16-byte chunk index inside a pixel's CK channels.  P =
// chunks per pixel.  16 lanes of one ds_read_b128 group read the same logical
// chunk of 16 pixels whose LDS pixel indices are distinct mod 16; after the
// swizzle they fall on 16 distinct 16-byte slots of the 256-byte bank row.
template <int P>
__device__ __forceinline__ int swz(int q) {
	if constexpr (P == 8) return (q >> 1) & 7;
	else if constexpr (P == 4) return (q >> 2) & 3;
	else return (q >> 3) & 1;
}

// LDS -> register fragments of MFMA step s (tap s / KS, k-step s % KS) of a staged
// channel chunk: A = kernel-ready weights, B = 32 pixels of each of the wave's rows.
// asm reads: hipcc sinks plain LDS loads back in front of their MFMA (one LDS round
// trip per MFMA); the consumer waits with convWait.  wAddr/iAddr are LDS byte
// addresses of the stage's weights / tile.
template <typename T, int TAPS, int CK, int NB, int RW>
__device__ __forceinline__ void convFetch(unsigned wAddr, unsigned iAddr, int s, Vec8<T> (&a)[NB],
    Vec8<T> (&b)[RW], int wave, int px, int hh) {
	constexpr int HALO = (TAPS == 9) ? 1 : 0;
	constexpr int IW = kTW + 2 * HALO;
	constexpr int P = CK / 8;
	constexpr int KS = CK / 16;
	constexpr int COG = 32 * NB;
	const int tap = s / KS, ks = s % KS;
	const int dy = (TAPS == 9) ? tap / 3 : 0;
	const int dx = (TAPS == 9) ? tap % 3 : 0;
	// weights: one lane base + an immediate offset (two bases: the field is 16 bits)
#pragma unroll
	for (int nb = 0; nb < NB; ++nb) {
		constexpr int kHi = 40960;
		const int off = (((tap * KS + ks) * 2) * COG + nb * 32) << 4;
		if (off < 65536) {
			asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[nb]) : "v"(wAddr), "n"(off));
		} else {
			asm volatile("ds_read_b128 %0, %1 offset:%2"
			             : "=v"(a[nb]) : "v"(wAddr + kHi), "n"(off - kHi));
		}
	}
#pragma unroll
	for (int rw = 0; rw < RW; ++rw) {
		const int q = (wave * RW + rw + dy) * IW + px + dx;
		const int c = ks * 2 + hh;
		const unsigned addr = iAddr + q * (CK * 2) + ((c ^ swz<P>(q)) << 4);
		asm volatile("ds_read_b128 %0, %1" : "=v"(b[rw]) : "v"(addr));
	}
}

// s_waitcnt lgkmcnt(N) that the fragments' consumers cannot be hoisted over (the
// registers are tied through the asm).
template <int N, typename V, int NA, int NBB>
__device__ __forceinline__ void convWait(V (&a)[NA], V (&b)[NBB]) {
	static_assert(N >= 0 && N <= 15, "lgkmcnt range");
#pragma unroll
	for (int i = 0; i < NA; ++i) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a[i]) : "n"(N));
#pragma unroll
	for (int i = 0; i < NBB; ++i) asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(b[i]) : "n"(N));
}

// MFMAs of one staged channel chunk, software-pipelined over the TAPS*KS steps: the
// fragments of step s+1 are read into the other register set before the MFMAs of
// step s issue, so a wave does not sit out an LDS round trip per MFMA.
template <typename T, int TAPS, int CK, int NB, int RW>
__device__ __forceinline__ void convChunkMfma(const unsigned char *smW, const unsigned char *smI,
    f32x16 (&acc)[NB][RW], int wave, int px, int hh) {
	constexpr int S = TAPS * (CK / 16);
	constexpr int COG = 32 * NB;
	unsigned wAddr = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (const __attribute__((address_space(3))) unsigned char *)smW)) + ((hh * COG + px) << 4);
	unsigned iAddr = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (const __attribute__((address_space(3))) unsigned char *)smI));
	// opaque per call: otherwise every read address of every step is computed once,
	// outside the chunk loop, and parked in ~100 VGPRs
	asm volatile("" : "+v"(wAddr), "+v"(iAddr));
	Vec8<T> a0[NB], b0[RW], a1[NB], b1[RW];
	convFetch<T, TAPS, CK, NB, RW>(wAddr, iAddr, 0, a0, b0, wave, px, hh);
#pragma unroll
	for (int s = 0; s < S; s += 2) {
		if (s + 1 < S) {
			convFetch<T, TAPS, CK, NB, RW>(wAddr, iAddr, s + 1, a1, b1, wave, px, hh);
			convWait<NB + RW>(a0, b0);
		} else {
			convWait<0>(a0, b0);
		}
#pragma unroll
		for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
			for (int nb = 0; nb < NB; ++nb) acc[nb][rw] = mfma32(a0[nb], b0[rw], acc[nb][rw]);
		}
		if (s + 1 < S) {
			if (s + 2 < S) {
				convFetch<T, TAPS, CK, NB, RW>(wAddr, iAddr, s + 2, a0, b0, wave, px, hh);
				convWait<NB + RW>(a1, b1);
			} else {
				convWait<0>(a1, b1);
			}
#pragma unroll
			for (int rw = 0; rw < RW; ++rw) {
#pragma unroll
				for (int nb = 0; nb < NB; ++nb) acc[nb][rw] = mfma32(a1[nb], b1[rw], acc[nb][rw]);
			}
		}
	}
}

// DBUF (layers with several channel chunks, two stages must fit LDS): the global
// loads of chunk c+1 are issued into registers before the MFMAs of chunk c and
// written to the other LDS stage afterwards -- one barrier per chunk, memory
// latency hidden behind the matrix cores.  Otherwise staging is synchronous and a
// second resident workgroup provides the overlap.
//
// UPS (flow decoder, models.py:412-447): the input tensor is the HALF-resolution
// activation [H/2][W/2][cin] and the TF1 bilinear x2 upsampling (keras_layers.py:12-61,
// src = dst/2, edge clamp) happens while the tile is staged: the low-resolution
// patch under the tile goes to LDS once, then every thread builds its tile elements
// from 4 LDS reads with the arithmetic of upsample2_kernel (same rounding: the
// fused and the two-kernel paths are bit-identical).  No 4x larger tensor in HBM,
// no upsample launch.
template <typename T, int TAPS, int CK, int NB, int RW, bool DBUF, bool UPS = false>
__global__ __launch_bounds__(kConvThreads) void conv_mfma_kernel(ConvParams p) {
	// (a look-ahead launch, ConvParams::items > 1: grid.z = frames x cout groups, the frame's tensors `item` strides on)
	int cogOfBlock = blockIdx.z;
	if (p.items > 1) {
		const int ncog = gridDim.z / p.items;
		const int item = cogOfBlock / ncog;
		cogOfBlock -= item * ncog;
		p.in = static_cast<const unsigned char *>(p.in) + item * p.inItemBytes;
		p.out = static_cast<unsigned char *>(p.out) + item * p.outItemBytes;
	}
	constexpr int HALO = (TAPS == 9) ? 1 : 0;
	constexpr int TH = 4 * RW;            // tile rows: 4 waves x RW rows each
	constexpr int IW = kTW + 2 * HALO;    // staged tile incl. halo
	constexpr int IH = TH + 2 * HALO;
	constexpr int P = CK / 8;             // 16-B chunks per pixel
	constexpr int COG = 32 * NB;          // couts per workgroup
	constexpr int W_BYTES = TAPS * CK * COG * 2;

	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *smW = smem;
	unsigned char *smI = smem + W_BYTES;

	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int tx0 = blockIdx.x * kTW;
	const int ty0 = blockIdx.y * TH;
	const int cog = cogOfBlock;
	const int nCC = p.cin / CK;
	const int inPitch = p.inPitch ? p.inPitch : p.W;
	const int outPitch = p.outPitch ? p.outPitch : p.W;
	const int resPitch = p.resPitch ? p.resPitch : p.W;
	const T *__restrict__ in = static_cast<const T *>(p.in);
	const T *__restrict__ wgt = static_cast<const T *>(p.wgt);

	// accumulators start at the (BN-folded) bias: rows of D are output channels
	f32x16 acc[NB][RW];
#pragma unroll
	for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const f32x4 b = *reinterpret_cast<const f32x4 *>(
			    p.bias + cog * COG + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
			for (int rw = 0; rw < RW; ++rw) {
				acc[nb][rw][4 * g + 0] = b[0];
				acc[nb][rw][4 * g + 1] = b[1];
				acc[nb][rw][4 * g + 2] = b[2];
				acc[nb][rw][4 * g + 3] = b[3];
			}
		}
	}

	if constexpr (DBUF) {
		constexpr int TILE_BYTES = IH * IW * CK * 2;
		constexpr int STAGE = W_BYTES + TILE_BYTES;
		constexpr int WN = W_BYTES / 16;
		static_assert(WN % kConvThreads == 0, "weight chunk must split evenly over the threads");
		constexpr int WITER = WN / kConvThreads;
		constexpr int N = IH * IW * P;
		constexpr int ITER = (N + kConvThreads - 1) / kConvThreads;
		// chunk-independent addressing of this thread's tile elements
		int srcOff[ITER], dstOff[ITER];
		bool inb[ITER];
#pragma unroll
		for (int k = 0; k < ITER; ++k) {
			const int i = min(tid + k * kConvThreads, N - 1);
			const int q = i / P;
			const int c = i % P;
			const int r = q / IW;
			const int x = q - r * IW;
			const int gy = ty0 - HALO + r;
			const int gx = tx0 - HALO + x;
			inb[k] = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			const int cy = min(max(gy, 0), p.H - 1);
			const int cx = min(max(gx, 0), p.W - 1);
			srcOff[k] = (cy * inPitch + cx) * p.cin + c * 8;
			dstOff[k] = (tid + k * kConvThreads < N) ? q * (CK * 2) + ((c ^ swz<P>(q)) << 4) : -1;
		}
		// it = -1 is the prologue (load + store chunk 0); iteration it loads chunk
		// it+1 (clamped, so the loads are unconditional), computes chunk it, then
		// parks the loaded registers in the other stage.  With ONE stage (launches with
		// more workgroups than CUs: two co-resident workgroups matter more than the
		// second stage) the registers wait for a barrier after the MFMAs instead.
		const int stageStride = p.stages == 2 ? STAGE : 0;
		for (int it = -1; it < nCC; ++it) {
			const int lc = min(it + 1, nCC - 1);
			// (named scalars, not an array: hipcc leaves a weight-prefetch array in
			// scratch memory once the asm-scheduled MFMA section sits between its
			// loads and its LDS stores)
			static_assert(WITER == 9, "double-buffered staging is written for 64x32 weight chunks");
			uint4 iv[ITER];
			const uint4 *wsrc = reinterpret_cast<const uint4 *>(
			    wgt + (size_t)(cog * nCC + lc) * (TAPS * CK * COG)) + tid;
			const uint4 w0 = wsrc[0 * kConvThreads], w1 = wsrc[1 * kConvThreads],
			            w2 = wsrc[2 * kConvThreads], w3 = wsrc[3 * kConvThreads],
			            w4 = wsrc[4 * kConvThreads], w5 = wsrc[5 * kConvThreads],
			            w6 = wsrc[6 * kConvThreads], w7 = wsrc[7 * kConvThreads],
			            w8 = wsrc[8 * kConvThreads];
#pragma unroll
			for (int k = 0; k < ITER; ++k) {
				iv[k] = *reinterpret_cast<const uint4 *>(in + srcOff[k] + lc * CK);
			}
			if (it >= 0) {
				const unsigned char *cw = smem + (it & 1) * stageStride;
				convChunkMfma<T, TAPS, CK, NB, RW>(cw, cw + W_BYTES, acc, wave, px, hh);
				if (stageStride == 0) __syncthreads();  // everyone is done reading the only stage
			}
			if (it + 1 < nCC) {
				unsigned char *sw = smem + ((it + 1) & 1) * stageStride;
				uint4 *wdst = reinterpret_cast<uint4 *>(sw) + tid;
				wdst[0 * kConvThreads] = w0;
				wdst[1 * kConvThreads] = w1;
				wdst[2 * kConvThreads] = w2;
				wdst[3 * kConvThreads] = w3;
				wdst[4 * kConvThreads] = w4;
				wdst[5 * kConvThreads] = w5;
				wdst[6 * kConvThreads] = w6;
				wdst[7 * kConvThreads] = w7;
				wdst[8 * kConvThreads] = w8;
#pragma unroll
				for (int k = 0; k < ITER; ++k) {
					// zero padding applied here, not at the load: a select right after
					// the load would wait for it in front of the MFMAs
					if (dstOff[k] >= 0) {
						*reinterpret_cast<uint4 *>(sw + W_BYTES + dstOff[k]) =
						    inb[k] ? iv[k] : make_uint4(0, 0, 0, 0);
					}
				}
			}
			__syncthreads();
		}
	} else {
	for (int cc = 0; cc < nCC; ++cc) {
			if (cc > 0) __syncthreads();
			// ---- stage the weight chunk (already in fragment order) ----
			// Up to 9 x 16 B per thread: the loads are issued here and written to LDS
			// together with the input tile below -- ONE memory round trip per chunk
			// instead of one for the weights (two for more than 6 x 16 B) plus one for
			// the tile.  Larger chunks (nb = 2 with 64 channels) keep the copy loop.
			constexpr int WN = W_BYTES / 16;
			constexpr int WITER = (WN + kConvThreads - 1) / kConvThreads;
			constexpr bool W_IN_REGS = WITER <= 9;
			uint4 wv[W_IN_REGS ? WITER : 1];
			{
				const uint4 *src = reinterpret_cast<const uint4 *>(
				    wgt + (size_t)(cog * nCC + cc) * (TAPS * CK * COG));
				if constexpr (W_IN_REGS) {
#pragma unroll
					for (int k = 0; k < WITER; ++k) wv[k] = src[min(tid + k * kConvThreads, WN - 1)];
				} else {
					uint4 *dst = reinterpret_cast<uint4 *>(smW);
#pragma unroll 6
					for (int i = tid; i < WN; i += kConvThreads) dst[i] = src[i];
				}
			}
// (a macro, not a lambda: capturing wv[] makes hipcc keep the array in scratch memory)
#define JU_STORE_WEIGHTS()                                                                   \
	if constexpr (W_IN_REGS) {                                                               \
		_Pragma("unroll") for (int k = 0; k < WITER; ++k) {                                  \
			if (tid + k * kConvThreads < WN)                                                 \
				reinterpret_cast<uint4 *>(smW)[tid + k * kConvThreads] = wv[k];              \
		}                                                                                    \
	}
			if constexpr (UPS) {
				// ---- low-resolution patch -> LDS (rows/cols clamped into the tensor) ----
				constexpr int LH = IH / 2 + 2, LW = IW / 2 + 2;
				constexpr int LN = LH * LW * P;
				constexpr int LITER = (LN + kConvThreads - 1) / kConvThreads;
				unsigned char *smL = smI + IH * IW * CK * 2;
				const int lh = p.H >> 1, lw = p.W >> 1;
				const int lowPitch = p.inPitch ? p.inPitch : lw;  // (inPitch defaults to the HI-res W)
				const int ly0 = (ty0 >> 1) - 1, lx0 = (tx0 >> 1) - 1;  // patch origin (may be -1)
				uint4 lv[LITER];
#pragma unroll
				for (int k = 0; k < LITER; ++k) {
					const int i = min(tid + k * kConvThreads, LN - 1);
					const int q = i / P, c = i % P;
					const int r = q / LW, x = q - r * LW;
					const int cy = min(max(ly0 + r, 0), lh - 1);
					const int cx = min(max(lx0 + x, 0), lw - 1);
					lv[k] = *reinterpret_cast<const uint4 *>(
					    in + ((size_t)cy * lowPitch + cx) * p.cin + cc * CK + c * 8);
				}
				JU_STORE_WEIGHTS()
#pragma unroll
				for (int k = 0; k < LITER; ++k) {
					const int i = tid + k * kConvThreads;
					if (i < LN) reinterpret_cast<uint4 *>(smL)[i] = lv[k];
				}
				__syncthreads();
				// ---- expand: tile element (hi-res pixel, 8 channels) from its sources ----
				// One pass per parity class of (row, column): src = dst/2 makes the lerp
				// weights 0 or 1/2, so a class is a copy, a 2-tap or a 4-tap average and
				// each pass is straight-line code without divergence.  The expressions
				// are upsample2_kernel's with the zero-weight terms dropped (a + (b-a)*0
				// == a exactly), so the result is bit-identical.
				constexpr int CH = IH / 2, CW = IW / 2, CN = CH * CW * P;  // per class
				static_assert(IH % 2 == 0 && IW % 2 == 0, "tile must split into parity classes");
#pragma unroll
				for (int cls = 0; cls < 4; ++cls) {
					// tile row r has gy = ty0 - 1 + r: odd r <=> even gy (ty0 is even)
					const int oddY = cls >> 1, oddX = cls & 1;  // parity of gy, gx
					for (int i = tid; i < CN; i += kConvThreads) {
						const int c = i % P, q2 = i / P;
						const int r = 2 * (q2 / CW) + 1 - oddY, x = 2 * (q2 % CW) + 1 - oddX;
						const int q = r * IW + x;
						const int gy = ty0 - HALO + r;
						const int gx = tx0 - HALO + x;
						Vec8<T> o;
						if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
							const int y0 = gy >> 1, x0 = gx >> 1;
							const int y1 = min(y0 + 1, lh - 1), x1 = min(x0 + 1, lw - 1);
							auto at = [&](int yy, int xx) {
								return *reinterpret_cast<const Vec8<T> *>(
								    smL + (((yy - ly0) * LW + (xx - lx0)) * P + c) * 16);
							};
							const Vec8<T> tl = at(y0, x0);
							if (!oddY && !oddX) {
								o = tl;
							} else if (!oddY) {
								const Vec8<T> tr = at(y0, x1);
#pragma unroll
								for (int j = 0; j < 8; ++j) {
									const float a = static_cast<float>(tl[j]), b2 = static_cast<float>(tr[j]);
									o[j] = static_cast<T>(a + (b2 - a) * 0.5f);
								}
							} else if (!oddX) {
								const Vec8<T> bl = at(y1, x0);
#pragma unroll
								for (int j = 0; j < 8; ++j) {
									const float a = static_cast<float>(tl[j]), d = static_cast<float>(bl[j]);
									o[j] = static_cast<T>(a + (d - a) * 0.5f);
								}
							} else {
								const Vec8<T> tr = at(y0, x1), bl = at(y1, x0), br = at(y1, x1);
#pragma unroll
								for (int j = 0; j < 8; ++j) {
									const float a = static_cast<float>(tl[j]), b2 = static_cast<float>(tr[j]);
									const float d = static_cast<float>(bl[j]), e = static_cast<float>(br[j]);
									const float top = a + (b2 - a) * 0.5f;
									const float bot = d + (e - d) * 0.5f;
									o[j] = static_cast<T>(top + (bot - top) * 0.5f);
								}
							}
						} else {
#pragma unroll
							for (int j = 0; j < 8; ++j) o[j] = static_cast<T>(0.f);
						}
						*reinterpret_cast<Vec8<T> *>(smI + q * (CK * 2) + ((c ^ swz<P>(q)) << 4)) = o;
					}
				}
			} else {
				// ---- stage the input tile (+halo), zero outside the image ----
				// Loads are issued unconditionally on clamped coordinates and zeroed by a
				// select: a load under `if (in bounds)` makes hipcc wait vmcnt(0) per element,
				// i.e. one serial memory round trip per 16 bytes per thread.
				{
					constexpr int N = IH * IW * P;
					constexpr int ITER = (N + kConvThreads - 1) / kConvThreads;
					uint4 v[ITER];
					int dstOff[ITER];
#pragma unroll
					for (int k = 0; k < ITER; ++k) {
						const int i = min(tid + k * kConvThreads, N - 1);
						const int q = i / P;
						const int c = i % P;
						const int r = q / IW;
						const int x = q - r * IW;
						const int gy = ty0 - HALO + r;
						const int gx = tx0 - HALO + x;
						const bool inb = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
						const int cy = min(max(gy, 0), p.H - 1);
						const int cx = min(max(gx, 0), p.W - 1);
						v[k] = *reinterpret_cast<const uint4 *>(
						    in + ((size_t)cy * inPitch + cx) * p.cin + cc * CK + c * 8);
						if (!inb) v[k] = make_uint4(0, 0, 0, 0);
						dstOff[k] = (tid + k * kConvThreads < N) ? q * (CK * 2) + ((c ^ swz<P>(q)) << 4) : -1;
					}
					JU_STORE_WEIGHTS()
#pragma unroll
					for (int k = 0; k < ITER; ++k) {
						if (dstOff[k] >= 0) *reinterpret_cast<uint4 *>(smI + dstOff[k]) = v[k];
					}
				}
			}
			__syncthreads();
			convChunkMfma<T, TAPS, CK, NB, RW>(smW, smI, acc, wave, px, hh);
		}
#undef JU_STORE_WEIGHTS
	}

	// ---- epilogue with the 2x2 max-pool folded in (flow encoder, models.py:377-410):
	// a wave owns rows 2j, 2j+1 (vertical max inside the lane), the horizontal
	// partner pixel sits in the neighbouring lane.  ReLU and LeakyReLU (slope >= 0:
	// monotonic, checked by the loader) commute with max.  Both
	// lanes of a pair end up with the pooled pixel; each stores half its channels.
	if constexpr (RW == 2) {
		if (p.pool) {
			const int gx = tx0 + px;
			const int gy = ty0 + wave * 2;
			const bool live = gy < p.H && gx < p.W;  // H, W even: the partner row/pixel exists too
			const int poolPitch = p.outPitch ? p.outPitch : p.W / 2;
			T *out = static_cast<T *>(p.out) +
			         ((size_t)(gy >> 1) * poolPitch + (gx >> 1)) * p.cout + cog * COG;
#pragma unroll
			for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					float v[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						float m = fmaxf(acc[nb][0][4 * g + i], acc[nb][1][4 * g + i]);
						m = fmaxf(m, __shfl_xor(m, 1));
						v[i] = p.relu == 1 ? fmaxf(m, 0.0f) : (p.relu == 2 ? leaky(m, p.slope) : m);
					}
					if (live && (g >> 1) == (px & 1)) {
						Vec4<T> o = {static_cast<T>(v[0]), static_cast<T>(v[1]), static_cast<T>(v[2]),
						    static_cast<T>(v[3])};
						*reinterpret_cast<Vec4<T> *>(out + nb * 32 + 8 * g + 4 * hh) = o;
					}
				}
			}
			return;
		}
	}

	// ---- epilogue: residual, activation, NHWC store (4 channels per lane) ----
	const int gx = tx0 + px;
#pragma unroll
	for (int rw = 0; rw < RW; ++rw) {
		const int gy = ty0 + wave * RW + rw;
		if (gy >= p.H || gx >= p.W) continue;
		const size_t pixOff = ((size_t)gy * outPitch + gx) * p.cout + cog * COG;
		const size_t resOff = ((size_t)gy * resPitch + gx) * p.cout + cog * COG;
#pragma unroll
		for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const size_t off = pixOff + nb * 32 + 8 * g + 4 * hh;
				const size_t roff = resOff + nb * 32 + 8 * g + 4 * hh;
				float v[4];
#pragma unroll
				for (int i = 0; i < 4; ++i) v[i] = acc[nb][rw][4 * g + i];
				if (p.res != nullptr) {
					const Vec4<T> r =
					    *reinterpret_cast<const Vec4<T> *>(static_cast<const T *>(p.res) + roff);
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] += static_cast<float>(r[i]);
				}
				if (p.relu == 1) {
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
				} else if (p.relu == 2) {
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = leaky(v[i], p.slope);
				}
				if (p.outHead) {  // the flow head is f16 whatever the compute type
					*reinterpret_cast<Vec4<f16> *>(static_cast<f16 *>(p.out) + off) = pack4<f16>(v[0], v[1], v[2], v[3]);
				} else {
					Vec4<T> o = {static_cast<T>(v[0]), static_cast<T>(v[1]), static_cast<T>(v[2]),
					    static_cast<T>(v[3])};
					*reinterpret_cast<Vec4<T> *>(static_cast<T *>(p.out) + off) = o;
				}
			}
		}
	}
}

template <int TAPS, int CK, int NB, int RW>
constexpr int convLdsBytes() {
	constexpr int HALO = (TAPS == 9) ? 1 : 0;
	return TAPS * CK * 32 * NB * 2 + (4 * RW + 2 * HALO) * (kTW + 2 * HALO) * CK * 2;
}

template <typename T, int TAPS, int CK, int NB, int RW, bool DBUF = false, bool UPS = false>
void launchConvInst(const ConvParams &p, hipStream_t stream) {
	// UPS: + the low-resolution patch (IH/2+2) x (IW/2+2) pixels
	constexpr int ldsMax = convLdsBytes<TAPS, CK, NB, RW>() * (DBUF ? 2 : 1) +
	                       (UPS ? ((4 * RW + 2) / 2 + 2) * ((kTW + 2) / 2 + 2) * CK * 2 : 0);
	static_assert(ldsMax <= 160 * 1024, "conv stages do not fit LDS");
	const int lds = (DBUF && p.stages == 1) ? ldsMax / 2 : ldsMax;
	auto kern = conv_mfma_kernel<T, TAPS, CK, NB, RW, DBUF, UPS>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), ldsMax, &ldsDone, "conv");
	if (launchesAreDry()) return;
	dim3 grid((p.W + kTW - 1) / kTW, (p.H + 4 * RW - 1) / (4 * RW), p.cout / (32 * NB) * (p.items > 1 ? p.items : 1));
	hipLaunchKernelGGL(kern, grid, dim3(kConvThreads), lds, stream, p);
	hipCheckLaunch("conv_mfma");
}

template <typename T>
void launchConvT(const ConvParams &p, hipStream_t stream) {
	const int ck = convCK(p.cin);
	if (p.cin % 16 != 0 || p.cout % 32 != 0 || (p.nb != 1 && p.nb != 2) ||
	    (p.rw != 1 && p.rw != 2) || p.cout % (32 * p.nb) != 0) {
		throw std::invalid_argument("conv: cin must be a multiple of 16, cout of 32*nb");
	}
	if (p.pool && (p.rw != 2 || p.H % 2 || p.W % 2 || p.res != nullptr || p.outHead)) {
		throw std::invalid_argument("conv: fused max-pool needs rw = 2, even H and W, 16-bit output");
	}
	// Several 64-channel chunks and a launch that leaves at most one workgroup per CU
	// (the coarsest flow levels): double-buffered staging, the only way such a
	// workgroup overlaps its loads with its MFMAs.  Larger launches keep the
	// single-stage kernel: two co-resident workgroups hide each other's latency and
	// the doubled LDS would cost a second round of workgroups.  (JU_CONV_DBUF=0/1
	// forces it off/on for A/B timing.)
	static const char *dbufEnv = devSwitch(Dev::ConvDbuf);
	const int cus = currentDeviceCUs();
	// (a look-ahead launch covers `items` frames: that many times the workgroups, as fbLaunchCost and the split-K rule count)
	const long wgs = (long)((p.W + kTW - 1) / kTW) * ((p.H + 4 * p.rw - 1) / (4 * p.rw)) *
	                 (p.cout / (32 * p.nb)) * (p.items > 1 ? p.items : 1);
	// several 64-channel chunks: the loads of chunk c+1 travel behind the MFMAs of chunk
	// c (register prefetch).  Two LDS stages when the launch has at most one workgroup
	// per CU anyway, one stage (and a second barrier) when two can be co-resident.
	const bool multi = ck == 64 && p.cin > 64 && p.nb == 1 && p.taps == 9;
	int stages = multi ? (wgs <= cus ? 2 : 1) : 0;
	if (dbufEnv && multi) stages = dbufEnv[0] - '0';  // A/B: 0 = plain staging, 1, 2
	const bool dbuf = stages > 0;
	if (p.upsample) {
		if (p.taps != 9 || ck != 64 || p.nb != 1 || p.H % 2 || p.W % 2 || p.pool) {
			throw std::invalid_argument("conv: fused upsampling needs 3x3, cin % 64 == 0, nb = 1, even H and W");
		}
		if (p.rw == 2) return launchConvInst<T, 9, 64, 1, 2, false, true>(p, stream);
		return launchConvInst<T, 9, 64, 1, 1, false, true>(p, stream);
	}
	if (dbuf) {
		ConvParams q = p;
		q.stages = stages;
		if (p.rw == 2) return launchConvInst<T, 9, 64, 1, 2, true>(q, stream);
		return launchConvInst<T, 9, 64, 1, 1, true>(q, stream);
	}
#define JU_CONV_CASE(TAPS_, CK_)                                                   \
	if (p.taps == TAPS_ && ck == CK_) {                                            \
		if (p.nb == 2 && p.rw == 2) launchConvInst<T, TAPS_, CK_, 2, 2>(p, stream);  \
		else if (p.nb == 2) launchConvInst<T, TAPS_, CK_, 2, 1>(p, stream);          \
		else if (p.rw == 2) launchConvInst<T, TAPS_, CK_, 1, 2>(p, stream);          \
		else launchConvInst<T, TAPS_, CK_, 1, 1>(p, stream);                         \
		return;                                                                    \
	}
	JU_CONV_CASE(9, 64)
	JU_CONV_CASE(9, 32)
	JU_CONV_CASE(9, 16)
	JU_CONV_CASE(1, 64)
	JU_CONV_CASE(1, 32)
#undef JU_CONV_CASE
	throw std::invalid_argument("conv: unsupported shape");
}

}  // namespace

void launchConv(DType dt, const ConvParams &p, hipStream_t stream) {
	if (p.items > 1 && p.res != nullptr) throw std::invalid_argument("launchConv: a look-ahead launch (items > 1) takes no residual");
	if (dt == kF16) launchConvT<f16>(p, stream);
	else launchConvT<bf16>(p, stream);
}


}  // namespace ju
