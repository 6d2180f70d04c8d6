// gfx950 (CDNA4, MI355X) kernels of the per-frame recurrent super-resolution
// step.  What each one computes follows the reference's Keras graph
// (scripts/training/models.py:680-829 and the layers it uses); how it is
// computed is MI355X-first:
//
//  * conv_mfma_kernel   implicit-GEMM 3x3 / 1x1 convolution on
//                       v_mfma_f32_32x32x16_{f16,bf16}: weights are the MFMA A
//                       operand (M = cout), activations the B operand
//                       (N = 32 consecutive pixels of one image row), so each
//                       lane's accumulator registers hold 4 consecutive output
//                       channels of ONE pixel and the NHWC store is 8 B/lane.
//                       Input tile (+halo) and the weight chunk are staged in
//                       LDS; the input image is XOR-swizzled per 16-B chunk so
//                       the ds_read_b128 fragment reads are bank-conflict free.
//  * pack_frames        u8 BGRX frame + frame history -> 16-channel flow input
//  * maxpool2/upsample2 flow auto-encoder resampling (TF1 asymmetric bilinear)
//  * warp_pack          dense bilinear warp of the previous HR output fused with
//                       space-to-depth(4), the concat with the LR frame and the
//                       16-bit pack (reference models.py:799-801, 523-530)
//  * tail               ConvT(2x2,s2,32->3)+bias, tanh, bilinear x4 skip, clip,
//                       HR state write and truncating BGRX u8 pack in one pass
//                       (reference models.py:573-593, keras_layers.py:211-230,
//                       core/src/cuda_convert.cc.cu:95-108)
//
// Wavefront = 64 lanes everywhere; no CUDA-isms, no portability layer.

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <type_traits>

#include "kernels.h"

#include <cstdlib>

namespace ju {

namespace {

using f16 = _Float16;
using bf16 = __bf16;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <typename T>
using Vec8 = T __attribute__((ext_vector_type(8)));
template <typename T>
using Vec4 = T __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x16 mfma32(Vec8<f16> a, Vec8<f16> b, f32x16 c) {
	return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma32(Vec8<bf16> a, Vec8<bf16> b, f32x16 c) {
	return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// ReLU on values already rounded to the 16-bit type, as a packed signed-int16 max
// with 0: a negative bf16/f16 (sign bit set, -0.0 included) is a negative int16.
// Rounding keeps the sign, so this equals relu-then-round bit for bit, and it is 2
// v_pk_max_i16 per 4 values instead of 8 v_max_f32 (hipcc puts a canonicalising
// max in front of every fmaxf on an MFMA result).
template <typename T>
__device__ __forceinline__ Vec4<T> reluPacked(Vec4<T> v) {
	typedef short s16x4 __attribute__((ext_vector_type(4)));
	s16x4 b = __builtin_bit_cast(s16x4, v);
	const s16x4 z = {0, 0, 0, 0};
	b = __builtin_elementwise_max(b, z);
	return __builtin_bit_cast(Vec4<T>, b);
}

// tanh(x) = 1 - 2 / (exp(2x) + 1) on the hardware exp2 / rcp units (1 ulp each):
// absolute error < 4e-7, far below the 1/255 output step; saturates correctly
// (exp -> inf: rcp -> 0; exp -> 0: 1 - 2).  libm's tanhf is ~40 VALU ops per value
// and the tail evaluates 6.2 M of them per frame.
__device__ __forceinline__ float fastTanh(float x) {
	const float t = __builtin_amdgcn_exp2f(x * 2.885390081777927f);  // 2 * log2(e)
	return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}

// Opt a kernel in to more than 64 KiB of dynamic LDS.  The attribute is per DEVICE:
// a process may hold runtimes on several GPUs, so "done" is tracked per device (one
// mask per kernel instantiation, passed in by the launcher).  The first launch on a
// device happens in the engine's constructor, before any graph capture.
inline void ensureDynamicLds(const void *kern, int bytes, std::atomic<std::uint64_t> *doneMask,
    const char *what) {
	int dev = 0;
	if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
	const std::uint64_t bit = 1ull << dev;
	if (doneMask->load(std::memory_order_acquire) & bit) return;
	const hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
	if (e != hipSuccess) {
		throw std::runtime_error(std::string("hipFuncSetAttribute(") + what + " LDS): " + hipGetErrorString(e));
	}
	doneMask->fetch_or(bit, std::memory_order_release);
}

inline void hipCheckLaunch(const char *what) {
	hipError_t e = hipGetLastError();
	if (e != hipSuccess) {
		throw std::runtime_error(std::string("HIP launch failed (") + what +
		                         "): " + hipGetErrorString(e));
	}
}

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
	const int cog = blockIdx.z;
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
	// partner pixel sits in the neighbouring lane.  ReLU commutes with max.  Both
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
						v[i] = p.relu ? fmaxf(m, 0.0f) : m;
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
				if (p.relu) {
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
				}
				if (p.outF32) {
					f32x4 o = {v[0], v[1], v[2], v[3]};
					*reinterpret_cast<f32x4 *>(static_cast<float *>(p.out) + off) = o;
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
	dim3 grid((p.W + kTW - 1) / kTW, (p.H + 4 * RW - 1) / (4 * RW), p.cout / (32 * NB));
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
	if (p.pool && (p.rw != 2 || p.H % 2 || p.W % 2 || p.res != nullptr || p.outF32)) {
		throw std::invalid_argument("conv: fused max-pool needs rw = 2, even H and W, 16-bit output");
	}
	// Several 64-channel chunks and a launch that leaves at most one workgroup per CU
	// (the coarsest flow levels): double-buffered staging, the only way such a
	// workgroup overlaps its loads with its MFMAs.  Larger launches keep the
	// single-stage kernel: two co-resident workgroups hide each other's latency and
	// the doubled LDS would cost a second round of workgroups.  (JU_CONV_DBUF=0/1
	// forces it off/on for A/B timing.)
	static const char *dbufEnv = std::getenv("JU_CONV_DBUF");
	static const int cus = [] {
		int dev = 0, n = 256;
		if (hipGetDevice(&dev) == hipSuccess) {
			(void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
		}
		return n;
	}();
	const long wgs = (long)((p.W + kTW - 1) / kTW) * ((p.H + 4 * p.rw - 1) / (4 * p.rw)) *
	                 (p.cout / (32 * p.nb));
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

// ---------------------------------------------------------------------------
// persistent 3x3 64->64 convolution of the generator's residual tower
// ---------------------------------------------------------------------------
// One workgroup per CU (grid = min(#tiles, 256)), 4 waves, one per SIMD.  The
// 72 KiB of kernel-ready weights are DMA'd into LDS once per workgroup; input
// tiles (8 rows x 32 px + halo = 10 x 34 px x 128 B) are double-buffered and
// fetched with global_load_lds (no VGPR round trip) while the previous tile is on
// the matrix cores.  Activations live in the zero-bordered tower layout, so tile
// staging has no bounds checks.  LDS: 73728 + 2 * 43520 = 160768 B of 163840.
//
// LDS-DMA writes lane-linear (base + lane*16), so the bank-conflict swizzle is
// applied to each lane's SOURCE chunk and again on the fragment read.
constexpr int kTowerThreads = 256;
constexpr int kTowerWBytes = 9 * 64 * 64 * 2;
constexpr int kTowerTileBytes = 10 * 34 * 128;
constexpr int kTowerLds = kTowerWBytes + 2 * kTowerTileBytes;

__device__ __forceinline__ void glds16(const void *g, void *l) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
	    (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

struct TowerParams {
	const void *in;    // tower-layout allocation start (row -1, col -1 of the image)
	const void *wgt;
	const float *bias;
	const void *res;   // allocation start or nullptr
	void *out;         // allocation start
	int H, W, pitch;   // pitch in pixels
	int tilesX, numTiles;
	int relu;
};

// VARIANT is a timing-only ablation switch (tools/tower_ablation.py); 0 is the
// product kernel.  1: no MFMA loop, 2: no epilogue loads/stores, 3: no tile
// staging, 4: no weight staging.  Variants != 0 compute garbage by design.
template <typename T, int VARIANT>
__global__ __launch_bounds__(kTowerThreads, 1) void conv_tower_kernel(TowerParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *smW = smem;
	unsigned char *smT = smem + kTowerWBytes;
	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const T *__restrict__ in = static_cast<const T *>(p.in);

	// XCD-aware tile order: workgroups b, b+8, b+16, ... share an XCD (and its L2);
	// give each XCD a contiguous run of tiles so vertically adjacent tiles find
	// their shared halo rows in L2.
	const int nwg = gridDim.x;
	const int bid = blockIdx.x;
	const int perX = (nwg + 7) >> 3;
	const int slot = (bid & 7) * perX + (bid >> 3);  // may exceed nwg-1 when nwg % 8 != 0

	auto stageTile = [&](int tile, int buf) {
		const int ty = tile / p.tilesX;
		const int tx = tile - ty * p.tilesX;
		const T *base = in + ((size_t)(ty * 8) * p.pitch + tx * 32) * 64;
		unsigned char *dst = smT + buf * kTowerTileBytes;
#pragma unroll
		for (int k = 0; k < (VARIANT == 3 ? 0 : 11); ++k) {
			const int i = wave + 4 * k;          // wave-instruction index: 8 pixels each
			const int q = i * 8 + (lane >> 3);   // pixel index inside the 10 x 34 tile
			if (i < 43 && q < 340) {
				const int r = q / 34;
				const int x = q - r * 34;
				const int c = (lane & 7) ^ ((q >> 1) & 7);
				glds16(base + ((size_t)r * p.pitch + x) * 64 + c * 8, dst + i * 1024);
			}
		}
	};

	// ---- prologue: weights + first tile in flight together ----
	{
		const unsigned char *wsrc = static_cast<const unsigned char *>(p.wgt);
#pragma unroll
		for (int k = 0; k < (VARIANT == 4 ? 0 : 18); ++k) {
			const int i = wave + 4 * k;
			glds16(wsrc + (size_t)i * 1024 + lane * 16, smW + i * 1024);
		}
	}
	int tile = slot;
	if (tile < p.numTiles) stageTile(tile, 0);

	f32x4 biasv[2][4];
#pragma unroll
	for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			biasv[nb][g] = *reinterpret_cast<const f32x4 *>(p.bias + nb * 32 + 8 * g + 4 * hh);
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	// LDS byte addresses for the hand-issued fragment reads
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));
	const unsigned wAddr = ldsBase + hh * 1024 + px * 16;
	const unsigned wAddrHi = wAddr + 40960;
	const int q0 = (wave * 2) * 34 + px;

	int buf = 0;
	const int stride = perX * 8;
	for (; tile < p.numTiles; tile += stride, buf ^= 1) {
		const int next = tile + stride;
		if (next < p.numTiles) stageTile(next, buf ^ 1);  // async, lands during the MFMAs

		const int ty = tile / p.tilesX;
		const int tx = tile - ty * p.tilesX;
		f32x16 acc[2][2];
#pragma unroll
		for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
#pragma unroll
				for (int rw = 0; rw < 2; ++rw) {
					acc[nb][rw][4 * g + 0] = biasv[nb][g][0];
					acc[nb][rw][4 * g + 1] = biasv[nb][g][1];
					acc[nb][rw][4 * g + 2] = biasv[nb][g][2];
					acc[nb][rw][4 * g + 3] = biasv[nb][g][3];
				}
			}
		}
		// residual prefetch (second conv of a block): issued now, consumed after the
		// K loop, so its latency hides behind the MFMAs
		Vec4<T> resv[2][2][4];
		if (VARIANT != 2 && p.res != nullptr) {
			const int gxr = tx * 32 + px;
#pragma unroll
			for (int rw = 0; rw < 2; ++rw) {
				const int gyr = ty * 8 + wave * 2 + rw;  // rows beyond H read the zero border
				const T *rp = static_cast<const T *>(p.res) +
				              ((size_t)(gyr + 1) * p.pitch + gxr + 1) * 64 + 4 * hh;
#pragma unroll
				for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						resv[rw][nb][g] = *reinterpret_cast<const Vec4<T> *>(rp + nb * 32 + 8 * g);
					}
				}
			}
		}

		// K loop as 12 macro-steps m = (dx, ks): the 3 vertical taps x 2 cout blocks
		// of weights (6 fragments) and the wave's 4 input rows (4 fragments) feed
		// 12 MFMAs.  With one wave per SIMD nothing but this wave's own instruction
		// stream can hide LDS latency, and hipcc schedules fragment reads just in
		// time (ds_read; s_waitcnt lgkmcnt(0); mfma), so the loop is hand-scheduled:
		// the 10 fragment reads of step m+1 are issued, one behind each of the first
		// 10 MFMAs of step m, IN THE ORDER step m+1 consumes them; each MFMA waits
		// with a counted lgkmcnt for exactly the reads it needs (LDS reads return in
		// order).  Reads are asm (hipcc must not count or move them); each wait is
		// followed by sched_barrier(0) so no MFMA is hoisted above it.
		//
		// consumption order of a step's reads:   a00 b0 a01 b1 a10 a11 b2 a20 a21 b3
		// MFMA k = (dy, rw, nb) = (k>>2, (k>>1)&1, k&1) uses a[dy][nb], b[rw+dy]
		Vec8<T> fa[2][3][2], fb[2][4];
		const unsigned tileAddr = ldsBase + kTowerWBytes + buf * kTowerTileBytes;
		auto issueRead = [&](int m, int set, int idx) {
			// idx = position in the consumption order above
			const int dx = m >> 2, ks = m & 3;
			constexpr int kind[10] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1};   // 0 = weight, 1 = activation
			constexpr int sub[10] = {0, 0, 1, 1, 2, 3, 2, 4, 5, 3};    // a: dy*2+nb ; b: row
			if (kind[idx] == 0) {
				const int dy = sub[idx] >> 1, nb = sub[idx] & 1;
				const int widx = (dy * 3 + dx) * 4 + ks;
				const int off = widx * 2048 + nb * 512;
				if (off < 65536 - 512) {
					asm volatile("ds_read_b128 %0, %1 offset:%2"
					             : "=v"(fa[set][dy][nb]) : "v"(wAddr), "n"(off));
				} else {
					asm volatile("ds_read_b128 %0, %1 offset:%2"
					             : "=v"(fa[set][dy][nb]) : "v"(wAddrHi), "n"(off - 40960));
				}
			} else {
				const int r = sub[idx];
				const int q = q0 + r * 34 + dx;
				const unsigned a = tileAddr + q * 128 + (((ks * 2 + hh) ^ ((q >> 1) & 7)) << 4);
				asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][r]) : "v"(a));
			}
		};
		// reads that must have landed before MFMA k of a step: index of the last one
		// it needs in the consumption order (-1: nothing new)
		constexpr int needs[12] = {1, 2, 3, -1, 4, 5, 6, -1, 7, 8, 9, -1};
		// start from an empty LGKM counter: the counted waits below must see only
		// this loop's own reads (compiler-issued scalar loads would skew them)
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int i = 0; i < (VARIANT == 1 ? 0 : 10); ++i) issueRead(0, 0, i);
#pragma unroll
		for (int m = 0; m < (VARIANT == 1 ? 0 : 12); ++m) {
			const int set = m & 1;
			const bool more = (m + 1 < 12);
#pragma unroll
			for (int k = 0; k < 12; ++k) {
				if (needs[k] >= 0) {
					// outstanding reads allowed = (this step's reads younger than needs[k])
					//                           + (next step's reads already issued = k)
					const int allowed = (9 - needs[k]) + (more ? (k < 10 ? k : 10) : 0);
					if (allowed >= 10) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
					else if (allowed == 9) asm volatile("s_waitcnt lgkmcnt(9)" ::: "memory");
					else if (allowed == 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
					else if (allowed == 7) asm volatile("s_waitcnt lgkmcnt(7)" ::: "memory");
					else if (allowed == 6) asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
					else if (allowed == 5) asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
					else if (allowed == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
					else if (allowed == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
					else if (allowed == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
					else if (allowed == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
					else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					__builtin_amdgcn_sched_barrier(0);
				}
				const int dy = k >> 2, rw = (k >> 1) & 1, nb = k & 1;
				acc[nb][rw] = mfma32(fa[set][dy][nb], fb[set][rw + dy], acc[nb][rw]);
				if (more && k < 10) issueRead(m + 1, set ^ 1, k);
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		// ---- epilogue ----
		// Each lane holds, per (row rw, cout block nb, group g), 4 consecutive output
		// channels of ONE pixel: stored directly that is an 8-byte write into each of
		// 32 different 128-byte pixel records per instruction, and a layer becomes ~1M
		// partial-line L2 requests (measured: 9.4 of 19.5 us).  Instead the wave
		// transposes its 2 rows x 32 px x 64 ch through LDS (its own 8 KiB slice of
		// the input buffer it has just finished with) and stores whole records,
		// 16 B per lane, 1 KiB contiguous per instruction.
		__syncthreads();  // every wave is done reading this tile's input (halo rows are shared)
		{
			unsigned char *slice = smT + buf * kTowerTileBytes + wave * 8192;
#pragma unroll
			for (int rw = 0; rw < 2; ++rw) {
				const int pi = rw * 32 + px;  // pixel index inside the slice
#pragma unroll
				for (int nb = 0; nb < 2; ++nb) {
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						float v[4];
#pragma unroll
						for (int i = 0; i < 4; ++i) v[i] = acc[nb][rw][4 * g + i];
						if (VARIANT != 2 && p.res != nullptr) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] += static_cast<float>(resv[rw][nb][g][i]);
						}
						if (p.relu) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
						}
						Vec4<T> o = {static_cast<T>(v[0]), static_cast<T>(v[1]),
						    static_cast<T>(v[2]), static_cast<T>(v[3])};
						const int c = nb * 4 + g;  // 16-byte chunk = channels 8c .. 8c+7
						*reinterpret_cast<Vec4<T> *>(
						    slice + pi * 128 + ((c ^ (pi & 7)) << 4) + hh * 8) = o;
					}
				}
			}
			// same-wave exchange through LDS: order the writes before the reads
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			if (VARIANT != 2) {
				T *outp = static_cast<T *>(p.out);
#pragma unroll
				for (int i = 0; i < 8; ++i) {
					const int pi = i * 8 + (lane >> 3);
					const int c = (lane & 7) ^ (pi & 7);
					typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
					const u32x4 val = *reinterpret_cast<const u32x4 *>(slice + i * 1024 + lane * 16);
					const int gy = ty * 8 + wave * 2 + (pi >> 5);
					const int gxo = tx * 32 + (pi & 31);
					if (gy < p.H && gxo < p.W) {
						// (non-temporal stores measured no better: 18.2 vs 17.4 us)
						*reinterpret_cast<u32x4 *>(
						    outp + ((size_t)(gy + 1) * p.pitch + gxo + 1) * 64 + c * 8) = val;
					}
				}
			} else {
				asm volatile("" ::"v"(acc[0][0]), "v"(acc[1][0]), "v"(acc[0][1]), "v"(acc[1][1]));
			}
		}
		// End of iteration.  The next tile's DMA was already drained by the
		// __syncthreads() in front of the epilogue (its fence waits vmcnt(0) while an
		// LDS-DMA is pending), so nothing here waits on memory: in particular not on
		// the record stores' acknowledgements (an s_waitcnt vmcnt(0) here cost ~1.5 us
		// per tile).  Only the staging slices must be read out before the next
		// iteration's DMA refills this buffer: LDS wait + raw s_barrier.
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		__builtin_amdgcn_s_barrier();
		__builtin_amdgcn_sched_barrier(0);
	}
}

int g_TowerVariant = 0;
int g_ResidentFault = 0;  // test hook: launch the resident tower this many workgroups short

template <typename T, int VARIANT>
void launchTowerT(const ConvParams &p, hipStream_t stream) {
	auto kern = conv_tower_kernel<T, VARIANT>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kTowerLds, &ldsDone, "tower");
	const int pitch = towerPitch(p.W);
	const size_t origin = towerOrigin(p.W) * 64 * 2;  // bytes from allocation start to pixel (0,0)
	TowerParams t{};
	t.in = static_cast<const unsigned char *>(p.in) - origin;
	t.wgt = p.wgt;
	t.bias = p.bias;
	t.res = p.res ? static_cast<const unsigned char *>(p.res) - origin : nullptr;
	t.out = static_cast<unsigned char *>(p.out) - origin;
	t.H = p.H;
	t.W = p.W;
	t.pitch = pitch;
	t.tilesX = (p.W + 31) / 32;
	t.numTiles = t.tilesX * ((p.H + 7) / 8);
	t.relu = p.relu;
	// The XCD remap in the kernel is a bijection only on grids that are a multiple
	// of 8; surplus workgroups find no tile and exit after the weight prologue.
	const int grid = ((t.numTiles < 256 ? t.numTiles : 256) + 7) / 8 * 8;
	hipLaunchKernelGGL(kern, dim3(grid), dim3(kTowerThreads), kTowerLds, stream, t);
	hipCheckLaunch("conv_tower");
}

// ---------------------------------------------------------------------------
// resident tower: all 2*B convolutions of the generator's residual blocks in
// ONE launch, activations never leaving the CU
// ---------------------------------------------------------------------------
// Per-layer launches are memory- and latency-chain-bound:
// a 64->64 layer moves ~46 MB for 9.6 GFLOP and pays a kernel boundary plus the
// flush of 16.6 MB of dirty L2.  Here each workgroup (one per CU) owns a region
// of 32 x RH (<= 16) pixels for the whole tower (480x270: 15 x 17 = 255 regions):
//   * two LDS buffers of (RH+2) x 34 px x 128 B hold the block input X and the
//     intermediate T, including a one-pixel halo ring; conv1 reads X writes T,
//     conv2 reads T, adds the residual from X and writes X in place; the 16-byte
//     chunk swizzle is keyed on the COLUMN ((cc>>1)&7): a row is 272 x 16 B, a
//     multiple of the 16-slot bank row, so rows do not shift the slot pattern;
//   * the wave's weights (its 32 output channels x 576) are the MFMA A operand
//     straight from 144 VGPRs, double-buffered (288) so the next layer's weights
//     stream in from L2 behind the current layer's MFMAs;
//   * after every layer only the edge ring (<= 94 px, 12 KB) is exchanged with the
//     <= 8 neighbouring workgroups through a global mailbox of self-validating
//     16-byte slots (epoch tag in the 8 free sign bits of post-ReLU values), one
//     write-through (sc1) store each, no drain and no release; a relaxed flag is
//     only a hint to start looking; the consumer reads with 16-byte sc1 loads and
//     retries slots whose tag is still old (cdna_hip_programming.md Guideline 16:
//     the "data is the flag" form R2, at 16 B).  Measured alternatives: drained
//     stores + flag + load (R1) 4.1 us per layer; 8-byte {tag,data} granules 13 us
//     (write-through stores are one fabric transaction each, so width matters).
// Nothing depends on dispatch order or XCD placement; all workgroups must be
// co-resident (grid <= #CUs, one workgroup per CU by LDS size); every wait is
// bounded in time and reports through *error.
constexpr int kResRW = 32;                                          // region width = one MFMA block
constexpr int kResMaxRH = 16;                                       // 8 row pairs, 4 per wave group
constexpr int kResPitch = kResRW + 2;                               // LDS row: 32 px + halo column each side
constexpr int kResRowBytes = kResPitch * 128;                       // 4352
constexpr int kResBufBytes = (kResMaxRH + 2) * kResRowBytes;       // 78336
constexpr int kResOffA = 0;
constexpr int kResOffB = kResBufBytes;
constexpr int kResOffMisc = kResOffB + kResBufBytes;
static_assert(kResRowBytes == 4352, "the ds_read immediates in tower_resident_kernel assume a 4352-byte row");
constexpr int kResLds = kResOffMisc + 64 + 512;                     // flag, 2 bias slots
constexpr int kResMailSlots = 4 * 32 * 8;                           // 16-byte slots per region per parity
constexpr unsigned long long kResTimeoutTicks = 20000000ull;        // 0.2 s of s_memrealtime

struct ResidentParams {
	const void *in;           // first layer's input, addressed at image pixel (0,0)
	int inPitch;              // its row pitch in pixels (dense W, or towerPitch(W))
	int hasHead;              // 1: layer 0 is the generator's conv_1 (no residual), blocks follow
	void *out;                // tower-layout tensor, allocation start (last layer's output)
	const void *weights;      // nLayers x 73728 B, kernel-ready (packConvWeights)
	const float *bias;        // nLayers x 64
	uint4 *mail;              // [regions][2][kResMailSlots] 16-byte slots
	unsigned *flag;           // [regions] {generation<<8 | layers published}
	const unsigned *gen;      // launch generation (bumped by bump_generation_kernel)
	unsigned *error;          // host-visible word, 0 = ok
	unsigned long long *debug;  // VARIANT 4 only: [regions][4 waves][8] cycle sums
	int H, W, pitch;
	int GX, GY, RH;
	int nLayers;
	int bumpGeneration;  // host-side only: launch bump_generation_kernel first
};

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__global__ void bump_generation_kernel(unsigned *gen) {
	*gen = *gen + 1;
}

// VARIANT: timing ablation only (0 = product; bit 0 = no halo exchange, bit 1 = no MFMA loop)
// HEAD: layer 0 is the generator's conv_1 (plain conv + ReLU), residual blocks follow
template <typename T, int VARIANT, bool HEAD>
__global__ __launch_bounds__(256, 1) void tower_resident_kernel(ResidentParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	constexpr bool xchg = !(VARIANT & 1);
	const int tid = threadIdx.x;
	const int wave = tid >> 6;
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int ch = wave & 1;   // cout half of this wave
	const int rp = wave >> 1;  // row-pair parity of this wave
	// Workgroups are dealt to the 8 XCDs round-robin (b % 8).  Give each XCD a
	// contiguous, row-major run of regions, so that most of a region's 8 neighbours
	// live on the same XCD and their mailbox lines are served by that XCD's L2.  A
	// bijection for any grid size: XCD x holds count_x = ceil((n - x) / 8) workgroups.
	// (Measured: 654-656 us per frame against 654-661 with region = blockIdx.x -- the
	// sc1 mailbox traffic mostly bypasses L2 either way.)
	int region;
	{
		const int n = gridDim.x, x = blockIdx.x & 7;
		int start = 0;
		for (int y = 0; y < x; ++y) start += (n - y + 7) >> 3;
		region = start + (blockIdx.x >> 3);
	}
	const int gxr = region % p.GX;
	const int gyr = region / p.GX;
	const int x0 = gxr * kResRW;
	const int y0 = gyr * p.RH;
	const int rwv = min(kResRW, p.W - x0);  // valid columns / rows of this region
	const int rhv = min(p.RH, p.H - y0);
	volatile int *failFlag = reinterpret_cast<volatile int *>(smem + kResOffMisc);
	float *ldsBias = reinterpret_cast<float *>(smem + kResOffMisc + 64);
	const unsigned genTag = (*p.gen) << 8;  // uniform; never matches a previous launch

	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	    (__attribute__((address_space(3))) unsigned char *)smem));

	// ---- zero both buffers (border, out-of-image area and overrun pads stay zero) ----
	for (int i = tid; i < kResOffMisc / 16; i += 256) {
		reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
	}
	if (tid == 0) *failFlag = 0;
	__syncthreads();

	// ---- first layer's input: region + halo straight from the complete global tensor;
	//      pixels outside the image stay zero (the buffers were just cleared) ----
	{
		const T *in = static_cast<const T *>(p.in);
		const int nPix = (rhv + 2) * kResPitch;
		const int nInstr = (nPix + 7) / 8;  // 8 pixels (1 KiB) per wave-instruction, rows back to back
		for (int i = wave; i < nInstr; i += 4) {
			const int q = i * 8 + (lane >> 3);
			const int rr = q / kResPitch, cc = q - rr * kResPitch;
			const int c = (lane & 7) ^ ((cc >> 1) & 7);
			const int gy = y0 - 1 + rr, gx = x0 - 1 + cc;
			if (q < nPix && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
				glds16(in + ((size_t)gy * p.inPitch + gx) * 64 + c * 8, smem + kResOffA + i * 1024);
			}
		}
	}

	// ---- register-resident weights: A fragment f = (dy*3+dx)*4+ks of this wave's cout half ----
	// Buffer loads: per-lane byte offset in ONE VGPR, fragment/layer offset scalar, so
	// a load costs no address VALU and no temporaries (a flat load 2048*f bytes away is
	// out of immediate range and needs a 64-bit add per load).
	typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
	const __amdgpu_buffer_rsrc_t wRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    const_cast<void *>(p.weights), 0, p.nLayers * 73728, 0x00020000);
	const unsigned wLaneOff = (unsigned)((hh * 64 + ch * 32 + px) * 16);
	auto loadWeightFrag = [&](int layer, int f) -> Vec8<T> {
		const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(wRsrc, wLaneOff, layer * 73728 + f * 2048, 0);
		return __builtin_bit_cast(Vec8<T>, v);
	};
	Vec8<T> w0[36], w1[36];
	auto loadWeights = [&](int layer, Vec8<T>(&w)[36]) {
#pragma unroll
		for (int f = 0; f < 36; ++f) w[f] = loadWeightFrag(layer, f);
	};
	loadWeights(0, w0);
	float biasNext = 0.f;  // wave 0: next layer's bias in flight (one value per lane)
	if (wave == 0) ldsBias[lane] = p.bias[lane];

	// per-lane LDS address parts (the swizzle depends only on the column: rows are 32 px)
	// B fragment of macro-step (dx, ks): byte offset inside a row =
	// colBase[dx] + (((ks*2+hh) ^ colSwz[dx]) << 4); 6 registers instead of 12
	unsigned colBase[3], colSwz[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
		const int cq = px + dx;
		colBase[dx] = cq * 128;
		colSwz[dx] = (cq >> 1) & 7;
	}
	unsigned outsw[4];  // [g]: byte offset of this lane's 4 output channels inside a row
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		const int cq = px + 1;
		outsw[g] = cq * 128 + (((ch * 4 + g) ^ ((cq >> 1) & 7)) << 4) + hh * 8;
	}

	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	// VARIANT 4 (diagnostic build only): per-wave cycle sums of the phases below
	u64 prof[7] = {0, 0, 0, 0, 0, 0, 0};
	auto stamp = [&]() -> u64 {
		if constexpr (VARIANT == 4) {
			__builtin_amdgcn_sched_barrier(0);
			u64 t;
			asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
			__builtin_amdgcn_sched_barrier(0);
			return t;
		} else {
			return 0;
		}
	};
	// ------------------------------------------------------------------------
	// one convolution layer over the region: in/out are LDS buffer offsets
	// ------------------------------------------------------------------------
	auto computeLayer = [&](auto residualTag, const int layer, const int inOff, const int outOff,
	                        const Vec8<T>(&w)[36]) {
		constexpr bool residual = decltype(residualTag)::value;

		const float *biasPtr = ldsBias + (layer & 1) * 64 + ch * 32 + 4 * hh;

		// rows of this wave: full pairs j = rp, rp+2, ... and, for an odd region
		// height, the last row as a single-row unit on the wave group with fewer pairs
		const int np2 = rhv >> 1;
		const int nUnits = np2 + (rhv & 1);
		Vec8<T> fb[2][4];
		auto issue = [&](unsigned rowAddr, int m, int set, int j) {
			const int dx = m >> 2, ks = m & 3;
			const unsigned a = rowAddr + colBase[dx] + (((unsigned)(ks * 2 + hh) ^ colSwz[dx]) << 4);
			if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][0]) : "v"(a));
			else if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:4352" : "=v"(fb[set][1]) : "v"(a));
			else if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:8704" : "=v"(fb[set][2]) : "v"(a));
			else asm volatile("ds_read_b128 %0, %1 offset:13056" : "=v"(fb[set][3]) : "v"(a));
		};
		auto rowAddrOf = [&](int unit) { return ldsBase + inOff + (2 * unit) * kResRowBytes; };

		// ROWS = 2: a row pair; ROWS = 1: the odd last row.  `primed`: this unit's
		// first 4 fragments were already issued by the previous unit's last step.
		auto unitBody = [&](auto rowsTag, const int unit, const bool primed, const int nextUnit) {
			constexpr int ROWS = decltype(rowsTag)::value;
			const u64 tu0 = stamp();
			constexpr int NR = ROWS + 2;      // input rows / fragment reads per macro-step
			constexpr int NM = 3 * ROWS;      // MFMAs per macro-step
			const int ra = 1 + 2 * unit;      // first output row (buffer row index)
			f32x16 acc[ROWS];
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const f32x4 bg = *reinterpret_cast<const f32x4 *>(biasPtr + 8 * g);
#pragma unroll
				for (int r = 0; r < ROWS; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = bg[i];
				}
			}
			const unsigned rowAddr = rowAddrOf(unit);
			if (!(VARIANT & 2)) {
				if (!primed) {
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
					__builtin_amdgcn_sched_barrier(0);
#pragma unroll
					for (int j = 0; j < NR; ++j) issue(rowAddr, 0, 0, j);
				}
#pragma unroll
				for (int m = 0; m < 12; ++m) {
					const int set = m & 1;
					const bool more = (m + 1 < 12);
					const int dx = m >> 2, ks = m & 3;
#pragma unroll
					for (int k = 0; k < NM; ++k) {
						// MFMA k = (dy, r): ROWS=2 -> (k>>1, k&1); ROWS=1 -> (k, 0); it needs
						// fragment r+dy, fragments are read (and return) in order 0..NR-1
						const int dy = ROWS == 2 ? (k >> 1) : k;
						const int r = ROWS == 2 ? (k & 1) : 0;
						const int need = r + dy;
						const bool fresh = ROWS == 2 ? (k == 0 || k == 1 || k == 3 || k == 5) : true;
						if (fresh) {
							// outstanding allowed = younger reads of this step + next step's issued so far
							const int issuedNext = more ? (k < NR ? k : NR) : 0;
							const int allowed = (NR - 1 - need) + issuedNext;
							if (allowed >= 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
							else if (allowed == 3) asm volatile("s_waitcnt lgkmcnt(3)" ::: "memory");
							else if (allowed == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
							else if (allowed == 1) asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
							else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
							__builtin_amdgcn_sched_barrier(0);
						}
						acc[r] = mfma32(w[(dy * 3 + dx) * 4 + ks], fb[set][need], acc[r]);
						if (more && k < NR) issue(rowAddr, m + 1, set ^ 1, k);
						__builtin_amdgcn_sched_barrier(0);
					}
				}
				// prime the next unit (always a pair or single with >= 3 input rows): its
				// first fragments travel while this unit's epilogue runs
				if (nextUnit >= 0) {
					const unsigned na = rowAddrOf(nextUnit);
					const bool nextSingle = (nextUnit == np2);
#pragma unroll
					for (int j = 0; j < 3; ++j) issue(na, 0, 0, j);
					if (!nextSingle) issue(na, 0, 0, 3);
					__builtin_amdgcn_sched_barrier(0);
				}
			}
			const u64 tu1 = stamp();
			// ---- epilogue: (+residual) ReLU, 16-bit, into the output buffer interior ----
			// (row ra + r <= rhv always: units are whole pairs, or the odd last row)
			if (px < rwv) {
				// The residual is read-modify-write in place.  All reads of the unit
				// first, then the arithmetic and the writes: left to the compiler every
				// group is read -> wait -> write -> next read (it cannot prove the groups
				// do not alias), i.e. 8 exposed LDS round trips per unit.
				Vec4<T> rv[ROWS][4];
				if (residual) {
#pragma unroll
					for (int r = 0; r < ROWS; ++r) {
#pragma unroll
						for (int g = 0; g < 4; ++g) {
							rv[r][g] = *reinterpret_cast<const Vec4<T> *>(
							    smem + outOff + (ra + r) * kResRowBytes + outsw[g]);
						}
					}
					asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				}
#pragma unroll
				for (int r = 0; r < ROWS; ++r) {
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						float v[4];
#pragma unroll
						for (int i = 0; i < 4; ++i) v[i] = acc[r][4 * g + i];
						if (residual) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] += static_cast<float>(rv[r][g][i]);
						}
						Vec4<T> o = {static_cast<T>(v[0]), static_cast<T>(v[1]), static_cast<T>(v[2]),
						    static_cast<T>(v[3])};
						*reinterpret_cast<Vec4<T> *>(smem + outOff + (ra + r) * kResRowBytes + outsw[g]) =
						    reluPacked<T>(o);
					}
				}
			}
			const u64 tu2 = stamp();
			prof[5] += tu1 - tu0;
			prof[6] += tu2 - tu1;
		};

		// pairs u = rp, rp+2, ... ; the odd last row goes to the wave group with fewer
		// pairs (group 0 when both have the same number)
		using R2 = std::integral_constant<int, 2>;
		using R1 = std::integral_constant<int, 1>;
		const bool mySingle = (rhv & 1) && rp == (np2 & 1);
		bool primed = false;
		for (int u = rp; u < np2; u += 2) {
			const int nu = (u + 2 < np2) ? u + 2 : (mySingle ? np2 : -1);
			unitBody(R2{}, u, primed, nu);
			primed = nu >= 0 && !(VARIANT & 2);
		}
		if (mySingle) unitBody(R1{}, np2, primed, -1);
		(void)nUnits;
	};

	// ------------------------------------------------------------------------
	// edge ring -> mailbox (publish) and neighbours' mailboxes -> halo ring
	// ------------------------------------------------------------------------
	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	const __amdgpu_buffer_rsrc_t mailRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)p.mail, 0, (int)((size_t)p.GX * p.GY * 2 * kResMailSlots * 16), 0x00020000);
	constexpr int kSc1 = 16;  // cache-policy bit of sc1 (write-through store / L2-bypassing load)
	// LDS address of 16-byte chunk c of pixel (rr, cc) in buffer `off`
	auto ldsChunk = [&](int off, int rr, int cc, int c) -> unsigned char * {
		return smem + off + rr * kResRowBytes + cc * 128 + ((c ^ ((cc >> 1) & 7)) << 4);
	};
	// Self-validating slots: every tower output is post-ReLU (>= 0), so the sign bit
	// of each of the 8 values in a 16-byte slot is free; it carries one bit of an
	// 8-bit epoch tag {generation & 3, layer + 1}.  Consecutive writes to the same
	// slot (layers l-2, l, l+2, ... and the previous launch) always differ in tag,
	// so a consumer can tell "new" from "old" from the payload itself and the
	// producer needs neither a drain nor a release: ONE hop instead of
	// store-ack -> flag -> load.  The flag below is only a hint that keeps 65k
	// threads from polling the fabric before the data can possibly be there.
	auto tagMasks = [&](int layer, u32x4 *m) {
		const unsigned t = ((genTag >> 8) & 3u) << 6 | (unsigned)((layer + 1) & 63);
#pragma unroll
		for (int k = 0; k < 4; ++k) {
			(*m)[k] = ((t >> (2 * k)) & 1u) << 15 | ((t >> (2 * k + 1)) & 1u) << 31;
		}
	};
	// `layer`: the layer whose output (in buffer `off`) is published
	auto publish = [&](int off, int layer) {
		// (the caller has just passed the workgroup barrier: the region's output is in LDS)
		u32x4 tm;
		tagMasks(layer, &tm);
		const unsigned base = (unsigned)((region * 2 + ((layer + 1) & 1)) * kResMailSlots) * 16u;
#pragma unroll
		for (int it = 0; it < kResMailSlots / 256; ++it) {
			const int idx = it * 256 + tid;
			const int strip = idx >> 8, e = (idx >> 3) & 31, c = idx & 7;
			int rr, cc;
			bool valid;
			if (strip == 0) { rr = 1; cc = e + 1; valid = e < rwv; }
			else if (strip == 1) { rr = rhv; cc = e + 1; valid = e < rwv; }
			else if (strip == 2) { rr = e + 1; cc = 1; valid = e < rhv; }
			else { rr = e + 1; cc = rwv; valid = e < rhv; }
			if (valid) {
				u32x4 v = *reinterpret_cast<const u32x4 *>(ldsChunk(off, rr, cc, c));
				v = (v & 0x7fff7fffu) | tm;
				__builtin_amdgcn_raw_buffer_store_b128(v, mailRsrc, base + idx * 16, 0, kSc1);
			}
		}
		if (tid == 0) {  // hint only: no drain, no barrier
			__hip_atomic_store((gu32 *)(p.flag + region), genTag | (unsigned)(layer + 1),
			    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		}
	};
	// fills the halo ring of buffer `off` with the neighbours' output of layer `layer`;
	// returns false on timeout (uniform across the workgroup)
	constexpr bool kUseHint = false;  // measured: polling a hint flag first is not faster than sweeping
	auto fillHalo = [&](int off, int layer) -> bool {
		const unsigned want = genTag | (unsigned)(layer + 1);
		(void)want;
		const u64 t0 = __builtin_amdgcn_s_memrealtime();
		if (wave == 0 && kUseHint) {
			bool ready = true;
			const gu32 *f = nullptr;
			if (lane < 8) {
				const int k = lane < 4 ? lane : lane + 1;  // skip the centre of the 3x3
				const int nx = gxr + (k % 3) - 1, ny = gyr + (k / 3) - 1;
				if (nx >= 0 && nx < p.GX && ny >= 0 && ny < p.GY) {
					f = (const gu32 *)(p.flag + ny * p.GX + nx);
					ready = false;
				}
			}
			while (!__all(ready)) {
				if (!ready) {
					const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
					ready = (v - want) <= 1u;  // a neighbour is at most one layer ahead
				}
				if (__builtin_amdgcn_s_memrealtime() - t0 > kResTimeoutTicks) break;  // the sweep reports it
				__builtin_amdgcn_s_sleep(1);
			}
		}
		if (kUseHint) __syncthreads();
		const int par = (layer + 1) & 1;
		u32x4 tm;
		tagMasks(layer, &tm);
		constexpr int NS = kResMailSlots / 256 + 1;  // 4 sides x 32 entries x 8 chunks, + the 4 corners
		unsigned hsrc[NS];
		unsigned char *hd[NS];
		unsigned pending = 0;
#pragma unroll
		for (int it = 0; it < NS; ++it) {
			int nx = gxr, ny = gyr, strip, se, rr, cc, c;
			bool valid;
			if (it < NS - 1) {
				const int idx = it * 256 + tid;
				const int hp = idx >> 3;
				c = idx & 7;
				const int side = hp >> 5, e = hp & 31;
				// side 0: row above, 1: row below, 2: column left, 3: column right
				if (side < 2) {
					ny += side == 0 ? -1 : 1;
					strip = side == 0 ? 1 : 0;  // their bottom row / their top row
					rr = side == 0 ? 0 : rhv + 1;
					se = e;
					cc = e + 1;
					valid = e < rwv;
				} else {
					nx += side == 2 ? -1 : 1;
					strip = side == 2 ? 3 : 2;  // their right column / their left column
					se = e;
					rr = e + 1;
					cc = side == 2 ? 0 : rwv + 1;
					valid = e < rhv;
				}
			} else {
				// corners: threads 0..31 = 4 corners x 8 chunks; the diagonal neighbour's
				// bottom/top row strip, last/first entry (interior columns are 32 wide)
				const int k = tid >> 3;
				c = tid & 7;
				const bool up = k < 2, left = (k & 1) == 0;
				ny += up ? -1 : 1;
				nx += left ? -1 : 1;
				strip = up ? 1 : 0;
				se = left ? kResRW - 1 : 0;
				rr = up ? 0 : rhv + 1;
				cc = left ? 0 : rwv + 1;
				valid = tid < 32;
			}
			valid = valid && nx >= 0 && nx < p.GX && ny >= 0 && ny < p.GY;
			const int nreg = valid ? ny * p.GX + nx : region;
			hsrc[it] = (unsigned)((nreg * 2 + par) * kResMailSlots + (strip * 32 + se) * 8 + c) * 16u;
			hd[it] = ldsChunk(off, rr, cc, c);
			if (valid) pending |= 1u << it;
		}
		// sweep: all loads of a pass in flight together, sc1 (never a stale L1/L2 line);
		// a slot is accepted only when all 8 tag bits match
		while (__any(pending != 0)) {
			u32x4 hv[NS];
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				hv[it] = __builtin_amdgcn_raw_buffer_load_b128(mailRsrc, hsrc[it], 0, kSc1);
			}
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				const u32x4 tg = hv[it] & 0x80008000u;
				const bool ok = tg[0] == tm[0] && tg[1] == tm[1] && tg[2] == tm[2] && tg[3] == tm[3];
				if ((pending >> it & 1u) && ok) {
					*reinterpret_cast<u32x4 *>(hd[it]) = hv[it] & 0x7fff7fffu;
					pending &= ~(1u << it);
				}
			}
			if (pending != 0) {
				if (__builtin_amdgcn_s_memrealtime() - t0 > kResTimeoutTicks) {
					*failFlag = 1;
					__hip_atomic_store((gu32 *)p.error, 0x700u + (unsigned)layer, __ATOMIC_RELAXED,
					    __HIP_MEMORY_SCOPE_SYSTEM);
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
		}
		__syncthreads();
		return *failFlag == 0;
	};

	// ------------------------------------------------------------------------
	// the tower: conv1 X->T (weights w0), conv2 T->X (+X) (weights w1)
	// ------------------------------------------------------------------------
	// Layer i reads buffer (i even ? A : B) and writes the other one: conv1 X->T, and
	// conv2 T->X adds the residual already sitting in its output buffer.  With a head
	// layer (generator conv_1) everything shifts by one.  Layer i uses weight set i&1.
	const int L = p.nLayers;
	// RES / PAR are compile-time: a runtime `if (residual)` around the epilogue's LDS
	// read makes hipcc branch and wait per element (+1 us per layer, measured).
	auto layerStep = [&](auto resTag, auto parTag, const int i, const Vec8<T>(&wc)[36],
	                     Vec8<T>(&wn)[36]) -> bool {
		constexpr int PAR = decltype(parTag)::value;
		constexpr int inOff = PAR ? kResOffB : kResOffA;
		constexpr int outOff = PAR ? kResOffA : kResOffB;
		const bool more = i + 1 < L;
		const u64 t0 = stamp();
		// vmcnt is in-order: the next layer's weight stream (36 loads per lane) is issued
		// AFTER the halo loads so they never queue behind it, and lands behind the MFMAs.
		if (i > 0 && xchg) {
			if (!fillHalo(inOff, i - 1)) return false;
		}
		const u64 t1 = stamp();
		// (Interleaving these 36 loads into the first unit's MFMA loop was tried: the
		// burst costs ~2.2k cycles of issue stall per layer -- four waves push 144 KB
		// through the CU's 64 B/clk address path -- but the interleaved form was no
		// faster end to end and doubled the code.)
		if (more) {
			loadWeights(i + 1, wn);
			if (wave == 0) biasNext = p.bias[(i + 1) * 64 + lane];
		}
		const u64 t2 = stamp();
		computeLayer(resTag, i, inOff, outOff, wc);
		const u64 t3 = stamp();
		if (more && wave == 0) ldsBias[((i + 1) & 1) * 64 + lane] = biasNext;
		if (more && xchg) {
			__syncthreads();  // (publish starts with this barrier; split out for the profile)
			const u64 t4 = stamp();
			publish(outOff, i);
			const u64 t5 = stamp();
			prof[3] += t4 - t3;
			prof[4] += t5 - t4;
		} else {
			__syncthreads();
		}
		prof[0] += t1 - t0;
		prof[1] += t2 - t1;
		prof[2] += t3 - t2;
		return true;
	};
	using No = std::false_type;
	using Yes = std::true_type;
	using P0 = std::integral_constant<int, 0>;
	using P1 = std::integral_constant<int, 1>;
	if (HEAD) {  // conv_1, then (conv1, conv2+skip) pairs: L is odd
		if (!layerStep(No{}, P0{}, 0, w0, w1)) return;
		for (int i = 1; i + 1 < L; i += 2) {
			if (!layerStep(No{}, P1{}, i, w1, w0)) return;
			if (!layerStep(Yes{}, P0{}, i + 1, w0, w1)) return;
		}
	} else {  // (conv1, conv2+skip) pairs: L is even
		for (int i = 0; i + 1 < L; i += 2) {
			if (!layerStep(No{}, P0{}, i, w0, w1)) return;
			if (!layerStep(Yes{}, P1{}, i + 1, w1, w0)) return;
		}
	}
	const int finalOff = (L & 1) ? kResOffB : kResOffA;
	if constexpr (VARIANT == 4) {
		if (lane == 0 && p.debug != nullptr) {
			for (int k = 0; k < 7; ++k) p.debug[(region * 4 + wave) * 8 + k] = prof[k];
		}
	}

	// ---- last block output: region interior -> global tower-layout tensor ----
	{
		T *out = static_cast<T *>(p.out);
		for (int i = tid; i < rhv * kResRW * 8; i += 256) {
			const int c = i & 7;
			const int pxl = (i >> 3) % kResRW;
			const int row = (i >> 3) / kResRW;
			if (pxl < rwv) {
				const int rr = row + 1, cc = pxl + 1;
				const uint4 v = *reinterpret_cast<const uint4 *>(
				    smem + finalOff + rr * kResRowBytes + cc * 128 + ((c ^ ((cc >> 1) & 7)) << 4));
				*reinterpret_cast<uint4 *>(
				    out + ((size_t)(y0 + rr) * p.pitch + x0 + cc) * 64 + c * 8) = v;
			}
		}
	}
}

template <typename T, int VARIANT, bool HEAD>
void launchResidentT(const ResidentParams &p, hipStream_t stream) {
	auto kern = tower_resident_kernel<T, VARIANT, HEAD>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kResLds, &ldsDone, "resident tower");
	if (p.bumpGeneration) {
		hipLaunchKernelGGL(bump_generation_kernel, dim3(1), dim3(1), 0, stream,
		    const_cast<unsigned *>(p.gen));
	}
	// (g_ResidentFault > 0, tests only: some regions are never computed, their neighbours'
	// bounded waits expire and the error path runs)
	const int grid = p.GX * p.GY - (g_ResidentFault < p.GX * p.GY ? g_ResidentFault : 0);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kResLds, stream, p);
	hipCheckLaunch("tower_resident");
}

// ---------------------------------------------------------------------------
// flow input packing
// ---------------------------------------------------------------------------
__device__ __forceinline__ float preprocessU8(unsigned v) {
	// PreprocessLayer: x / 255 - 0.5 (reference keras_layers.py:208).  Multiply by the
	// f32 reciprocal: at most 1 ulp (6e-8) from the correctly rounded quotient, far below
	// the 16-bit activations and the 1/255 output step, and one op instead of the ~10 of
	// an IEEE division (the tail evaluates 12 per lane); 0 and 255 map to -0.5 and 0.5
	// exactly.
	return static_cast<float>(v) * (1.0f / 255.0f) - 0.5f;
}

// normalize_brightness (reference models.py:772-779, utils.py:151): the scalar
// b = mean(x * BGR_LUMA * 3) over H, W, C of the preprocessed frame
//   = sum_c luma_c * (S_c / (255 N) - 0.5)
// from the three exact integer channel sums S_c (order-independent, so the
// reduction is deterministic).  sums == nullptr: feature off, b = 0.
__device__ __forceinline__ float brightnessOf(const unsigned *__restrict__ sums, float invN) {
	if (sums == nullptr) return 0.0f;
	const float mb = static_cast<float>(sums[0]) * invN / 255.0f - 0.5f;
	const float mg = static_cast<float>(sums[1]) * invN / 255.0f - 0.5f;
	const float mr = static_cast<float>(sums[2]) * invN / 255.0f - 0.5f;
	return 0.114f * mb + 0.587f * mg + 0.2989f * mr;
}

__global__ __launch_bounds__(1024) void frame_sums_kernel(const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, int H, int W, unsigned *__restrict__ sums) {
	__shared__ unsigned part[3][16];
	unsigned s0 = 0, s1 = 0, s2 = 0;
	for (int i = threadIdx.x; i < H * W; i += 1024) {
		const int y = i / W, x = i - y * W;
		const unsigned v = *reinterpret_cast<const unsigned *>(frame + y * frameStride + x * 4);
		s0 += v & 0xff;
		s1 += (v >> 8) & 0xff;
		s2 += (v >> 16) & 0xff;
	}
#pragma unroll
	for (int o = 32; o > 0; o >>= 1) {
		s0 += __shfl_down(s0, o);
		s1 += __shfl_down(s1, o);
		s2 += __shfl_down(s2, o);
	}
	const int wv = threadIdx.x >> 6;
	if ((threadIdx.x & 63) == 0) {
		part[0][wv] = s0;
		part[1][wv] = s1;
		part[2][wv] = s2;
	}
	__syncthreads();
	if (threadIdx.x < 3) {
		unsigned t = 0;
		for (int k = 0; k < 16; ++k) t += part[threadIdx.x][k];
		sums[threadIdx.x] = t;
	}
}

template <typename T>
__global__ __launch_bounds__(256) void pack_frames_kernel(const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, const T *__restrict__ prev, T *__restrict__ cur, int H, int W,
    int PH, int PW, int padTop, int padLeft, int numInputs, const unsigned *__restrict__ sums,
    unsigned *generation) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	// first kernel of every frame: bump the launch generation the resident tower tags
	// its halo slots with (saves a 1-thread launch)
	if (idx == 0 && generation != nullptr) *generation = *generation + 1;
	if (idx >= PH * PW) return;
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));
	const int py = idx / PW;
	const int pxx = idx - py * PW;
	const int y = py - padTop;
	const int x = pxx - padLeft;
	float c0 = 0.f, c1 = 0.f, c2 = 0.f;  // ZeroPadding2D after preprocess: 0.0 in the border
	if (y >= 0 && y < H && x >= 0 && x < W) {
		const unsigned v = *reinterpret_cast<const unsigned *>(frame + y * frameStride + x * 4);
		// the flow net sees the brightness-normalised frame (models.py:779); the pad
		// border stays exactly zero (ZeroPadding2D comes after the subtraction)
		c0 = preprocessU8(v & 0xff) - bright;
		c1 = preprocessU8((v >> 8) & 0xff) - bright;
		c2 = preprocessU8((v >> 16) & 0xff) - bright;
	}
	const Vec8<T> p0 = *reinterpret_cast<const Vec8<T> *>(prev + (size_t)idx * 16);
	const Vec8<T> p1 = *reinterpret_cast<const Vec8<T> *>(prev + (size_t)idx * 16 + 8);
	T pv[16];
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		pv[i] = p0[i];
		pv[8 + i] = p1[i];
	}
	const int nch = 3 * numInputs;
	T o[16];
	o[0] = static_cast<T>(c0);
	o[1] = static_cast<T>(c1);
	o[2] = static_cast<T>(c2);
#pragma unroll
	for (int k = 3; k < 16; ++k) o[k] = (k < nch) ? pv[k - 3] : static_cast<T>(0.f);
	Vec8<T> o0, o1;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		o0[i] = o[i];
		o1[i] = o[8 + i];
	}
	*reinterpret_cast<Vec8<T> *>(cur + (size_t)idx * 16) = o0;
	*reinterpret_cast<Vec8<T> *>(cur + (size_t)idx * 16 + 8) = o1;
}

// ---------------------------------------------------------------------------
// 2x2 max-pool and TF1 bilinear x2 (8 channels = 16 B per thread)
// ---------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_kernel(
    const T *__restrict__ in, T *__restrict__ out, int H, int W, int C) {
	const int OH = H / 2, OW = W / 2, CC = C / 8;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= OH * OW * CC) return;
	const int c = idx % CC;
	const int pix = idx / CC;
	const int ox = pix % OW;
	const int oy = pix / OW;
	const T *base = in + ((size_t)(2 * oy) * W + 2 * ox) * C + c * 8;
	const Vec8<T> a = *reinterpret_cast<const Vec8<T> *>(base);
	const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(base + C);
	const Vec8<T> d = *reinterpret_cast<const Vec8<T> *>(base + (size_t)W * C);
	const Vec8<T> e = *reinterpret_cast<const Vec8<T> *>(base + (size_t)W * C + C);
	Vec8<T> o;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const float m = fmaxf(fmaxf(static_cast<float>(a[i]), static_cast<float>(b[i])),
		    fmaxf(static_cast<float>(d[i]), static_cast<float>(e[i])));
		o[i] = static_cast<T>(m);
	}
	*reinterpret_cast<Vec8<T> *>(out + (size_t)pix * C + c * 8) = o;
}

template <typename T>
__global__ __launch_bounds__(256) void upsample2_kernel(
    const T *__restrict__ in, T *__restrict__ out, int H, int W, int C) {
	// tf.compat.v1.image.resize_bilinear(align_corners=False,
	// half_pixel_centers=False): src = dst / 2 (reference keras_layers.py:46-52)
	const int OH = H * 2, OW = W * 2, CC = C / 8;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= OH * OW * CC) return;
	const int c = idx % CC;
	const int pix = idx / CC;
	const int ox = pix % OW;
	const int oy = pix / OW;
	const int y0 = oy >> 1, x0 = ox >> 1;
	const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
	const float fy = (oy & 1) * 0.5f, fx = (ox & 1) * 0.5f;
	const Vec8<T> tl = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y0 * W + x0) * C + c * 8);
	const Vec8<T> tr = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y0 * W + x1) * C + c * 8);
	const Vec8<T> bl = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y1 * W + x0) * C + c * 8);
	const Vec8<T> br = *reinterpret_cast<const Vec8<T> *>(in + ((size_t)y1 * W + x1) * C + c * 8);
	Vec8<T> o;
#pragma unroll
	for (int i = 0; i < 8; ++i) {
		const float a = static_cast<float>(tl[i]), b = static_cast<float>(tr[i]);
		const float d = static_cast<float>(bl[i]), e = static_cast<float>(br[i]);
		const float top = a + (b - a) * fx;
		const float bot = d + (e - d) * fx;
		o[i] = static_cast<T>(top + (bot - top) * fy);
	}
	*reinterpret_cast<Vec8<T> *>(out + (size_t)pix * C + c * 8) = o;
}

// ---------------------------------------------------------------------------
// dense warp + space-to-depth + concat + pack
// ---------------------------------------------------------------------------
// One thread per (LR pixel, HR row i of its 4x4 block): 4 warped HR pixels x 3
// channels plus 4 spare slots = one 32-byte quarter of the pixel's 128-byte
// generator-input record.  Four consecutive lanes fill one record, a wavefront
// writes 2 KiB contiguously.
template <typename T>
__global__ __launch_bounds__(256) void warp_pack_kernel(const f16 *__restrict__ state,
    const float *__restrict__ flow, const std::uint8_t *__restrict__ frame,
    std::ptrdiff_t frameStride, T *__restrict__ out, int H, int W, int PW, int padTop,
    int padLeft, const unsigned *__restrict__ sums, f16 *__restrict__ preWarpOut) {
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= H * W * 4) return;
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));  // pre_warp += b (models.py:803)
	const int i = idx & 3;
	const int pix = idx >> 2;
	const int w = pix % W;
	const int h = pix / W;
	const int HH = H * 4, WW = W * 4;
	// depth-to-space(4) of the flow head is just this channel addressing:
	// flow[4h+i, 4w+j, k] = head[h, w, (i*4+j)*2 + k]  (keras_layers.py:175)
	const float *fp = flow + ((size_t)(h + padTop) * PW + (w + padLeft)) * 32 + i * 8;
	const f32x4 f0 = *reinterpret_cast<const f32x4 *>(fp);
	const f32x4 f1 = *reinterpret_cast<const f32x4 *>(fp + 4);
	const float fl[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
	T o[16];
	Vec4<f16> pw[4];  // the same 4 HR pixels in [4H][4W][4] f16 for the temporal filter
	const int Y = 4 * h + i;
#pragma unroll
	for (int j = 0; j < 4; ++j) {
		const int X = 4 * w + j;
		// tfa/dense_image_warp.py:232-245, 116-171
		const float qy = static_cast<float>(Y) - fl[2 * j];
		const float qx = static_cast<float>(X) - fl[2 * j + 1];
		const float fy = fminf(fmaxf(0.0f, floorf(qy)), static_cast<float>(HH - 2));
		const float fx = fminf(fmaxf(0.0f, floorf(qx)), static_cast<float>(WW - 2));
		const float ay = fminf(fmaxf(0.0f, qy - fy), 1.0f);
		const float ax = fminf(fmaxf(0.0f, qx - fx), 1.0f);
		const int y0 = static_cast<int>(fy), x0 = static_cast<int>(fx);
		const f16 *s0 = state + ((size_t)y0 * WW + x0) * 4;
		const f16 *s1 = s0 + (size_t)WW * 4;
		const Vec4<f16> tl = *reinterpret_cast<const Vec4<f16> *>(s0);
		const Vec4<f16> tr = *reinterpret_cast<const Vec4<f16> *>(s0 + 4);
		const Vec4<f16> bl = *reinterpret_cast<const Vec4<f16> *>(s1);
		const Vec4<f16> br = *reinterpret_cast<const Vec4<f16> *>(s1 + 4);
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			const float a = static_cast<float>(tl[c]), b = static_cast<float>(tr[c]);
			const float d = static_cast<float>(bl[c]), e = static_cast<float>(br[c]);
			const float top = ax * (b - a) + a;
			const float bot = ax * (e - d) + d;
			const float v = ay * (bot - top) + top + bright;
			o[j * 3 + c] = static_cast<T>(v);
			pw[j][c] = static_cast<f16>(v);
		}
		pw[j][3] = static_cast<f16>(0.f);
	}
	if (preWarpOut != nullptr) {
		f16 *d = preWarpOut + ((size_t)Y * WW + 4 * w) * 4;
#pragma unroll
		for (int j = 0; j < 4; ++j) *reinterpret_cast<Vec4<f16> *>(d + 4 * j) = pw[j];
	}
	float l0 = 0.f, l1 = 0.f, l2 = 0.f;
	if (i == 0) {
		const unsigned v = *reinterpret_cast<const unsigned *>(frame + h * frameStride + w * 4);
		l0 = preprocessU8(v & 0xff);
		l1 = preprocessU8((v >> 8) & 0xff);
		l2 = preprocessU8((v >> 16) & 0xff);
	}
	// spare slots: the LR frame rides in quarter 0, zeros elsewhere (x/255-0.5 of
	// a real pixel is never needed for i != 0, and 0.0 weights nothing)
	o[12] = static_cast<T>(l0);
	o[13] = static_cast<T>(l1);
	o[14] = static_cast<T>(l2);
	o[15] = static_cast<T>(0.f);
	Vec8<T> o0, o1;
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		o0[k] = o[k];
		o1[k] = o[8 + k];
	}
	T *dst = out + (size_t)pix * 64 + i * 16;
	*reinterpret_cast<Vec8<T> *>(dst) = o0;
	*reinterpret_cast<Vec8<T> *>(dst + 8) = o1;
}

// ---------------------------------------------------------------------------
// temporal moving-average output filter (scripts/inference/onnx/frame_moving_avg.py
// :146-302, its default mode: global L1 scene-cut gate with a sign function)
// ---------------------------------------------------------------------------
// Runs after the tail: `state` holds gen - b (f16, b = brightness scalar or 0),
// `preWarp` the warped previous output (+ b).  Pass 1 sums |gen - pre_warp| over
// every element into a 32.32 fixed-point accumulator (integer atomics: the result
// does not depend on the order, so the gate is deterministic); pass 2 blends and
// rewrites the state and the u8 frame unless the gate says "scene cut" (then the
// generator output written by the tail already is the result).
constexpr double kTemporalScale = 4294967296.0;  // 2^32

__global__ __launch_bounds__(256) void temporal_reduce_kernel(const f16 *__restrict__ state,
    const f16 *__restrict__ preWarp, size_t nPix, int lrPixels, const unsigned *__restrict__ sums,
    unsigned long long *__restrict__ acc) {
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(lrPixels));
	float s = 0.f;
	for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nPix; i += (size_t)gridDim.x * 256) {
		const Vec4<f16> g = *reinterpret_cast<const Vec4<f16> *>(state + i * 4);
		const Vec4<f16> q = *reinterpret_cast<const Vec4<f16> *>(preWarp + i * 4);
#pragma unroll
		for (int c = 0; c < 3; ++c) {
			s += fabsf(static_cast<float>(g[c]) + bright - static_cast<float>(q[c]));
		}
	}
#pragma unroll
	for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
	if ((threadIdx.x & 63) == 0) {
		atomicAdd(acc, static_cast<unsigned long long>(static_cast<double>(s) * kTemporalScale + 0.5));
	}
}

__global__ __launch_bounds__(256) void temporal_blend_kernel(f16 *__restrict__ state,
    const f16 *__restrict__ preWarp, std::uint8_t *__restrict__ outU8, std::ptrdiff_t outStride,
    int HH, int WW, int lrPixels, const unsigned *__restrict__ sums,
    const unsigned long long *__restrict__ acc, float strength, float threshold) {
	const double mean = static_cast<double>(*acc) / kTemporalScale / (3.0 * HH * WW);
	const double d = mean - static_cast<double>(threshold);
	const float c = d > 0.0 ? 1.0f : (d < 0.0 ? -1.0f : 0.0f);  // Sign (:229-232)
	if (c > 0.0f) return;                                        // scene cut: out = gen
	const float half = 0.5f * strength;
	const float m1 = half - c * half;         // weight of pre_warp (:272-279)
	const float m2 = c * half + 1.0f - half;  // weight of the generator output (:280-285)
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(lrPixels));
	const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (i >= (size_t)HH * WW) return;
	const int y = static_cast<int>(i / WW), x = static_cast<int>(i - (size_t)y * WW);
	const Vec4<f16> g = *reinterpret_cast<const Vec4<f16> *>(state + i * 4);
	const Vec4<f16> q = *reinterpret_cast<const Vec4<f16> *>(preWarp + i * 4);
	Vec4<f16> st;
	unsigned packed = 0;
#pragma unroll
	for (int ch = 0; ch < 3; ++ch) {
		const float r = static_cast<float>(q[ch]) * m1 + (static_cast<float>(g[ch]) + bright) * m2;
		st[ch] = static_cast<f16>(r - bright);
		const unsigned u = static_cast<unsigned>((r + 0.5f) * 255.0f);  // postprocess, truncating
		packed |= (u & 0xff) << (8 * ch);
	}
	st[3] = static_cast<f16>(0.f);
	*reinterpret_cast<Vec4<f16> *>(state + i * 4) = st;
	*reinterpret_cast<unsigned *>(outU8 + y * outStride + (std::ptrdiff_t)x * 4) = packed;
}

// ---------------------------------------------------------------------------
// generator tail
// ---------------------------------------------------------------------------
// One thread per mid-resolution pixel (2h+a, 2w+b): 32 channels in, 2x2 HR
// pixels x 3 channels out.
template <typename T>
__global__ __launch_bounds__(256) void tail_kernel(const T *__restrict__ y,
    const float *__restrict__ w2, const float *__restrict__ b2,
    const std::uint8_t *__restrict__ frame, std::ptrdiff_t frameStride,
    f16 *__restrict__ stateOut, std::uint8_t *__restrict__ outU8, std::ptrdiff_t outStride,
    int H, int W, const unsigned *__restrict__ sums) {
	const float bright = brightnessOf(sums, 1.0f / static_cast<float>(H * W));
	const int MW = 2 * W, MH = 2 * H;
	const int idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= MW * MH) return;
	const int mx = idx % MW;
	const int my = idx / MW;
	const int h = my >> 1, a = my & 1;
	const int w = mx >> 1, b = mx & 1;
	const T *src = y + ((size_t)h * W + w) * 128 + (a * 2 + b) * 32;
	float in[32];
#pragma unroll
	for (int k = 0; k < 4; ++k) {
		const Vec8<T> v = *reinterpret_cast<const Vec8<T> *>(src + k * 8);
#pragma unroll
		for (int i = 0; i < 8; ++i) in[k * 8 + i] = static_cast<float>(v[i]);
	}
	// LR neighbourhood for the bilinear x4 skip (UpscaleLayer, keras_layers.py:46-52)
	const int h1 = min(h + 1, H - 1), w1 = min(w + 1, W - 1);
	float lr[2][2][3];
#pragma unroll
	for (int yy = 0; yy < 2; ++yy) {
#pragma unroll
		for (int xx = 0; xx < 2; ++xx) {
			const unsigned v = *reinterpret_cast<const unsigned *>(
			    frame + (yy ? h1 : h) * frameStride + (xx ? w1 : w) * 4);
			lr[yy][xx][0] = preprocessU8(v & 0xff);
			lr[yy][xx][1] = preprocessU8((v >> 8) & 0xff);
			lr[yy][xx][2] = preprocessU8((v >> 16) & 0xff);
		}
	}
	const int WW = 4 * W;
#pragma unroll
	for (int a2 = 0; a2 < 2; ++a2) {
		const int Y = 2 * my + a2;
		const float fy = static_cast<float>(Y & 3) * 0.25f;
		Vec8<f16> st;
		unsigned pk[2];
#pragma unroll
		for (int b2i = 0; b2i < 2; ++b2i) {
			const int X = 2 * mx + b2i;
			const float fx = static_cast<float>(X & 3) * 0.25f;
			unsigned packed = 0;
#pragma unroll
			for (int c = 0; c < 3; ++c) {
				// ConvT 2x2 s2: y[2h+a,2w+b,o] = sum_c x[h,w,c] K[a,b,o,c] (+bias)
				float acc = b2[c];
				const float *wk = w2 + ((a2 * 2 + b2i) * 3 + c) * 32;
#pragma unroll
				for (int k = 0; k < 32; ++k) acc = fmaf(in[k], wk[k], acc);
				const float top = lr[0][0][c] + (lr[0][1][c] - lr[0][0][c]) * fx;
				const float bot = lr[1][0][c] + (lr[1][1][c] - lr[1][0][c]) * fx;
				const float skip = top + (bot - top) * fy;
				float r = fastTanh(acc) + skip;
				r = fminf(fmaxf(r, -0.5f), 0.5f);  // ClipLayer
				st[b2i * 4 + c] = static_cast<f16>(r - bright);  // fed-back state: output_raw - b (models.py:810)
				// PostprocessLayer + truncating cast (cuda_convert.cc.cu:76-81)
				const unsigned u = static_cast<unsigned>((r + 0.5f) * 255.0f);
				packed |= (u & 0xff) << (8 * c);
			}
			st[b2i * 4 + 3] = static_cast<f16>(0.f);
			pk[b2i] = packed;  // X byte = 0
		}
		*reinterpret_cast<Vec8<f16> *>(stateOut + ((size_t)Y * WW + 2 * mx) * 4) = st;
		*reinterpret_cast<uint2 *>(outU8 + Y * outStride + 2 * mx * 4) = make_uint2(pk[0], pk[1]);
	}
}


// ---------------------------------------------------------------------------
// fused generator tail on the matrix cores
// ---------------------------------------------------------------------------
// trunk [H][W][64] -> ConvT(2x2,s2,64->32)+BN+ReLU -> ConvT(2x2,s2,32->3)+bias -> tanh
// -> + bilinear x4 of the LR frame -> clip -> HR state (f16) and BGRX u8, in ONE
// pass (reference models.py:559-593, keras_layers.py:211-230, cuda_convert.cc.cu:76-81).
// The two-kernel form wrote and re-read a [H][W][128] tensor (66 MB per frame).
//   stage 1  D1[128][32 px] = W1[128][64] x X[64][32 px]: 4 cout blocks x 4 k-steps.
//            Cout block nb = (a*2+b) IS the mid-resolution pixel (2h+a, 2w+b)'s 32
//            channels, so after bias + ReLU each block goes to LDS pixel-major
//            (64 B per mid pixel) and is directly the B operand of
//   stage 2  D2[16][32 mid px] = W2[16][32] x Y[32][32 mid px], rows m = 4*(a'*2+b') + c:
//            lane (mid px, hh), register group g2 then holds the 3 channels of ONE HR
//            pixel (a' = g2, b' = hh).
// One workgroup = 8 LR rows x 32 px (4 waves x 2 rows); outputs are staged in LDS
// and written as whole rows, 16 B per lane.
constexpr int kTailLdsIn = 8 * 32 * 128;          // 32 KiB input tile
constexpr int kTailLdsW1 = 64 * 128 * 2;          // 16 KiB convT1 weights (fragment order)
constexpr int kTailLdsMid = 4 * 4 * 32 * 64;      // per wave: 4 mid-pixel groups x 32 px x 64 B = 8 KiB
constexpr int kTailLdsOut = 4 * (4 * 128 * 8 + 4 * 128 * 4);  // per wave: 4 HR rows x 128 px x (8 + 4) B
constexpr int kTailLds = kTailLdsIn + kTailLdsW1 + kTailLdsMid + kTailLdsOut;

struct TailFusedParams {
	const void *x;        // trunk, addressed at image pixel (0,0)
	int xPitch;           // row pitch in pixels
	const void *w1;       // convT1 as 1x1 conv 64->128, packConvWeights order with nb = 2
	const float *b1;      // [128]
	const void *w2;       // A fragments of convT2: [2 ks][64 lanes][8] 16-bit
	const float *b2;      // [3]
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	void *state;          // f16 [4H][4W][4]
	std::uint8_t *outU8;
	std::ptrdiff_t outStride;
	const unsigned *sums;
	int H, W;
};

template <typename T>
__global__ __launch_bounds__(256) void tail_fused_kernel(TailFusedParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *smI = smem;
	unsigned char *smW = smem + kTailLdsIn;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	unsigned char *smMid = smem + kTailLdsIn + kTailLdsW1 + wave * (kTailLdsMid / 4);
	unsigned char *smOut = smem + kTailLdsIn + kTailLdsW1 + kTailLdsMid + wave * (kTailLdsOut / 4);
	const int tx0 = blockIdx.x * 32, ty0 = blockIdx.y * 8;
	const T *__restrict__ x = static_cast<const T *>(p.x);
	const float bright = brightnessOf(p.sums, 1.0f / static_cast<float>(p.H * p.W));

	// ---- stage weights (linear) and the 8 x 32 input tile (swizzled chunks, zero outside) ----
	{
		const uint4 *src = reinterpret_cast<const uint4 *>(p.w1);
		uint4 *dst = reinterpret_cast<uint4 *>(smW);
#pragma unroll
		for (int k = 0; k < kTailLdsW1 / 16 / 256; ++k) dst[tid + k * 256] = src[tid + k * 256];
		uint4 v[8];
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int i = tid + k * 256;  // 8 rows x 32 px x 8 chunks = 2048
			const int q = i >> 3, c = i & 7;
			const int r = q >> 5, xx = q & 31;
			const int gy = min(ty0 + r, p.H - 1), gx = min(tx0 + xx, p.W - 1);
			v[k] = *reinterpret_cast<const uint4 *>(x + ((size_t)gy * p.xPitch + gx) * 64 + c * 8);
		}
#pragma unroll
		for (int k = 0; k < 8; ++k) {
			const int i = tid + k * 256;
			const int q = i >> 3, c = i & 7;
			*reinterpret_cast<uint4 *>(smI + q * 128 + ((c ^ ((q >> 1) & 7)) << 4)) = v[k];
		}
	}
	// convT2 A fragments (2 k-steps) and biases in registers
	Vec8<T> a2[2];
	a2[0] = reinterpret_cast<const Vec8<T> *>(p.w2)[lane];
	a2[1] = reinterpret_cast<const Vec8<T> *>(p.w2)[64 + lane];
	const float b2v[3] = {p.b2[0], p.b2[1], p.b2[2]};
	__syncthreads();

	for (int rw = 0; rw < 2; ++rw) {
		const int lr = wave * 2 + rw;  // LR row inside the tile
		const int h = ty0 + lr;
		// ---- stage 1: 128 couts x 32 px ----
		f32x16 acc[4];
#pragma unroll
		for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const f32x4 b = *reinterpret_cast<const f32x4 *>(p.b1 + nb * 32 + 8 * g + 4 * hh);
#pragma unroll
				for (int i = 0; i < 4; ++i) acc[nb][4 * g + i] = b[i];
			}
		}
#pragma unroll
		for (int ks = 0; ks < 4; ++ks) {
			const int q = lr * 32 + px;
			const int c = ks * 2 + hh;
			const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(smI + q * 128 + ((c ^ ((q >> 1) & 7)) << 4));
#pragma unroll
			for (int nb = 0; nb < 4; ++nb) {
				// weights: [cog = nb>>1][tap 0][ks][h][n = 64][8]
				const Vec8<T> a = *reinterpret_cast<const Vec8<T> *>(
				    smW + (nb >> 1) * (64 * 64 * 2) + (((ks * 2 + hh) * 64 + (nb & 1) * 32 + px) << 4));
				acc[nb] = mfma32(a, b, acc[nb]);
			}
		}
		// ReLU, 16-bit, to LDS as mid pixels: group nb, pixel px, 64 B (4 chunks, P = 4 swizzle)
#pragma unroll
		for (int nb = 0; nb < 4; ++nb) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				Vec4<T> o = {static_cast<T>(acc[nb][4 * g + 0]), static_cast<T>(acc[nb][4 * g + 1]),
				    static_cast<T>(acc[nb][4 * g + 2]), static_cast<T>(acc[nb][4 * g + 3])};
				*reinterpret_cast<Vec4<T> *>(smMid + nb * 2048 + px * 64 +
				                             ((g ^ ((px >> 2) & 3)) << 4) + hh * 8) = reluPacked<T>(o);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

		// LR neighbourhood for the bilinear x4 skip of this lane's LR pixel (px)
		const int w = tx0 + px;
		const int hc = min(h, p.H - 1), wc = min(w, p.W - 1);
		const int h1 = min(hc + 1, p.H - 1), w1 = min(wc + 1, p.W - 1);
		float lrv[2][2][3];
#pragma unroll
		for (int yy = 0; yy < 2; ++yy) {
#pragma unroll
			for (int xx = 0; xx < 2; ++xx) {
				const unsigned v = *reinterpret_cast<const unsigned *>(
				    p.frame + (yy ? h1 : hc) * p.frameStride + (xx ? w1 : wc) * 4);
				lrv[yy][xx][0] = preprocessU8(v & 0xff);
				lrv[yy][xx][1] = preprocessU8((v >> 8) & 0xff);
				lrv[yy][xx][2] = preprocessU8((v >> 16) & 0xff);
			}
		}
		// ---- stage 2 per mid-pixel group (a, b) ----
#pragma unroll
		for (int nb = 0; nb < 4; ++nb) {
			const int a = nb >> 1, bb = nb & 1;
			f32x16 d = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
			for (int ks = 0; ks < 2; ++ks) {
				const int c = ks * 2 + hh;
				const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(
				    smMid + nb * 2048 + px * 64 + ((c ^ ((px >> 2) & 3)) << 4));
				d = mfma32(a2[ks], b, d);
			}
			// lane (px, hh), g2: HR pixel (4h + 2a + g2, 4w + 2b + hh), channels d[4*g2 + 0..2]
#pragma unroll
			for (int g2 = 0; g2 < 2; ++g2) {
				const int yq = 2 * a + g2, xq = 2 * bb + hh;  // position inside the 4x4 HR block
				const float fy = yq * 0.25f, fx = xq * 0.25f;
				Vec4<f16> st;
				unsigned packed = 0;
#pragma unroll
				for (int c = 0; c < 3; ++c) {
					const float top = lrv[0][0][c] + (lrv[0][1][c] - lrv[0][0][c]) * fx;
					const float bot = lrv[1][0][c] + (lrv[1][1][c] - lrv[1][0][c]) * fx;
					const float skip = top + (bot - top) * fy;
					float r = fastTanh(d[4 * g2 + c] + b2v[c]) + skip;
					r = fminf(fmaxf(r, -0.5f), 0.5f);
					st[c] = static_cast<f16>(r - bright);
					const unsigned u = static_cast<unsigned>((r + 0.5f) * 255.0f);
					packed |= (u & 0xff) << (8 * c);
				}
				st[3] = static_cast<f16>(0.f);
				const int xcol = 4 * px + xq;  // HR column inside the 128-px row segment
				*reinterpret_cast<Vec4<f16> *>(smOut + yq * 1024 + xcol * 8) = st;
				*reinterpret_cast<unsigned *>(smOut + 4096 + yq * 512 + xcol * 4) = packed;
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
		// ---- coalesced output: 4 HR rows x 128 px ----
		if (h < p.H) {
			const int WW = 4 * p.W;
			const int nValidPx = min(128, 4 * (p.W - tx0));
			f16 *stateOut = static_cast<f16 *>(p.state);
#pragma unroll
			for (int yq = 0; yq < 4; ++yq) {
				const int Y = 4 * h + yq;
				// state: 1024 B per row = 64 lanes x 16 B (2 px per lane)
				if (2 * lane < nValidPx) {
					const uint4 v = *reinterpret_cast<const uint4 *>(smOut + yq * 1024 + lane * 16);
					*reinterpret_cast<uint4 *>(stateOut + ((size_t)Y * WW + 4 * tx0 + 2 * lane) * 4) = v;
				}
				// u8: 512 B per row = 64 lanes x 8 B (2 px per lane)
				if (2 * lane < nValidPx) {
					const uint2 v = *reinterpret_cast<const uint2 *>(smOut + 4096 + yq * 512 + lane * 8);
					*reinterpret_cast<uint2 *>(p.outU8 + Y * p.outStride + (4 * tx0 + 2 * lane) * 4) = v;
				}
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
	}
}

template <typename T>
void launchTailFusedT(const TailFusedParams &p, hipStream_t stream) {
	auto kern = tail_fused_kernel<T>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kTailLds, &ldsDone, "tail");
	dim3 grid((p.W + 31) / 32, (p.H + 7) / 8);
	hipLaunchKernelGGL(kern, grid, dim3(256), kTailLds, stream, p);
	hipCheckLaunch("tail_fused");
}

// ---------------------------------------------------------------------------
// staging helpers
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_rows_kernel(const std::uint8_t *__restrict__ src,
    std::ptrdiff_t srcStride, std::uint8_t *__restrict__ dst, std::ptrdiff_t dstStride,
    unsigned wordsPerRow, unsigned rows) {
	const unsigned idx = blockIdx.x * 256 + threadIdx.x;
	if (idx >= wordsPerRow * rows) return;
	const unsigned r = idx / wordsPerRow;
	const unsigned c = idx - r * wordsPerRow;
	const unsigned v = *reinterpret_cast<const unsigned *>(
	    src + static_cast<std::ptrdiff_t>(r) * srcStride + c * 4);
	*reinterpret_cast<unsigned *>(dst + static_cast<std::ptrdiff_t>(r) * dstStride + c * 4) = v;
}

template <typename T>
__global__ __launch_bounds__(256) void to_float_kernel(
    const T *__restrict__ in, float *__restrict__ out, size_t n) {
	const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (idx < n) out[idx] = static_cast<float>(in[idx]);
}

inline unsigned blocksFor(size_t n) { return static_cast<unsigned>((n + 255) / 256); }

}  // namespace

// ===========================================================================
// launchers
// ===========================================================================
void launchConv(DType dt, const ConvParams &p, hipStream_t stream) {
	if (dt == kF16) launchConvT<f16>(p, stream);
	else launchConvT<bf16>(p, stream);
}

void launchConvTower(DType dt, const ConvParams &p, hipStream_t stream) {
	const int pitch = towerPitch(p.W);
	const bool fits = p.taps == 9 && p.cin == 64 && p.cout == 64 && !p.outF32 && p.nb == 2 &&
	                  p.inPitch == pitch && p.outPitch == pitch &&
	                  (p.res == nullptr || p.resPitch == pitch);
	if (!fits) {
		launchConv(dt, p, stream);
		return;
	}
	if (dt == kF16) {
		launchTowerT<f16, 0>(p, stream);
		return;
	}
	switch (g_TowerVariant) {
	case 1: launchTowerT<bf16, 1>(p, stream); break;
	case 2: launchTowerT<bf16, 2>(p, stream); break;
	case 3: launchTowerT<bf16, 3>(p, stream); break;
	case 4: launchTowerT<bf16, 4>(p, stream); break;
	default: launchTowerT<bf16, 0>(p, stream); break;
	}
}

void setTowerVariant(int v) { g_TowerVariant = v; }
void setResidentFault(int n) { g_ResidentFault = n; }

void launchBumpGeneration(unsigned *generation, hipStream_t stream) {
	hipLaunchKernelGGL(bump_generation_kernel, dim3(1), dim3(1), 0, stream, generation);
	hipCheckLaunch("bump_generation");
}

bool residentTowerGeometry(int H, int W, int numCUs, int *GX, int *GY, int *RH) {
	const int gx = (W + kResRW - 1) / kResRW;
	const int gy = (H + kResMaxRH - 1) / kResMaxRH;
	if (gx * gy > numCUs) return false;
	*GX = gx;
	*GY = gy;
	*RH = (H + gy - 1) / gy;  // <= kResMaxRH, balances the last row of regions
	return true;
}

std::size_t residentMailboxBytes(int GX, int GY) {
	return static_cast<std::size_t>(GX) * GY * 2 * kResMailSlots * 16;
}

void launchResidentTower(DType dt, const ResidentTowerParams &q, hipStream_t stream) {
	ResidentParams p{};
	const std::size_t origin = towerOrigin(q.W) * 64 * 2;
	p.in = q.in;
	p.inPitch = q.inPitch ? q.inPitch : q.W;
	p.hasHead = q.hasHead;
	p.out = static_cast<unsigned char *>(q.out) - origin;
	p.weights = q.weights;
	p.bias = q.bias;
	p.mail = static_cast<uint4 *>(q.mailbox);
	p.gen = q.generation;
	p.flag = q.generation + 16;  // same small buffer: word 0 = generation, flags from byte 64
	p.debug = static_cast<unsigned long long *>(q.debug);
	p.error = q.error;
	p.H = q.H;
	p.W = q.W;
	p.pitch = towerPitch(q.W);
	p.GX = q.GX;
	p.GY = q.GY;
	p.RH = q.RH;
	p.nLayers = q.nLayers;
	p.bumpGeneration = q.bumpGeneration;
	if ((p.nLayers & 1) != (p.hasHead ? 1 : 0)) {
		throw std::invalid_argument("resident tower: layer count must be 2*blocks (+1 with a head)");
	}
	if (!p.hasHead) {
		if (dt == kF16) launchResidentT<f16, 0, false>(p, stream);
		else launchResidentT<bf16, 0, false>(p, stream);
		return;
	}
	if (dt == kF16) {
		launchResidentT<f16, 0, true>(p, stream);
		return;
	}
	switch (g_TowerVariant) {
	case 1: launchResidentT<bf16, 1, true>(p, stream); break;
	case 2: launchResidentT<bf16, 2, true>(p, stream); break;
	case 3: launchResidentT<bf16, 3, true>(p, stream); break;
	case 4: launchResidentT<bf16, 4, true>(p, stream); break;
	default: launchResidentT<bf16, 0, true>(p, stream); break;
	}
}

void launchPackFrames(DType dt, const std::uint8_t *frame, std::ptrdiff_t frameStride,
    const void *prevPacked, void *curPacked, int H, int W, int PH, int PW, int padTop,
    int padLeft, int numInputs, const unsigned *sums, unsigned *generation,
    hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)PH * PW);
	if (dt == kF16) {
		hipLaunchKernelGGL(pack_frames_kernel<f16>, dim3(nb), dim3(256), 0, stream, frame,
		    frameStride, static_cast<const f16 *>(prevPacked), static_cast<f16 *>(curPacked), H, W,
		    PH, PW, padTop, padLeft, numInputs, sums, generation);
	} else {
		hipLaunchKernelGGL(pack_frames_kernel<bf16>, dim3(nb), dim3(256), 0, stream, frame,
		    frameStride, static_cast<const bf16 *>(prevPacked), static_cast<bf16 *>(curPacked), H,
		    W, PH, PW, padTop, padLeft, numInputs, sums, generation);
	}
	hipCheckLaunch("pack_frames");
}

void launchMaxPool2(DType dt, const void *in, void *out, int H, int W, int C, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)(H / 2) * (W / 2) * (C / 8));
	if (dt == kF16) {
		hipLaunchKernelGGL(maxpool2_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), static_cast<f16 *>(out), H, W, C);
	} else {
		hipLaunchKernelGGL(maxpool2_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), static_cast<bf16 *>(out), H, W, C);
	}
	hipCheckLaunch("maxpool2");
}

void launchUpsample2(DType dt, const void *in, void *out, int H, int W, int C, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)(H * 2) * (W * 2) * (C / 8));
	if (dt == kF16) {
		hipLaunchKernelGGL(upsample2_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), static_cast<f16 *>(out), H, W, C);
	} else {
		hipLaunchKernelGGL(upsample2_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), static_cast<bf16 *>(out), H, W, C);
	}
	hipCheckLaunch("upsample2");
}

void launchWarpPack(DType dt, const void *state, const float *flow, const std::uint8_t *frame,
    std::ptrdiff_t frameStride, void *out, int H, int W, int PW, int padTop, int padLeft,
    const unsigned *sums, void *preWarpOut, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)H * W * 4);
	if (dt == kF16) {
		hipLaunchKernelGGL(warp_pack_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(state), flow, frame, frameStride, static_cast<f16 *>(out), H,
		    W, PW, padTop, padLeft, sums, static_cast<f16 *>(preWarpOut));
	} else {
		hipLaunchKernelGGL(warp_pack_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(state), flow, frame, frameStride, static_cast<bf16 *>(out), H,
		    W, PW, padTop, padLeft, sums, static_cast<f16 *>(preWarpOut));
	}
	hipCheckLaunch("warp_pack");
}

void launchTemporalFilter(void *state, const void *preWarp, std::uint8_t *outU8,
    std::ptrdiff_t outStride, int H, int W, const unsigned *sums, unsigned long long *acc,
    float strength, float threshold, hipStream_t stream) {
	const int HH = 4 * H, WW = 4 * W;
	const size_t nPix = (size_t)HH * WW;
	hipError_t e = hipMemsetAsync(acc, 0, sizeof(unsigned long long), stream);
	if (e != hipSuccess) throw std::runtime_error(std::string("hipMemsetAsync: ") + hipGetErrorString(e));
	hipLaunchKernelGGL(temporal_reduce_kernel, dim3(2048), dim3(256), 0, stream,
	    static_cast<const f16 *>(state), static_cast<const f16 *>(preWarp), nPix, H * W, sums, acc);
	hipCheckLaunch("temporal_reduce");
	hipLaunchKernelGGL(temporal_blend_kernel, dim3(blocksFor(nPix)), dim3(256), 0, stream,
	    static_cast<f16 *>(state), static_cast<const f16 *>(preWarp), outU8, outStride, HH, WW,
	    H * W, sums, acc, strength, threshold);
	hipCheckLaunch("temporal_blend");
}

void launchTail(DType dt, const void *y, const float *w2, const float *b2,
    const std::uint8_t *frame, std::ptrdiff_t frameStride, void *stateOut, std::uint8_t *outU8,
    std::ptrdiff_t outStride, int H, int W, const unsigned *sums, hipStream_t stream) {
	const unsigned nb = blocksFor((size_t)4 * H * W);
	if (dt == kF16) {
		hipLaunchKernelGGL(tail_kernel<f16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const f16 *>(y), w2, b2, frame, frameStride, static_cast<f16 *>(stateOut),
		    outU8, outStride, H, W, sums);
	} else {
		hipLaunchKernelGGL(tail_kernel<bf16>, dim3(nb), dim3(256), 0, stream,
		    static_cast<const bf16 *>(y), w2, b2, frame, frameStride, static_cast<f16 *>(stateOut),
		    outU8, outStride, H, W, sums);
	}
	hipCheckLaunch("tail");
}

void launchTailFused(DType dt, const TailFusedLaunch &q, hipStream_t stream) {
	TailFusedParams p{};
	p.x = q.x;
	p.xPitch = q.xPitch ? q.xPitch : q.W;
	p.w1 = q.w1;
	p.b1 = q.b1;
	p.w2 = q.w2;
	p.b2 = q.b2;
	p.frame = q.frame;
	p.frameStride = q.frameStride;
	p.state = q.state;
	p.outU8 = q.outU8;
	p.outStride = q.outStride;
	p.sums = q.sums;
	p.H = q.H;
	p.W = q.W;
	if (dt == kF16) launchTailFusedT<f16>(p, stream);
	else launchTailFusedT<bf16>(p, stream);
}

void launchFrameSums(const std::uint8_t *frame, std::ptrdiff_t frameStride, int H, int W,
    unsigned *sums, hipStream_t stream) {
	hipLaunchKernelGGL(frame_sums_kernel, dim3(1), dim3(1024), 0, stream, frame, frameStride, H, W,
	    sums);
	hipCheckLaunch("frame_sums");
}

void launchCopyRows(const std::uint8_t *src, std::ptrdiff_t srcStride, std::uint8_t *dst,
    std::ptrdiff_t dstStride, std::size_t rowBytes, std::size_t rows, hipStream_t stream) {
	const unsigned words = static_cast<unsigned>(rowBytes / 4);
	hipLaunchKernelGGL(copy_rows_kernel, dim3(blocksFor((size_t)words * rows)), dim3(256), 0,
	    stream, src, srcStride, dst, dstStride, words, static_cast<unsigned>(rows));
	hipCheckLaunch("copy_rows");
}

void launchToFloat(DType dt, const void *in, float *out, std::size_t n, hipStream_t stream) {
	if (dt == kF16) {
		hipLaunchKernelGGL(to_float_kernel<f16>, dim3(blocksFor(n)), dim3(256), 0, stream,
		    static_cast<const f16 *>(in), out, n);
	} else {
		hipLaunchKernelGGL(to_float_kernel<bf16>, dim3(blocksFor(n)), dim3(256), 0, stream,
		    static_cast<const bf16 *>(in), out, n);
	}
	hipCheckLaunch("to_float");
}

}  // namespace ju
