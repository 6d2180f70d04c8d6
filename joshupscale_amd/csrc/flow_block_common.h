// Shared pieces of the conv-pair kernels (flow_block_kernel, res_block_kernel, res_block_pipe_kernel) and of
// conv_splitk_kernel: the 34-pixel-wide LDS tile, its chunk swizzle, the LDS-DMA copy, the hand-issued
// fragment pipeline of a row pair (FbPair) and the one-multiplier activation.  Device code only; every
// translation unit that includes it gets its own (anonymous-namespace) copy.
#pragma once

#include "kernel_common.h"

namespace ju {

// res_block_kernels.hip: the persistent one-launch-per-block form of a 64-filter residual block
// (launchFlowBlock in flow_kernels.hip routes FlowBlockLaunch::residual there)
void launchResBlockPersistent(DType dt, const FlowBlockLaunch &q, hipStream_t stream);

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
	if constexpr (PB == 256) return col & 15u;  // (one pixel = one whole bank row: the 16 columns of a group take the 16 slots)
	else if constexpr (PB == 128) return (col >> 1) & 7u;
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

}  // namespace

}  // namespace ju
