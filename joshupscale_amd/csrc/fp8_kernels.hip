// gfx950 (CDNA4, MI355X): 8-bit (OCP e4m3) form of the generator's residual-block
// convolutions (reference scripts/training/models.py:193-254; the reference's own 8-bit
// deployment is TensorRT INT8, scripts/inference/tensorrt/quantize_int8.py:140-209).
// Quantisation scheme and operand packing: fp8.h.
//
//  * conv_tower_fp8_kernel   one 3x3 64->64 layer per launch on the block-scaled
//                            matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4:
//                            K = 64 is exactly one tap of the 64-channel input, so a
//                            32 px x 32 cout tile is 9 instructions
//  * quantize_tower_kernel   16-bit tower tensor -> e4m3 copy (once per frame, after
//                            the generator's conv_1)
#include "kernel_common.h"

namespace ju {

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// Persistent, TWO workgroups per CU (4 waves each, two waves per SIMD): with one wave per
// SIMD the tile pipeline is serial -- fragment reads, 36 matrix instructions, barrier,
// epilogue -- and the matrix cores idle for 2/3 of it (measured: 6400 cycles per tile
// against 2300 of MFMA); a second resident workgroup fills those gaps.  That caps a wave
// at 256 registers and a workgroup at 80 KiB of LDS, hence the split:
//   wave = (ch, rp): output channels 32 ch .. 32 ch + 31 of tile rows 4 rp .. 4 rp + 3.
// Each wave keeps the weights of its 32 output channels in registers as A fragments
// (9 taps x 8 VGPRs), so the LDS serves pixel fragments only (36 ds_read_b128 per 36
// matrix instructions).  Input tiles (8 rows x 32 px + halo = 10 x 34 records of 64 B)
// are double-buffered and fetched with global_load_lds; the wave's share of the skip
// connection (4 rows x 32 px x 64 B) is DMA'd into its private 8 KiB LDS slice at the top
// of the tile and the same slice then transposes the outputs, so that memory only sees
// whole 64-byte (stream) and 32-byte (e4m3) half records.  One barrier per tile.
constexpr int kF8Threads = 256;
constexpr int kF8TileBytes = 22 * 1024;  // 340 records of 64 B, rounded up to whole 1 KiB DMA writes
constexpr int kF8Slice = 8192;           // per wave
constexpr int kF8Lds = 2 * kF8TileBytes + 4 * kF8Slice;  // 77824: two workgroups per CU

// Addresses are "uniform base + uniform offset + 32-bit lane offset": the loop-invariant
// lane offsets cost one VGPR each.  (Buffer instructions -- descriptor + voffset + soffset
// -- would save the 64-bit adds, and were tried: with the e4m3 copy's stores as
// `buffer_store_dwordx4 ... sN offen` this kernel produced rare stale tiles at two workgroups
// per CU.  Round 3 bisected it (tools/probes/fp8_stale_tile.sh PROBE_SET=bisect: soffset 0,
// `s_nop 7` behind the store or a vmcnt(0) behind it each cure it) and reproduced the mechanism
// standalone (tools/probes/mubuf_store_data.hip): a buffer store with an SGPR soffset reads its
// data registers late when a second wave of the SIMD competes for the vector-memory issue, and
// hipcc's hazard recogniser exempts exactly that form from the store-data wait states -- the
// next store's address arithmetic / LDS read then reuses the registers too early.  Rule: no
// soffset-SGPR buffer stores in kernels with more than one wave per SIMD)
// 16 bytes per lane, memory -> LDS without a VGPR round trip (lands at l + lane * 16)
// Probe builds only (tools/probes/fp8_stale_tile.sh, never the product): JU_FP8_MUBUF_LD / _ST (or
// JU_FP8_MUBUF for both) move the same bytes with buffer instructions, JU_FP8_NOWAIT drops the
// explicit DMA waits in front of the barriers -- the ingredients of the stale-tile report,
// separately switchable.
#if defined(JU_FP8_MUBUF)
#define JU_FP8_MUBUF_LD 1
#define JU_FP8_MUBUF_ST 1
#endif
#if defined(JU_FP8_MUBUF_LD) || defined(JU_FP8_MUBUF_ST)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t probeBuffer(const void *base) {
	return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, 0xffc00000u, 0x00020000);  // (the loaders admit tensors up to 0xFFC00000 bytes: model.cpp)
}
#endif
__device__ __forceinline__ void dmaToLds16(const unsigned char *base, unsigned uniformOff, unsigned laneOff,
    void *l) {
#if defined(JU_FP8_MUBUF_LD)
	__builtin_amdgcn_raw_ptr_buffer_load_lds(probeBuffer(base), (__attribute__((address_space(3))) void *)l, 16,
	    static_cast<int>(laneOff), static_cast<int>(uniformOff), 0, 0);
#else
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + uniformOff + laneOff),
	    (__attribute__((address_space(3))) void *)l, 16, 0, 0);
#endif
}
// GROUP: 1 = the 16-bit stream's stores, 2 = the e4m3 copy's.  Further probe switches (bisecting
// the MUBUF-store anomaly, tools/probes/fp8_stale_tile.sh): JU_FP8_ST_GROUPS = mask of the groups
// that use buffer stores (default both), JU_FP8_ST_NOSOFF = whole offset in the VGPR (soffset 0),
// JU_FP8_ST_NOP = `s_nop 7` behind every buffer store, JU_FP8_ST_DRAIN = vmcnt(0) behind it.
#ifndef JU_FP8_ST_GROUPS
#define JU_FP8_ST_GROUPS 3
#endif
template <int GROUP>
__device__ __forceinline__ void store16(unsigned char *base, unsigned uniformOff, unsigned laneOff, i32x4 v) {
#if defined(JU_FP8_MUBUF_ST)
	if constexpr ((JU_FP8_ST_GROUPS & GROUP) != 0) {
#if defined(JU_FP8_ST_NOSOFF)
		__builtin_amdgcn_raw_buffer_store_b128(v, probeBuffer(base), static_cast<int>(laneOff + uniformOff), 0, 0);
#else
		__builtin_amdgcn_raw_buffer_store_b128(v, probeBuffer(base), static_cast<int>(laneOff), static_cast<int>(uniformOff), 0);
#endif
#if defined(JU_FP8_ST_NOP)
		asm volatile("s_nop 7" ::: "memory");
#endif
#if defined(JU_FP8_ST_DRAIN)
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
	} else {
		*reinterpret_cast<i32x4 *>(base + uniformOff + laneOff) = v;
	}
#else
	*reinterpret_cast<i32x4 *>(base + uniformOff + laneOff) = v;
#endif
}
#if defined(JU_FP8_NOWAIT)
#define JU_F8_DMA_WAIT() ((void)0)
#else
#define JU_F8_DMA_WAIT() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#endif

// e4m3 of four values, saturating (the hardware conversion returns NaN above 448:
// tools/probes/fp8_mfma_probe.hip), packed into one dword.  ReLU models: the values are
// non-negative, one min each; LEAKY (`activation: lrelu`): clamped on both sides.
template <bool LEAKY = false>
__device__ __forceinline__ int quantize4(float a, float b, float c, float d, float mul) {
	if constexpr (LEAKY) {
		a = __builtin_amdgcn_fmed3f(a * mul, -448.0f, 448.0f);
		b = __builtin_amdgcn_fmed3f(b * mul, -448.0f, 448.0f);
		c = __builtin_amdgcn_fmed3f(c * mul, -448.0f, 448.0f);
		d = __builtin_amdgcn_fmed3f(d * mul, -448.0f, 448.0f);
	} else {
		a = fminf(a * mul, 448.0f);
		b = fminf(b * mul, 448.0f);
		c = fminf(c * mul, 448.0f);
		d = fminf(d * mul, 448.0f);
	}
	int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
	return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

struct Fp8KernelParams {
	const unsigned char *in8;  // e4m3 tower-layout allocation start (row -1, col -1 of the image)
	const unsigned char *wgt;  // fp8.h packFp8TowerWeights
	const int *scaleA;         // [64] E8M0 codes, per output channel
	const float *bias;         // [64]
	const void *res;           // 16-bit stream, allocation start; nullptr: first conv of a block
	void *outT;                // 16-bit stream out (may alias res), allocation start; nullptr: none
	unsigned char *out8;       // e4m3 out, allocation start
	int scaleB;                // E8M0 code of the input tensor's scale 2^-ea
	float outMul;              // 2^ea of the output tensor
	float slope;               // LEAKY instantiations: LeakyReLU negative slope
	int H, W, pitch;           // pitch in pixels
	int tilesX, numTiles;
};

// STREAM: second conv of a block: + skip connection, writes the 16-bit stream too
template <typename T, bool STREAM, bool LEAKY = false>
__global__ __launch_bounds__(kF8Threads, 2) void conv_tower_fp8_kernel(Fp8KernelParams p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *smT = smem;
	const int tid = threadIdx.x;
	// wave-uniform values in SGPRs: addresses below are "uniform base + 32-bit lane offset"
	const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
	const int lane = tid & 63;
	const int px = lane & 31;
	const int hh = lane >> 5;
	const int ch = wave & 1;
	const int rp = wave >> 1;
	unsigned char *slice = smem + 2 * kF8TileBytes + wave * kF8Slice;
	// half-record transfers (skip DMA, stream store): instruction i moves pixels
	// 16 i + (lane >> 2) of the wave's 4 x 32; the chunk swizzle (pi >> 2) & 3 = (lane >> 4) & 3
	// does not depend on i
	const unsigned halfOff = (lane >> 2) * 128 + (((lane & 3) ^ ((lane >> 4) & 3)) << 4);  // bytes

	// XCD-aware tile order: workgroups b, b+8, ... share an XCD (and its L2); in every full
	// round each XCD gets a contiguous run of tiles.  The tiles of the last, partial round
	// are dealt out one per workgroup in launch order instead, which spreads them over all
	// XCDs and CUs (the contiguous order gave them all to the first XCDs: 6 tiles on some
	// CUs, 4 on the others).  The launcher keeps the grid a multiple of 8.
	const int nwg = gridDim.x;
	const int bid = blockIdx.x;
	const int slot = nwg >= 8 ? (bid & 7) * (nwg >> 3) + (bid >> 3) : bid;
	const int fullTiles = p.numTiles / nwg * nwg;
	auto tileAt = [&](int k) {
		const int base = k * nwg;
		if (base < fullTiles) return base + slot;
		if (fullTiles == 0) return (k == 0 && slot < p.numTiles) ? slot : -1;  // one round, surplus workgroups idle
		return (base == fullTiles && base + bid < p.numTiles) ? base + bid : -1;
	};


	// Tile in LDS: record q = r * 34 + x at q * 64; 16-byte chunk c of column x sits at
	// position c ^ ((x >> 2) & 3): a fragment read (16 consecutive columns per quarter wave)
	// covers all 64 banks, and the swizzle does not depend on the row, so the 6 rows of a
	// tap are immediate offsets from one address.
	auto stageTile = [&](int tile, int buf) {
		const int ty = tile / p.tilesX;
		const int tx = tile - ty * p.tilesX;
		const unsigned base = static_cast<unsigned>(((ty * 8) * p.pitch + tx * 32) * 64);
		unsigned char *dst = smT + buf * kF8TileBytes;
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int i = wave + 4 * k;          // wave-instruction index: 16 records each
			const int q = i * 16 + (lane >> 2);  // record index inside the 10 x 34 tile
			if (i < 22 && q < 340) {
				const int r = q / 34;
				const int x = q - r * 34;
				const int c = (lane & 3) ^ ((x >> 2) & 3);  // swizzled on the SOURCE side (DMA writes lane-linear)
				dmaToLds16(p.in8, base, static_cast<unsigned>((r * p.pitch + x) * 64 + c * 16), dst + i * 1024);
			}
		}
	};

	int round = 0;
	int tile = tileAt(0);
	if (tile >= 0) stageTile(tile, 0);

	// ---- the weights of this wave's 32 output channels: 9 A fragments of 32 bytes per lane ----
	i32x8 wf[9];
	{
		const i32x4 *wsrc = reinterpret_cast<const i32x4 *>(p.wgt);
#pragma unroll
		for (int t = 0; t < 9; ++t) {
			const i32x4 lo = wsrc[((t * 2 + ch) * 64 + lane) * 2];
			const i32x4 hi = wsrc[((t * 2 + ch) * 64 + lane) * 2 + 1];
			wf[t] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
		}
	}
	const int scA = p.scaleA[ch * 32 + px];
	const int scB = p.scaleB;
	f32x4 biasv[4];
#pragma unroll
	for (int g = 0; g < 4; ++g) {
		biasv[g] = *reinterpret_cast<const f32x4 *>(p.bias + ch * 32 + 8 * g + 4 * hh);
	}

	// the first tile has landed.  hipcc does not count a buffer load to LDS as an LDS write
	// the barrier's fence must wait for (it emitted vmcnt(23) here: rare stale tiles)
	JU_F8_DMA_WAIT();
	__syncthreads();

	int buf = 0;
	for (; tile >= 0; buf ^= 1) {
		const int next = tileAt(++round);
		const int ty = tile / p.tilesX;
		const int tx = tile - ty * p.tilesX;
		const int gy0 = ty * 8 + rp * 4;  // first image row of this wave
		const int gx0 = tx * 32;

		// skip connection: this wave's 4 rows x 32 px half records (channels 32 ch ..), DMA'd
		// into its slice; lands during the K loop.  Slice layout: pixel pi = rw * 32 + px at
		// pi * 64, 16-byte chunk c at position c ^ ((pi >> 2) & 3) (conflict-free b64 access).
		if constexpr (STREAM) {
			// rows beyond H and columns beyond W read the zero border / the next tile:
			// harmless, those pixels are never stored
			const unsigned base = static_cast<unsigned>(((gy0 + 1) * p.pitch + gx0 + 1) * 128 + ch * 64);
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				dmaToLds16(static_cast<const unsigned char *>(p.res),
				    base + static_cast<unsigned>(((i >> 1) * p.pitch + (i & 1) * 16) * 128), halfOff, slice + i * 1024);
			}
		}
		// every wave is done with the other tile buffer (barrier of the previous
		// iteration): refill it
		if (next >= 0) stageTile(next, buf ^ 1);
		f32x16 acc[4];
#pragma unroll
		for (int rw = 0; rw < 4; ++rw) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				acc[rw][4 * g + 0] = biasv[g][0];
				acc[rw][4 * g + 1] = biasv[g][1];
				acc[rw][4 * g + 2] = biasv[g][2];
				acc[rw][4 * g + 3] = biasv[g][3];
			}
		}

		// ---- K loop: per horizontal tap dx, the wave's 6 input rows (6 B fragments of
		// 32 bytes: channels 32h .. 32h+31 of the lane's pixel) feed 12 instructions ----
		const unsigned char *tileBase = smT + buf * kF8TileBytes;
#pragma unroll
		for (int t = 0; t < 3; ++t) {
			const int dx = t == 0 ? 1 : (t == 1 ? 0 : 2);  // tap order of all three 8-bit kernels: 1, 0, 2
			// (keeps hipcc from hoisting all 18 fragments above the first instruction:
			// 144 VGPRs, spills)
			__builtin_amdgcn_sched_barrier(0);
			i32x8 fb[6];
			const int x = px + dx;
			const int sw = (x >> 2) & 3;
			const unsigned char *col = tileBase + ((rp * 4) * 34 + x) * 64;
			const unsigned char *colLo = col + (((2 * hh) ^ sw) << 4);
			const unsigned char *colHi = col + (((2 * hh + 1) ^ sw) << 4);
#pragma unroll
			for (int r = 0; r < 6; ++r) {
				const i32x4 lo = *reinterpret_cast<const i32x4 *>(colLo + r * (34 * 64));
				const i32x4 hi = *reinterpret_cast<const i32x4 *>(colHi + r * (34 * 64));
				fb[r] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
			}
#pragma unroll
			for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
				for (int rw = 0; rw < 4; ++rw) {
					acc[rw] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(
					    wf[dy * 3 + dx], fb[rw + dy], acc[rw], 0, 0, 0, scA, 0, scB);
				}
			}
		}

		// every wave is done with this tile's input; the next tile and the skip records
		// have landed (explicit wait: see the prologue)
		JU_F8_DMA_WAIT();
		__syncthreads();

		// ---- epilogue: bias is in the accumulator; + skip, ReLU ----
		if constexpr (STREAM) {
			// skip values out of the slice, results back into the same places
#pragma unroll
			for (int rw = 0; rw < 4; ++rw) {
				const int pi = rw * 32 + px;
				unsigned char *rec = slice + pi * 64 + hh * 8;
				const int sw = (pi >> 2) & 3;
				Vec4<T> rv[4];
#pragma unroll
				for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const Vec4<T> *>(rec + ((g ^ sw) << 4));
#pragma unroll
				for (int g = 0; g < 4; ++g) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						acc[rw][4 * g + i] = act8<LEAKY>(acc[rw][4 * g + i] + static_cast<float>(rv[g][i]), p.slope);
					}
					*reinterpret_cast<Vec4<T> *>(rec + ((g ^ sw) << 4)) =
					    pack4<T>(acc[rw][4 * g], acc[rw][4 * g + 1], acc[rw][4 * g + 2], acc[rw][4 * g + 3]);
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			const unsigned base = static_cast<unsigned>(((gy0 + 1) * p.pitch + gx0 + 1) * 128 + ch * 64);
#pragma unroll
			for (int i = 0; i < 8; ++i) {
				const i32x4 val = *reinterpret_cast<const i32x4 *>(slice + i * 1024 + lane * 16);
				const int gy = gy0 + (i >> 1);
				const int gx = gx0 + (i & 1) * 16 + (lane >> 2);
				if (gy < p.H && gx < p.W) {
					store16<1>(static_cast<unsigned char *>(p.outT),
					    base + static_cast<unsigned>(((i >> 1) * p.pitch + (i & 1) * 16) * 128), halfOff, val);
				}
			}
		} else {
#pragma unroll
			for (int rw = 0; rw < 4; ++rw) {
#pragma unroll
				for (int i = 0; i < 16; ++i) acc[rw][i] = act8<LEAKY>(acc[rw][i], p.slope);
			}
		}
		// e4m3 copy through the slice: [rw][g][px][hh] dwords (every write instruction covers
		// all 64 banks once); a half record (32 B) is read back as four 8-byte pieces
#pragma unroll
		for (int rw = 0; rw < 4; ++rw) {
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				*reinterpret_cast<int *>(slice + ((rw * 4 + g) * 64 + px * 2 + hh) * 4) =
				    quantize4<LEAKY>(acc[rw][4 * g], acc[rw][4 * g + 1], acc[rw][4 * g + 2], acc[rw][4 * g + 3], p.outMul);
			}
		}
		__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
		__builtin_amdgcn_wave_barrier();
		__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
		for (int i = 0; i < 4; ++i) {  // one tile row per instruction: lane = (pixel, 16-byte half)
			const int pxo = lane >> 1;
			const int c = lane & 1;  // channels 32 ch + 16 c .. + 15  =  g in {2c, 2c+1}
			typedef int i32x2 __attribute__((ext_vector_type(2)));
			const i32x2 a = *reinterpret_cast<const i32x2 *>(slice + ((i * 4 + 2 * c) * 64 + pxo * 2) * 4);
			const i32x2 b = *reinterpret_cast<const i32x2 *>(slice + ((i * 4 + 2 * c + 1) * 64 + pxo * 2) * 4);
			const int gy = gy0 + i;
			const int gx = gx0 + pxo;
			if (gy < p.H && gx < p.W) {
				store16<2>(p.out8, static_cast<unsigned>(((gy + 1) * p.pitch + gx0 + 1) * 64 + ch * 32),
				    static_cast<unsigned>(pxo * 64 + c * 16), i32x4{a[0], a[1], b[0], b[1]});
			}
		}
		// the slice is read out before the next tile's skip DMA refills it: LDS reads of
		// one wave complete in order, but the DMA is a memory operation -- wait for them
		asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		tile = next;
	}
}

// ---------------------------------------------------------------------------
// res_block_fp8_kernel: one residual block of the 8-bit tower per launch
// ---------------------------------------------------------------------------
// The per-layer form above moves every e4m3 tensor through HBM twice and its second
// convolution runs at the memory rate.  Here a block is ONE persistent
// launch (one workgroup of 8 waves per CU, two per SIMD) over TH x 30-pixel tiles:
//   * a convolution's A fragments (9 taps x 8 VGPRs) are in registers while it runs; they
//     are re-fetched per tile and convolution (18 KB per wave from L2) -- holding both sets
//     for the whole launch does not fit two waves per SIMD (256 registers) without spills;
//   * conv A: X8 tile ((TH + 4) x 34 records of 64 B, LDS) -> ReLU -> e4m3 with the
//     intermediate tensor's scale -> T8 tile ((TH + 2) x 34, LDS), zero outside the image;
//     the intermediate tensor never reaches memory;
//   * conv B: T8 -> + skip (16-bit stream, read back from memory at the output pixel) ->
//     ReLU -> the stream in place, and its e4m3 copy with the NEXT block's scale into the
//     other e4m3 tensor (a neighbouring tile still reads this block's input as its halo);
//   * the next tile's X8 travels by LDS-DMA while conv B computes (X8 is dead then).
// The arithmetic per element is exactly the per-layer kernels' (same instruction sequence
// per output), so both paths produce the same bytes.
struct Fp8BlockParams {
	const unsigned char *in8;   // e4m3 tower-layout tensor, allocation start
	unsigned char *out8;        // the other e4m3 tower-layout tensor
	void *stream;               // 16-bit residual stream, tower layout, updated in place
	const unsigned char *w1, *w2;
	const int *scaleA1, *scaleA2;
	const float *b1, *b2;
	int scaleB1, scaleB2;       // E8M0 codes of the input / intermediate tensor scales
	float mulT, mulOut;         // 2^e of the intermediate / output e4m3 tensors
	float slope;                // LEAKY instantiations: LeakyReLU negative slope
	int H, W, pitch;
	int tilesX, numTiles;
	int skip;                   // timing ablation (JU_FB_SKIP, developer only)
	int prio;                   // wave priority scheme (kernel_common.h applyWavePriority)
};

// DUO (round 4): the same block as TWO workgroups of 4 waves per CU instead of one of 8.  The eight waves of one
// workgroup run the same phases in lockstep -- both waves of a SIMD in their K loops, then both in their epilogues
// (MFMA ~4.4 us + VALU ~3.4 us + LDS ~4.9 us per tile add up to the measured 11.9 us: nothing overlaps; a static
// wave priority changes nothing, profiles/r04_wave_priority_ab.txt) -- two independent workgroups are in different
// phases.  That caps a workgroup at 80 KiB of LDS: conv B's fragments cannot live there, so BOTH convolutions'
// A fragments are register operands, re-fetched from L2 per tile and convolution (18 KB per wave) behind the
// previous convolution's last K loop.  Same instruction sequence per output element: same bytes.
template <int TH, bool DUO = false>
struct Fp8BlockGeom {
	static constexpr int NWAVES = DUO ? 4 : 8;
	static constexpr int XR = TH + 4, TR = TH + 2;
	static constexpr int XBYTES = (XR * 34 * 64 + 1023) / 1024 * 1024;  // whole 1 KiB DMA writes
	static constexpr int TBYTES = TR * 34 * 64;
	static constexpr int WBYTES = 9 * 2 * 64 * 32;  // a convolution's fragments (both cout blocks): 36 KiB
	static constexpr int WLDS = DUO ? 0 : WBYTES;   // conv B's set in LDS (one workgroup per CU only)
	// per wave: the skip records of a row pair (2 x 32 px x 64 B; the results overwrite them
	// in place; a row's e4m3 copy is then staged in the same 2 KiB once its stream row has
	// been read out)
	static constexpr int STAGE_WAVE = 4096;
	static constexpr int OFF_T = XBYTES;
	static constexpr int OFF_W = OFF_T + TBYTES;
	static constexpr int OFF_STAGE = OFF_W + WLDS;
	static constexpr int OFF_BIAS = OFF_STAGE + NWAVES * STAGE_WAVE;  // 2 x 64 floats
	static constexpr int LDS = OFF_BIAS + 512;
	static constexpr bool FITS = LDS <= (DUO ? 80 : 160) * 1024;
};

template <typename T, int TH, bool LEAKY = false, bool DUO = false>
__global__ __launch_bounds__(DUO ? 256 : 512, 2) void res_block_fp8_kernel(Fp8BlockParams p) {
	using G = Fp8BlockGeom<TH, DUO>;
	static_assert(G::FITS, "fp8 block tile");
	constexpr int NWV = G::NWAVES, PLS = NWV / 2;  // waves; row-pair lanes (pairs pl, pl + PLS, ...)
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	unsigned char *smX = smem, *smT = smem + G::OFF_T, *smW = smem + G::OFF_W;
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	const int cb = wave & 1, pl = wave >> 1;  // cout block; pairs pl, pl + PLS, ...
	unsigned char *stage = smem + G::OFF_STAGE + wave * G::STAGE_WAVE;
	if constexpr (!DUO) applyWavePriority(p.prio, wave, 8);

	// conv A's fragments of this wave's cout block: registers, for the whole launch (buffer
	// loads: ONE lane offset register, the tap offset scalar -- flat loads keep 18 64-bit
	// addresses alive: spills).  conv B's fragments: LDS, shared by the workgroup (both sets
	// in registers do not fit two waves per SIMD).
	typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
	i32x8 wa[9];
	const unsigned wLane = static_cast<unsigned>((cb * 64 + lane) * 32);
	// (DUO: both convolutions' fragments take turns in these registers)
	auto loadW = [&](const unsigned char *base) __attribute__((always_inline)) {
		const __amdgpu_buffer_rsrc_t rsrc =
		    __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(base), 0, G::WBYTES, 0x00020000);
#pragma unroll
		for (int t = 0; t < 9; ++t) {
			const u32x4w lo = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wLane, t * 4096, 0);
			const u32x4w hi = __builtin_amdgcn_raw_buffer_load_b128(rsrc, wLane, t * 4096 + 16, 0);
			wa[t] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
		}
	};
	loadW(p.w1);
	if constexpr (!DUO) {
		for (int i = wave; i < G::WBYTES / 1024; i += 8) dmaToLds16(p.w2, 0u, static_cast<unsigned>(i * 1024 + lane * 16), smW + i * 1024);
	}
	const int scA1 = p.scaleA1[cb * 32 + px], scA2 = p.scaleA2[cb * 32 + px];
	float *smBias = reinterpret_cast<float *>(smem + G::OFF_BIAS);
	if (tid < 64) smBias[tid] = p.b1[tid];
	else if (tid < 128) smBias[tid] = p.b2[tid - 64];
	const float *biasA = smBias + cb * 32 + 4 * hh, *biasB = smBias + 64 + cb * 32 + 4 * hh;
	const unsigned char *wbLane = smW + (cb * 64 + lane) * 32;  // + tap * 4096

	// X8 tile: record q = r * 34 + k = image pixel (y0 - 2 + r, x0 - 2 + k) at q * 64, chunk c of
	// column k at position c ^ ((k >> 2) & 3).  Image pixels by LDS-DMA (16 records per
	// wave-instruction), the rest zeroed by hand (disjoint locations).
	auto stageX = [&](int tile) {
		const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
		const int y0 = ty * TH, x0 = tx * 30;
		constexpr int NREC = G::XR * 34;
		constexpr int NINSTR = (NREC + 15) / 16;
		const bool border = y0 - 2 < 0 || y0 + TH + 2 > p.H || x0 - 2 < 0 || x0 + 32 > p.W;
		for (int i = wave; i < NINSTR; i += NWV) {
			const int q = i * 16 + (lane >> 2);
			const int r = q / 34, k = q - r * 34;
			const int gy = y0 - 2 + r, gx = x0 - 2 + k;
			const int c = (lane & 3) ^ ((k >> 2) & 3);
			const bool inside = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
			if (q < NREC && inside) {
				dmaToLds16(p.in8, 0u, (static_cast<unsigned>(gy + 1) * static_cast<unsigned>(p.pitch) + static_cast<unsigned>(gx + 1)) * 64u + static_cast<unsigned>(c) * 16u,
				    smX + i * 1024);  // (unsigned arithmetic: tensors of 2-4 GiB wrap by definition, not by signed overflow)
			} else if (border && q < NREC) {
				*reinterpret_cast<i32x4 *>(smX + i * 1024 + lane * 16) = i32x4{0, 0, 0, 0};
			}
		}
	};
	// B fragments of one horizontal tap of a row pair: 4 input rows x 32 bytes per lane
	auto loadFrags = [&](const unsigned char *tile, int pair, int dx, i32x8(&fb)[4]) {
		const int x = px + dx;
		const int sw = (x >> 2) & 3;
		const unsigned char *col = tile + ((2 * pair) * 34 + x) * 64;
		const unsigned char *colLo = col + (((2 * hh) ^ sw) << 4);
		const unsigned char *colHi = col + (((2 * hh + 1) ^ sw) << 4);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const i32x4 lo = *reinterpret_cast<const i32x4 *>(colLo + r * (34 * 64));
			const i32x4 hi = *reinterpret_cast<const i32x4 *>(colHi + r * (34 * 64));
			fb[r] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
		}
	};

	// DUO: a row pair's 18 matrix instructions with the fragment reads pinned one tap ahead (left to itself
	// hipcc sinks every read next to its first use and waits lgkmcnt(0) in front of each instruction pair: nine
	// exposed LDS round trips per pair).  Tap order 1, 0, 2 and dy inside: the other forms' order, same bytes.
	auto kLoop = [&](const unsigned char *tileBase, const int pair, f32x16(&acc)[2], const int scA, const int scB)
	                 __attribute__((always_inline)) {
		i32x8 fa[4], fb[4];
		loadFrags(tileBase, pair, 1, fa);
		loadFrags(tileBase, pair, 0, fb);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
			for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 1], fa[r + dy], acc[r], 0, 0, 0, scA, 0, scB);
		}
		__builtin_amdgcn_sched_barrier(0);
		loadFrags(tileBase, pair, 2, fa);
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
			for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 0], fb[r + dy], acc[r], 0, 0, 0, scA, 0, scB);
		}
		__builtin_amdgcn_sched_barrier(0);
#pragma unroll
		for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
			for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 2], fa[r + dy], acc[r], 0, 0, 0, scA, 0, scB);
		}
		__builtin_amdgcn_sched_barrier(0);
	};

	int tile = blockIdx.x;
	if (tile < p.numTiles && !(JU_SKIP(p) & 1)) stageX(tile);
	for (; tile < p.numTiles; tile += gridDim.x) {
		const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
		const int y0 = ty * TH, x0 = tx * 30;
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this tile's X8 (first tile: and conv B's weights) landed
		__syncthreads();                                  // ... for every wave; all are done with T8
		// ---- conv A: (TH + 2) rows x 32 columns -> ReLU -> e4m3 -> T8, zero outside the image ----
		for (int pair = pl; pair < G::TR / 2; pair += PLS) {
			f32x16 acc[2];
#pragma unroll
			for (int g = 0; g < 4; ++g) {
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = biasA[8 * g + i];
				}
			}
			// the next tap's fragments travel behind the current tap's 6 instructions
			i32x8 f0[4], f1[4];
			if (JU_SKIP(p) & 2) goto epiA;
			if constexpr (DUO) {
				kLoop(smX, pair, acc, scA1, p.scaleB1);
				goto epiA;
			}
			loadFrags(smX, pair, 0, f0);
			loadFrags(smX, pair, 1, f1);
			// (tap order 1, 0, 2: see tower8_kernels.hip)
#pragma unroll
			for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
				for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 1], f1[r + dy], acc[r], 0, 0, 0, scA1, 0, p.scaleB1);
			}
			loadFrags(smX, pair, 2, f1);
#pragma unroll
			for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
				for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 0], f0[r + dy], acc[r], 0, 0, 0, scA1, 0, p.scaleB1);
			}
#pragma unroll
			for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
				for (int r = 0; r < 2; ++r) acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wa[dy * 3 + 2], f1[r + dy], acc[r], 0, 0, 0, scA1, 0, p.scaleB1);
			}
		epiA:
			if (JU_SKIP(p) & 32) continue;
			const int gx = x0 - 1 + px;
			const bool colIn = gx >= 0 && gx < p.W;
			const int sw = (px >> 2) & 3;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int tr = 2 * pair + r;
				const int gy = y0 - 1 + tr;
				const bool inside = colIn && gy >= 0 && gy < p.H;
				unsigned char *rec = smT + (tr * 34 + px) * 64;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					int q = quantize4<LEAKY>(act8<LEAKY>(acc[r][4 * g], p.slope), act8<LEAKY>(acc[r][4 * g + 1], p.slope),
					    act8<LEAKY>(acc[r][4 * g + 2], p.slope), act8<LEAKY>(acc[r][4 * g + 3], p.slope), p.mulT);
					if (!inside) q = 0;
					// bytes 32 cb + 8 g + 4 hh .. + 3 of the record: chunk 2 cb + (g >> 1)
					*reinterpret_cast<int *>(rec + (((2 * cb + (g >> 1)) ^ sw) << 4) + (g & 1) * 8 + hh * 4) = q;
				}
			}
		}
		if constexpr (DUO) loadW(p.w2);  // conv A's fragments are dead: conv B's travel across the barrier
		__syncthreads();  // T8 complete, X8 dead
		if (tile + static_cast<int>(gridDim.x) < p.numTiles && !(JU_SKIP(p) & 1)) stageX(tile + gridDim.x);
		// ---- conv B: TH rows x 32 columns (30 valid) + skip -> ReLU -> stream, e4m3 copy ----
		for (int pair = pl; pair < TH / 2; pair += PLS) {
			// the pair's skip records (this wave's half: channels 32 cb ..) by LDS-DMA into the
			// staging slice, pixel pi = r * 32 + px at pi * 64, chunk c at c ^ ((pi >> 2) & 3) -- keyed on pi >> 2: the epilogue's
			// 8-byte accesses of 32 pixels (64-byte stride: pi and pi + 4 share a bank group) then conflict 2 ways, not 8
			// (PMC, round 4: 45 % of this kernel's LDS cycles were bank conflicts with the key pi & 3); they land
			// during the K loop, and the results go back into the same places
			if (!(JU_SKIP(p) & 8)) {
				const unsigned char *src = static_cast<const unsigned char *>(p.stream);
#pragma unroll
				for (int i = 0; i < 4; ++i) {
					const int pi = i * 16 + (lane >> 2);
					const int gy = min(y0 + 2 * pair + (pi >> 5), p.H - 1), gx = min(x0 + (pi & 31), p.W - 1);
					const unsigned c = static_cast<unsigned>(lane & 3) ^ ((static_cast<unsigned>(pi) >> 2) & 3u);
					dmaToLds16(src, 0u, ((static_cast<unsigned>(gy + 1) * static_cast<unsigned>(p.pitch) + static_cast<unsigned>(gx + 1)) * 64u + static_cast<unsigned>(cb) * 32u) * 2u + c * 16u,
					    stage + i * 1024);
				}
			}
			f32x16 acc[2];
#pragma unroll
			for (int g = 0; g < 4; ++g) {
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = biasB[8 * g + i];
				}
			}
			i32x8 f0[4], f1[4];
			if constexpr (DUO) {
				if (!(JU_SKIP(p) & 4)) kLoop(smT, pair, acc, scA2, p.scaleB2);
			}
			if (!DUO && !(JU_SKIP(p) & 4)) {
			loadFrags(smT, pair, 0, f0);
			loadFrags(smT, pair, 1, f1);
			}
#pragma unroll
			for (int t = 0; t < ((DUO || (JU_SKIP(p) & 4)) ? 0 : 3); ++t) {
				const int dx = t == 0 ? 1 : (t == 1 ? 0 : 2);
				if (t == 1) loadFrags(smT, pair, 2, f1);
#pragma unroll
				for (int dy = 0; dy < 3; ++dy) {
					i32x8 w;
					if constexpr (DUO) {
						w = wa[dy * 3 + dx];
					} else {
						const i32x4 lo = *reinterpret_cast<const i32x4 *>(wbLane + (dy * 3 + dx) * 4096);
						const i32x4 hi = *reinterpret_cast<const i32x4 *>(wbLane + (dy * 3 + dx) * 4096 + 16);
						w = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
					}
#pragma unroll
					for (int r = 0; r < 2; ++r) {
						acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w, (t == 1 ? f0 : f1)[r + dy], acc[r], 0, 0, 0, scA2, 0, p.scaleB2);
					}
				}
			}
			// (timing ablation bit 128: WITHOUT this wait -- vmcnt counts stores too, so it also waits for the
			// previous pair's output stores to be acknowledged and for the next tile's X8 DMA; wrong values)
			if (!(JU_SKIP(p) & 128)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the skip records have landed (this wave's own DMA)
			if (JU_SKIP(p) & 64) continue;
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int pi = r * 32 + px;
				unsigned char *rec = stage + pi * 64 + hh * 8;
				const unsigned sw = (static_cast<unsigned>(pi) >> 2) & 3u;
				Vec4<T> rv[4];
#pragma unroll
				for (int g = 0; g < 4; ++g) rv[g] = *reinterpret_cast<const Vec4<T> *>(rec + ((static_cast<unsigned>(g) ^ sw) << 4));
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					float v[4];
#pragma unroll
					for (int i = 0; i < 4; ++i) v[i] = act8<LEAKY>(acc[r][4 * g + i] + static_cast<float>(rv[g][i]), p.slope);
					*reinterpret_cast<Vec4<T> *>(rec + ((static_cast<unsigned>(g) ^ sw) << 4)) = pack4<T>(v[0], v[1], v[2], v[3]);
					acc[r][4 * g] = v[0];
					acc[r][4 * g + 1] = v[1];
					acc[r][4 * g + 2] = v[2];
					acc[r][4 * g + 3] = v[3];
				}
			}
			__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
			__builtin_amdgcn_wave_barrier();
			__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
			// stream rows: 2 x 32 px x 64 B = 4 wave-instructions of 16 B per lane
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				const int pi = i * 16 + (lane >> 2);
				const unsigned slot = static_cast<unsigned>(lane & 3);
				const unsigned chunk = slot ^ ((static_cast<unsigned>(pi) >> 2) & 3u);
				const i32x4 val = *reinterpret_cast<const i32x4 *>(stage + pi * 64 + (slot << 4));
				const int gy = y0 + 2 * pair + (pi >> 5), gx = x0 + (pi & 31);
				if ((pi & 31) < 30 && gy < p.H && gx < p.W && !(JU_SKIP(p) & 16)) {
					*reinterpret_cast<i32x4 *>(static_cast<unsigned char *>(p.stream) +
					    (((size_t)(gy + 1) * p.pitch + gx + 1) * 64 + cb * 32) * 2 + chunk * 16) = val;
				}
			}
			// e4m3 copy, one row at a time through [g][px][hh] dwords (every write instruction
			// covers all 64 banks once) in the row's own, now free, 2 KiB; lane = (pixel, 16-byte
			// half of the 32-byte half record)
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				unsigned char *stage8 = stage + r * 2048;
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					*reinterpret_cast<int *>(stage8 + (g * 64 + px * 2 + hh) * 4) =
					    quantize4<LEAKY>(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3], p.mulOut);
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
				__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
				const int pxo = lane >> 1, c = lane & 1;
				typedef int i32x2 __attribute__((ext_vector_type(2)));
				const i32x2 a = *reinterpret_cast<const i32x2 *>(stage8 + ((2 * c) * 64 + pxo * 2) * 4);
				const i32x2 b = *reinterpret_cast<const i32x2 *>(stage8 + ((2 * c + 1) * 64 + pxo * 2) * 4);
				const int gy = y0 + 2 * pair + r, gx = x0 + pxo;
				if (pxo < 30 && gy < p.H && gx < p.W && !(JU_SKIP(p) & 16)) {
					*reinterpret_cast<i32x4 *>(p.out8 + ((size_t)(gy + 1) * p.pitch + gx + 1) * 64 + cb * 32 + c * 16) =
					    i32x4{a[0], a[1], b[0], b[1]};
				}
				__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
				__builtin_amdgcn_wave_barrier();
			}
			// the slice is read out before the next pair's skip DMA refills it: LDS reads of one
			// wave complete in order, but the DMA is a memory operation -- wait for them
			asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
		}
		if constexpr (DUO) {
			if (tile + static_cast<int>(gridDim.x) < p.numTiles) loadW(p.w1);  // the next tile's conv A
		}
	}
}

constexpr bool kFp8BlockDuoDefault = true;  // measured, 640x448: 35.7-36.4 us per block against 38.0-38.5 (profiles/r04_fp8_duo.txt)

template <typename T, int TH, bool LEAKY, bool DUO = false>
void launchFp8BlockT(Fp8BlockParams k, int cus, hipStream_t stream) {
	using G = Fp8BlockGeom<TH, DUO>;
	auto kern = res_block_fp8_kernel<T, TH, LEAKY, DUO>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), G::LDS, &ldsDone, "fp8 block");
	k.tilesX = (k.W + 29) / 30;
	k.numTiles = k.tilesX * ((k.H + TH - 1) / TH);
	const int slots = DUO ? 2 * cus : cus;  // DUO: two workgroups per CU
	const int grid = k.numTiles < slots ? k.numTiles : slots;
	hipLaunchKernelGGL(kern, dim3(grid), dim3(DUO ? 256 : 512), G::LDS, stream, k);
	hipCheckLaunch("res_block_fp8");
}

// 16-bit tower tensor -> e4m3 copy (interior pixels only: the border stays zero)
template <typename T, bool LEAKY>
__global__ __launch_bounds__(256) void quantize_tower_kernel(const T *__restrict__ in,
    unsigned char *__restrict__ out, int H, int W, int pitch, float mul) {
	const int i = blockIdx.x * 256 + threadIdx.x;  // one 16-channel chunk per thread
	const int chunk = i & 3;
	const int pix = i >> 2;
	if (pix >= H * W) return;
	const int y = pix / W, x = pix - y * W;
	const size_t rec = (size_t)(y + 1) * pitch + x + 1;
	const Vec8<T> a = *reinterpret_cast<const Vec8<T> *>(in + rec * 64 + chunk * 16);
	const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(in + rec * 64 + chunk * 16 + 8);
	float v[16];
#pragma unroll
	for (int k = 0; k < 8; ++k) {
		// (the tensor is a layer's OUTPUT: the activation has been applied; ReLU models clamp
		// at 0 all the same -- a no-op that keeps -0.0 out of the e4m3 copy)
		v[k] = LEAKY ? static_cast<float>(a[k]) : fmaxf(static_cast<float>(a[k]), 0.0f);
		v[8 + k] = LEAKY ? static_cast<float>(b[k]) : fmaxf(static_cast<float>(b[k]), 0.0f);
	}
	i32x4 o;
#pragma unroll
	for (int k = 0; k < 4; ++k) o[k] = quantize4<LEAKY>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3], mul);
	*reinterpret_cast<i32x4 *>(out + rec * 64 + chunk * 16) = o;
}

template <typename T, bool STREAM, bool LEAKY>
void launchFp8T(const Fp8KernelParams &k, int grid, hipStream_t stream) {
	auto kern = conv_tower_fp8_kernel<T, STREAM, LEAKY>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kF8Lds, &ldsDone, "fp8 tower");
	hipLaunchKernelGGL(kern, dim3(grid), dim3(kF8Threads), kF8Lds, stream, k);
	hipCheckLaunch("conv_tower_fp8");
}

}  // namespace

void launchConvTowerFp8(DType dt, const Fp8TowerParams &q, hipStream_t stream) {
	Fp8KernelParams k{};
	k.in8 = static_cast<const unsigned char *>(q.in8);
	k.wgt = static_cast<const unsigned char *>(q.weights);
	k.scaleA = q.scaleA;
	k.bias = q.bias;
	k.res = q.stream;
	k.outT = q.stream;  // in place: a lane's skip values are read before its wave writes the records
	k.out8 = static_cast<unsigned char *>(q.out8);
	k.scaleB = 127 - q.inExp;
	k.outMul = std::ldexp(1.0f, q.outExp);
	k.slope = q.slope;
	k.H = q.H;
	k.W = q.W;
	k.pitch = towerPitch(q.W);
	k.tilesX = (q.W + 31) / 32;
	k.numTiles = k.tilesX * ((q.H + 7) / 8);
	// CU count of the current device, looked up once per device (49 launches per frame)
	static std::atomic<int> cuCache[64];
	int dev = 0, cus = 256;
	if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
		cus = cuCache[dev].load(std::memory_order_relaxed);
		if (cus == 0) {
			cus = 256;
			(void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
			cuCache[dev].store(cus, std::memory_order_relaxed);
		}
	}
	int grid = k.numTiles < 2 * cus ? k.numTiles : 2 * cus;  // two workgroups per CU
	// the kernel's XCD tile order needs a multiple of 8: round up when everything fits one
	// round (surplus workgroups exit), down otherwise
	if (grid > 8) grid = k.numTiles <= 2 * cus ? (grid + 7) / 8 * 8 : grid - grid % 8;
	// JU_FP8_GRID=n (tests): any grid computes the same bytes, a race would not
	if (const char *g = devSwitch(Dev::Fp8Grid)) grid = std::atoi(g) > 0 ? std::atoi(g) : grid;
	if (q.leaky) {  // `activation: lrelu` models: own instantiations, the ReLU kernels are unchanged
		if (q.stream != nullptr) {
			if (dt == kF16) launchFp8T<f16, true, true>(k, grid, stream);
			else launchFp8T<bf16, true, true>(k, grid, stream);
		} else {
			if (dt == kF16) launchFp8T<f16, false, true>(k, grid, stream);
			else launchFp8T<bf16, false, true>(k, grid, stream);
		}
		return;
	}
	if (q.stream != nullptr) {
		if (dt == kF16) launchFp8T<f16, true, false>(k, grid, stream);
		else launchFp8T<bf16, true, false>(k, grid, stream);
	} else {
		if (dt == kF16) launchFp8T<f16, false, false>(k, grid, stream);
		else launchFp8T<bf16, false, false>(k, grid, stream);
	}
}

// 0: by geometry; 1: one 8-wave workgroup per CU; 2: two 4-wave workgroups per CU (JU_FP8_BLOCK=solo / duo, tests)
static std::atomic<int> g_Fp8BlockForm{[] {
	const char *e = devSwitch(Dev::Fp8Block);
	return e == nullptr ? 0 : (std::string(e) == "duo" ? 2 : (std::string(e) == "solo" ? 1 : 0));
}()};
void setFp8BlockForm(int form) { g_Fp8BlockForm = form; }

void launchResBlockFp8(DType dt, const Fp8BlockLaunch &q, hipStream_t stream) {
	Fp8BlockParams k{};
	k.in8 = static_cast<const unsigned char *>(q.in8);
	k.out8 = static_cast<unsigned char *>(q.out8);
	k.stream = q.stream;
	k.w1 = static_cast<const unsigned char *>(q.w1);
	k.w2 = static_cast<const unsigned char *>(q.w2);
	k.scaleA1 = q.scaleA1;
	k.scaleA2 = q.scaleA2;
	k.b1 = q.b1;
	k.b2 = q.b2;
	k.scaleB1 = 127 - q.inExp;
	k.scaleB2 = 127 - q.midExp;
	k.mulT = std::ldexp(1.0f, q.midExp);
	k.mulOut = std::ldexp(1.0f, q.outExp);
	k.slope = q.slope;
	k.H = q.H;
	k.W = q.W;
	k.pitch = towerPitch(q.W);
	k.skip = ablationSkipBits();
#ifdef JU_ABLATE
	{
		// JU_FB_SKIP_ALT (probe builds only): every second launch takes these bits instead of JU_FB_SKIP -- the
		// bytes-only ablation of "two residual blocks per launch" (tools/fp8_pair_ceiling.sh: the first block of
		// a pair stores nothing, the second stages and fetches nothing: what the pair's halved stream traffic could
		// buy at best, with none of the fusion's halo recompute)
		static const int alt = [] { const char *e = std::getenv("JU_FB_SKIP_ALT"); return e ? std::atoi(e) : -1; }();
		static std::atomic<unsigned> launches{0};
		if (alt >= 0 && (launches.fetch_add(1) & 1u)) k.skip = alt;
	}
#endif
	k.prio = wavePriorityMode(0);
	const int cus = currentDeviceCUs();
	// Tile height: a CU works through ceil(tiles / CUs) tiles one after the other, each
	// costing about (rows + 5) row-times (recompute ring + per-tile fixed work): take the
	// height with the shortest makespan (480x270: 18 rows, 240 tiles, one each; 640x448: 14
	// rows, 704 tiles, three rounds -- 18 rows would be three rounds of taller tiles).
	const long tilesX = (q.W + 29) / 30;
	// Two workgroups of 4 waves per CU (DUO, above) where the frame has more tiles than one round of the
	// one-workgroup form can take (640x448: 990 tiles of 10 rows on 512 slots); JU_FP8_BLOCK=solo / duo forces a form.
	const int form = g_Fp8BlockForm.load();
	const bool duo = form == 2 || (form == 0 && kFp8BlockDuoDefault && tilesX * ((q.H + 17) / 18) > cus);
	if (duo) {
		int bestD = 6;
		long bestCostD = -1;
		for (int th : {10, 8, 6}) {
			const long tiles = tilesX * ((q.H + th - 1) / th);
			const long cost = ((tiles + 2 * cus - 1) / (2 * cus)) * (th + 5);
			if (bestCostD < 0 || cost < bestCostD) {
				bestCostD = cost;
				bestD = th;
			}
		}
#define JU_F8D(TH_)                                                             \
	if (bestD == TH_) {                                                         \
		if (q.leaky) {                                                          \
			if (dt == kF16) launchFp8BlockT<f16, TH_, true, true>(k, cus, stream);  \
			else launchFp8BlockT<bf16, TH_, true, true>(k, cus, stream);        \
		} else {                                                                \
			if (dt == kF16) launchFp8BlockT<f16, TH_, false, true>(k, cus, stream); \
			else launchFp8BlockT<bf16, TH_, false, true>(k, cus, stream);       \
		}                                                                       \
		return;                                                                 \
	}
		JU_F8D(10)
		JU_F8D(8)
		JU_F8D(6)
#undef JU_F8D
	}
	int best = 6;
	long bestCost = -1;
	for (int th : {18, 14, 10, 6}) {
		const long tiles = tilesX * ((q.H + th - 1) / th);
		const long cost = ((tiles + cus - 1) / cus) * (th + 5);
		if (bestCost < 0 || cost < bestCost) {
			bestCost = cost;
			best = th;
		}
	}
#define JU_F8B(TH_)                                                       \
	if (best == TH_) {                                                    \
		if (q.leaky) {                                                    \
			if (dt == kF16) launchFp8BlockT<f16, TH_, true>(k, cus, stream);  \
			else launchFp8BlockT<bf16, TH_, true>(k, cus, stream);        \
		} else {                                                          \
			if (dt == kF16) launchFp8BlockT<f16, TH_, false>(k, cus, stream); \
			else launchFp8BlockT<bf16, TH_, false>(k, cus, stream);       \
		}                                                                 \
		return;                                                           \
	}
	JU_F8B(18)
	JU_F8B(14)
	JU_F8B(10)
	JU_F8B(6)
#undef JU_F8B
}

void launchQuantizeTower(DType dt, const void *in, void *out8, int H, int W, int exponent, bool leaky,
    hipStream_t stream) {
	const int n = H * W * 4;
	const float mul = std::ldexp(1.0f, exponent);
	const int pitch = towerPitch(W);
	const dim3 grid((n + 255) / 256), block(256);
	auto *dst = static_cast<unsigned char *>(out8);
	if (dt == kF16) {
		const auto *src = static_cast<const f16 *>(in);
		if (leaky) hipLaunchKernelGGL((quantize_tower_kernel<f16, true>), grid, block, 0, stream, src, dst, H, W, pitch, mul);
		else hipLaunchKernelGGL((quantize_tower_kernel<f16, false>), grid, block, 0, stream, src, dst, H, W, pitch, mul);
	} else {
		const auto *src = static_cast<const bf16 *>(in);
		if (leaky) hipLaunchKernelGGL((quantize_tower_kernel<bf16, true>), grid, block, 0, stream, src, dst, H, W, pitch, mul);
		else hipLaunchKernelGGL((quantize_tower_kernel<bf16, false>), grid, block, 0, stream, src, dst, H, W, pitch, mul);
	}
	hipCheckLaunch("quantize_tower");
}

}  // namespace ju
