// gfx950 (CDNA4, MI355X): the generator's residual tower (reference
// scripts/training/models.py:193-254, 531-550).
//
//  * tower_resident_kernel  conv_1 + all 48 3x3 64->64 convolutions in ONE launch,
//                           activations resident in LDS, halo exchange through a
//                           global mailbox of self-validating slots (the product path)
//  * conv_tower_kernel      one 64->64 layer per launch, persistent, LDS-DMA staged
//                           (fallback: more regions than CUs, JU_TOWER=layers)
#include "kernel_common.h"

namespace ju {

namespace {

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
	int relu;     // 0 none, 1 ReLU, 2 LeakyReLU(slope)
	float slope;
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
		// every wave is done reading this tile's input (halo rows are shared), and the next
		// tile's DMA has landed.  Explicit wait: hipcc does not reliably count an LDS-DMA as
		// something the barrier's fence must wait for (fp8_kernels.hip got vmcnt(23)).
		asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
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
						if (p.relu == 1) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], 0.0f);
						} else if (p.relu == 2) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] = leaky(v[i], p.slope);
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
	t.slope = p.slope;
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
// The MFMA shape of the resident tower's K loops.  1 (round 5): v_mfma_f32_16x16x32 -- per (dx, 32-channel k-step,
// pixel half) macro-step the same four row fragments (16 pixels x 32 channels each: the same LDS bytes per FLOP) feed
// twelve 16-cycle MFMAs on quarters (pixel half, cout quarter) of the row accumulators.  The shape draws less power per
// FLOP (a dense loop of this kernel's form holds 2.0 GHz instead of 1.74 on random data, tools/probes/mfma_shape.hip),
// and the tower runs against the socket's power limit: +0.9 % frames/s at equal cycles (profiles/r05_m16b_tower_ab.txt).
// 0: v_mfma_f32_32x32x16 (rounds 1-4; A/B builds).  The weights are packed for the shape (packTowerWeights, model.cpp).
#ifndef JU_TOWER_M16
#define JU_TOWER_M16 1
#endif
// A unit row's 16 accumulator values per lane.  16x16x32 form: FOUR independent 4-register tuples (pixel half, cout
// quarter), each the C / D operand of its own MFMAs -- as quarters of one 16-register vector hipcc renamed every MFMA's
// destination and reused the freed quarters for the epilogue's temporaries, a hazard s_nop in front of each.
#if JU_TOWER_M16
struct TowerAcc {
	f32x4 q[4];
};
__device__ __forceinline__ float accElem(const TowerAcc &a, const int i) { return a.q[i >> 2][i & 3]; }
__device__ __forceinline__ void setAccElem(TowerAcc &a, const int i, const float v) { a.q[i >> 2][i & 3] = v; }
#else
typedef f32x16 TowerAcc;
__device__ __forceinline__ float accElem(const TowerAcc &a, const int i) { return a[i]; }
__device__ __forceinline__ void setAccElem(TowerAcc &a, const int i, const float v) { a[i] = v; }
#endif
constexpr int kResRW = 32;                                          // region width = one MFMA block
constexpr int kResMaxRH = 16;                                       // 8 row pairs, 4 per wave group
constexpr int kResPitch = kResRW + 2;                               // LDS row: 32 px + halo column each side
constexpr int kResRowBytes = kResPitch * 128;                       // 4352
constexpr int kResBufBytes = (kResMaxRH + 2) * kResRowBytes;       // 78336
// 16 spare bytes BEHIND each buffer, at the same offset from its base (kResDummyOff): where the branch-free sweep of
// the fast schedule sends the lanes that have no halo cell to fill, whichever buffer the layer writes
constexpr int kResDummyOff = kResBufBytes;
constexpr int kResOffA = 0;
constexpr int kResOffB = kResBufBytes + 16;
constexpr int kResOffMisc = kResOffB + kResBufBytes + 16;
static_assert(kResRowBytes == 4352, "the ds_read immediates in tower_resident_kernel assume a 4352-byte row");
constexpr int kResLds = kResOffMisc + 64 + 512;                     // flag, 2 bias slots
constexpr int kResMailSlots = 4 * 32 * 8;                           // 16-byte slots per region per parity
constexpr unsigned long long kResTimeoutTicks = 20000000ull;        // 0.2 s of s_memrealtime
// Pre-run (see the kernel): units per wave whose halo-independent steps run inside the halo exchange,
// and how many of them go in front of the halo loads.  -D overrides are for A/B builds only
// (measured: 2 / 1 best; 1 / 0 +6 %, 2 / 2 +2 %, 3 / 1 +5 %).
#ifndef JU_PRERUN
#define JU_PRERUN 2
#endif
#ifndef JU_PREBEFORE
#define JU_PREBEFORE 1
#endif
// The fast schedule (FAST instantiations): pre-run units per wave; a wave keeps at least one whole unit.
// (Measured: 2 -> 348-352 us per tower; 3 -> 350-358 us once the fragment addresses were kept out of the
// registers (before: 394 us, hoisted LDS addresses spilled to scratch): what the third unit takes out of
// the finish phase the halo fill gives back, the exchange's ~2.1 k cycles beside the pre-run stay.
// JU_PREBEFORE 0 / 1 / 2 with two units: 366 / 352 / 354 us, 0.80 / 0.33 / 0.18 extra sweep passes per layer.)
#ifndef JU_FAST_PRERUN
#define JU_FAST_PRERUN 2
#endif

// The fast schedule's region shape (tower_resident_kernel, FAST): with two pre-run pairs per wave both
// row-pair parities are left with one or two whole pairs, there is no odd row, and the left and the right
// edge column are different lanes.  16- and 14-row regions qualify (480x270: 15 x 17 regions of 32 x 16,
// the last row 14 high), 12-row ones too.
__host__ __device__ inline bool residentFastShape(int rhv, int rwv) {
	if ((rhv & 1) || rwv < 2) return false;
#if JU_TOWER_M16
	// (16x16x32 form: full-width regions only -- the fast schedule then writes its groups without a lane mask; a frame
	// with a ragged last column of regions runs the general schedule)
	if (rwv != 32) return false;
#endif
	const int np2 = rhv >> 1;
	for (int par = 0; par < 2; ++par) {
		const int first = par ? 1 : 2;
		int n = 0;
		for (int k = 0; k < 2; ++k) {
			const int u = first + 2 * k;
			if (u < np2 && 2 * u + 3 <= rhv) n = k + 1;
		}
		const int mine = (np2 - par + 1) >> 1;  // pairs of this parity
		if (n != 2 || (mine != 3 && mine != 4)) return false;
	}
	return true;
}

struct ResidentParams {
	const void *in;           // first layer's input, addressed at image pixel (0,0)
	int inPitch;              // its row pitch in pixels (dense W, or towerPitch(W))
	int hasHead;              // 1: layer 0 is the generator's conv_1 (no residual), blocks follow
	void *out;                // tower-layout tensor, allocation start (last layer's output)
	const void *weights;      // nLayers x 73728 B, kernel-ready (packConvWeights)
	const float *bias;        // nLayers x 64
	uint4 *mail;              // [regions][2][kResMailSlots] 16-byte slots
	unsigned *count;          // [regions][2] publishes so far into the region's two slot parities (persistent)
	unsigned *error;          // host-visible word, 0 = ok
	unsigned long long *debug;  // VARIANT 4 only: [regions][4 waves][8] cycle sums
	int H, W, pitch;
	int GX, GY, RH;
	int nLayers;
	float slope;              // LEAKY instantiations: the LeakyReLU negative slope, in [0, 1]
	// fused generator tail (tailW1 != nullptr), see ResidentTowerParams
	const void *tailW1;
	const float *tailB1;
	const void *tailW2;
	const float *tailB2;
	const std::uint8_t *frame;
	std::ptrdiff_t frameStride;
	void *state;
	std::uint8_t *outU8;
	std::ptrdiff_t outStride;
	const unsigned *sums;
};

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

// VARIANT: timing ablation only (0 = product; bit 0 = no halo exchange, bit 1 = no MFMA loop)
// HEAD: layer 0 is the generator's conv_1 (plain conv + ReLU), residual blocks follow
// TAIL: the generator tail runs on the LDS-resident last layer (JU_TAIL=tower); a
// separate instantiation so that the default kernel carries none of its code
// LEAKY: `activation: lrelu` models (models.py:24-27, 36-60): the epilogue applies
// x < 0 ? slope * x : x in f32, and -- the outputs having no free sign bit -- the halo slots
// carry their epoch beside the values instead of inside them (see publish / fillHalo).  A
// separate instantiation: the ReLU kernel is byte for byte what it was.
// FAST: every region of the frame has the "deferred" shape (residentTowerFastGeometry: two pre-run
// pairs and one or two whole pairs per wave, no odd row, at least two columns) -- the instantiation
// holds ONLY that schedule: epilogues behind the next unit's MFMAs, edges published from the
// epilogues.  The general schedule below it serves every other geometry (and LeakyReLU models,
// calibration, the ablation variants) and is what it was.
template <typename T, int VARIANT, bool HEAD, bool TAIL = false, bool LEAKY = false, bool FAST = false>
__global__ __launch_bounds__(256, 1) void tower_resident_kernel(ResidentParams p) {
	static_assert(!FAST || ((VARIANT == 0 || VARIANT == 4) && (JU_FAST_PRERUN == 2 || JU_FAST_PRERUN == 3)), "the fast schedule is built for the product only");
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	// (variants 1 .. 3 are bit masks; 4, 5, 8 are NOT -- round 2 tested `VARIANT & 1` and ran the
	// calibration build, variant 5, without the halo exchange: its maxima drifted by up to 10 %)
	constexpr bool kNoMfma = VARIANT == 2 || VARIANT == 3;
#ifdef JU_NOXCHG_DEV  // developer timing build only (tools/dev_tower_lib.sh x -DJU_NOXCHG_DEV): no halo exchange at all, wrong frames
	constexpr bool xchg = false;
#else
	constexpr bool xchg = !(VARIANT == 1 || VARIANT == 3);
#endif
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
	// Publishes into this region's two slot parities so far, over ALL launches (zeroed with
	// the mailbox).  Every region publishes the same layers, so a consumer knows its
	// neighbours' counts from its own.
	unsigned pubCount[2] = {p.count[region * 2], p.count[region * 2 + 1]};

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
#if JU_TOWER_M16
	// fragment f = (tap, kf = ks32 * 2 + c16): 16 output channels ch * 32 + c16 * 16 + (lane & 15) x 32 input channels
	// ks32 * 32 + 8 (lane >> 4) .. + 7; packed [tap][ks32][c16][ch][lane][8] (packTowerWeights)
	const unsigned wLaneOff = (unsigned)((ch * 64 + lane) * 16);
#else
	const unsigned wLaneOff = (unsigned)((hh * 64 + ch * 32 + px) * 16);
#endif
	auto loadWeightFrag = [&](int layer, int f) __attribute__((always_inline)) -> Vec8<T> {
		const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(wRsrc, wLaneOff, layer * 73728 + f * 2048, 0);
		return __builtin_bit_cast(Vec8<T>, v);
	};
	// Fragment f = (dy*3+dx)*4+ks, ONE set of 36 (144 registers).  A layer's registers are
	// free once its last unit is through the K loop; the next layer's set streams in right
	// after the publish, the 12 fragments with dx = 1 first: the pre-run (below) uses them,
	// and by the time the other 24 and the halo loads have been issued they have landed.
	Vec8<T> wm[12], ws[24];
	auto loadMidWeights = [&](int layer) __attribute__((always_inline)) {
#pragma unroll
		for (int f = 0; f < 36; ++f) {
			if ((f >> 2) % 3 == 1) wm[(f / 12) * 4 + (f & 3)] = loadWeightFrag(layer, f);
		}
	};
	auto loadSideWeights = [&](int layer) __attribute__((always_inline)) {
#pragma unroll
		for (int f = 0; f < 36; ++f) {
			const int dx = (f >> 2) % 3;
			if (dx != 1) ws[(f / 12) * 8 + (dx >> 1) * 4 + (f & 3)] = loadWeightFrag(layer, f);
		}
	};
	loadMidWeights(0);
	loadSideWeights(0);
	// The layer's bias, as this lane's 16 accumulator values of a row (channel ch*32 + 8g + 4hh
	// + i at index 4g + i): the C operand of every unit's first MFMAs, so no accumulator is
	// ever initialised by moves.  Fetched at the head of the weight stream.
	const __amdgpu_buffer_rsrc_t biasRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    const_cast<float *>(p.bias), 0, p.nLayers * 256, 0x00020000);
	TowerAcc biasVec;
	auto loadBias = [&](int layer) __attribute__((always_inline)) {
#if JU_TOWER_M16
		// quarter g = (pixel half, cout quarter c16 = g & 1) of a row accumulator: channels ch * 32 + c16 * 16 + 4 (lane >> 4) + i
#pragma unroll
		for (int c16 = 0; c16 < 2; ++c16) {
			const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(biasRsrc, (unsigned)((ch * 32 + c16 * 16 + 4 * (lane >> 4)) * 4), layer * 256, 0);
#pragma unroll
			for (int i = 0; i < 4; ++i) {
				setAccElem(biasVec, 4 * c16 + i, __uint_as_float(v[i]));
				setAccElem(biasVec, 8 + 4 * c16 + i, __uint_as_float(v[i]));
			}
		}
#else
#pragma unroll
		for (int g = 0; g < 4; ++g) {
			const u32x4w v = __builtin_amdgcn_raw_buffer_load_b128(biasRsrc, (unsigned)((ch * 32 + 4 * hh + 8 * g) * 4), layer * 256, 0);
#pragma unroll
			for (int i = 0; i < 4; ++i) setAccElem(biasVec, 4 * g + i, __uint_as_float(v[i]));  // (bit_cast of a vector element reads element 0)
		}
#endif
	};
	loadBias(0);

	// per-lane LDS address parts (the swizzle depends only on the column: rows are 32 px)
	// B fragment of macro-step (dx, ks): byte offset inside a row =
	// colBase[dx] + (((ks*2+hh) ^ colSwz[dx]) << 4); 6 registers instead of 12
	unsigned colBase[3], colSwz[3];
#pragma unroll
	for (int dx = 0; dx < 3; ++dx) {
#if JU_TOWER_M16  // a B fragment is 16 pixels x 32 channels: the lane's pixel is lane & 15
		const int cq = (lane & 15) + dx;
#else
		const int cq = px + dx;
#endif
		colBase[dx] = cq * 128;
		colSwz[dx] = (cq >> 1) & 7;
	}
	unsigned outsw[4];  // [g]: byte offset of this lane's 4 output channels inside a row
#pragma unroll
	for (int g = 0; g < 4; ++g) {
#if JU_TOWER_M16
		// accumulator quarter g = (pixel half g >> 1, cout quarter g & 1): pixel (g >> 1) * 16 + (lane & 15), channels
		// ch * 32 + (g & 1) * 16 + 4 (lane >> 4) .. + 3 = chunk ch * 4 + (g & 1) * 2 + (lane >> 5), 8-byte half (lane >> 4) & 1
		const int cq = (g >> 1) * 16 + (lane & 15) + 1;
		outsw[g] = cq * 128 + (((ch * 4 + (g & 1) * 2 + (lane >> 5)) ^ ((cq >> 1) & 7)) << 4) + ((lane >> 4) & 1) * 8;
#else
		const int cq = px + 1;
		outsw[g] = cq * 128 + (((ch * 4 + g) ^ ((cq >> 1) & 7)) << 4) + hh * 8;
#endif
	}

	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();

	// VARIANT 4 (diagnostic build only): per-wave cycle sums of the phases below
	u64 prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	u64 extraPasses = 0;  // VARIANT 4: sweep passes after the first
	float calibMax = 0.f;  // VARIANT 5: largest post-ReLU output of the current layer (this lane)
	auto stamp = [&]() __attribute__((always_inline)) -> u64 {
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
	// one convolution layer over the region, in segments
	// ------------------------------------------------------------------------
	// A unit is a row pair (or the odd last row) x 32 columns x this wave's 32 output
	// channels: 12 macro-steps (dx, ks) of 6 (3) MFMAs.  The steps run dx = 1 FIRST: those
	// read no halo column, so for a unit that touches no halo row (every unit but the
	// region's first and last pair) they depend on nothing a neighbour sends.  Up to
	// kPreRun such units per wave run these four steps -- a third of their MFMAs -- while
	// the halo loads of the layer's input are in flight ("pre-run"), and finish after the
	// sweep; the exchange round trip (~2.8k cycles per layer) hides behind them.
	constexpr int kOrder[12] = {4, 5, 6, 7, 0, 1, 2, 3, 8, 9, 10, 11};
	// VARIANT 8 (tests): the PLAIN schedule -- no pre-run, weights as a burst after the publish, units
	// in their natural order -- with the same arithmetic per output element (same tap order, bias as
	// the C operand), so its bytes must equal the product kernel's: the cross-check that the
	// product's schedule (accumulators kept across the sweep, registers refilled behind MFMAs, asm
	// fragment reads under its register pressure) loses or corrupts nothing.
	constexpr bool kPlain = VARIANT == 8;
	constexpr int kPreRun = kPlain ? 0 : FAST ? JU_FAST_PRERUN : JU_PRERUN;
	constexpr int kPreBefore = kPlain ? 0 : JU_PREBEFORE;  // how many of them run BEFORE the halo loads are issued
	TowerAcc accPre0[2], accPre1[2], accPre2[2];  // (separate objects: an array indexed by the slot would live in scratch)
	static_assert(kPreRun >= 0 && kPreRun <= 3, "at most three interior units per wave");
	Vec8<T> fb[2][4];
	// (The reads are asm so that they can be issued a macro-step ahead with counted waits.  To the
	// compiler an asm output is defined when the statement ends -- it may copy the register before
	// the data has landed -- so a fragment's live range is kept to the one macro-step between its
	// read and its MFMAs and nothing else is read this way: reading the residual early the same way
	// was tried and produced rare wrong values under this kernel's register pressure.  The
	// determinism soak and the parity suite guard it.)
	auto issue = [&](unsigned rowAddr, int m, int set, int j) __attribute__((always_inline)) {
#if JU_TOWER_M16
		// macro-step m = (dx, ks32, pixel half): the fragment of row j is 16 pixels x 32 channels; the second
		// pixel half is 16 columns = 2048 bytes further (same swizzle: ((c + 16) >> 1) & 7 == (c >> 1) & 7)
		const int dx = m >> 2, ks32 = (m >> 1) & 1, ph = m & 1;
		const unsigned a = rowAddr + colBase[dx] + (((unsigned)(ks32 * 4 + (lane >> 4)) ^ colSwz[dx]) << 4);
		if (ph) {  // (the pixel half as the instruction's immediate offset: no address arithmetic)
			if (j == 0) asm volatile("ds_read_b128 %0, %1 offset:2048" : "=v"(fb[set][0]) : "v"(a));
			else if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:6400" : "=v"(fb[set][1]) : "v"(a));
			else if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:10752" : "=v"(fb[set][2]) : "v"(a));
			else asm volatile("ds_read_b128 %0, %1 offset:15104" : "=v"(fb[set][3]) : "v"(a));
			return;
		}
#else
		const int dx = m >> 2, ks = m & 3;
		const unsigned a = rowAddr + colBase[dx] + (((unsigned)(ks * 2 + hh) ^ colSwz[dx]) << 4);
#endif
		if (j == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(fb[set][0]) : "v"(a));
		else if (j == 1) asm volatile("ds_read_b128 %0, %1 offset:4352" : "=v"(fb[set][1]) : "v"(a));
		else if (j == 2) asm volatile("ds_read_b128 %0, %1 offset:8704" : "=v"(fb[set][2]) : "v"(a));
		else asm volatile("ds_read_b128 %0, %1 offset:13056" : "=v"(fb[set][3]) : "v"(a));
	};
	// rows of this wave: full pairs u = rp, rp+2, ... and, for an odd region height, the
	// last row as a single-row unit on the wave group with fewer pairs
	const int np2 = rhv >> 1;
	const bool mySingle = (rhv & 1) && rp == (np2 & 1);
	// pre-run units: preFirst, preFirst + 2, ... (pairs whose four input rows are interior)
	const int preFirst = rp ? 1 : 2;
	int nPre = 0;
#pragma unroll
	for (int k = 0; k < kPreRun; ++k) {
		const int u = preFirst + 2 * k;
		if (u < np2 && 2 * u + 3 <= rhv) nPre = k + 1;
	}
	if constexpr (FAST && kPreRun == 3) {
		// (one whole unit stays: the next layer's weights stream behind it)
		const int mine = (np2 - rp + 1) >> 1;
		nPre = nPre < mine - 1 ? nPre : mine - 1;
	}
	const int preEnd = preFirst + 2 * nPre;  // first unit of this wave's parity at or after preFirst that is NOT pre-run
	auto isPre = [&](int u) __attribute__((always_inline)) { return u >= preFirst && u < preEnd; };
	// next unit of this wave after `u` that runs whole (not pre-run), or -1
	auto nextWhole = [&](int u) __attribute__((always_inline)) {
		int v = u + 2;
		if (isPre(v)) v = preEnd;
		return v < np2 ? v : -1;
	};
	const int firstWhole = isPre(rp) ? (preEnd < np2 ? preEnd : -1) : (rp < np2 ? rp : -1);
	const bool streamsInUnit = (firstWhole >= 0 || mySingle) && !kNoMfma && !kPlain;  // the next layer's weights stream behind the last unit
	constexpr bool kDefer = FAST;
	const bool deferShape = residentFastShape(rhv, rwv);  // (region-uniform: a function of the region's size only)
	if constexpr (FAST) {
		if (!deferShape) {  // (the host chose the wrong instantiation: fail loudly, never compute something else)
			if (tid == 0) __hip_atomic_store((gu32 *)p.error, 0x6f0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
			return;
		}
	}

	// KIND 0: pre-run (accumulator init + positions 0..3); 1: finish (positions 4..11 +
	// epilogue); 2: whole unit.  `primed`: the first step's fragments are already in flight
	// in set 0.  nextUnit >= 0: prime that unit's first step at the end -- position 0
	// (nextFinish false: a pre-run or whole unit) or 4 (a finish) -- so that its fragments
	// travel while this unit's epilogue runs.
	// Deferred epilogue (DEF): the previous unit of this wave (accumulators `dacc`, row pair `dunit`)
	// has not been written yet -- its epilogue runs one (r, g) group per macro-step BEHIND this
	// unit's MFMAs (VALU and the LDS write issue while the matrix pipe works on the macro-step's
	// last two MFMAs), instead of as ~560 exposed cycles between the units.  Same arithmetic, same
	// order per element.  EPI false: this unit leaves its own epilogue to its successor.
	// is this lane's pixel of accumulator group g inside the region (ragged right edge)?
#if JU_TOWER_M16
	const bool validLo = (lane & 15) < rwv, validHi = 16 + (lane & 15) < rwv;
	auto groupValid = [&](const int g) __attribute__((always_inline)) { return FAST || ((g >> 1) ? validHi : validLo); };
#else
	const bool lanesValid = px < rwv;
	auto groupValid = [&](const int) __attribute__((always_inline)) { return lanesValid; };
#endif
	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	const __amdgpu_buffer_rsrc_t mailRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)p.mail, 0, (int)((size_t)p.GX * p.GY * 2 * (LEAKY ? 2 : 1) * kResMailSlots * 16), 0x00020000);
	constexpr int kSc1 = 16;  // cache-policy bit of sc1 (write-through store / L2-bypassing load)
	// (Measured and dropped: publishing the edges from the epilogue groups themselves -- 8-byte stores of
	// the edge lanes, no cooperative publish after the barrier.  Bytes equal, 434 against 361 us: a
	// sparse write-through store costs ~90 cycles of issue wherever it stands
	// (tools/probes/mfma_valu_overlap.hip: the chip accepts ~13 such instructions per clock), and the
	// epilogues need 36 per wave and layer where the cooperative publish needs 4 dense ones.)
	Vec4<T> rvNext[2][4];  // residual of the unit whose epilogue is pending
	auto epiValue = [&](auto resTag, const TowerAcc &a, const int g, const Vec4<T> &rv, float(&v)[4]) __attribute__((always_inline)) {
#pragma unroll
		for (int i = 0; i < 4; ++i) v[i] = accElem(a, 4 * g + i);
		if constexpr (decltype(resTag)::value) {
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] += static_cast<float>(rv[i]);
		}
		if constexpr (VARIANT == 5 && !LEAKY) {  // calibration build: range of the layer's output
			if (groupValid(g)) {
#pragma unroll
				for (int i = 0; i < 4; ++i) calibMax = fmaxf(calibMax, v[i]);  // (= max of the ReLU'd values)
			}
		}
	};
	auto epiStore = [&](auto outTag, const int row, const int g, float(&v)[4]) __attribute__((always_inline)) {
		constexpr int outOff = decltype(outTag)::value;
		if constexpr (LEAKY) {
			// slope in [0, 1] (model.cpp): max(x, slope * x) IS x < 0 ? slope * x : x
#pragma unroll
			for (int i = 0; i < 4; ++i) v[i] = fmaxf(v[i], v[i] * p.slope);
			if constexpr (VARIANT == 5) {
#pragma unroll
				for (int i = 0; i < 4; ++i) calibMax = fmaxf(calibMax, fabsf(v[i]));
			}
			*reinterpret_cast<Vec4<T> *>(smem + outOff + row * kResRowBytes + outsw[g]) = pack4<T>(v[0], v[1], v[2], v[3]);
		} else {
			*reinterpret_cast<Vec4<T> *>(smem + outOff + row * kResRowBytes + outsw[g]) =
			    reluPacked<T>(pack4<T>(v[0], v[1], v[2], v[3]));
		}
	};
	auto unitSeg = [&](auto rowsTag, auto kindTag, auto resTag, auto inTag, auto outTag, TowerAcc(&acc)[2],
	                   const int layer, const int unit, const bool primed, const int nextUnit,
	                   const bool nextFinish, auto streamTag, auto defTag, TowerAcc(&dacc)[2], const int dunit,
	                   auto epiTag) __attribute__((always_inline)) {
		constexpr bool streamNext = decltype(streamTag)::value;  // (compile-time: no branch per macro-step)
		constexpr bool DEF = decltype(defTag)::value;   // (a deferring segment is always primed and deferred units are row pairs)
		constexpr bool EPI = decltype(epiTag)::value;
		constexpr int ROWS = decltype(rowsTag)::value;
		constexpr int KIND = decltype(kindTag)::value;
		constexpr bool residual = decltype(resTag)::value;
		constexpr int inOff = decltype(inTag)::value;
		constexpr int outOff = decltype(outTag)::value;
		constexpr int P0 = KIND == 1 ? 4 : 0;
		constexpr int P1 = KIND == 0 ? 4 : 12;
		const u64 tu0 = stamp();
		constexpr int NR = ROWS + 2;      // input rows / fragment reads per macro-step
		constexpr int NM = 3 * ROWS;      // MFMAs per macro-step
		const int ra = 1 + 2 * unit;      // first output row (buffer row index)
		if constexpr (KIND != 1 && kNoMfma) {
#pragma unroll
			for (int r = 0; r < ROWS; ++r) acc[r] = biasVec;
		}
		// (opaque: otherwise every (unit, macro-step) fragment address is loop-invariant over the layers and is
		// kept in a register of its own -- 67 registers of the fast instantiation; with them gone a third
		// pre-run unit fits without scratch, and was measured no faster: 350-358 against 348-352 us)
		unsigned rowAddr = ldsBase + inOff + (2 * unit) * kResRowBytes;
		asm volatile("" : "+v"(rowAddr));
		// A unit whose epilogue is deferred (EPI false) fetches its residual behind the MFMAs of its own
		// last macro-step into rvNext (plain C++ loads: the compiler's own waits count only what it
		// knows of and are therefore never too lenient); its successor (DEF) consumes them.
		constexpr bool kFetchRes = !EPI && residual && KIND != 0;
		const int dra = 1 + 2 * dunit;
		// (by value, constant indices: an element read of `dacc[j >> 2]` is a dynamic index until the loops
		// are unrolled, by when the loads had been merged into partial vectors and the array stayed in scratch)
		const TowerAcc dacc0 = dacc[0], dacc1 = dacc[1];
		float dv[4];
		unsigned dlo = 0, dhi = 0;
		// The deferred epilogue's slot behind MFMA group k of macro-step `pos` (DEF segments): the previous unit's group
		// (r, g) = step index -- MFMA groups 0..3 value i = accumulator (+ residual), 4 convert, 5 ReLU + write.
		// `part` (16x16x32 form): 0 = the slot between the group's two MFMAs, 1 = behind the second one (the LDS write of
		// the last group only: with its lane mask it does not fit the 8 cycles between two 16-cycle MFMAs)
		auto defSlot = [&](const int pos, const int k, const int part = -1) __attribute__((always_inline)) {
					if constexpr (DEF) {
						// group (r, g) = step index, at most three plain VALU instructions behind each MFMA
						// (tools/probes/mfma_valu_overlap.hip: that many issue in an MFMA's shadow for
						// free, a fourth and every packed-f32 one cost their full issue time):
						// MFMA 0..3 value i = accumulator (+ residual), 4 convert, 5 ReLU + write
						const int j = pos - P0;
						if (j < 8) {
							const int r = j >> 2, g = j & 3;
							if (k < 5 && part == 1) {
							} else if (k < 4) {
								dv[k] = accElem(r ? dacc1 : dacc0, 4 * g + k);
								if constexpr (residual) {
									dv[k] += static_cast<float>(rvNext[r][g][k]);
									asm volatile("" : "+v"(dv[k]));  // (keeps the four adds scalar and in their slots)
								}
								if constexpr (LEAKY) {
									// max(x, slope x), slope in [0, 1]; the instruction itself (fmaxf() canonicalises
									// both operands first: two more instructions in a slot that has room for three)
									const float t = dv[k] * p.slope;
									asm volatile("v_max_f32 %0, %1, %2" : "=v"(dv[k]) : "v"(dv[k]), "v"(t));
								}
							} else if (k == 4) {
								const Vec4<T> pk = pack4<T>(dv[0], dv[1], dv[2], dv[3]);
								typedef unsigned u32x2d __attribute__((ext_vector_type(2)));
								const u32x2d w = __builtin_bit_cast(u32x2d, pk);
								dlo = w[0];
								dhi = w[1];
								asm volatile("" : "+v"(dlo), "+v"(dhi));
							} else {
								typedef unsigned u32x2d __attribute__((ext_vector_type(2)));
								if (part != 1) {
									const Vec4<T> pk = __builtin_bit_cast(Vec4<T>, u32x2d{dlo, dhi});
									const u32x2d w = __builtin_bit_cast(u32x2d, LEAKY ? pk : reluPacked<T>(pk));
									dlo = w[0];
									dhi = w[1];
									if (part == 0) asm volatile("" : "+v"(dlo), "+v"(dhi));
								}
								if (part != 0) {
									const Vec4<T> o = __builtin_bit_cast(Vec4<T>, u32x2d{dlo, dhi});
									if (groupValid(g)) *reinterpret_cast<Vec4<T> *>(smem + outOff + (dra + r) * kResRowBytes + outsw[g]) = o;
								}
							}
						}
					}
		};
		if (!kNoMfma) {
			if (!primed) {
				asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
				__builtin_amdgcn_sched_barrier(0);
#pragma unroll
				for (int j = 0; j < NR; ++j) issue(rowAddr, kOrder[P0], 0, j);
			}
#pragma unroll
			for (int pos = P0; pos < P1; ++pos) {
				const int m = kOrder[pos];
				const int set = (pos - P0) & 1;
				const bool more = (pos + 1 < P1);
				const int dx = m >> 2;
				[[maybe_unused]] const int ks = m & 3;  // (32x32x16 form: the 16-channel k-step)
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
						// deferred epilogue: the residual reads sit behind the first step's fragments, a
						// group's LDS write behind the fragments of the step after it
						const int younger = (DEF && pos > P0 && pos - P0 <= 8 ? 1 : 0) +
						                    (kFetchRes && pos == P1 - 1 ? 2 * (k < 4 ? k : 4) : 0);
						const int allowed = (NR - 1 - need) + issuedNext + younger;
						if (allowed >= 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
						else if (allowed == 11) asm volatile("s_waitcnt lgkmcnt(11)" ::: "memory");
						else if (allowed == 10) asm volatile("s_waitcnt lgkmcnt(10)" ::: "memory");
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
#if JU_TOWER_M16
					{
						// the same 12 macro-steps as (dx, ks32, pixel half): two 16x16x32 MFMAs (cout halves) on
						// quarters (ph, c16) of the row's accumulator; A fragment index ks = ks32 * 2 + c16
						const int ks32 = (m >> 1) & 1, ph = m & 1;
						const bool first = KIND != 1 && (pos == P0 || pos == P0 + 1) && dy == 0;
#pragma unroll
						for (int c16 = 0; c16 < 2; ++c16) {
							const int kf = ks32 * 2 + c16, sl = ph * 2 + c16;
							acc[r].q[sl] = mfma16(dx == 1 ? wm[dy * 4 + kf] : ws[dy * 8 + (dx >> 1) * 4 + kf], fb[set][need],
							    first ? biasVec.q[sl] : acc[r].q[sl]);
							if (c16 == 0) {
								// (the deferred epilogue's slot goes BETWEEN the group's two MFMAs: a 16-cycle MFMA leaves 8 cycles
								// of issue, and behind the second one come the next fragment read and the next group's wait)
								defSlot(pos, k, 0);
								__builtin_amdgcn_sched_barrier(0);
							}
						}
						defSlot(pos, k, 1);
						if constexpr (KIND == 2) {
							// The wave's LAST unit of the layer, second pixel half of (dx, ks32): vertical tap dy's two fragments
							// have fed their last MFMA (group k = 2 dy + 1) -- the next layer's go into the same registers now,
							// two loads behind this group instead of six in a row behind the step
							if (streamNext && (m & 1) && (ROWS == 2 ? (k & 1) : true)) {
								const int sdy = ROWS == 2 ? (k >> 1) : k;
#pragma unroll
								for (int c16 = 0; c16 < 2; ++c16) {
									const int kf = ks32 * 2 + c16;
									const Vec8<T> nf = loadWeightFrag(layer + 1, (sdy * 3 + dx) * 4 + kf);
									if (dx == 1) wm[sdy * 4 + kf] = nf;
									else ws[sdy * 8 + (dx >> 1) * 4 + kf] = nf;
								}
							}
						}
					}
#else
					// (a unit's first MFMA of each row takes the bias as its C operand)
					acc[r] = mfma32(dx == 1 ? wm[dy * 4 + ks] : ws[dy * 8 + (dx >> 1) * 4 + ks], fb[set][need],
					    (KIND != 1 && pos == P0 && dy == 0) ? biasVec : acc[r]);
#endif
					if (more && k < NR) issue(rowAddr, kOrder[pos + 1 < 12 ? pos + 1 : 11], set ^ 1, k);
#if !JU_TOWER_M16
					defSlot(pos, k);
#endif
					if constexpr (kFetchRes) {
						if (pos == P1 - 1 && k < 4) {
							const int ra0 = 1 + 2 * unit;
							rvNext[(2 * k) >> 2][(2 * k) & 3] = *reinterpret_cast<const Vec4<T> *>(
							    smem + outOff + (ra0 + ((2 * k) >> 2)) * kResRowBytes + outsw[(2 * k) & 3]);
							rvNext[(2 * k + 1) >> 2][(2 * k + 1) & 3] = *reinterpret_cast<const Vec4<T> *>(
							    smem + outOff + (ra0 + ((2 * k + 1) >> 2)) * kResRowBytes + outsw[(2 * k + 1) & 3]);
						}
					}
					__builtin_amdgcn_sched_barrier(0);
				}
				// The wave's LAST unit of the layer: the three fragments of this step (and, after
				// the first step, the bias) are dead now -- the next layer's go into the same
				// registers from here, one step at a time, so the 36 loads per lane travel
				// behind this unit's MFMAs instead of as a burst between the layers (four waves
				// x 36 KB through the CU's 64 B/clk address path: ~2.4k cycles of issue).
				if constexpr (KIND == 2) {
					if (streamNext) {
#if JU_TOWER_M16
						if (pos == 1) loadBias(layer + 1);  // (the bias is the C operand of the first TWO steps: one per pixel half)
#else
						if (pos == 0) loadBias(layer + 1);
#endif
#if JU_TOWER_M16
						// (the six fragments of (dx, ks32) go out from INSIDE the second pixel half's step, two behind each
						// vertical tap's last MFMA group -- above, in the k loop -- not as six loads in a row here)
#else
#pragma unroll
						for (int dy = 0; dy < 3; ++dy) {
							const Vec8<T> nf = loadWeightFrag(layer + 1, (dy * 3 + dx) * 4 + ks);
							if (dx == 1) wm[dy * 4 + ks] = nf;
							else ws[dy * 8 + (dx >> 1) * 4 + ks] = nf;
						}
#endif
						__builtin_amdgcn_sched_barrier(0);
					}
				}
			}
			// prime the next segment (a pair, or the single row: >= 3 input rows)
			if (nextUnit >= 0) {
				const unsigned na = ldsBase + inOff + (2 * nextUnit) * kResRowBytes;
				const bool nextSingle = (nextUnit == np2);
				if (nextFinish) {
#pragma unroll
					for (int j = 0; j < 4; ++j) issue(na, kOrder[4], 0, j);
				} else {
#pragma unroll
					for (int j = 0; j < 3; ++j) issue(na, kOrder[0], 0, j);
					if (!nextSingle) issue(na, kOrder[0], 0, 3);
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
		const u64 tu1 = stamp();
		if constexpr (KIND == 0) prof[7] += tu1 - tu0;
#ifdef JU_TOWER_SEGPROF  // developer builds: the finish segments one by one (slots 3..6 carry them instead of their usual sums)
		else if constexpr (KIND == 1 && !DEF) prof[3] += tu1 - tu0;
		else if constexpr (KIND == 1 && DEF) prof[4] += tu1 - tu0;
		else if constexpr (KIND == 2 && !streamNext) prof[5] += tu1 - tu0;
		else prof[6] += tu1 - tu0;
#else
		else prof[5] += tu1 - tu0;
#endif
		if constexpr (KIND != 0 && EPI) {
			// ---- epilogue: (+residual) ReLU, 16-bit, into the output buffer interior ----
			// (row ra + r <= rhv always: units are whole pairs, or the odd last row)
			{
				// The residual is read-modify-write in place.  All reads of the unit
				// first, then the arithmetic and the writes: left to the compiler every
				// group is read -> wait -> write -> next read (it cannot prove the groups
				// do not alias), i.e. 8 exposed LDS round trips per unit.  (Lanes whose pixel lies beyond a
				// ragged right edge read inside the buffer and store nothing.)
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
						epiValue(resTag, acc[r], g, rv[r][g], v);
						if (groupValid(g)) epiStore(outTag, ra + r, g, v);
					}
				}
			}
			const u64 tu2 = stamp();
#ifndef JU_TOWER_SEGPROF
			prof[6] += tu2 - tu1;
#endif
		}
	};
	using R2 = std::integral_constant<int, 2>;
	using R1 = std::integral_constant<int, 1>;
	using KPre = std::integral_constant<int, 0>;
	using KFin = std::integral_constant<int, 1>;
	using KWhole = std::integral_constant<int, 2>;
	// the halo-independent third of up to kPreRun units of `layer` (input buffer inTag)
	// slots [LO, HI) of the pre-run
	auto preRun = [&](auto inTag, auto outTag, auto loTag, auto hiTag, const int layer, const bool primedFirst)
	                  __attribute__((always_inline)) {
		constexpr int LO = decltype(loTag)::value, HI = decltype(hiTag)::value;
		auto slot = [&](TowerAcc(&acc)[2], const int k) __attribute__((always_inline)) {
			if (k < nPre) {
				// (k + 1 < kPreRun folds at compile time: without it the last slot carries a never-taken branch that
				// primes a successor nobody reads -- four LDS reads into dead registers which the halo loads then
				// reuse, the one thing tools/lds_wait_check.py found in the built kernels)
				const int nu = (k + 1 < kPreRun && k + 1 < nPre) ? preFirst + 2 * (k + 1) : -1;
				unitSeg(R2{}, KPre{}, std::false_type{}, inTag, outTag, acc, layer, preFirst + 2 * k,
				    (k > 0 || primedFirst) && !kNoMfma, nu, false, std::false_type{}, std::false_type{}, acc, 0, std::true_type{});
			}
		};
		if constexpr (LO <= 0 && 0 < HI) slot(accPre0, 0);
		if constexpr (LO <= 1 && 1 < HI) slot(accPre1, 1);
		if constexpr (LO <= 2 && 2 < HI) slot(accPre2, 2);
	};
	// the rest of the layer: the pre-run units' remaining steps, then the other units whole
	auto finishLayer = [&](auto resTag, auto inTag, auto outTag, const int layer, const bool streamW) __attribute__((always_inline)) {
		const int afterPre = firstWhole >= 0 ? firstWhole : (mySingle ? np2 : -1);
		if constexpr (kDefer) {
			// The fast schedule's shape (two pre-run pairs, then one or two whole pairs: every wave of a
			// 16- or 14-row region): each unit's epilogue runs behind its successor's MFMAs, only the
			// last one is exposed.  The accumulators rotate through the two pre-run sets -- a set is
			// free again once its epilogue has run, one segment later.
			// In every shape the LAST unit accumulates into accPre0 and leaves its epilogue too: that one runs
			// once, below the branches (emitted per branch, the identical tails are merged by the optimiser into
			// one that picks its accumulator array through a pointer -- and the arrays then live in scratch).
			{
				using Y = std::true_type;
				using N = std::false_type;
				const int ua = firstWhole, ub = nextWhole(firstWhole);
				int lastUnit = ua;
				unitSeg(R2{}, KFin{}, resTag, inTag, outTag, accPre0, layer, preFirst, false, preFirst + 2, true, N{}, N{}, accPre0, 0, N{});
				if (kPreRun == 3 && nPre == 3) {
					// (16-row regions: three pre-run pairs, one whole pair)
					unitSeg(R2{}, KFin{}, resTag, inTag, outTag, accPre1, layer, preFirst + 2, true, preFirst + 4, true, N{}, Y{}, accPre0, preFirst, N{});
					unitSeg(R2{}, KFin{}, resTag, inTag, outTag, accPre2, layer, preFirst + 4, true, ua, false, N{}, Y{}, accPre1, preFirst + 2, N{});
					if (streamW) unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ua, true, -1, false, Y{}, Y{}, accPre2, preFirst + 4, N{});
					else unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ua, true, -1, false, N{}, Y{}, accPre2, preFirst + 4, N{});
				} else {
					unitSeg(R2{}, KFin{}, resTag, inTag, outTag, accPre1, layer, preFirst + 2, true, ua, false, N{}, Y{}, accPre0, preFirst, N{});
					if (ub >= 0) {
						// (the third set takes the first whole pair, so that the last one finds accPre0 free)
						unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre2, layer, ua, true, ub, false, N{}, Y{}, accPre1, preFirst + 2, N{});
						if (streamW) unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ub, true, -1, false, Y{}, Y{}, accPre2, ua, N{});
						else unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ub, true, -1, false, N{}, Y{}, accPre2, ua, N{});
						lastUnit = ub;
					} else {
						if (streamW) unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ua, true, -1, false, Y{}, Y{}, accPre1, preFirst + 2, N{});
						else unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, accPre0, layer, ua, true, -1, false, N{}, Y{}, accPre1, preFirst + 2, N{});
					}
				}
				// the one exposed epilogue of the layer (its residual was fetched behind the unit's last MFMAs)
				const u64 te0 = stamp();
				const int ra = 1 + 2 * lastUnit;
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						float v[4];
						epiValue(resTag, accPre0[r], g, rvNext[r][g], v);
						if (groupValid(g)) epiStore(outTag, ra + r, g, v);
					}
				}
#ifndef JU_TOWER_SEGPROF
				prof[6] += stamp() - te0;
#else
				(void)te0;
#endif
			}
		} else {
		auto slot = [&](TowerAcc(&acc)[2], const int k) __attribute__((always_inline)) {
			if (k < nPre) {
				const bool nextIsPre = k + 1 < nPre;
				const int nu = nextIsPre ? preFirst + 2 * (k + 1) : afterPre;
				unitSeg(R2{}, KFin{}, resTag, inTag, outTag, acc, layer, preFirst + 2 * k, k > 0 && !kNoMfma, nu, nextIsPre,
				    std::false_type{}, std::false_type{}, acc, 0, std::true_type{});
			}
		};
		if constexpr (kPreRun > 0) slot(accPre0, 0);
		if constexpr (kPreRun > 1) slot(accPre1, 1);
		if constexpr (kPreRun > 2) slot(accPre2, 2);
		// (every finish segment primes its successor when there is one)
		bool primed = nPre > 0 && afterPre >= 0 && !kNoMfma;
		TowerAcc acc[2];
		for (int u = firstWhole; u >= 0;) {
			const int nw = nextWhole(u);
			const int nu = nw >= 0 ? nw : (mySingle ? np2 : -1);
			if (streamW && nu < 0) unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, acc, layer, u, primed, nu, false, std::true_type{}, std::false_type{}, acc, 0, std::true_type{});
			else unitSeg(R2{}, KWhole{}, resTag, inTag, outTag, acc, layer, u, primed, nu, false, std::false_type{}, std::false_type{}, acc, 0, std::true_type{});
			primed = nu >= 0 && !kNoMfma;
			u = nw;
		}
		if (mySingle) {
			if (streamW) unitSeg(R1{}, KWhole{}, resTag, inTag, outTag, acc, layer, np2, primed, -1, false, std::true_type{}, std::false_type{}, acc, 0, std::true_type{});
			else unitSeg(R1{}, KWhole{}, resTag, inTag, outTag, acc, layer, np2, primed, -1, false, std::false_type{}, std::false_type{}, acc, 0, std::true_type{});
		}
		}
	};

	// ------------------------------------------------------------------------
	// edge ring -> mailbox (publish) and neighbours' mailboxes -> halo ring
	// ------------------------------------------------------------------------
	// Self-validating slots: every tower output is post-ReLU (>= 0), so the sign bit
	// of each of the 8 values in a 16-byte slot is free.  EVERY dword carries the same
	// 2-bit epoch e = (number of writes to this slot so far) & 3 in its two sign bits:
	// consecutive writes to a slot differ in e, so in every dword -- a slot is accepted only
	// when all four dwords show the expected epoch, which a store or load torn at dword
	// granularity between two consecutive writes cannot produce (the previous 8-bit tag
	// spread over the dwords left 12 of 16 bytes unprotected against that).  The slot always
	// holds the previous write or the expected one (the producer cannot run further ahead:
	// it needs this region's next layer first), and the mailbox starts zeroed with the
	// counts (e = 0 never matches a first write, e = 1).  The producer needs neither a drain
	// nor a release: ONE hop instead of store-ack -> flag -> load.
	auto epochMask = [&](int par) __attribute__((always_inline)) -> unsigned {
		const unsigned e = pubCount[par] & 3u;
		return (e & 1u) << 15 | (e >> 1) << 31;
	};
	// Slot descriptors, computed ONCE: they depend on the lane and the region only.  The
	// buffer (A / B) enters as the LDS instructions' immediate offset and the slot parity as
	// the buffer instructions' scalar offset, so publish and sweep are loads, stores and the
	// tag arithmetic, nothing else -- in this kernel every other instruction is serial time.
	// LEAKY: a LeakyReLU output has no free bit, so a 16-byte slot carries FOUR values, each
	// dword = value16 | epoch16 << 16 with epoch = (writes to this slot so far) & 0xffff --
	// every dword still validates itself, the mailbox holds twice the slots (16 per pixel
	// record instead of 8) and publish / sweep move 8-byte half chunks on the LDS side.
	// The exchange is branch-free (round 6).  Measured with stamps inside the fill (profiles/r06_tower_fill.txt):
	// the sweep's loads have a raw latency of ~730 cycles and are back before the pre-run unit behind them ends -- what
	// was called "exposed exchange" was the CODE that checks them: per slot four v_cmp -> s_and chains, a saveexec and a
	// branch, ~180 cycles each at one wave per SIMD, 900-1050 cycles per layer, and as many again where a second pass is
	// checked.  Now: one OR of the four (dword ^ expected epoch) per slot into a per-lane accumulator, ONE vote per pass;
	// every lane writes its slot's payload every pass (a slot holds the previous write or the expected one, never a newer
	// one, so a rewrite can only repeat the bytes; lanes without a halo cell write the 16 spare bytes behind the buffer
	// and read their OWN region's slot, which carries the same epoch); the publish stores unconditionally (an entry
	// beyond the region's rows lands in its own slot, which no consumer reads).
	// Every instantiation but the plain schedule (VARIANT 8: the tests' cross-check keeps the per-slot form, on the same
	// mailbox -- the slots both forms read are written by both) exchanges this way.
	constexpr bool kLean = VARIANT != 8;
	constexpr bool kCompact = kLean;  // (the column strips packed: below)
	constexpr int kSlots = LEAKY ? 2 * kResMailSlots : kResMailSlots;  // per region and parity
	constexpr int CPP = LEAKY ? 16 : 8;      // slots per pixel record
	constexpr int CSH = LEAKY ? 4 : 3;
	// (kCompact: a region is at most 16 rows high, so a column strip has 16 entries, not 32 -- ReLU: both columns share ONE
	// slot per thread (threads 0..127 the left column, 128..255 the right one): 3 publish stores and 4 sweep loads per
	// thread instead of 4 and 5; LeakyReLU (twice the slots per pixel): one slot per thread and column, 6 and 7 instead of
	// 8 and 9 -- every lane of every instruction at work; the mailbox's slot numbering is unchanged)
	constexpr int ROWITS = 2 * CPP / 8;      // slots per thread of the two row strips
	constexpr int COLPT = 16 * CPP;          // threads (= slots) of one packed column strip
	constexpr int NP = kCompact ? ROWITS + 2 * COLPT / 256 : kSlots / 256;  // publish: 4 strips x 32 entries x CPP slots
	constexpr int NS = NP + 1;               // sweep: 4 sides x 32 entries x CPP slots, + the 4 corners
	unsigned pubLds[NP];                     // LDS byte offset inside a buffer of the (half) chunk to publish
	unsigned pubOff[kCompact ? NP : 1];      // kCompact: the slot's byte offset in the mailbox, parity 0 (else pubBase + it * 4096)
	unsigned pubValid = 0;
#pragma unroll
	for (int it = 0; it < NP; ++it) {
		const int idx = it * 256 + tid;
		int strip = idx >> (5 + CSH), e = (idx >> CSH) & 31;
		const int cs = idx & (CPP - 1);
		if (kCompact && it >= ROWITS) {
			const int j = (it - ROWITS) * 256 + tid;
			strip = 2 + j / COLPT;
			e = (j / CPP) & 15;
		}
		if constexpr (kCompact) pubOff[it] = (unsigned)((region * 2) * kSlots + (strip * 32 + e) * CPP + cs) * 16u;
		const int c = LEAKY ? cs >> 1 : cs, half = LEAKY ? (cs & 1) * 8 : 0;
		int rr, cc;
		bool valid;
		if (strip == 0) { rr = 1; cc = e + 1; valid = e < rwv; }
		else if (strip == 1) { rr = rhv; cc = e + 1; valid = e < rwv; }
		else if (strip == 2) { rr = e + 1; cc = 1; valid = e < rhv; }
		else { rr = e + 1; cc = rwv; valid = e < rhv; }
		pubLds[it] = (unsigned)(rr * kResRowBytes + cc * 128 + ((c ^ ((cc >> 1) & 7)) << 4) + half);
		if (valid) pubValid |= 1u << it;
		// (the publish is branch-free: an entry beyond the region's rows or columns reads a harmless place and stores into
		// its own slot, which no consumer reads)
		if (kLean && !valid) pubLds[it] = 0u;
	}
	const unsigned pubBase = (unsigned)(region * 2 * kSlots) * 16u + (unsigned)tid * 16u;
	unsigned sweepSrc[NS];  // mailbox byte offset of the neighbour's slot, parity 0
	unsigned sweepLds[NS];  // LDS byte offset inside a buffer of the halo (half) chunk it fills
	unsigned sweepValid = 0;
#pragma unroll
	for (int it = 0; it < NS; ++it) {
		int nx = gxr, ny = gyr, strip, se, rr, cc, cs;
		bool valid;
		if (it < NS - 1) {
			const int idx = it * 256 + tid;
			const int hp = idx >> CSH;
			cs = idx & (CPP - 1);
			int side = hp >> 5, e = hp & 31;
			if (kCompact && it >= ROWITS) {  // (the packed column strips, 16 entries each)
				const int j = (it - ROWITS) * 256 + tid;
				side = 2 + j / COLPT;
				e = (j / CPP) & 15;
			}
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
			// corners: threads 0 .. 4 * CPP - 1 = 4 corners x CPP slots; the diagonal neighbour's
			// bottom/top row strip, last/first entry (interior columns are 32 wide)
			const int k = tid >> CSH;
			cs = tid & (CPP - 1);
			const bool up = k < 2, left = (k & 1) == 0;
			ny += up ? -1 : 1;
			nx += left ? -1 : 1;
			strip = up ? 1 : 0;
			se = left ? kResRW - 1 : 0;
			rr = up ? 0 : rhv + 1;
			cc = left ? 0 : rwv + 1;
			valid = tid < 4 * CPP;
		}
		const int c = LEAKY ? cs >> 1 : cs, half = LEAKY ? (cs & 1) * 8 : 0;
		valid = valid && nx >= 0 && nx < p.GX && ny >= 0 && ny < p.GY;
		const int nreg = valid ? ny * p.GX + nx : region;
		sweepSrc[it] = (unsigned)((nreg * 2) * kSlots + (strip * 32 + se) * CPP + cs) * 16u;
		sweepLds[it] = (unsigned)(rr * kResRowBytes + cc * 128 + ((c ^ ((cc >> 1) & 7)) << 4) + half);
		if (valid) sweepValid |= 1u << it;
		if (kLean && !valid) {
			// this lane has no neighbour slot to fetch: it reads a slot of its own region (published for the same layer into
			// the same parity: the epoch the check expects) and parks the bytes behind the buffer.  Entry 0 of the top
			// row strip: written in every geometry and by the plain schedule's publish too
			sweepSrc[it] = (unsigned)((region * 2) * kSlots + (tid & (CPP - 1))) * 16u;
			sweepLds[it] = (unsigned)kResDummyOff;
		}
	}
	constexpr unsigned kParityBytes = kSlots * 16u;
	// `layer`: the layer whose output (in buffer `off`) is published
	auto publish = [&](auto offTag, int layer) __attribute__((always_inline)) {
		constexpr int off = decltype(offTag)::value;
		// (the caller has just passed the workgroup barrier: the region's output is in LDS)
		const int ppar = (layer + 1) & 1;
		pubCount[ppar] += 1u;
		const unsigned soff = ppar ? kParityBytes : 0u;
		if constexpr (LEAKY) {
			typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
			const unsigned tg = pubCount[ppar] << 16;
			u32x2 v[NP];
#pragma unroll
			for (int it = 0; it < NP; ++it) v[it] = *reinterpret_cast<const u32x2 *>(smem + off + pubLds[it]);
#pragma unroll
			for (int it = 0; it < NP; ++it) {
				if (kLean || (pubValid >> it & 1u)) {
					const u32x4 d = {(v[it][0] & 0xffffu) | tg, (v[it][0] >> 16) | tg, (v[it][1] & 0xffffu) | tg,
					    (v[it][1] >> 16) | tg};
					__builtin_amdgcn_raw_buffer_store_b128(d, mailRsrc, kCompact ? pubOff[it] : pubBase + it * 4096, soff, kSc1);
				}
			}
		} else {
			const unsigned tm = epochMask(ppar);
			u32x4 v[NP];
#pragma unroll
			for (int it = 0; it < NP; ++it) v[it] = *reinterpret_cast<const u32x4 *>(smem + off + pubLds[it]);
#pragma unroll
			for (int it = 0; it < NP; ++it) {
				if constexpr (kCompact) {
					// (post-ReLU values: the sign bits are clear already)
					__builtin_amdgcn_raw_buffer_store_b128(v[it] | tm, mailRsrc, pubOff[it], soff, kSc1);
				} else if (pubValid >> it & 1u) {
					__builtin_amdgcn_raw_buffer_store_b128((v[it] & 0x7fff7fffu) | tm, mailRsrc,
					    pubBase + it * 4096, soff, kSc1);
				}
			}
		}
		// MUBUF stores with an SGPR soffset read their data registers late when a second wave of
		// the SIMD competes for the vector-memory issue, and hipcc inserts no wait state for this
		// form (tools/probes/mubuf_store_data.hip).  This kernel runs ONE wave per
		// SIMD, where the probe never saw it (gap 0, every configuration), and every slot is
		// tag-checked by its consumer; the two wait states cost nothing and keep the store's data
		// registers untouched for as long as the probe's worst case needed.
		asm volatile("s_nop 1" ::: "memory");
	};
	// fills the halo ring of buffer `off` with the neighbours' output of layer `layer`;
	// returns false on timeout (uniform across the workgroup)
	auto fillHalo = [&](auto offTag, int layer, auto &&behindFirstPass) __attribute__((always_inline)) -> bool {
		constexpr int off = decltype(offTag)::value;
		u64 t0 = 0;  // (the clock is read only once a pass has failed: a scalar-memory round trip)
		const int par = (layer + 1) & 1;
		// (this region published the same layer a moment ago: its count is the neighbours')
		const unsigned tm = LEAKY ? (pubCount[par] & 0xffffu) : epochMask(par);
		const unsigned soff = par ? kParityBytes : 0u;
		unsigned pending = sweepValid;
		// sweep: all loads of a pass in flight together, sc1 (never a stale L1/L2 line);
		// a slot is accepted only when all four dwords carry the expected epoch
		u32x4 hvFirst[NS];
		auto loadPassTo = [&](u32x4(&hv)[NS]) __attribute__((always_inline)) {
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				hv[it] = __builtin_amdgcn_raw_buffer_load_b128(mailRsrc, sweepSrc[it], soff, kSc1);
			}
		};
		auto checkPassOf = [&](const u32x4(&hv)[NS]) __attribute__((always_inline)) {
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				if constexpr (LEAKY) {
					typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
					const u32x4 tg = hv[it] >> 16;
					const bool ok = tg[0] == tm && tg[1] == tm && tg[2] == tm && tg[3] == tm;
					if ((pending >> it & 1u) && ok) {
						*reinterpret_cast<u32x2 *>(smem + off + sweepLds[it]) =
						    u32x2{(hv[it][0] & 0xffffu) | (hv[it][1] << 16), (hv[it][2] & 0xffffu) | (hv[it][3] << 16)};
						pending &= ~(1u << it);
					}
				} else {
					const u32x4 tg = hv[it] & 0x80008000u;
					const bool ok = tg[0] == tm && tg[1] == tm && tg[2] == tm && tg[3] == tm;
					if ((pending >> it & 1u) && ok) {
						*reinterpret_cast<u32x4 *>(smem + off + sweepLds[it]) = hv[it] & 0x7fff7fffu;
						pending &= ~(1u << it);
					}
				}
			}
		};
		// The first pass is straight-line code (not the loop's first iteration): what the caller
		// runs behind its loads -- a pre-run unit, 32 accumulator registers -- is then defined on
		// one path only and needs no copies where the paths would join.
		// the branch-free check: every lane writes, one accumulator, one vote (see kLean)
		auto checkPassFast = [&](const u32x4(&hv)[NS]) __attribute__((always_inline)) {
			unsigned bad = 0;
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				if constexpr (LEAKY) {
					// dword = value16 | epoch16 << 16: XOR with the expected epoch clears the upper half of the expected write
					typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
					const u32x4 x = hv[it] ^ (tm << 16);
					bad |= x[0] | x[1] | x[2] | x[3];
					*reinterpret_cast<u32x2 *>(smem + off + sweepLds[it]) = u32x2{(x[0] & 0xffffu) | (x[1] << 16), (x[2] & 0xffffu) | (x[3] << 16)};
				} else {
					// the expected write XOR the expected epoch IS the payload (post-ReLU values: sign bits clear); anything
					// else leaves epoch bits standing, is written all the same and overwritten by the pass that succeeds
					const u32x4 x = hv[it] ^ tm;
					bad |= x[0] | x[1] | x[2] | x[3];
					*reinterpret_cast<u32x4 *>(smem + off + sweepLds[it]) = x;
				}
			}
			pending = bad & (LEAKY ? 0xffff0000u : 0x80008000u);  // != 0: some slot of this lane still holds the previous write
		};
		auto loadPass = [&]() __attribute__((always_inline)) { loadPassTo(hvFirst); };
		auto checkPass = [&]() __attribute__((always_inline)) {
			if constexpr (kLean) checkPassFast(hvFirst);
			else checkPassOf(hvFirst);
		};
		loadPass();
		__builtin_amdgcn_sched_barrier(0);
		behindFirstPass();  // (work the caller wants done while the loads travel; vmcnt is in-order)
		__builtin_amdgcn_sched_barrier(0);
#ifdef JU_TOWER_FILLPROF  // developer diagnostic (variant 4): where inside the fill the time goes -- slots 1, 3, 4 of the profile
		const u64 fp0 = stamp();
#endif
		if constexpr (FAST) {
			// A second pass in flight before the first is checked: when the first came too early for a slot
			// (0.3 times per layer), its data is one pre-run unit behind instead of a whole round trip
			// (-1 % per tower; twice the sweep's read traffic, 5 MB more per layer over the chip).
			u32x4 hvSecond[NS];
			loadPassTo(hvSecond);
			__builtin_amdgcn_sched_barrier(0);
			if constexpr (kLean) {
				checkPassFast(hvFirst);
				if (__any(pending != 0)) {
					checkPassFast(hvSecond);
					if constexpr (VARIANT == 4) extraPasses += 1;
				}
			} else {
				checkPassOf(hvFirst);
				if (__any(pending != 0)) {
					checkPassOf(hvSecond);
					if constexpr (VARIANT == 4) extraPasses += 1;
				}
			}
		} else {
			checkPass();
		}
#ifdef JU_TOWER_FILLPROF
		const u64 fp1 = stamp();
#endif
		while (__any(pending != 0)) {
			const u64 now = __builtin_amdgcn_s_memrealtime();
			if (t0 == 0) t0 = now;
			if (now - t0 > kResTimeoutTicks) {
				if (pending != 0) {
					*failFlag = 1;
					__hip_atomic_store((gu32 *)p.error, 0x700u + (unsigned)layer, __ATOMIC_RELAXED,
					    __HIP_MEMORY_SCOPE_SYSTEM);
				}
				break;
			}
			__builtin_amdgcn_s_sleep(1);
			loadPass();
			checkPass();
			if constexpr (VARIANT == 4) extraPasses += 1;
		}
#ifdef JU_TOWER_FILLPROF
		const u64 fp2 = stamp();
#endif
		__syncthreads();
#ifdef JU_TOWER_FILLPROF
		const u64 fp3 = stamp();
		prof[1] += fp1 - fp0;  // the first check (waits for the first pass's loads)
		prof[3] += fp2 - fp1;  // the retry loop
		prof[4] += fp3 - fp2;  // the barrier behind the fill
#endif
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
	auto layerStep = [&](auto resTag, auto parTag, const int i) __attribute__((always_inline)) -> bool {
		constexpr int PAR = decltype(parTag)::value;
		using InT = std::integral_constant<int, PAR ? kResOffB : kResOffA>;
		using OutT = std::integral_constant<int, PAR ? kResOffA : kResOffB>;
		const bool more = i + 1 < L;
		const u64 t0 = stamp();
		finishLayer(resTag, InT{}, OutT{}, i, more && streamsInUnit);
		if constexpr (VARIANT == 5) {
			// per-layer maximum over the frame -> debug[i] (non-negative floats order like
			// their bit patterns, so an integer atomic max does it)
			float m = calibMax;
#pragma unroll
			for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
			if (lane == 0) atomicMax(reinterpret_cast<unsigned *>(p.debug) + i, __float_as_uint(m));
			calibMax = 0.f;
		}
		const u64 t1 = stamp();
		__syncthreads();
		const u64 t2 = stamp();
		if (more && xchg) publish(OutT{}, i);
		// the pre-run's first fragments (interior rows of this layer's output) travel
		// while the weight stream is issued
		const bool primePre = more && nPre > 0 && !kNoMfma;
		if (primePre) {
			const unsigned na = ldsBase + OutT::value + (2 * preFirst) * kResRowBytes;
#pragma unroll
			for (int j = 0; j < 4; ++j) issue(na, kOrder[0], 0, j);
			__builtin_amdgcn_sched_barrier(0);
		}
		const u64 t3 = stamp();
		// The weight stream goes out BETWEEN the publish and the sweep: its ~2.2k cycles of
		// issue (four waves push 144 KB through the CU's 64 B/clk address path) pass while
		// the neighbours' write-through stores travel (vmcnt is in-order: the halo loads
		// return behind the weights, which is when they would have been valid anyway).
		// This layer's weight registers are free: the next layer's set goes into them.  Measured: sweep first / between the two sets / after
		// both = 452 / 438 / 430 us per tower; an extra 256-1024 cycles of sleep before the
		// sweep -3 ... +10 us.  (Round 1 had the stream after the sweep: ~4k + 2.2k cycles in
		// series; streaming from inside the K loop was no faster.)
		if (more && !streamsInUnit) {  // (a wave without a whole unit: tiny regions)
			loadBias(i + 1);
			loadMidWeights(i + 1);
			loadSideWeights(i + 1);
		}
		const u64 t4 = stamp();
		if (more && xchg) {
			// the next layer's halo-independent steps run behind the first pass's loads
			using Z = std::integral_constant<int, 0>;
			using Split = std::integral_constant<int, kPreBefore>;
			using N = std::integral_constant<int, kPreRun>;
			preRun(OutT{}, InT{}, Z{}, Split{}, i + 1, primePre);
			const bool okFill = fillHalo(OutT{}, i, [&]() __attribute__((always_inline)) {
				preRun(OutT{}, InT{}, Split{}, N{}, i + 1, primePre);
			});
			if (!okFill) return false;
		} else {
			if (more) preRun(OutT{}, InT{}, std::integral_constant<int, 0>{}, std::integral_constant<int, kPreRun>{}, i + 1, primePre);
			__syncthreads();
		}
		const u64 t5 = stamp();
		prof[0] += t5 - t4;
#ifndef JU_TOWER_FILLPROF
		prof[1] += (VARIANT == 4 && FAST) ? extraPasses * 1000 : t4 - t3;  // (fast schedule: sweep passes beyond the first, x 1000)
#endif
		extraPasses = 0;
		prof[2] += t1 - t0;
#if !defined(JU_TOWER_SEGPROF) && !defined(JU_TOWER_FILLPROF)
		prof[3] += t2 - t1;
		prof[4] += t3 - t2;
#endif
		return true;
	};
	using No = std::false_type;
	using Yes = std::true_type;
	using P0 = std::integral_constant<int, 0>;
	using P1 = std::integral_constant<int, 1>;
	preRun(std::integral_constant<int, kResOffA>{}, std::integral_constant<int, kResOffB>{}, std::integral_constant<int, 0>{},
	    std::integral_constant<int, kPreRun>{}, 0, false);
	if (HEAD) {  // conv_1, then (conv1, conv2+skip) pairs: L is odd
		if (!layerStep(No{}, P0{}, 0)) return;
		for (int i = 1; i + 1 < L; i += 2) {
			if (!layerStep(No{}, P1{}, i)) return;
			if (!layerStep(Yes{}, P0{}, i + 1)) return;
		}
	} else {  // (conv1, conv2+skip) pairs: L is even
		for (int i = 0; i + 1 < L; i += 2) {
			if (!layerStep(No{}, P0{}, i)) return;
			if (!layerStep(Yes{}, P1{}, i + 1)) return;
		}
	}
	const int finalOff = (L & 1) ? kResOffB : kResOffA;
	if (tid == 0 && xchg) {  // the next launch continues the slot epochs (plain stores: kernel boundary)
		p.count[region * 2] = pubCount[0];
		p.count[region * 2 + 1] = pubCount[1];
	}
	if constexpr (VARIANT == 4) {
		if (lane == 0 && p.debug != nullptr) {
			for (int k = 0; k < 8; ++k) p.debug[(region * 4 + wave) * 8 + k] = prof[k];
		}
	}

	if constexpr (TAIL) {
		// ---- fused generator tail on the LDS-resident last layer (models.py:552-593):
		// the buffer the last layer did NOT write is free now: 16 KiB of convT1 weights,
		// then per wave 8 KiB mid pixels + 4 KiB state staging + 2 KiB u8 staging.
		// Saves the trunk's 16.6 MB write and re-read and one launch.
		const int scratchOff = finalOff == kResOffA ? kResOffB : kResOffA;
		static_assert(kTailLdsW1 + 4 * (kTailLdsMid / 4 + kTailLdsRow + kTailLdsU8 / 4) <= kResBufBytes,
		    "tail scratch must fit one activation buffer");
		unsigned char *smW = smem + scratchOff;
		unsigned char *waveBase = smW + kTailLdsW1 + wave * (kTailLdsMid / 4 + kTailLdsRow + kTailLdsU8 / 4);
		unsigned char *smMid = waveBase;
		unsigned char *smState = waveBase + kTailLdsMid / 4;
		unsigned char *smU8 = smState + kTailLdsRow;
		{
			const uint4 *src = reinterpret_cast<const uint4 *>(p.tailW1);
			uint4 *dst = reinterpret_cast<uint4 *>(smW);
#pragma unroll
			for (int k = 0; k < kTailLdsW1 / 16 / 256; ++k) dst[tid + k * 256] = src[tid + k * 256];
		}
		Vec8<T> a2[2];
		a2[0] = reinterpret_cast<const Vec8<T> *>(p.tailW2)[lane];
		a2[1] = reinterpret_cast<const Vec8<T> *>(p.tailW2)[64 + lane];
		const float b2v[3] = {p.tailB2[0], p.tailB2[1], p.tailB2[2]};
		// (the fused tail is instantiated for ReLU models only: launchResidentTower refuses TAIL + LEAKY)
		const TailRowArgs args{p.tailB1, p.frame, p.frameStride, p.state, p.outU8, p.outStride, p.H, p.W, -1.0f};
		const float bright = brightnessOf(p.sums, 1.0f / static_cast<float>(p.H * p.W));
		__syncthreads();
		for (int r = wave; r < rhv; r += 4) {
			const unsigned char *rowBase = smem + finalOff + (r + 1) * kResRowBytes;
			const auto fetchB = [&](int ks) {
				const int cc = px + 1;
				return *reinterpret_cast<const Vec8<T> *>(
				    rowBase + cc * 128 + (((ks * 2 + hh) ^ ((cc >> 1) & 7)) << 4));
			};
			tailRow<T>(fetchB, smW, smMid, smState, smU8, a2, b2v, args, bright, x0, y0 + r, lane);
		}
		return;
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

template <typename T, int VARIANT, bool HEAD, bool TAIL = false, bool LEAKY = false, bool FAST = false>
void launchResidentT(const ResidentParams &p, hipStream_t stream) {
	auto kern = tower_resident_kernel<T, VARIANT, HEAD, TAIL, LEAKY, FAST>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kResLds, &ldsDone, "resident tower");
	// (g_ResidentFault > 0, tests only: some regions are never computed, their neighbours'
	// bounded waits expire and the error path runs)
	const int grid = p.GX * p.GY - (g_ResidentFault < p.GX * p.GY ? g_ResidentFault : 0);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kResLds, stream, p);
	hipCheckLaunch("tower_resident");
}

}  // namespace

void launchConvTower(DType dt, const ConvParams &p, hipStream_t stream) {
	const int pitch = towerPitch(p.W);
	const bool fits = p.taps == 9 && p.cin == 64 && p.cout == 64 && !p.outHead && p.nb == 2 &&
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

bool residentTowerM16() { return JU_TOWER_M16 != 0; }
void setTowerVariant(int v) { g_TowerVariant = v; }
// the fast schedule where the geometry allows it (default); 0 (JU_TOWER_FAST=0 or the tests' switch): the
// general schedule everywhere
static std::atomic<int> g_TowerFast{[] { const char *e = devSwitch(Dev::TowerFast); return (e != nullptr && e[0] == '0') ? 0 : 1; }()};
void setResidentTowerFast(int on) { g_TowerFast = on ? 1 : 0; }
bool residentTowerFast() { return g_TowerFast.load() != 0; }
int towerVariant() { return g_TowerVariant; }
void setResidentFault(int n) { g_ResidentFault = n; }
int residentFaultForTests() { return g_ResidentFault; }

bool residentTowerGeometry(int H, int W, int numCUs, int *GX, int *GY, int *RH) {
	const int gx = (W + kResRW - 1) / kResRW;
	const int gy = (H + kResMaxRH - 1) / kResMaxRH;
	if (gx * gy > numCUs) return false;
	*GX = gx;
	*GY = gy;
	*RH = (H + gy - 1) / gy;  // <= kResMaxRH, balances the last row of regions
	return true;
}

bool residentTowerFastGeometry(int H, int W, int GX, int GY, int RH) {
	const int lastH = H - (GY - 1) * RH, lastW = W - (GX - 1) * kResRW;
	return residentFastShape(RH, kResRW) && residentFastShape(lastH, kResRW) && residentFastShape(RH, lastW) &&
	       residentFastShape(lastH, lastW);
}

std::size_t residentMailboxBytes(int GX, int GY, bool leaky) {
	return static_cast<std::size_t>(GX) * GY * 2 * kResMailSlots * 16 * (leaky ? 2 : 1);
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
	p.count = q.counters;
	p.debug = static_cast<unsigned long long *>(q.debug);
	p.error = q.error;
	p.H = q.H;
	p.W = q.W;
	p.pitch = towerPitch(q.W);
	p.GX = q.GX;
	p.GY = q.GY;
	p.RH = q.RH;
	p.nLayers = q.nLayers;
	p.slope = q.slope;
	p.tailW1 = q.tailW1;
	p.tailB1 = q.tailB1;
	p.tailW2 = q.tailW2;
	p.tailB2 = q.tailB2;
	p.frame = q.frame;
	p.frameStride = q.frameStride;
	p.state = q.state;
	p.outU8 = q.outU8;
	p.outStride = q.outStride;
	p.sums = q.sums;
	if ((p.nLayers & 1) != (p.hasHead ? 1 : 0)) {
		throw std::invalid_argument("resident tower: layer count must be 2*blocks (+1 with a head)");
	}
	// the fast schedule where every region has its shape (JU_TOWER_FAST=0: the general one, for A/B runs)
	const bool fast = residentTowerFast() && residentTowerFastGeometry(p.H, p.W, p.GX, p.GY, p.RH);
#ifdef JU_TOWER_DEV  // developer builds: the bf16 ReLU instantiations only (compile time)
	if (q.leaky || dt == kF16 || !p.hasHead) throw std::invalid_argument("resident tower: developer build");
	if (p.tailW1 != nullptr) {
		if (fast) launchResidentT<bf16, 0, true, true, false, true>(p, stream);
		else launchResidentT<bf16, 0, true, true>(p, stream);
	} else if (g_TowerVariant == 8) launchResidentT<bf16, 8, true>(p, stream);
	else if (g_TowerVariant == 4) {
		if (fast) launchResidentT<bf16, 4, true, false, false, true>(p, stream);
		else launchResidentT<bf16, 4, true>(p, stream);
	} else if (fast) launchResidentT<bf16, 0, true, false, false, true>(p, stream);
	else launchResidentT<bf16, 0, true>(p, stream);
	return;
#else
	if (q.leaky) {
		// `activation: lrelu` models: own instantiations (epoch beside the values, f32 LeakyReLU);
		// variant 8 = the plain schedule (tests), 5 = calibration maxima
		if (!(q.slope >= 0.0f && q.slope <= 1.0f)) throw std::invalid_argument("resident tower: negative slope outside [0, 1]");
		if (p.tailW1 != nullptr) throw std::invalid_argument("resident tower: the fused tail is built for ReLU models");
		if (!p.hasHead) {
			if (fast) {
				if (dt == kF16) launchResidentT<f16, 0, false, false, true, true>(p, stream);
				else launchResidentT<bf16, 0, false, false, true, true>(p, stream);
			} else {
				if (dt == kF16) launchResidentT<f16, 0, false, false, true>(p, stream);
				else launchResidentT<bf16, 0, false, false, true>(p, stream);
			}
		} else if (g_TowerVariant == 8) {
			if (dt == kF16) launchResidentT<f16, 8, true, false, true>(p, stream);
			else launchResidentT<bf16, 8, true, false, true>(p, stream);
		} else if (g_TowerVariant == 5 && dt == kBF16) {
			(void)hipMemsetAsync(p.debug, 0, static_cast<std::size_t>(p.nLayers) * sizeof(unsigned), stream);
			launchResidentT<bf16, 5, true, false, true>(p, stream);
		} else if (fast && g_TowerVariant == 0) {
			if (dt == kF16) launchResidentT<f16, 0, true, false, true, true>(p, stream);
			else launchResidentT<bf16, 0, true, false, true, true>(p, stream);
		} else {
			if (dt == kF16) launchResidentT<f16, 0, true, false, true>(p, stream);
			else launchResidentT<bf16, 0, true, false, true>(p, stream);
		}
		return;
	}
	if (!p.hasHead) {
		if (fast) {
			if (dt == kF16) launchResidentT<f16, 0, false, false, false, true>(p, stream);
			else launchResidentT<bf16, 0, false, false, false, true>(p, stream);
		} else {
			if (dt == kF16) launchResidentT<f16, 0, false>(p, stream);
			else launchResidentT<bf16, 0, false>(p, stream);
		}
		return;
	}
	if (p.tailW1 != nullptr) {  // fused tail: product kernel only (the ablation variants keep the split)
		if (fast) {
			if (dt == kF16) launchResidentT<f16, 0, true, true, false, true>(p, stream);
			else launchResidentT<bf16, 0, true, true, false, true>(p, stream);
		} else {
			if (dt == kF16) launchResidentT<f16, 0, true, true>(p, stream);
			else launchResidentT<bf16, 0, true, true>(p, stream);
		}
		return;
	}
	if (dt == kF16) {
		if (g_TowerVariant == 8) launchResidentT<f16, 8, true>(p, stream);
		else if (fast) launchResidentT<f16, 0, true, false, false, true>(p, stream);
		else launchResidentT<f16, 0, true>(p, stream);
		return;
	}
	switch (g_TowerVariant) {
	case 8: launchResidentT<bf16, 8, true>(p, stream); break;
	case 1: launchResidentT<bf16, 1, true>(p, stream); break;
	case 2: launchResidentT<bf16, 2, true>(p, stream); break;
	case 3: launchResidentT<bf16, 3, true>(p, stream); break;
	case 4:
		if (fast) launchResidentT<bf16, 4, true, false, false, true>(p, stream);
		else launchResidentT<bf16, 4, true>(p, stream);
		break;
	case 5:  // calibration: per-layer output maxima of this frame in debug[0 .. nLayers)
		(void)hipMemsetAsync(p.debug, 0, static_cast<std::size_t>(p.nLayers) * sizeof(unsigned), stream);
		launchResidentT<bf16, 5, true>(p, stream);
		break;
	default:
		if (fast) launchResidentT<bf16, 0, true, false, false, true>(p, stream);
		else launchResidentT<bf16, 0, true>(p, stream);
		break;
	}
#endif
}


}  // namespace ju
