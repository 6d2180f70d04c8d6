// gfx950 (CDNA4, MI355X): a 64-filter residual block (reference scripts/training/models.py:193-254) as ONE
// persistent launch -- the generator's (and a 64-filter flow-resnet's) tower wherever tower_resident_kernel
// cannot run (more 32 x 16 regions than CUs, e.g. 640 x 448; after a resident fallback).
//   res_block_kernel       every activation; the plain schedule
//   res_block_pipe_kernel  ReLU blocks: a row pair's epilogue behind the next pair's MFMAs
// (moved out of flow_kernels.hip in round 4: these are generator kernels)
#include "flow_block_common.h"

namespace ju {

namespace {


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
	// res_block_pipe_kernel: bytes of `in` / `out` from the pointer to the end of image row H - 1 (the range of its
	// buffer descriptors; everything behind it is zero padding or never touched)
	unsigned inBytes, outBytes;
};
// The pipelined kernel addresses its tensors with 32-bit byte offsets (descriptor + scalar + lane offset) and gets the
// rows above the image from offsets that wrap below zero, i.e. land just under 4 GiB and out of the descriptor's
// range: a tensor may reach 4 GiB minus the two rows of the widest frame (2 x 8226 px x 128 B) and a margin.
constexpr unsigned long long kRpMaxTensorBytes = 0xFFC00000ull;

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
	// 32-bit register, the tile / row / fragment part is scalar (a tensor is < 4 GiB: the launcher checks).
	// (Until round 4 the descriptors said 2 GiB - 16 and no launcher checked: beyond 16.7 M pixels -- 4096 x 4096 -- the
	// rows past 2 GiB read as zeros and their stores were dropped; found by tests/test_gpu_presets.py's crop property.)
	const __amdgpu_buffer_rsrc_t rsrcWa = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w1), 0, 2 * 36 * 1024, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcWb = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.w2), 0, 2 * 36 * 1024, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcIn = __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p.in), 0, p.inBytes, 0x00020000);
	const __amdgpu_buffer_rsrc_t rsrcOut = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, p.outBytes, 0x00020000);
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
static std::atomic<int> g_ResBlockPlain{[] { const char *e = devSwitch(Dev::ResBlock); return (e != nullptr && std::string(e) == "plain") ? 1 : 0; }()};
}  // namespace
void setResBlockPlain(int on) { g_ResBlockPlain = on ? 1 : 0; }
bool resBlockPlain() { return g_ResBlockPlain.load() != 0; }
namespace {

template <typename T>
void launchResBlockT(const FlowBlockLaunch &q, hipStream_t stream) {
	// the pipelined form (epilogues behind the next pair's MFMAs); JU_RES_BLOCK=plain: the plain kernel (tests, A/B).
	// A slope outside [0, 1] (no model the container accepts has one) also takes the plain kernel.
	const int inPitch = q.inPitch ? q.inPitch : q.W, outPitch = q.outPitch ? q.outPitch : q.W;
	const unsigned long long inBytes = 128ull * q.H * inPitch, outBytes = 128ull * q.H * outPitch;
	// (tensors of 4 GiB and more: the plain kernel, which addresses with 64-bit pointers)
	const bool pipe = !resBlockPlain() && q.act1 == 1 && q.act2 == 1 && ablationSkipBits() == 0 &&  // (ReLU blocks)
	                  inBytes <= kRpMaxTensorBytes && outBytes <= kRpMaxTensorBytes;
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
	p.inPitch = inPitch;
	p.outPitch = outPitch;
	p.inBytes = static_cast<unsigned>(inBytes < kRpMaxTensorBytes ? inBytes : kRpMaxTensorBytes);
	p.outBytes = static_cast<unsigned>(outBytes < kRpMaxTensorBytes ? outBytes : kRpMaxTensorBytes);
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

}  // namespace

void launchResBlockPersistent(DType dt, const FlowBlockLaunch &q, hipStream_t stream) {
	if (dt == kF16) launchResBlockT<f16>(q, stream);
	else launchResBlockT<bf16>(q, stream);
}

}  // namespace ju
