// gfx950 (CDNA4, MI355X): the 8-bit (OCP e4m3) residual tower as ONE launch
// (reference scripts/training/models.py:193-254, 538-550; quantisation scheme: fp8.h).
//
//  * tower8_resident_kernel   all 2 B convolutions of the B residual blocks, activations
//                             resident in LDS, halo exchange through the global mailbox
//
// The per-block form (res_block_fp8_kernel) moves the 16-bit residual stream and the
// e4m3 copies through memory once per block: 50 MB per block at 480 x 270, i.e. the memory
// rate (16 us per block), although the matrix instructions of a block are 5 us.  Here each
// workgroup (one per CU) owns a region of 32 x RH (<= 16) pixels for the whole tower, as in
// tower_resident_kernel, and keeps three tiles in LDS:
//     S   the 16-bit residual stream            (RH + 2) x 34 px x 128 B   (pointwise: no halo used)
//     X8  e4m3(S * 2^ex), conv A's input         (RH + 2) x 34 px x 64 B    with a one-pixel halo ring
//     T8  e4m3(relu(conv A) * 2^et)              the same
//   conv A:  X8 -> ReLU -> e4m3 -> T8 interior;   exchange T8's edge ring
//   conv B:  T8 -> + S -> ReLU -> S (in place) and X8 interior (next block's scale);
//            exchange X8's edge ring
// on v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64: one tap of the 64-channel input per
// instruction; both power-of-two scales are undone by its E8M0 scale operands).  A wave =
// (cout half, row-pair parity) keeps its 9 A fragments per layer in registers (72 VGPRs,
// double-buffered: the next layer's stream in behind the current layer's instructions).
// The halo exchange is tower_resident_kernel's (self-validating 16-byte slots, one
// write-through store each, sc1 loads, epochs from persistent publish counts) on records
// of half the size: e4m3 values of post-ReLU tensors are non-negative, so every BYTE's
// sign bit is free and each dword carries the 2-bit epoch twice.
// Per output element the instruction sequence is that of conv_tower_fp8_kernel /
// res_block_fp8_kernel, so all three paths produce the same bytes.
#include "kernel_common.h"

namespace ju {

namespace {

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) unsigned gu32;

constexpr int kT8RW = 32, kT8MaxRH = 16, kT8Pitch = 34;
constexpr int kT8SRow = kT8Pitch * 128;                 // stream row, bytes
constexpr int kT8QRow = kT8Pitch * 64;                  // e4m3 row, bytes
constexpr int kT8OffS = 0;
constexpr int kT8OffX = (kT8MaxRH + 2) * kT8SRow;       // 78336
// (16 spare bytes behind each e4m3 buffer, at the same offset from its base: where the branch-free sweep parks the
// lanes that have no halo cell to fill -- tower_kernels.hip, kLean)
constexpr int kT8DummyOff = (kT8MaxRH + 2) * kT8QRow;
constexpr int kT8OffT = kT8OffX + (kT8MaxRH + 2) * kT8QRow + 16;
constexpr int kT8OffMisc = kT8OffT + (kT8MaxRH + 2) * kT8QRow + 16;  // 156704
// misc: fail flag (64 B), per-layer bias x 2 slots (512 B), per-layer weight scale codes x 2 slots (512 B)
constexpr int kT8Lds = kT8OffMisc + 64 + 512 + 512;
constexpr int kT8MailSlots = 4 * 32 * 4;                // 16-byte slots per region per parity
constexpr unsigned long long kT8TimeoutTicks = 20000000ull;  // 0.2 s of s_memrealtime

__device__ __forceinline__ void t8Glds16(const void *g, void *l) {
	__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
	    (__attribute__((address_space(3))) void *)l, 16, 0, 0);
}

// e4m3 of relu(x) * mul for four values, saturating (the hardware conversion returns NaN
// above 448): one multiply and one v_med3_f32 (clamp to [0, 448]) per value.  The same
// value as min(max(x, 0) * mul, 448) of the other 8-bit kernels: mul is a positive power
// of two.  (371 -> 361 us per tower against separate max / min.)
// (the multiplies as two packed v_pk_mul_f32: same values, half the instructions)
// LEAKY (`activation: lrelu`): the caller has applied the LeakyReLU; clamp on both sides.
typedef float t8f32x2 __attribute__((ext_vector_type(2)));
template <bool LEAKY = false>
__device__ __forceinline__ int t8Quantize4(float a, float b, float c, float d, float mul) {
	const t8f32x2 m = {mul, mul};
	const t8f32x2 ab = t8f32x2{a, b} * m, cd = t8f32x2{c, d} * m;
	constexpr float lo = LEAKY ? -448.0f : 0.0f;
	a = __builtin_amdgcn_fmed3f(ab[0], lo, 448.0f);
	b = __builtin_amdgcn_fmed3f(ab[1], lo, 448.0f);
	c = __builtin_amdgcn_fmed3f(cd[0], lo, 448.0f);
	d = __builtin_amdgcn_fmed3f(cd[1], lo, 448.0f);
	int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
	return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

// the same value with single multiplies (for the shadow of an MFMA, where packed f32 is not hidden)
template <bool LEAKY = false>
__device__ __forceinline__ int t8Quantize4s(float a, float b, float c, float d, float mul) {
	constexpr float lo = LEAKY ? -448.0f : 0.0f;
	a = __builtin_amdgcn_fmed3f(a * mul, lo, 448.0f);
	b = __builtin_amdgcn_fmed3f(b * mul, lo, 448.0f);
	c = __builtin_amdgcn_fmed3f(c * mul, lo, 448.0f);
	d = __builtin_amdgcn_fmed3f(d * mul, lo, 448.0f);
	int r = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
	return __builtin_amdgcn_cvt_pk_fp8_f32(c, d, r, true);
}

struct Tower8Params {
	const void *in;            // 16-bit stream (generator conv_1's output), tower layout, allocation start
	void *out;                 // 16-bit stream after the last block, tower layout, allocation start
	const unsigned char *weights;  // nLayers x 36864 B (packFp8TowerWeights)
	const int *scaleA;         // nLayers x 64 E8M0 codes
	const float *bias;         // nLayers x 64
	const int *scaleB;         // nLayers: E8M0 code of each convolution's input tensor scale
	const float *outMul;       // nLayers: 2^e of the e4m3 tensor each convolution's output feeds
	uint4 *mail;               // [regions][2][kT8MailSlots]
	unsigned *count;           // [regions][2] publishes so far (persistent)
	unsigned *error;
	int H, W, pitch;
	int GX, GY, RH;
	int nLayers;               // 2 x blocks
	float slope;               // LEAKY instantiations: LeakyReLU negative slope
	unsigned long long *debug; // JU_T8_PROF developer builds: [regions][4][8] cycle sums
	int fault;                 // test hook: workgroups launched short (they never publish)
	int skip;                  // timing ablation (JU_FB_SKIP, developer only): 1 exchange, 2 K loop, 4 epilogue
};

// LEAKY: `activation: lrelu` generators (models.py:24-27): LeakyReLU in f32, two-sided e4m3
// clamp, and -- e4m3 bytes of such tensors having no free sign bit -- halo slots that carry a
// 16-bit epoch beside every two value bytes (twice the slots; tower_resident_kernel's LEAKY form)
template <typename T, bool LEAKY = false>
__global__ __launch_bounds__(256, 1) void tower8_resident_kernel(Tower8Params p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, px = lane & 31, hh = lane >> 5;
	const int ch = wave & 1, rp = wave >> 1;
	int region;
	{  // XCD-contiguous regions (tower_resident_kernel)
		const int n = gridDim.x, x = blockIdx.x & 7;
		int start = 0;
		for (int y = 0; y < x; ++y) start += (n - y + 7) >> 3;
		region = start + (blockIdx.x >> 3);
	}
	const int gxr = region % p.GX, gyr = region / p.GX;
	const int x0 = gxr * kT8RW, y0 = gyr * p.RH;
	const int rwv = min(kT8RW, p.W - x0), rhv = min(p.RH, p.H - y0);
	volatile int *failFlag = reinterpret_cast<volatile int *>(smem + kT8OffMisc);
	float *ldsBias = reinterpret_cast<float *>(smem + kT8OffMisc + 64);
	int *ldsScale = reinterpret_cast<int *>(smem + kT8OffMisc + 64 + 512);
	unsigned pubCount[2] = {p.count[region * 2], p.count[region * 2 + 1]};

	// ---- zero everything (borders, out-of-image area, overrun pads stay zero) ----
	for (int i = tid; i < kT8OffMisc / 16; i += 256) reinterpret_cast<uint4 *>(smem)[i] = make_uint4(0, 0, 0, 0);
	if (tid == 0) *failFlag = 0;
	__syncthreads();

	// ---- the stream: region + halo from the complete global tensor (pixels outside the image stay zero) ----
	{
		const T *in = static_cast<const T *>(p.in);
		const int nPix = (rhv + 2) * kT8Pitch;
		const int nInstr = (nPix + 7) / 8;  // 8 pixels (1 KiB) per wave-instruction
		for (int i = wave; i < nInstr; i += 4) {
			const int q = i * 8 + (lane >> 3);
			const int rr = q / kT8Pitch, cc = q - rr * kT8Pitch;
			const int c = (lane & 7) ^ ((cc >> 1) & 7);
			const int gy = y0 - 1 + rr, gx = x0 - 1 + cc;
			if (q < nPix && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
				t8Glds16(in + ((size_t)(gy + 1) * p.pitch + gx + 1) * 64 + c * 8, smem + kT8OffS + i * 1024);
			}
		}
	}
	// ---- weights of layer 0, bias / scale codes of layer 0 ----
	typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
	const __amdgpu_buffer_rsrc_t wRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    const_cast<unsigned char *>(p.weights), 0, p.nLayers * 36864, 0x00020000);
	const unsigned wLane = static_cast<unsigned>((ch * 64 + lane) * 32);
	auto loadWeights = [&](int layer, i32x8(&w)[9]) {
#pragma unroll
		for (int t = 0; t < 9; ++t) {
			const u32x4w lo = __builtin_amdgcn_raw_buffer_load_b128(wRsrc, wLane, layer * 36864 + t * 4096, 0);
			const u32x4w hi = __builtin_amdgcn_raw_buffer_load_b128(wRsrc, wLane, layer * 36864 + t * 4096 + 16, 0);
			w[t] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
		}
	};
	i32x8 w0[9], w1[9];
	loadWeights(0, w0);
	if (p.nLayers > 1) loadWeights(1, w1);
	if (wave == 0) {
		ldsBias[lane] = p.bias[lane];
		ldsScale[lane] = p.scaleA[lane];
		if (p.nLayers > 1) {
			ldsBias[64 + lane] = p.bias[64 + lane];
			ldsScale[64 + lane] = p.scaleA[64 + lane];
		}
	}
	asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	// ---- X8 = e4m3(relu(S) * 2^e0) over the whole tile incl. halo (zero stays zero) ----
	{
		const float mul0 = p.outMul[p.nLayers];  // scale of the tower's input tensor (slot nLayers)
		const int nChunk = (rhv + 2) * kT8Pitch * 4;  // 16-byte e4m3 chunks = 16 channels each
		for (int e = tid; e < nChunk; e += 256) {
			const int c = e & 3, q = e >> 2;
			const int rr = q / kT8Pitch, cc = q - rr * kT8Pitch;
			const unsigned char *src = smem + kT8OffS + rr * kT8SRow + cc * 128;
			const unsigned sw = (cc >> 1) & 7;
			const Vec8<T> a = *reinterpret_cast<const Vec8<T> *>(src + (((2 * c) ^ sw) << 4));
			const Vec8<T> b = *reinterpret_cast<const Vec8<T> *>(src + (((2 * c + 1) ^ sw) << 4));
			float v[16];
#pragma unroll
			for (int k = 0; k < 8; ++k) {
				v[k] = LEAKY ? static_cast<float>(a[k]) : fmaxf(static_cast<float>(a[k]), 0.0f);
				v[8 + k] = LEAKY ? static_cast<float>(b[k]) : fmaxf(static_cast<float>(b[k]), 0.0f);
			}
			i32x4 o;
#pragma unroll
			for (int k = 0; k < 4; ++k) o[k] = t8Quantize4<LEAKY>(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3], mul0);
			*reinterpret_cast<i32x4 *>(smem + kT8OffX + rr * kT8QRow + cc * 64 + ((c ^ ((cc >> 2) & 3)) << 4)) = o;
		}
	}
	__syncthreads();

	// JU_T8_PROF (developer builds: tools/dev_kernel_lib.sh tower8_kernels.hip x -DJU_T8_PROF, tools/tower8_phases.py): s_memtime
	// sums per wave -- 0 sweep, 1 weight issue, 2 the layer's units, 3 barrier, 4 bias + publish, 5 K loops, 6 writes between
	// the units + the exposed epilogue, 7 unused
	unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
	auto stamp = [&]() __attribute__((always_inline)) -> unsigned long long {
#ifdef JU_T8_PROF
		__builtin_amdgcn_sched_barrier(0);
		unsigned long long t;
		asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
		__builtin_amdgcn_sched_barrier(0);
		return t;
#else
		return 0;
#endif
	};
	// B fragments of one horizontal tap of a row pair: 4 input rows x 32 bytes per lane
	auto loadFrags = [&](int off, int unit, int dx, i32x8(&fb)[4]) {
		const int x = px + dx;
		const int sw = (x >> 2) & 3;
		const unsigned char *col = smem + off + (2 * unit) * kT8QRow + x * 64;
		const unsigned char *colLo = col + (((2 * hh) ^ sw) << 4);
		const unsigned char *colHi = col + (((2 * hh + 1) ^ sw) << 4);
#pragma unroll
		for (int r = 0; r < 4; ++r) {
			const i32x4 lo = *reinterpret_cast<const i32x4 *>(colLo + r * kT8QRow);
			const i32x4 hi = *reinterpret_cast<const i32x4 *>(colHi + r * kT8QRow);
			fb[r] = i32x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
		}
	};

	// one convolution over the region; SECOND: conv B of a block (+ stream, both outputs)
	// A unit's epilogue -- ~150 VALU instructions of residual add, activation, two roundings -- no longer
	// runs between two units' MFMAs (1.9 of 6.7 us per layer at one wave per SIMD): its VALU part is computed
	// in the shadow of the NEXT unit's 18 MFMAs (32x32x64: 16 passes each), two accumulator sets and two sets
	// of stream values alternating; only the 16 masked LDS writes stay between the units, and the last unit
	// of the layer keeps its epilogue.  A group of four values (row r, channel group g) goes behind two MFMAs:
	// sums and activation behind the first, the two roundings behind the second.  (Left to the scheduler with
	// sched_group_barrier pipelines the VALU block stayed where it was.)  Same arithmetic per element in the same order
	// (the packed f32 multiplies / adds of the plain form as single ones: same values): the bytes do not
	// change -- JU_T8_DEFER=0 at build time keeps the plain form for A/B runs.  (Also measured, no faster: the
	// groups' LDS writes behind the same MFMAs and the bias as the first MFMAs' C operand, 284-288 against
	// 283-285 us.)
#ifndef JU_T8_DEFER
#define JU_T8_DEFER 1
#endif
	auto computeLayer = [&](auto secondTag, const int layer, const i32x8(&w)[9]) {
		constexpr bool SECOND = decltype(secondTag)::value;
		constexpr int inOff = SECOND ? kT8OffT : kT8OffX;
		constexpr int outOff = SECOND ? kT8OffX : kT8OffT;
		const float *biasPtr = ldsBias + (layer & 1) * 64 + ch * 32 + 4 * hh;
		const int scA = ldsScale[(layer & 1) * 64 + ch * 32 + px];
		const int scB = p.scaleB[layer];
		const float mul = p.outMul[layer];
		const int np2 = (rhv + 1) >> 1;  // row pairs (an odd last row: its partner row is masked)
		const int sw = ((px + 1) >> 2) & 3;
		const unsigned ssw = ((px + 1) >> 1) & 7;
		auto streamRec = [&](int u) __attribute__((always_inline)) {
			return smem + kT8OffS + (2 * u + 1) * kT8SRow + (px + 1) * 128 + hh * 8;
		};
		// the unit's 18 MFMAs; `behind`: VALU-only work the scheduler spreads behind them
		auto kloop = [&](const int u, f32x16(&acc)[2], Vec4<T>(&rv)[2][4], auto deferTag, auto &&behind) __attribute__((always_inline)) {
			constexpr bool DEFER = decltype(deferTag)::value;
#pragma unroll
			for (int g = 0; g < 4; ++g) {
				const f32x4 bg = *reinterpret_cast<const f32x4 *>(biasPtr + 8 * g);
#pragma unroll
				for (int r = 0; r < 2; ++r) {
#pragma unroll
					for (int i = 0; i < 4; ++i) acc[r][4 * g + i] = bg[i];
				}
			}
			// the next tap's fragments travel behind the current tap's 6 instructions.  (Also
			// fetching the NEXT unit's first fragments in front of the epilogue measured slower,
			// 402 / 378 against 363 us per tower with / without the reading epilogues included:
			// LDS operations return in order.)
			i32x8 f0[4], f1[4];
			if (!(JU_SKIP(p) & 2)) {
				loadFrags(inOff, u, 0, f0);
				loadFrags(inOff, u, 1, f1);
			}
			// conv B: this lane's pieces of the stream (2 rows x 4 groups of 4 channels) are
			// read behind the last fragments, so they return during the remaining 12 instructions
			// (LDS operations return in order); unconditional reads, masked writes: a read under
			// a lane condition makes hipcc wait per element
			const unsigned char *srec = streamRec(u);
#pragma unroll
			// taps in the order dx = 1, 0, 2 (all three 8-bit kernels: the fp32 summation order is
			// part of their byte equality): the middle tap reads no halo column
			for (int t = 0; t < ((JU_SKIP(p) & 2) ? 0 : 3); ++t) {
				const int dx = t == 0 ? 1 : (t == 1 ? 0 : 2);
				if (t == 1) {
					loadFrags(inOff, u, 2, f1);
					if constexpr (SECOND) {
#pragma unroll
						for (int r = 0; r < 2; ++r) {
#pragma unroll
							for (int g = 0; g < 4; ++g) {
								rv[r][g] = *reinterpret_cast<const Vec4<T> *>(srec + r * kT8SRow + ((static_cast<unsigned>(ch * 4 + g) ^ ssw) << 4));
							}
						}
					}
				}
#pragma unroll
				for (int dy = 0; dy < 3; ++dy) {
#pragma unroll
					for (int r = 0; r < 2; ++r) {
						acc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w[dy * 3 + dx], (t == 1 ? f0 : f1)[r + dy], acc[r],
						    0, 0, 0, scA, 0, scB);
						if constexpr (DEFER) {
							// (an MFMA on plainly loaded fragments has no ordered neighbour: without a use that
							// is ordered, instruction selection emits all 18 in front of the barriers)
							asm volatile("" : "+v"(acc[r]));
							__builtin_amdgcn_sched_barrier(0);  // (the MFMA first, the deferred work in its shadow)
							behind(t * 6 + dy * 2 + r);
							__builtin_amdgcn_sched_barrier(0);
						}
					}
				}
			}
		};
		// the epilogue's arithmetic: accumulators (+ stream) -> activation -> 16-bit stream values / e4m3 bytes
		// the deferred form: group j = (r, g) behind MFMAs 2 j (sums and activation) and 2 j + 1 (the roundings)
		float dv[4];
		auto epiHalf = [&](const f32x16(&acc)[2], const Vec4<T>(&rv)[2][4], int(&o8)[2][4], Vec4<T>(&o16)[2][4], const int k)
		                   __attribute__((always_inline)) {
			const int j = k >> 1;
			if (j < 8) {
				const int r = j >> 2, g = j & 3;
				if ((k & 1) == 0) {
#pragma unroll
					for (int i = 0; i < 4; ++i) {
						float x = acc[r][4 * g + i];
						if constexpr (SECOND) x += static_cast<float>(rv[r][g][i]);
						if constexpr (SECOND || LEAKY) x = act8<LEAKY>(x, p.slope);
						dv[i] = x;
						// (pinned: the results are used under the lane mask of the stores, and the optimiser
						// sinks unpinned arithmetic into that branch -- out of the MFMAs' shadow)
						asm volatile("" : "+v"(dv[i]));
					}
				} else {
					if constexpr (SECOND) {
						typedef unsigned u32x2t __attribute__((ext_vector_type(2)));
						u32x2t w16 = __builtin_bit_cast(u32x2t, pack4<T>(dv[0], dv[1], dv[2], dv[3]));
						asm volatile("" : "+v"(w16));
						o16[r][g] = __builtin_bit_cast(Vec4<T>, w16);
					}
					o8[r][g] = t8Quantize4s<LEAKY>(dv[0], dv[1], dv[2], dv[3], mul);
					asm volatile("" : "+v"(o8[r][g]));
				}
			}
		};
		auto epiValues = [&](const f32x16(&acc)[2], const Vec4<T>(&rv)[2][4], int(&o8)[2][4], Vec4<T>(&o16)[2][4], auto singleTag)
		                     __attribute__((always_inline)) {
			constexpr bool SINGLE = decltype(singleTag)::value;  // single f32 instructions (the deferred form)
#pragma unroll
			for (int r = 0; r < 2; ++r) {
#pragma unroll
				for (int g = 0; g < 4; ++g) {
					if constexpr (SECOND) {
						float v[4];
						if constexpr (SINGLE) {
#pragma unroll
							for (int i = 0; i < 4; ++i) v[i] = act8<LEAKY>(acc[r][4 * g + i] + static_cast<float>(rv[r][g][i]), p.slope);
						} else {
							// (the residual adds as two packed v_pk_add_f32)
							const t8f32x2 s01 = t8f32x2{acc[r][4 * g], acc[r][4 * g + 1]} +
							                    t8f32x2{static_cast<float>(rv[r][g][0]), static_cast<float>(rv[r][g][1])};
							const t8f32x2 s23 = t8f32x2{acc[r][4 * g + 2], acc[r][4 * g + 3]} +
							                    t8f32x2{static_cast<float>(rv[r][g][2]), static_cast<float>(rv[r][g][3])};
							v[0] = act8<LEAKY>(s01[0], p.slope);
							v[1] = act8<LEAKY>(s01[1], p.slope);
							v[2] = act8<LEAKY>(s23[0], p.slope);
							v[3] = act8<LEAKY>(s23[1], p.slope);
						}
						o16[r][g] = pack4<T>(v[0], v[1], v[2], v[3]);
						o8[r][g] = SINGLE ? t8Quantize4s<LEAKY>(v[0], v[1], v[2], v[3], mul) : t8Quantize4<LEAKY>(v[0], v[1], v[2], v[3], mul);
					} else if constexpr (LEAKY) {
						const float v0 = act8<true>(acc[r][4 * g], p.slope), v1 = act8<true>(acc[r][4 * g + 1], p.slope),
						            v2 = act8<true>(acc[r][4 * g + 2], p.slope), v3 = act8<true>(acc[r][4 * g + 3], p.slope);
						o8[r][g] = SINGLE ? t8Quantize4s<true>(v0, v1, v2, v3, mul) : t8Quantize4<true>(v0, v1, v2, v3, mul);
					} else {
						o8[r][g] = SINGLE ? t8Quantize4s(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3], mul)
						                  : t8Quantize4(acc[r][4 * g], acc[r][4 * g + 1], acc[r][4 * g + 2], acc[r][4 * g + 3], mul);
					}
				}
			}
		};
		auto epiStores = [&](const int u, const int(&o8)[2][4], const Vec4<T>(&o16)[2][4]) __attribute__((always_inline)) {
			unsigned char *srec = streamRec(u);
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				const int row = 2 * u + r;  // region row; buffer row index row + 1
				const bool valid = px < rwv && row < rhv;
				unsigned char *q8 = smem + outOff + (row + 1) * kT8QRow + (px + 1) * 64;
				if (valid) {
#pragma unroll
					for (int g = 0; g < 4; ++g) {
						if constexpr (SECOND) {
							*reinterpret_cast<Vec4<T> *>(srec + r * kT8SRow + ((static_cast<unsigned>(ch * 4 + g) ^ ssw) << 4)) = o16[r][g];
						}
						*reinterpret_cast<int *>(q8 + (((2 * ch + (g >> 1)) ^ sw) << 4) + (g & 1) * 8 + hh * 4) = o8[r][g];
					}
				}
			}
		};
		const auto nothing = [](int) __attribute__((always_inline)) {};
		int o8[2][4];
		Vec4<T> o16[2][4];
		if (JU_T8_DEFER == 0 || (JU_SKIP(p) & 6) != 0) {
			// the plain form: a unit's epilogue right behind its MFMAs
			for (int u = rp; u < np2; u += 2) {
				f32x16 acc[2];
				Vec4<T> rv[2][4];
				kloop(u, acc, rv, std::false_type{}, nothing);
				if (!(JU_SKIP(p) & 4)) {
					epiValues(acc, rv, o8, o16, std::false_type{});
					epiStores(u, o8, o16);
				}
			}
			return;
		}
		int u = rp;
		if (u >= np2) return;
		f32x16 A[2], B[2];
		Vec4<T> rvA[2][4], rvB[2][4];
		unsigned long long ts = stamp();
		auto lap = [&](int slot) __attribute__((always_inline)) {
			const unsigned long long t = stamp();
			prof[slot] += t - ts;
			ts = t;
		};
		kloop(u, A, rvA, std::false_type{}, nothing);
		lap(5);
		bool lastInB = false;
		for (u += 2; u < np2;) {
			kloop(u, B, rvB, std::true_type{}, [&](int k) __attribute__((always_inline)) { epiHalf(A, rvA, o8, o16, k); });
			lap(5);
			epiStores(u - 2, o8, o16);
			lap(6);
			u += 2;
			if (u >= np2) {
				lastInB = true;
				break;
			}
			kloop(u, A, rvA, std::true_type{}, [&](int k) __attribute__((always_inline)) { epiHalf(B, rvB, o8, o16, k); });
			lap(5);
			epiStores(u - 2, o8, o16);
			lap(6);
			u += 2;
		}
		// the layer's last unit: its epilogue is exposed.  ONE code path, the set selected by value (two
		// branches with the same code on different arrays are merged into one that takes the array through a
		// pointer, and the arrays then live in scratch).
		{
			f32x16 L[2];
			Vec4<T> rvL[2][4];
#pragma unroll
			for (int r = 0; r < 2; ++r) {
				L[r] = lastInB ? B[r] : A[r];
#pragma unroll
				for (int g = 0; g < 4; ++g) rvL[r][g] = lastInB ? rvB[r][g] : rvA[r][g];
			}
			epiValues(L, rvL, o8, o16, std::false_type{});
			epiStores(u - 2, o8, o16);
			lap(6);
		}
	};

	// ---- edge ring -> mailbox, neighbours' mailboxes -> halo ring (64-byte records: 4 chunks) ----
	// LEAKY: dword = 2 value bytes | epoch16 << 16, so a 16-byte slot carries 8 value bytes and a
	// 64-byte record takes 8 slots instead of 4
	constexpr int kSlots = LEAKY ? 2 * kT8MailSlots : kT8MailSlots;
	constexpr int CPP = LEAKY ? 8 : 4, CSH = LEAKY ? 3 : 2;
	const __amdgpu_buffer_rsrc_t mailRsrc = __builtin_amdgcn_make_buffer_rsrc(
	    (void *)p.mail, 0, (int)((size_t)p.GX * p.GY * 2 * kSlots * 16), 0x00020000);
	constexpr int kSc1 = 16;
	// 2-bit epoch (writes to the slot so far & 3) in the sign bits of every byte pair: each
	// dword carries it twice (bits 7, 23 = e & 1; bits 15, 31 = e >> 1)
	auto epochMask = [&](int par) -> unsigned {
		const unsigned e = pubCount[par] & 3u;
		return (e & 1u) * 0x00800080u | (e >> 1) * 0x80008000u;
	};
	// slot descriptors, computed once (tower_kernels.hip): the buffer enters as the LDS
	// instructions' immediate offset, the slot parity as the buffer instructions' scalar offset
	// (cs: slot index inside the pixel record; LEAKY: chunk cs >> 1, 8-byte half cs & 1)
	auto chunkOff = [&](int rr, int cc, int cs) -> unsigned {
		const int c = LEAKY ? cs >> 1 : cs;
		return (unsigned)(rr * kT8QRow + cc * 64 + ((c ^ ((cc >> 2) & 3)) << 4) + (LEAKY ? (cs & 1) * 8 : 0));
	};
	constexpr int NP = kSlots / 256;  // 4 strips x 32 entries x CPP slots
	constexpr int NS = NP + 1;        // 4 sides x 32 entries x CPP slots, + the 4 corners
	unsigned pubLds[NP];
	unsigned pubValid = 0;
#pragma unroll
	for (int it = 0; it < NP; ++it) {
		const int idx = it * 256 + tid;
		const int strip = idx >> (5 + CSH), e = (idx >> CSH) & 31, c = idx & (CPP - 1);
		int rr, cc;
		bool valid;
		if (strip == 0) { rr = 1; cc = e + 1; valid = e < rwv; }
		else if (strip == 1) { rr = rhv; cc = e + 1; valid = e < rwv; }
		else if (strip == 2) { rr = e + 1; cc = 1; valid = e < rhv; }
		else { rr = e + 1; cc = rwv; valid = e < rhv; }
		pubLds[it] = valid ? chunkOff(rr, cc, c) : 0u;  // (branch-free publish: an entry beyond the region reads a harmless place)
		if (valid) pubValid |= 1u << it;
	}
	const unsigned pubBase = (unsigned)(region * 2 * kSlots) * 16u + (unsigned)tid * 16u;
	unsigned sweepSrc[NS], sweepLds[NS];
	unsigned sweepValid = 0;
#pragma unroll
	for (int it = 0; it < NS; ++it) {
		int nx = gxr, ny = gyr, strip, se, rr, cc, c;
		bool valid;
		if (it < NS - 1) {
			const int idx = it * 256 + tid;
			const int hp = idx >> CSH;
			c = idx & (CPP - 1);
			const int side = hp >> 5, e = hp & 31;
			if (side < 2) {  // row above / below: their bottom / top row strip
				ny += side == 0 ? -1 : 1;
				strip = side == 0 ? 1 : 0;
				rr = side == 0 ? 0 : rhv + 1;
				se = e;
				cc = e + 1;
				valid = e < rwv;
			} else {  // column left / right: their right / left column strip
				nx += side == 2 ? -1 : 1;
				strip = side == 2 ? 3 : 2;
				se = e;
				rr = e + 1;
				cc = side == 2 ? 0 : rwv + 1;
				valid = e < rhv;
			}
		} else {  // corners: threads 0 .. 4 CPP - 1 = 4 corners x CPP slots, from the diagonal neighbour's row strips
			const int k = tid >> CSH;
			c = tid & (CPP - 1);
			const bool up = k < 2, left = (k & 1) == 0;
			ny += up ? -1 : 1;
			nx += left ? -1 : 1;
			strip = up ? 1 : 0;
			se = left ? kT8RW - 1 : 0;
			rr = up ? 0 : rhv + 1;
			cc = left ? 0 : rwv + 1;
			valid = tid < 4 * CPP;
		}
		valid = valid && nx >= 0 && nx < p.GX && ny >= 0 && ny < p.GY;
		const int nreg = valid ? ny * p.GX + nx : region;
		sweepSrc[it] = (unsigned)((nreg * 2) * kSlots + (strip * 32 + se) * CPP + c) * 16u;
		sweepLds[it] = chunkOff(rr, cc, c);
		if (valid) sweepValid |= 1u << it;
		if (!valid) {
			// (round 6, as in tower_kernels.hip: the check is branch-free -- a lane without a neighbour slot reads entry 0 of
			// its own region's top row strip, published for the same layer into the same parity, and parks the bytes in the
			// 16 spare bytes behind the buffer)
			sweepSrc[it] = (unsigned)((region * 2) * kSlots + (tid & (CPP - 1))) * 16u;
			sweepLds[it] = (unsigned)kT8DummyOff;
		}
	}
	(void)pubValid;
	(void)sweepValid;
	constexpr unsigned kParityBytes = kSlots * 16u;
	auto publish = [&](auto offTag, int layer) {
		constexpr int off = decltype(offTag)::value;
		const int ppar = (layer + 1) & 1;
		pubCount[ppar] += 1u;
		const unsigned soff = ppar ? kParityBytes : 0u;
		if constexpr (LEAKY) {
			typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
			const unsigned tg = pubCount[ppar] << 16;
			u32x2w v[NP];
#pragma unroll
			for (int it = 0; it < NP; ++it) v[it] = *reinterpret_cast<const u32x2w *>(smem + off + pubLds[it]);
#pragma unroll
			for (int it = 0; it < NP; ++it) {
				{  // (unconditional: an entry beyond the region lands in its own slot, which no consumer reads)
					const u32x4w d = {(v[it][0] & 0xffffu) | tg, (v[it][0] >> 16) | tg, (v[it][1] & 0xffffu) | tg,
					    (v[it][1] >> 16) | tg};
					__builtin_amdgcn_raw_buffer_store_b128(d, mailRsrc, pubBase + it * 4096, soff, kSc1);
				}
			}
		} else {
			const unsigned tm = epochMask(ppar);
			u32x4w v[NP];
#pragma unroll
			for (int it = 0; it < NP; ++it) v[it] = *reinterpret_cast<const u32x4w *>(smem + off + pubLds[it]);
#pragma unroll
			for (int it = 0; it < NP; ++it) {
				// (unconditional.  The mask stays: an e4m3 -0 byte, which the conversion may produce, must not pass for an
				// epoch bit -- one v_and_or per dword either way)
				__builtin_amdgcn_raw_buffer_store_b128((v[it] & 0x7f7f7f7fu) | tm, mailRsrc, pubBase + it * 4096, soff, kSc1);
			}
		}
		asm volatile("s_nop 1" ::: "memory");  // (store-data hazard of soffset-SGPR buffer stores: tower_kernels.hip, publish)
	};
	auto fillHalo = [&](auto offTag, int layer) -> bool {
		constexpr int off = decltype(offTag)::value;
		u64 t0 = 0;  // (the clock is read only once a pass has failed: a scalar-memory round trip)
		const int par = (layer + 1) & 1;
		const unsigned tm = LEAKY ? (pubCount[par] & 0xffffu) : epochMask(par);
		const unsigned soff = par ? kParityBytes : 0u;
		// Branch-free check (round 6; tower_kernels.hip measured the per-slot compare / branch form at ~180 cycles a slot
		// on one wave per SIMD): the expected write XOR the expected epoch IS the payload -- post-ReLU e4m3 bytes have
		// clear sign bits, LeakyReLU slots carry the epoch in their dwords' upper halves --, every lane writes every pass
		// (a slot holds the previous write or the expected one, never a newer one: a rewrite repeats the bytes), one
		// accumulator of the bits left standing, one vote per pass.
		unsigned pending = 1u;
		while (__any(pending != 0)) {
			u32x4w hv[NS];
#pragma unroll
			for (int it = 0; it < NS; ++it) hv[it] = __builtin_amdgcn_raw_buffer_load_b128(mailRsrc, sweepSrc[it], soff, kSc1);
			unsigned bad = 0;
#pragma unroll
			for (int it = 0; it < NS; ++it) {
				if constexpr (LEAKY) {
					typedef unsigned u32x2w __attribute__((ext_vector_type(2)));
					const u32x4w x = hv[it] ^ (tm << 16);
					bad |= x[0] | x[1] | x[2] | x[3];
					*reinterpret_cast<u32x2w *>(smem + off + sweepLds[it]) = u32x2w{(x[0] & 0xffffu) | (x[1] << 16), (x[2] & 0xffffu) | (x[3] << 16)};
				} else {
					const u32x4w x = hv[it] ^ tm;
					bad |= x[0] | x[1] | x[2] | x[3];
					*reinterpret_cast<u32x4w *>(smem + off + sweepLds[it]) = x;
				}
			}
			pending = bad & (LEAKY ? 0xffff0000u : 0x80808080u);
			if (pending != 0) {
				const u64 now = __builtin_amdgcn_s_memrealtime();
				if (t0 == 0) t0 = now;
				if (now - t0 > kT8TimeoutTicks) {
					*failFlag = 1;
					__hip_atomic_store((gu32 *)p.error, 0x800u + (unsigned)layer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
					break;
				}
				__builtin_amdgcn_s_sleep(1);
			}
		}
		__syncthreads();
		return *failFlag == 0;
	};

	// ---- the tower ----
	const int L = p.nLayers;
	float biasNext = 0.f;
	int scaleNext = 0;
	auto layerStep = [&](auto secondTag, const int i, i32x8(&wc)[9]) -> bool {
		constexpr bool SECOND = decltype(secondTag)::value;
		const bool more = i + 1 < L;
		const unsigned long long t0 = stamp();
		computeLayer(secondTag, i, wc);
		const unsigned long long t1 = stamp();
		__syncthreads();
		const unsigned long long t2 = stamp();
		// operands of layer i+1 (fetched one step ago) into the slots layer i-1 used; the
		// sweep's closing barrier orders them before the next layer's reads
		if (i >= 1 && more && wave == 0) {
			ldsBias[((i + 1) & 1) * 64 + lane] = biasNext;
			ldsScale[((i + 1) & 1) * 64 + lane] = scaleNext;
		}
		if (more && !(JU_SKIP(p) & 1)) publish(std::integral_constant<int, SECOND ? kT8OffX : kT8OffT>{}, i);
		const unsigned long long t3 = stamp();
		// this layer's weight registers are free: refill them for layer i+2 while the
		// neighbours' stores travel, THEN sweep (tower_kernels.hip, same order)
		if (i + 2 < L) {
			// (Round 6, tools/tower8_phases.py: these 18 loads of four waves are 1.1 k exposed cycles per layer here -- and the
			// cheapest cover there is for the flight of the publish's stores: streamed from inside the next layer's first unit
			// instead, the sweep's first pass comes too early and the layer is 5 us slower, profiles/r06_t8_wstream_ab.txt)
			loadWeights(i + 2, wc);
			if (wave == 0) {
				biasNext = p.bias[(i + 2) * 64 + lane];
				scaleNext = p.scaleA[(i + 2) * 64 + lane];
			}
		}
		const unsigned long long t4 = stamp();
		// the halo of the next layer's INPUT: this layer's output ring
		if (more && !(JU_SKIP(p) & 1)) {
			if (!fillHalo(std::integral_constant<int, SECOND ? kT8OffX : kT8OffT>{}, i)) return false;
		} else {
			__syncthreads();
		}
		const unsigned long long t5 = stamp();
		prof[0] += t5 - t4;
		prof[1] += t4 - t3;
		prof[2] += t1 - t0;
		prof[3] += t2 - t1;
		prof[4] += t3 - t2;
		return true;
	};
	using First = std::false_type;
	using Second = std::true_type;
	for (int i = 0; i + 1 < L; i += 2) {
		if (!layerStep(First{}, i, w0)) return;
		if (!layerStep(Second{}, i + 1, w1)) return;
	}
	if (tid == 0) {
		p.count[region * 2] = pubCount[0];
		p.count[region * 2 + 1] = pubCount[1];
	}
#ifdef JU_T8_PROF
	if (lane == 0 && p.debug != nullptr) {
		for (int k = 0; k < 8; ++k) p.debug[(region * 4 + wave) * 8 + k] = prof[k];
	}
#endif
	// ---- the stream's interior -> global tower-layout tensor ----
	{
		T *out = static_cast<T *>(p.out);
		for (int i = tid; i < rhv * kT8RW * 8; i += 256) {
			const int c = i & 7;
			const int pxl = (i >> 3) % kT8RW;
			const int row = (i >> 3) / kT8RW;
			if (pxl < rwv) {
				const int rr = row + 1, cc = pxl + 1;
				const uint4 v = *reinterpret_cast<const uint4 *>(
				    smem + kT8OffS + rr * kT8SRow + cc * 128 + ((c ^ ((cc >> 1) & 7)) << 4));
				*reinterpret_cast<uint4 *>(out + ((size_t)(y0 + rr) * p.pitch + x0 + cc) * 64 + c * 8) = v;
			}
		}
	}
}

template <typename T, bool LEAKY>
void launchTower8T(const Tower8Params &p, hipStream_t stream) {
	auto kern = tower8_resident_kernel<T, LEAKY>;
	static std::atomic<std::uint64_t> ldsDone{0};
	ensureDynamicLds(reinterpret_cast<const void *>(kern), kT8Lds, &ldsDone, "fp8 resident tower");
	const int grid = p.GX * p.GY - (p.fault < p.GX * p.GY ? p.fault : 0);
	hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kT8Lds, stream, p);
	hipCheckLaunch("tower8_resident");
}

}  // namespace

std::size_t residentMailboxBytes8(int GX, int GY, bool leaky) {
	return static_cast<std::size_t>(GX) * GY * 2 * kT8MailSlots * 16 * (leaky ? 2 : 1);
}

void launchResidentTower8(DType dt, const ResidentTower8Params &q, hipStream_t stream) {
	Tower8Params p{};
	p.in = q.in;
	p.out = q.out;
	p.weights = static_cast<const unsigned char *>(q.weights);
	p.scaleA = q.scaleA;
	p.bias = q.bias;
	p.scaleB = q.scaleB;
	p.outMul = q.outMul;
	p.mail = static_cast<uint4 *>(q.mailbox);
	p.count = q.counters;
	p.error = q.error;
	p.H = q.H;
	p.W = q.W;
	p.pitch = towerPitch(q.W);
	p.GX = q.GX;
	p.GY = q.GY;
	p.RH = q.RH;
	p.nLayers = q.nLayers;
	p.slope = q.slope;
	p.debug = static_cast<unsigned long long *>(q.debug);
	p.fault = residentFaultForTests();
	p.skip = ablationSkipBits();
	if (p.nLayers < 2 || (p.nLayers & 1)) throw std::invalid_argument("fp8 resident tower: layer count must be 2 x blocks");
	if (q.leaky) {
		if (dt == kF16) launchTower8T<f16, true>(p, stream);
		else launchTower8T<bf16, true>(p, stream);
		return;
	}
	if (dt == kF16) launchTower8T<f16, false>(p, stream);
	else launchTower8T<bf16, false>(p, stream);
}

}  // namespace ju
