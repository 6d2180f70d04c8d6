// Which MFMA shape should a 3x3 64->64 convolution loop of the resident tower use under the socket's power limit?
// (MI355X_MICROARCH.md "DVFS give-back" item 7: in bare loops on random data v_mfma_f32_16x16x32_bf16 delivered
// ~1.15x the FLOP/s of 32x32x16 at equal cycles per FLOP -- the chip holds a higher clock.)  This probe runs the
// TOWER's loop shape: one wave per SIMD, weights (32 couts x 576) as the A operand from 144 registers, activations
// as the B operand from an 18 x 34-pixel LDS tile (128 B per pixel, the tower's column swizzle) by hand-issued
// ds_read_b128 one macro-step ahead with counted waits, every CU busy, random bf16 data.
//
//   shape 0: v_mfma_f32_32x32x16_bf16, unit = ROWS output rows x 32 px x 32 couts; per (dx, ks16) macro-step
//            ROWS+2 row fragments feed 3*ROWS MFMAs (ROWS = 2: the tower today, 0.67 reads per MFMA)
//   shape 1: v_mfma_f32_16x16x32_bf16, same unit; per (dx, ks32, pixel half) macro-step ROWS+2 row fragments
//            (16 px x 32 ch each) feed 3*ROWS*2 MFMAs (both cout halves): the same LDS bytes per FLOP
//
// Printed per variant: us per launch, TFLOP/s, cycles per unit (s_memtime) and the in-kernel clock
// (s_memtime / s_memrealtime), after a warm-up of back-to-back launches.
//
//     hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form tools/probes/mfma_shape.hip -o build/mfma_shape
#include <hip/hip_runtime.h>

#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#define CHECK(x)                                                         \
	do {                                                                 \
		hipError_t e_ = (x);                                             \
		if (e_ != hipSuccess) {                                          \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			std::exit(2);                                                \
		}                                                                \
	} while (0)

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;

constexpr int kRowBytes = 34 * 128;  // 4352
constexpr int kTileRows = 18;
constexpr int kTileBytes = kTileRows * kRowBytes;

template <int N>
__device__ __forceinline__ void waitLgkm() {
	static_assert(N >= 0 && N <= 15, "lgkmcnt");
	asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

template <int J>
__device__ __forceinline__ void rd(bf16x8 &dst, unsigned a) {
	asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(a), "n"(kRowBytes * J));
}

template <int NR>
__device__ __forceinline__ void issueRow(bf16x8 (&fb)[NR], unsigned a, int j) {
	if (j == 0) rd<0>(fb[0], a);
	else if (j == 1) rd<1>(fb[1], a);
	else if (j == 2) rd<2>(fb[2], a);
	else if (j == 3) rd<3>(fb[3], a);
	else if (j == 4) { if constexpr (NR > 4) rd<4>(fb[4], a); }
	else { if constexpr (NR > 5) rd<5>(fb[5], a); }
}

template <int ALLOWED>
__device__ __forceinline__ void waitAllowed() {
	if constexpr (ALLOWED >= 12) waitLgkm<12>();
	else waitLgkm<ALLOWED>();
}

// SHAPE 0: 32x32x16.  12 macro-steps (dx, ks) per unit.
// FILL: the tower's deferred epilogue as filler -- in 8 of the 12 macro-steps 12 plain VALU instructions and one
// ds_write_b64 spread behind the step's MFMAs (values that touch neither operands nor results)
template <int ROWS, bool FILL>
__device__ __forceinline__ void unit32(unsigned rowAddr, const unsigned (&colBase)[3], const unsigned (&colSwz)[3], int hh,
    const bf16x8 (&w)[36], f32x16 (&acc)[ROWS], float (&fx)[4], unsigned wrAddr) {
	constexpr int NR = ROWS + 2;
	bf16x8 fb[2][NR];
	auto addr = [&](int m) {
		const int dx = m >> 2, ks = m & 3;
		return rowAddr + colBase[dx] + (((unsigned)(ks * 2 + hh) ^ colSwz[dx]) << 4);
	};
	waitLgkm<0>();
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int j = 0; j < NR; ++j) issueRow<NR>(fb[0], addr(0), j);
#pragma unroll
	for (int m = 0; m < 12; ++m) {
		const int set = m & 1;
		const bool more = m + 1 < 12;
		const int dx = m >> 2, ks = m & 3;
		int maxNeed = -1;
#pragma unroll
		for (int k = 0; k < 3 * ROWS; ++k) {
			const int dy = k / ROWS, r = k % ROWS;
			const int need = r + dy;
			if (need > maxNeed) {
				maxNeed = need;
				const int issuedNext = more ? (k < NR ? k : NR) : 0;
				const int allowed = (NR - 1 - need) + issuedNext;
				switch (allowed) {
				case 0: waitLgkm<0>(); break;
				case 1: waitLgkm<1>(); break;
				case 2: waitLgkm<2>(); break;
				case 3: waitLgkm<3>(); break;
				case 4: waitLgkm<4>(); break;
				case 5: waitLgkm<5>(); break;
				case 6: waitLgkm<6>(); break;
				case 7: waitLgkm<7>(); break;
				case 8: waitLgkm<8>(); break;
				case 9: waitLgkm<9>(); break;
				case 10: waitLgkm<10>(); break;
				default: waitLgkm<11>(); break;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
			acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[(dy * 3 + dx) * 4 + ks], fb[set][need], acc[r], 0, 0, 0);
			if (more && k < NR) issueRow<NR>(fb[set ^ 1], addr(m + 1), k);
			if constexpr (FILL) {
				if (m < 8 && k < 6) {
					if (k < 4) {
						asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_add_f32 %0, %0, %2" : "=&v"(fx[k]) : "v"(fx[(k + 1) & 3]), "v"(fx[(k + 2) & 3]));
					} else if (k == 4) {
						asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %1, %2, %0" : "+v"(fx[0]), "+v"(fx[1]) : "v"(fx[2]));
					} else {
						asm volatile("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0\n\tds_write_b64 %2, %3" : "+v"(fx[0]), "+v"(fx[1]) : "v"(wrAddr), "v"(*reinterpret_cast<u64 *>(&fx[2])) : "memory");
					}
				}
			}
			__builtin_amdgcn_sched_barrier(0);
		}
	}
}

// SHAPE 1: 16x16x32.  12 macro-steps (dx, ks32, pixel half) per unit; A fragment (tap, ks32, cout half) = 16 couts x 32 K.
template <int ROWS, bool FILL>
__device__ __forceinline__ void unit16(unsigned rowAddr, const unsigned (&colBase)[3][2], const unsigned (&colSwz)[3][2], int q,
    const bf16x8 (&w)[36], f32x4 (&acc)[ROWS][2][2], float (&fx)[4], unsigned wrAddr) {
	constexpr int NR = ROWS + 2;
	bf16x8 fb[2][NR];
	auto addr = [&](int m) {
		const int dx = m >> 2, ks = (m >> 1) & 1, ph = m & 1;
		return rowAddr + colBase[dx][ph] + (((unsigned)(ks * 4 + q) ^ colSwz[dx][ph]) << 4);
	};
	waitLgkm<0>();
	__builtin_amdgcn_sched_barrier(0);
#pragma unroll
	for (int j = 0; j < NR; ++j) issueRow<NR>(fb[0], addr(0), j);
#pragma unroll
	for (int m = 0; m < 12; ++m) {
		const int set = m & 1;
		const bool more = m + 1 < 12;
		const int dx = m >> 2, ks = (m >> 1) & 1, ph = m & 1;
		int maxNeed = -1;
#pragma unroll
		for (int k = 0; k < 3 * ROWS; ++k) {
			const int dy = k / ROWS, r = k % ROWS;
			const int need = r + dy;
			if (need > maxNeed) {
				maxNeed = need;
				const int issuedNext = more ? (k < NR ? k : NR) : 0;
				const int allowed = (NR - 1 - need) + issuedNext;
				switch (allowed) {
				case 0: waitLgkm<0>(); break;
				case 1: waitLgkm<1>(); break;
				case 2: waitLgkm<2>(); break;
				case 3: waitLgkm<3>(); break;
				case 4: waitLgkm<4>(); break;
				case 5: waitLgkm<5>(); break;
				case 6: waitLgkm<6>(); break;
				case 7: waitLgkm<7>(); break;
				case 8: waitLgkm<8>(); break;
				case 9: waitLgkm<9>(); break;
				case 10: waitLgkm<10>(); break;
				default: waitLgkm<11>(); break;
				}
				__builtin_amdgcn_sched_barrier(0);
			}
#pragma unroll
			for (int ch = 0; ch < 2; ++ch) {
				acc[r][ph][ch] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[((dy * 3 + dx) * 2 + ks) * 2 + ch], fb[set][need], acc[r][ph][ch], 0, 0, 0);
				if (ch == 0 && more && k < NR) issueRow<NR>(fb[set ^ 1], addr(m + 1), k);
				if constexpr (FILL) {
					const int kk = 2 * k + ch;  // 0 .. 6 ROWS - 1: the same 12 VALU + 1 LDS write per macro-step, two behind every second MFMA
					if (m < 8 && kk < 12) {
						if (kk < 8) {
							if ((kk & 1) == 0) asm volatile("v_lshlrev_b32 %0, 16, %1\n\tv_add_f32 %0, %0, %2" : "=&v"(fx[kk >> 1]) : "v"(fx[((kk >> 1) + 1) & 3]), "v"(fx[((kk >> 1) + 2) & 3]));
						} else if (kk == 8) {
							asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2\n\tv_cvt_pk_bf16_f32 %1, %2, %0" : "+v"(fx[0]), "+v"(fx[1]) : "v"(fx[2]));
						} else if (kk == 10) {
							asm volatile("v_pk_max_i16 %0, %0, 0\n\tv_pk_max_i16 %1, %1, 0" : "+v"(fx[0]), "+v"(fx[1]));
						} else if (kk == 11) {
							asm volatile("ds_write_b64 %0, %1" ::"v"(wrAddr), "v"(*reinterpret_cast<u64 *>(&fx[2])) : "memory");
						}
					}
				}
				__builtin_amdgcn_sched_barrier(0);
			}
		}
	}
}

template <int SHAPE, int ROWS, bool FILL>
__global__ __launch_bounds__(256, 1) void probe(const unsigned char *__restrict__ tile, const unsigned char *__restrict__ wgt, float *out,
    u64 *stamps, int units) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	for (int i = tid; i < kTileBytes / 16; i += 256) {
		reinterpret_cast<uint4 *>(smem)[i] = reinterpret_cast<const uint4 *>(tile)[i + (blockIdx.x & 3) * 64];
	}
	bf16x8 w[36];
#pragma unroll
	for (int f = 0; f < 36; ++f) w[f] = *reinterpret_cast<const bf16x8 *>(wgt + ((size_t)(wave & 1) * 36 + f) * 1024 + lane * 16);
	__syncthreads();
	const unsigned ldsBase = static_cast<unsigned>(reinterpret_cast<unsigned long long>((__attribute__((address_space(3))) unsigned char *)smem));
	const int nUnitsPerTile = (kTileRows - 2) / ROWS;
	u64 t0, t1, r0, r1;
	float sum = 0.f;
	float fx[4] = {1.f + lane, 2.f, 3.f + tid, 4.f};
	const unsigned wrAddr = ldsBase + kTileBytes + tid * 8;  // (a scratch area behind the tile)
	if constexpr (SHAPE == 0) {
		const int px = lane & 31, hh = lane >> 5;
		unsigned colBase[3], colSwz[3];
#pragma unroll
		for (int dx = 0; dx < 3; ++dx) {
			colBase[dx] = (px + dx) * 128;
			colSwz[dx] = ((px + dx) >> 1) & 7;
		}
		f32x16 acc[ROWS];
#pragma unroll
		for (int r = 0; r < ROWS; ++r)
#pragma unroll
			for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
		asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
		for (int u = 0; u < units; ++u) {
			unsigned rowAddr = ldsBase + ((u + (wave >> 1)) % nUnitsPerTile) * ROWS * kRowBytes;
			asm volatile("" : "+v"(rowAddr));
			unit32<ROWS, FILL>(rowAddr, colBase, colSwz, hh, w, acc, fx, wrAddr);
		}
		asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
#pragma unroll
		for (int r = 0; r < ROWS; ++r)
#pragma unroll
			for (int i = 0; i < 16; ++i) sum += acc[r][i];
	} else {
		const int p16 = lane & 15, q = lane >> 4;
		unsigned colBase[3][2], colSwz[3][2];
#pragma unroll
		for (int dx = 0; dx < 3; ++dx) {
#pragma unroll
			for (int ph = 0; ph < 2; ++ph) {
				const int c = ph * 16 + p16 + dx;
				colBase[dx][ph] = c * 128;
				colSwz[dx][ph] = (c >> 1) & 7;
			}
		}
		f32x4 acc[ROWS][2][2];
#pragma unroll
		for (int r = 0; r < ROWS; ++r)
#pragma unroll
			for (int a = 0; a < 4; ++a)
#pragma unroll
				for (int i = 0; i < 4; ++i) acc[r][a >> 1][a & 1][i] = 0.f;
		asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0)::"memory");
		for (int u = 0; u < units; ++u) {
			unsigned rowAddr = ldsBase + ((u + (wave >> 1)) % nUnitsPerTile) * ROWS * kRowBytes;
			asm volatile("" : "+v"(rowAddr));
			unit16<ROWS, FILL>(rowAddr, colBase, colSwz, q, w, acc, fx, wrAddr);
		}
		asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1)::"memory");
#pragma unroll
		for (int r = 0; r < ROWS; ++r)
#pragma unroll
			for (int a = 0; a < 4; ++a)
#pragma unroll
				for (int i = 0; i < 4; ++i) sum += acc[r][a >> 1][a & 1][i];
	}
	out[blockIdx.x * 256 + tid] = sum + fx[0] + fx[1] + fx[2] + fx[3];
	if (lane == 0) {
		stamps[(blockIdx.x * 4 + wave) * 2] = t1 - t0;
		stamps[(blockIdx.x * 4 + wave) * 2 + 1] = r1 - r0;
	}
}

template <int SHAPE, int ROWS, bool FILL>
void run(const char *name, const unsigned char *tile, const unsigned char *wgt, float *out, u64 *stamps, int units, int grid) {
	auto kern = probe<SHAPE, ROWS, FILL>;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kTileBytes + 2048));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	// warm-up: ~1.5 s of back-to-back launches, so that the clock has settled under this load
	for (int i = 0; i < 600; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kTileBytes + 2048, 0, tile, wgt, out, stamps, units);
	CHECK(hipDeviceSynchronize());
	const int reps = 200;
	CHECK(hipEventRecord(e0));
	for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kTileBytes + 2048, 0, tile, wgt, out, stamps, units);
	CHECK(hipEventRecord(e1));
	CHECK(hipEventSynchronize(e1));
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	std::vector<u64> st(grid * 8);
	CHECK(hipMemcpy(st.data(), stamps, st.size() * sizeof(u64), hipMemcpyDeviceToHost));
	std::vector<double> cyc, clk;
	for (int i = 0; i < grid * 4; ++i) {
		cyc.push_back((double)st[2 * i] / units);
		clk.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
	}
	std::sort(cyc.begin(), cyc.end());
	std::sort(clk.begin(), clk.end());
	const double us = ms * 1000.0 / reps;
	const double flop = (double)grid * 4 * units * ROWS * 32.0 * 32.0 * 576.0 * 2.0;
	std::printf("%-34s %8.1f us/launch  %7.1f TFLOP/s  %7.1f cycles/unit (%.2f per 32x32x16-equivalent MFMA)  in-kernel clock %6.0f MHz\n", name, us,
	    flop / us * 1e-6, cyc[cyc.size() / 2], cyc[cyc.size() / 2] / (36.0 * ROWS), clk[clk.size() / 2]);
	std::fflush(stdout);
}

// Operand layout of v_mfma_f32_16x16x32_bf16 as this probe (and the tower's 16x16x32 form) assumes it:
//   A (16 x 32): lane l holds A[l % 16][8 * (l / 16) + e], e = 0..7;  B (32 x 16): lane l holds B[8 * (l / 16) + e][l % 16];
//   D (16 x 16): lane l holds D[4 * (l / 16) + i][l % 16], i = 0..3.
__global__ void layoutCheck(const float *a, const float *b, float *d) {
	const int l = threadIdx.x;
	bf16x8 av, bv;
	for (int e = 0; e < 8; ++e) {
		av[e] = (__bf16)a[(l % 16) * 32 + 8 * (l / 16) + e];
		bv[e] = (__bf16)b[(8 * (l / 16) + e) * 16 + (l % 16)];
	}
	f32x4 acc = {0.f, 0.f, 0.f, 0.f};
	acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc, 0, 0, 0);
	for (int i = 0; i < 4; ++i) d[(4 * (l / 16) + i) * 16 + (l % 16)] = acc[i];
}

static void checkLayout() {
	std::vector<float> a(16 * 32), b(32 * 16), d(256), ref(256, 0.f);
	for (int i = 0; i < 512; ++i) {
		a[i] = (float)((i * 7) % 13 - 6);   // small integers: exact in bf16 and in the f32 sums
		b[i] = (float)((i * 5) % 11 - 5);
	}
	for (int i = 0; i < 16; ++i)
		for (int j = 0; j < 16; ++j)
			for (int k = 0; k < 32; ++k) ref[i * 16 + j] += a[i * 32 + k] * b[k * 16 + j];
	float *da, *db, *dd;
	CHECK(hipMalloc(&da, 2048));
	CHECK(hipMalloc(&db, 2048));
	CHECK(hipMalloc(&dd, 1024));
	CHECK(hipMemcpy(da, a.data(), 2048, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(db, b.data(), 2048, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(layoutCheck, dim3(1), dim3(64), 0, 0, da, db, dd);
	CHECK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
	int bad = 0;
	for (int i = 0; i < 256; ++i) bad += d[i] != ref[i];
	std::printf("v_mfma_f32_16x16x32_bf16 operand layout as assumed: %s (%d of 256 elements differ)\n", bad ? "NO" : "yes", bad);
}

int main(int argc, char **argv) {
	const int units = argc > 1 ? std::atoi(argv[1]) : 1000;
	const int rounds = argc > 2 ? std::atoi(argv[2]) : 3;
	int cus = 256;
	CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
	std::mt19937 rng(1234);
	std::normal_distribution<float> nd(0.f, 1.f);
	auto toBf16 = [](float f) {
		unsigned u;
		std::memcpy(&u, &f, 4);
		u += 0x7fff + ((u >> 16) & 1);
		return (unsigned short)(u >> 16);
	};
	std::vector<unsigned short> ht(kTileBytes / 2 + 4 * 512), hw(2 * 36 * 512);
	for (auto &v : ht) v = toBf16(std::fabs(nd(rng)) * 0.5f);   // post-ReLU activations
	for (auto &v : hw) v = toBf16(nd(rng) * 0.06f);             // He-scaled 3x3x64 weights
	unsigned char *tile, *wgt;
	float *out;
	u64 *stamps;
	CHECK(hipMalloc(&tile, ht.size() * 2));
	CHECK(hipMalloc(&wgt, hw.size() * 2));
	CHECK(hipMalloc(&out, cus * 256 * 4));
	CHECK(hipMalloc(&stamps, cus * 8 * sizeof(u64)));
	CHECK(hipMemcpy(tile, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
	CHECK(hipMemcpy(wgt, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
	checkLayout();
	std::printf("grid %d workgroups x 256 threads (one wave per SIMD), %d units per wave and launch\n", cus, units);
	for (int r = 0; r < rounds; ++r) {
		run<0, 2, false>("32x32x16, 2-row units (tower)", tile, wgt, out, stamps, units, cus);
		run<1, 2, false>("16x16x32, 2-row units", tile, wgt, out, stamps, units, cus);
		run<0, 2, true>("32x32x16, 2-row, epilogue filler", tile, wgt, out, stamps, units, cus);
		run<1, 2, true>("16x16x32, 2-row, epilogue filler", tile, wgt, out, stamps, units, cus);
		run<0, 4, false>("32x32x16, 4-row units", tile, wgt, out, stamps, units / 2, cus);
		run<1, 4, false>("16x16x32, 4-row units", tile, wgt, out, stamps, units / 2, cus);
	}
	return 0;
}
