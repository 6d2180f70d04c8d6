// How much other work fits in the shadow of an MFMA issued by the SAME wave (one wave per SIMD, gfx950)?
// A wave runs a chain-free stream of v_mfma_f32_32x32x16_bf16 (two accumulators alternating, 8 passes =
// 32 cycles each) and, behind every MFMA, N instructions of one kind that touch neither its operands nor
// its result.  Printed: cycles per MFMA (s_memtime, 100 MHz ticks scaled by the measured shader clock
// are NOT used -- the figure is relative to the N = 0 row).
//
//     hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o build/mfma_valu_overlap
//     (-DPROBE_FP8: the same behind v_mfma_scale_f32_32x32x64_f8f6f4, 16 passes)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
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
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

enum Kind { kNone, kCvtPk, kPkMax, kAddF32, kPkAddF32, kMov64, kDsWrite64, kDsRead128, kLshl, kSaveExec, kStore4, kStore64, kStore4Wb, kStoreEvery8, kDepChain, kIndep3, kAccRead };
static const char *kNames[] = {"(nothing)", "v_cvt_pk_bf16_f32", "v_pk_max_i16", "v_add_f32", "v_pk_add_f32", "v_mov_b64",
    "ds_write_b64", "ds_read_b128 (+wait at the end)", "v_lshlrev_b32", "s_and_saveexec + s_or exec",
    "buffer_store_dwordx2 sc1, 4 lanes", "buffer_store_dwordx2 sc1, 64 lanes", "buffer_store_dwordx2 (no sc1), 4 lanes", "buffer_store_dwordx2 sc1, 4 lanes, every 8th MFMA",
    "N x (add -> mul -> max, each on the last result)", "N x (add, mul, max, independent)", "N x v_add_f32 reading the OTHER accumulator"};

template <int KIND, int N>
__global__ __launch_bounds__(256, 1) void probe(float *out, unsigned long long *ticks, int iters, void *scratch) {
	__shared__ __attribute__((aligned(16))) unsigned char lds[65536];
	const int lane = threadIdx.x & 63;
	bf16x8 a, b;
	for (int i = 0; i < 8; ++i) {
		a[i] = (__bf16)(0.001f * (lane + i));
		b[i] = (__bf16)(0.002f * (lane - i));
	}
	typedef int i32x8p __attribute__((ext_vector_type(8)));
	i32x8p a8, b8;
	for (int i = 0; i < 8; ++i) {
		a8[i] = 0x38383838 + lane;
		b8[i] = 0x30303030 + i;
	}
	f32x16 acc0 = {}, acc1 = {};
	float x0 = lane, x1 = 2 * lane, x2 = 3, x3 = 4, y2 = 5;
	unsigned u0 = lane, u1 = lane * 3;
	u32x2 w = {u0, u1};
	unsigned ldsAddr = (threadIdx.x * 16) & 0xffff;
	typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
	u32x4 rd = {};
	const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(scratch, 0, 0x7fffffff, 0x00020000);
	unsigned long long t0, t1;
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
	for (int it = 0; it < iters; ++it) {
#pragma unroll
		for (int m = 0; m < 8; ++m) {
#ifdef PROBE_FP8  // v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 operands, 16 passes) instead of the bf16 instruction
			if (m & 1) acc1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc1, 0, 0, 0, 127, 0, 127);
			else acc0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc0, 0, 0, 0, 127, 0, 127);
#else
			if (m & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc1, 0, 0, 0);
			else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc0, 0, 0, 0);
#endif
#pragma unroll
			for (int n = 0; n < N; ++n) {
				if constexpr (KIND == kCvtPk) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(u0) : "v"(x0), "v"(x1));
				if constexpr (KIND == kPkMax) asm volatile("v_pk_max_i16 %0, %1, 0" : "=v"(u0) : "v"(u1));
				if constexpr (KIND == kAddF32) asm volatile("v_add_f32 %0, %1, %2" : "=v"(x2) : "v"(x0), "v"(x1));
				if constexpr (KIND == kPkAddF32) asm volatile("v_pk_add_f32 %0, %1, %1" : "=v"(w) : "v"(w));
				if constexpr (KIND == kMov64) asm volatile("v_mov_b64 %0, %1" : "=v"(w) : "v"(w));
				if constexpr (KIND == kDsWrite64) asm volatile("ds_write_b64 %0, %1" ::"v"(ldsAddr), "v"(w) : "memory");
				if constexpr (KIND == kDsRead128) asm volatile("ds_read_b128 %0, %1" : "=v"(rd) : "v"(ldsAddr) : "memory");
				if constexpr (KIND == kLshl) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(u0) : "v"(u1));
				if constexpr (KIND == kStore4 || KIND == kStore64 || KIND == kStore4Wb || KIND == kStoreEvery8) {
					if (KIND != kStoreEvery8 || m == 7) {
						const bool on = KIND == kStore64 || (lane & 31) == 0 || (lane & 31) == 31;
						const unsigned off = (blockIdx.x * 256 + threadIdx.x) * 8 + ((it * 8 + m) & 63) * 2048 * 256;
						if (on) __builtin_amdgcn_raw_buffer_store_b64(w, rsrc, off & 0x7ffffff, 0, KIND == kStore4Wb ? 0 : 16);
					}
				}
				if constexpr (KIND == kDepChain) {
					asm volatile("v_add_f32 %0, %1, %2" : "=v"(x2) : "v"(x0), "v"(x1));
					asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x3) : "v"(x2), "v"(x1));
					asm volatile("v_max_f32 %0, %1, %2" : "=v"(x2) : "v"(x2), "v"(x3));
				}
				if constexpr (KIND == kIndep3) {
					asm volatile("v_add_f32 %0, %1, %2" : "=v"(x2) : "v"(x0), "v"(x1));
					asm volatile("v_mul_f32 %0, %1, %2" : "=v"(x3) : "v"(x0), "v"(x1));
					asm volatile("v_max_f32 %0, %1, %2" : "=v"(y2) : "v"(x0), "v"(x1));
				}
				if constexpr (KIND == kAccRead) {
					// reads an element of the accumulator the CURRENT MFMA does not write
					float e = (m & 1) ? acc0[n & 15] : acc1[n & 15];
					asm volatile("v_add_f32 %0, %1, %2" : "=v"(x2) : "v"(e), "v"(x1));
				}
				if constexpr (KIND == kSaveExec) asm volatile("s_and_saveexec_b64 s[40:41], vcc\n\ts_or_b64 exec, exec, s[40:41]" ::: "s40", "s41");
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if constexpr (KIND == kDsRead128 || KIND == kDsWrite64) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
	}
	asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
	float s = 0;
	for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
	out[blockIdx.x * 256 + threadIdx.x] = s + x2 + x3 + y2 + u0 + w[0] + w[1] + rd[0] + lds[threadIdx.x];
	if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

static void *g_scratch = nullptr;

template <int KIND, int N>
double run(float *out, unsigned long long *ticks, int iters) {
	hipLaunchKernelGGL((probe<KIND, N>), dim3(256), dim3(256), 0, 0, out, ticks, iters, g_scratch);
	CHECK(hipDeviceSynchronize());
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	CHECK(hipEventRecord(e0));
	hipLaunchKernelGGL((probe<KIND, N>), dim3(256), dim3(256), 0, 0, out, ticks, iters, g_scratch);
	CHECK(hipEventRecord(e1));
	CHECK(hipDeviceSynchronize());
	float ms = 0;
	CHECK(hipEventElapsedTime(&ms, e0, e1));
	return ms * 1e6 / (iters * 8.0);  // ns per MFMA
}

template <int KIND>
void sweep(float *out, unsigned long long *ticks, int iters, double base) {
	const double r[] = {run<KIND, 1>(out, ticks, iters), run<KIND, 2>(out, ticks, iters), run<KIND, 3>(out, ticks, iters),
	    run<KIND, 4>(out, ticks, iters), run<KIND, 6>(out, ticks, iters), run<KIND, 8>(out, ticks, iters), run<KIND, 12>(out, ticks, iters)};
	std::printf("%-34s", kNames[KIND]);
	for (double v : r) std::printf(" %6.2f", v / base);
	std::printf("\n");
}

int main() {
	float *out;
	unsigned long long *ticks;
	CHECK(hipMalloc(&out, 256 * 256 * 4));
	CHECK(hipMalloc(&ticks, 256 * 8));
	CHECK(hipMalloc(&g_scratch, 256u << 20));
	const int iters = 20000;
	const double base = run<kNone, 0>(out, ticks, iters);
	#ifdef PROBE_FP8
	std::printf("one wave per SIMD, v_mfma_scale_f32_32x32x64_f8f6f4 back to back: %.2f ns per MFMA (= 64 cycles at %.2f GHz)\n", base, 64.0 / base);
#else
	std::printf("one wave per SIMD, v_mfma_f32_32x32x16_bf16 back to back: %.2f ns per MFMA (= 32 cycles at %.2f GHz)\n", base, 32.0 / base);
#endif
	std::printf("time per MFMA relative to that, with N instructions behind every MFMA:\n%-34s %6d %6d %6d %6d %6d %6d %6d\n", "N =", 1, 2, 3, 4, 6, 8, 12);
	sweep<kCvtPk>(out, ticks, iters, base);
	sweep<kPkMax>(out, ticks, iters, base);
	sweep<kAddF32>(out, ticks, iters, base);
	sweep<kPkAddF32>(out, ticks, iters, base);
	sweep<kMov64>(out, ticks, iters, base);
	sweep<kLshl>(out, ticks, iters, base);
	sweep<kDsWrite64>(out, ticks, iters, base);
	sweep<kDsRead128>(out, ticks, iters, base);
	sweep<kSaveExec>(out, ticks, iters, base);
	sweep<kStore4>(out, ticks, iters, base);
	sweep<kStore64>(out, ticks, iters, base);
	sweep<kStore4Wb>(out, ticks, iters, base);
	sweep<kStoreEvery8>(out, ticks, iters, base);
	sweep<kDepChain>(out, ticks, iters, base);
	sweep<kIndep3>(out, ticks, iters, base);
	sweep<kAccRead>(out, ticks, iters, base);
	return 0;
}
