// Standalone reproducer (no engine) for the MUBUF-store anomaly of DESIGN.md section 4b:
// conv_tower_fp8_kernel produced grid-dependent bytes as soon as its 16-byte output stores were
// `buffer_store_dwordx4 v, voffset, s[desc:desc+3], soffset offen` instead of
// `global_store_dwordx4` -- only with two workgroups per CU (two waves per SIMD).
//
// What the product ISA looks like around those stores (hipcc 7.2, -O3):
//
//     s_mov_b32 s14, s6 ; s_mov_b32 s15, s7          descriptor words 2, 3 (num_records, flags)
//     buffer_store_dwordx4 v[18:21], v198, s[12:15], s40 offen
//     s_or_b64  exec, exec, s[0:1]
//     s_and_b64 s[14:15], s[50:51], vcc              <- words 2, 3 of the descriptor REUSED as
//     s_and_saveexec_b64 s[0:1], s[14:15]               scratch two SALU instructions later
//
// (and the soffset register is recycled the same way: `buffer_store ... s0 offen` directly
// followed by `v_cmp_gt_i32_e64 s[0:1], ...`).  The compiler is entitled to do that: an
// instruction's SGPR operands are supposed to be read when it issues.  A MUBUF store with a
// 128-bit descriptor AND an soffset register reads FIVE SGPRs (a global store with an saddr
// pair reads two); this probe asks the hardware whether an SALU write to those SGPRs right
// behind the store can still be seen by it -- the hypothesis that would explain "buffer stores
// only", "two waves per SIMD only" (a second wave competing for the SIMD's scalar read port)
// and "rare".
//
// The kernel issues, per thread and iteration, ONE 16-byte store to a location that is its own
// (every location is written exactly once with a value that names it), in hand-written ISA:
//
//     mode 0   descriptor word 2 (num_records) := 0 right behind the store: if the store sees
//              it, it is range-checked away -> the location keeps its sentinel   ("lost")
//     mode 1   soffset := soffset + TRAP right behind the store: if the store sees it, the
//              bytes land in a trap region behind the data                        ("misplaced")
//     mode 2   the same store as `global_store_dwordx4` + the same SALU traffic   (control)
//
// with GAP = 0 .. 4 independent SALU instructions (`s_nop 0`) between the store and the
// overwrite, at 1, 2 and 4 waves per SIMD.  Build and run on the GPU box:
//
//     hipcc --offload-arch=gfx950 -O2 tools/probes/mubuf_sgpr_war.hip -o build/mubuf_sgpr_war
//     build/mubuf_sgpr_war            (prints one line per configuration, exit code 1 if any store
//                                      was lost or misplaced)
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
	do {                                                                              \
		hipError_t e_ = (x);                                                          \
		if (e_ != hipSuccess) {                                                       \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));              \
			std::exit(2);                                                             \
		}                                                                             \
	} while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n"
#define NOPS_2 "s_nop 0\ns_nop 0\n"
#define NOPS_3 "s_nop 0\ns_nop 0\ns_nop 0\n"
#define NOPS_4 "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n"

// One store + hazard sequence.  The descriptor lives in s[40:43], the soffset in s44: fixed
// registers, so that the instruction sequence is exactly the one written here.
template <int MODE, int GAP>
__device__ __forceinline__ void storeOnce(u32x4 data, unsigned voff, unsigned lo, unsigned hi, unsigned nrec,
    unsigned soff, unsigned trap, void *flat) {
	if constexpr (MODE == 2) {
		// control: the same bytes through a global store, the same scalar traffic behind it
		asm volatile(
		    "s_mov_b32 s40, %[lo]\n"
		    "s_mov_b32 s41, %[hi]\n"
		    "s_mov_b32 s42, %[nrec]\n"
		    "s_mov_b32 s43, 0x00020000\n"
		    "s_mov_b32 s44, %[soff]\n"
		    "s_nop 4\n"
		    "global_store_dwordx4 %[addr], %[d], off\n"
		    "s_mov_b32 s42, 0\n"
		    "s_add_u32 s44, s44, %[trap]\n"
		    :
		    : [lo] "s"(lo), [hi] "s"(hi), [nrec] "s"(nrec), [soff] "s"(soff), [trap] "s"(trap), [d] "v"(data),
		    [addr] "v"(flat)
		    : "s40", "s41", "s42", "s43", "s44", "memory");
		return;
	}
#define JU_SEQ(NOPS, CLOBBER)                                                                          \
	asm volatile("s_mov_b32 s40, %[lo]\n"                                                              \
	             "s_mov_b32 s41, %[hi]\n"                                                              \
	             "s_mov_b32 s42, %[nrec]\n"                                                            \
	             "s_mov_b32 s43, 0x00020000\n"                                                         \
	             "s_mov_b32 s44, %[soff]\n"                                                            \
	             "s_nop 4\n"                                                                           \
	             "buffer_store_dwordx4 %[d], %[voff], s[40:43], s44 offen\n" NOPS CLOBBER              \
	             :                                                                                     \
	             : [lo] "s"(lo), [hi] "s"(hi), [nrec] "s"(nrec), [soff] "s"(soff), [trap] "s"(trap),  \
	             [d] "v"(data), [voff] "v"(voff)                                                       \
	             : "s40", "s41", "s42", "s43", "s44", "memory")
	if constexpr (MODE == 0) {
		if constexpr (GAP == 0) JU_SEQ(NOPS_0, "s_mov_b32 s42, 0\n");
		if constexpr (GAP == 1) JU_SEQ(NOPS_1, "s_mov_b32 s42, 0\n");
		if constexpr (GAP == 2) JU_SEQ(NOPS_2, "s_mov_b32 s42, 0\n");
		if constexpr (GAP == 3) JU_SEQ(NOPS_3, "s_mov_b32 s42, 0\n");
		if constexpr (GAP == 4) JU_SEQ(NOPS_4, "s_mov_b32 s42, 0\n");
	} else {
		if constexpr (GAP == 0) JU_SEQ(NOPS_0, "s_add_u32 s44, s44, %[trap]\n");
		if constexpr (GAP == 1) JU_SEQ(NOPS_1, "s_add_u32 s44, s44, %[trap]\n");
		if constexpr (GAP == 2) JU_SEQ(NOPS_2, "s_add_u32 s44, s44, %[trap]\n");
		if constexpr (GAP == 3) JU_SEQ(NOPS_3, "s_add_u32 s44, s44, %[trap]\n");
		if constexpr (GAP == 4) JU_SEQ(NOPS_4, "s_add_u32 s44, s44, %[trap]\n");
	}
#undef JU_SEQ
}

// data region: iters * threads 16-byte locations; trap region of the same size behind it
template <int MODE, int GAP>
__global__ void probe_kernel(uint4 *out, unsigned regionBytes, int iters) {
	extern __shared__ unsigned char lds[];  // (only to set the number of workgroups per CU)
	const unsigned gtid = blockIdx.x * blockDim.x + threadIdx.x;
	const unsigned total = gridDim.x * blockDim.x;
	const unsigned long long base = reinterpret_cast<unsigned long long>(out);
	const unsigned lo = static_cast<unsigned>(base), hi = static_cast<unsigned>(base >> 32) & 0xffffu;
	float busy = static_cast<float>(gtid);
	for (int it = 0; it < iters; ++it) {
		const unsigned soff = static_cast<unsigned>(it) * total * 16u;  // uniform
		const unsigned voff = gtid * 16u;
		const u32x4 data = {gtid, static_cast<unsigned>(it), ~gtid, 0xc0ffee00u + GAP};
		storeOnce<MODE, GAP>(data, voff, lo, hi, 2u * regionBytes, soff, regionBytes,
		    reinterpret_cast<unsigned char *>(out) + soff + voff);
		// vector work with scalar operands between the stores (the product's epilogue has it too)
		busy = busy * 1.0001f + static_cast<float>(it);
	}
	if (busy == -1.0f) out[0].x = 0;  // (keeps `busy` alive)
	if (threadIdx.x == 100000) lds[0] = 0;
}

struct Result {
	long lost = 0, misplaced = 0, wrong = 0;
};

template <int MODE, int GAP>
Result run(int wavesPerSimd, int iters, int reps) {
	// 256 CUs x 4 SIMDs; a workgroup = 4 waves (one per SIMD); 90 KB of LDS each for 1 / CU, 40 KB for
	// 2 / CU ... (160 KB per CU)
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	const int block = 256;
	const int grid = cus * wavesPerSimd;
	const int ldsBytes = wavesPerSimd == 1 ? 90 * 1024 : (wavesPerSimd == 2 ? 60 * 1024 : 36 * 1024);
	const size_t threads = static_cast<size_t>(grid) * block;
	const size_t regionBytes = threads * iters * 16;
	uint4 *buf = nullptr;
	CHECK(hipMalloc(&buf, 2 * regionBytes));
	auto kern = probe_kernel<MODE, GAP>;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
	std::vector<uint4> host(2 * regionBytes / 16);
	Result r;
	for (int rep = 0; rep < reps; ++rep) {
		CHECK(hipMemset(buf, 0xef, 2 * regionBytes));  // (0xefefefef: neither data nor zero)
		hipLaunchKernelGGL(kern, dim3(grid), dim3(block), ldsBytes, nullptr, buf, static_cast<unsigned>(regionBytes), iters);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemcpy(host.data(), buf, 2 * regionBytes, hipMemcpyDeviceToHost));
		const size_t n = regionBytes / 16;
		for (size_t i = 0; i < n; ++i) {
			const unsigned it = static_cast<unsigned>(i / threads), gtid = static_cast<unsigned>(i % threads);
			const uint4 v = host[i];
			if (v.x == 0xefefefefu && v.y == 0xefefefefu) ++r.lost;
			else if (v.x != gtid || v.y != it || v.z != ~gtid) ++r.wrong;
			const uint4 t = host[n + i];
			if (t.x != 0xefefefefu || t.y != 0xefefefefu) ++r.misplaced;
		}
	}
	CHECK(hipFree(buf));
	return r;
}

template <int MODE, int GAP>
bool report(const char *what) {
	bool bad = false;
	for (int w : {1, 2, 4}) {
		const Result r = run<MODE, GAP>(w, 64, 6);
		std::printf("%-34s gap %d  waves/SIMD %d : lost %ld  misplaced %ld  wrong %ld\n", what, GAP, w, r.lost, r.misplaced,
		    r.wrong);
		bad = bad || r.lost || r.misplaced || r.wrong;
	}
	return bad;
}

int main() {
	bool bad = false;
	bad |= report<2, 0>("global_store (control)");
	bad |= report<0, 0>("buffer_store, num_records := 0");
	bad |= report<0, 1>("buffer_store, num_records := 0");
	bad |= report<0, 2>("buffer_store, num_records := 0");
	bad |= report<0, 3>("buffer_store, num_records := 0");
	bad |= report<0, 4>("buffer_store, num_records := 0");
	bad |= report<1, 0>("buffer_store, soffset += TRAP");
	bad |= report<1, 1>("buffer_store, soffset += TRAP");
	bad |= report<1, 2>("buffer_store, soffset += TRAP");
	bad |= report<1, 3>("buffer_store, soffset += TRAP");
	bad |= report<1, 4>("buffer_store, soffset += TRAP");
	std::printf(bad ? "RESULT: a store saw an SGPR written BEHIND it\n" : "RESULT: no store was affected\n");
	return bad ? 1 : 0;
}
