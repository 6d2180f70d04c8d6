// Standalone reproducer #3 (no engine) for the MUBUF-store anomaly of DESIGN.md section 4b.
//
// What the bisection of the product kernel said (tools/probes/fp8_stale_tile.sh, PROBE_SET=bisect;
// conv_tower_fp8_kernel at the PS2 size, two workgroups per CU, buffer stores for ...):
//     both store groups                      FAIL      (grid-dependent bytes)
//     the 16-bit stream's stores only        pass
//     the e4m3 copy's stores only            FAIL
//     ... with soffset = 0 (offset in VGPR)  pass
//     ... with `s_nop 7` behind every store  pass
//     ... with `s_waitcnt vmcnt(0)` behind   pass
// and the two earlier probes: an SALU write to the descriptor / soffset SGPRs behind the store is
// not seen by it (mubuf_sgpr_war.hip), and the memory pattern alone -- cross-XCD ping-pong, LDS-DMA
// loads, buffer stores under EXEC masks, two workgroups per CU -- is clean (mubuf_pingpong.hip).
//
// The failing store group is the one whose DATA registers are rewritten soonest behind the store:
//
//     ds_read2_b64 v[18:21], ...            this store's 16 bytes
//     s_waitcnt lgkmcnt(0)
//     buffer_store_dwordx4 v[18:21], v198, s[12:15], s8 offen
//     s_or_b64 exec, exec, s[0:1]           4 scalar instructions ...
//     s_and_b64 s[14:15], s[46:47], vcc
//     s_and_saveexec_b64 s[0:1], s[14:15]
//     s_cbranch_execz ...
//     v_add_u32_e32 v18, v0, v197           ... then a VALU write of v18 (the next store's LDS address)
//
// LLVM's hazard recogniser (GCNHazardRecognizer::createsVALUHazard) requires a wait state between
// a VMEM store of more than 64 bits and a VALU write of its data registers ONLY when the store has
// NO soffset register ("for MUBUF/MTBUF this hazard only exists if the instruction is not using a
// register in the soffset field") -- which matches "soffset = 0: pass" (the compiler then inserts
// the wait state itself).  This probe asks the hardware how long a buffer_store_dwordx4 WITH an
// soffset register needs its data registers when the vector-memory queue is busy: GAP scalar
// instructions between the store and a VALU overwrite of its first and last data register, with K
// LDS-DMA loads (global_load_lds_dwordx4) issued in front of the store so that the store is not at
// the head of the queue, at 1, 2 and 4 waves per SIMD.  A location that ends up holding the
// overwrite value 0x0bad0bad means the store read its data AFTER the later VALU instruction wrote it.
//
//     hipcc --offload-arch=gfx950 -O2 tools/probes/mubuf_store_data.hip -o build/mubuf_store_data
//     build/mubuf_store_data          exit code 1 if any store picked up overwritten data
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

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kBad = 0x0bad0badu;

#define NOPS_0 ""
#define NOPS_1 "s_nop 0\n"
#define NOPS_2 "s_nop 0\ns_nop 0\n"
#define NOPS_4 "s_nop 0\ns_nop 0\ns_nop 0\ns_nop 0\n"
#define NOPS_8 NOPS_4 NOPS_4
#define NOPS_16 NOPS_8 NOPS_8
#define DMA_0 ""
#define DMA_1 "global_load_lds_dwordx4 %[ga], off\n"
#define DMA_4 DMA_1 DMA_1 DMA_1 DMA_1
#define DMA_8 DMA_4 DMA_4

// MODE 0: buffer_store with an soffset REGISTER; 1: buffer_store with soffset 0 (offset in the
// VGPR); 2: global_store.  The data sits in v[100:103] (fixed, so that the overwrite can name them).
// how the data registers are rewritten behind the store: by the VALU, or by an LDS read's RETURN
// (the product: the next store's `ds_read2_b64 v[18:21]` lands in the previous store's registers)
#define OVW_VALU "v_mov_b32 v100, 0x0bad0bad\nv_mov_b32 v103, 0x0bad0bad\n"
#define OVW_LDS "ds_read_b128 v[100:103], %[la]\n"
#define JU_BODY(STORE, DMA, NOPS, OVW)                                                                    \
	asm volatile("s_mov_b32 s40, %[lo]\n"                                                            \
	             "s_mov_b32 s41, %[hi]\n"                                                            \
	             "s_mov_b32 s42, 0x7ffffff0\n"                                                       \
	             "s_mov_b32 s43, 0x00020000\n"                                                       \
	             "s_mov_b32 s44, %[soff]\n"                                                          \
	             "s_mov_b32 m0, %[lds]\n"                                                            \
	             "v_mov_b32 v100, %[d0]\n"                                                           \
	             "v_mov_b32 v101, %[d1]\n"                                                           \
	             "v_mov_b32 v102, %[d2]\n"                                                           \
	             "v_mov_b32 v103, %[d3]\n"                                                           \
	             "s_nop 4\n" DMA STORE NOPS OVW                                                      \
	             "s_waitcnt vmcnt(0) lgkmcnt(0)\n"                                                   \
	             :                                                                                   \
	             : [lo] "s"(lo), [hi] "s"(hi), [soff] "s"(soff), [lds] "s"(ldsOff), [d0] "v"(d0), [d1] "v"(d1),      \
	             [d2] "v"(d2), [d3] "v"(d3), [voff] "v"(voff), [vall] "v"(vall), [ga] "v"(ga), [fa] "v"(fa), [la] "v"(la)  \
	             : "s40", "s41", "s42", "s43", "s44", "v100", "v101", "v102", "v103", "memory")

#define ST_SOFF "buffer_store_dwordx4 v[100:103], %[voff], s[40:43], s44 offen\n"
#define ST_NOSOFF "buffer_store_dwordx4 v[100:103], %[vall], s[40:43], 0 offen\n"
#define ST_GLOBAL "global_store_dwordx4 %[fa], v[100:103], off\n"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// SIB: every second group of four waves of the workgroup (waves 4-7, 12-15: the SAME SIMDs as waves
// 0-3, 8-11) runs a matrix-core loop instead of stores -- the product's situation: the other
// wave of the SIMD is in its MFMA K loop and streams accumulators through the VGPR ports.
template <int MODE, int K, int GAP, bool SIB = false>
__global__ void probe_kernel(u32x4 *out, const u32x4 *scratch, int iters) {
	extern __shared__ unsigned char lds[];
	if (SIB && ((threadIdx.x >> 8) & 1)) {
		f32x16 acc = {};
		bf16x8 a, b;
		for (int i = 0; i < 8; ++i) {
			a[i] = static_cast<__bf16>(static_cast<float>(threadIdx.x & 7) * 0.125f);
			b[i] = static_cast<__bf16>(static_cast<float>(i) * 0.25f);
		}
		for (int i = 0; i < iters * 24; ++i) {
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
			acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, acc, 0, 0, 0);
		}
		if (acc[0] == 12345.678f) lds[1] = 1;  // (keeps the loop alive)
		return;
	}
	// store role: its own dense index among the storing threads
	const unsigned stid = SIB ? ((threadIdx.x >> 9) << 8) + (threadIdx.x & 255) : threadIdx.x;
	const unsigned sper = SIB ? blockDim.x / 2 : blockDim.x;
	const unsigned gtid = blockIdx.x * sper + stid;
	const unsigned total = gridDim.x * sper;
	const unsigned long long base = reinterpret_cast<unsigned long long>(out);
	const unsigned lo = static_cast<unsigned>(base), hi = static_cast<unsigned>(base >> 32) & 0xffffu;
	const unsigned ldsOff = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<unsigned long long>(
	                            (__attribute__((address_space(3))) unsigned char *)lds)) + (threadIdx.x >> 6) * 1024u);
	const u32x4 *ga = scratch + gtid;  // the LDS-DMA loads' source (16 bytes per lane)
	// LDS words 32 KB.. hold the overwrite pattern (MODE 3 reads them back into the data registers)
	const unsigned la = 32768u + (threadIdx.x & 1023u) * 16u;
	*reinterpret_cast<u32x4 *>(lds + la) = u32x4{kBad, kBad, kBad, kBad};
	__syncthreads();
	for (int it = 0; it < iters; ++it) {
		const unsigned soff = static_cast<unsigned>(it) * total * 16u;
		const unsigned voff = gtid * 16u, vall = soff + voff;
		unsigned char *fa = reinterpret_cast<unsigned char *>(out) + soff + voff;
		const unsigned d0 = gtid, d1 = static_cast<unsigned>(it), d2 = ~gtid, d3 = gtid ^ 0x5a5a5a5au;
#define JU_PICK(DMA, NOPS)                                                \
	if constexpr (MODE == 0) JU_BODY(ST_SOFF, DMA, NOPS, OVW_VALU);        \
	else if constexpr (MODE == 1) JU_BODY(ST_NOSOFF, DMA, NOPS, OVW_VALU); \
	else if constexpr (MODE == 2) JU_BODY(ST_GLOBAL, DMA, NOPS, OVW_VALU); \
	else if constexpr (MODE == 3) JU_BODY(ST_SOFF, DMA, NOPS, OVW_LDS);    \
	else JU_BODY(ST_GLOBAL, DMA, NOPS, OVW_LDS)
#define JU_GAPS(DMA)                                    \
	if constexpr (GAP == 0) { JU_PICK(DMA, NOPS_0); }   \
	else if constexpr (GAP == 1) { JU_PICK(DMA, NOPS_1); } \
	else if constexpr (GAP == 2) { JU_PICK(DMA, NOPS_2); } \
	else if constexpr (GAP == 4) { JU_PICK(DMA, NOPS_4); } \
	else if constexpr (GAP == 8) { JU_PICK(DMA, NOPS_8); } \
	else { JU_PICK(DMA, NOPS_16); }
		if constexpr (K == 0) { JU_GAPS(DMA_0) }
		else if constexpr (K == 4) { JU_GAPS(DMA_4) }
		else { JU_GAPS(DMA_8) }
	}
	if (threadIdx.x == 100000) lds[0] = 0;
}

struct Result {
	long overwritten = 0, other = 0;
};

template <int MODE, int K, int GAP, bool SIB = false>
Result run(int wavesPerSimd, int iters, int reps) {
	hipDeviceProp_t prop;
	CHECK(hipGetDeviceProperties(&prop, 0));
	// SIB: ONE workgroup per CU of 4 * wavesPerSimd waves, half of them storing
	const int grid = SIB ? prop.multiProcessorCount : prop.multiProcessorCount * wavesPerSimd;
	const int block = SIB ? 256 * wavesPerSimd : 256;
	const int ldsBytes = SIB ? 90 * 1024 : (wavesPerSimd == 1 ? 90 * 1024 : (wavesPerSimd == 2 ? 60 * 1024 : 36 * 1024));
	const size_t threads = static_cast<size_t>(grid) * (SIB ? block / 2 : block);
	const size_t bytes = threads * iters * 16;
	u32x4 *buf = nullptr, *scratch = nullptr;
	CHECK(hipMalloc(&buf, bytes));
	CHECK(hipMalloc(&scratch, threads * 16));
	CHECK(hipMemset(scratch, 0x11, threads * 16));
	auto kern = probe_kernel<MODE, K, GAP, SIB>;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsBytes));
	std::vector<u32x4> host(bytes / 16);
	Result r;
	for (int rep = 0; rep < reps; ++rep) {
		CHECK(hipMemset(buf, 0xef, bytes));
		hipLaunchKernelGGL(kern, dim3(grid), dim3(block), ldsBytes, nullptr, buf, scratch, iters);
		CHECK(hipDeviceSynchronize());
		CHECK(hipMemcpy(host.data(), buf, bytes, hipMemcpyDeviceToHost));
		for (size_t i = 0; i < host.size(); ++i) {
			const unsigned it = static_cast<unsigned>(i / threads), gtid = static_cast<unsigned>(i % threads);
			const u32x4 v = host[i];
			if (v.x == kBad || v.w == kBad) ++r.overwritten;
			else if (v.x != gtid || v.y != it || v.z != ~gtid || v.w != (gtid ^ 0x5a5a5a5au)) ++r.other;
		}
	}
	CHECK(hipFree(buf));
	CHECK(hipFree(scratch));
	return r;
}

template <int MODE, int K, int GAP>
bool report(const char *what) {
	bool bad = false;
	for (int w : {1, 2, 4}) {
		const Result r = run<MODE, K, GAP>(w, 64, 4);
		std::printf("%-30s DMA loads in front %d  gap %d  waves/SIMD %d : overwritten data stored %ld  other %ld\n", what, K, GAP, w,
		    r.overwritten, r.other);
		bad = bad || r.overwritten || r.other;
	}
	return bad;
}

template <int MODE, int GAP>
bool reportSib(const char *what) {
	bool bad = false;
	for (int w : {2, 4}) {
		const Result r = run<MODE, 0, GAP, true>(w, 64, 4);
		std::printf("%-30s MFMA sibling on the SIMD      gap %2d  waves/SIMD %d : overwritten data stored %ld  other %ld\n", what, GAP, w,
		    r.overwritten, r.other);
		bad = bad || r.overwritten || r.other;
	}
	return bad;
}

template <int MODE>
bool sweepSib(const char *what) {
	bool bad = false;
	bad |= reportSib<MODE, 0>(what);
	bad |= reportSib<MODE, 1>(what);
	bad |= reportSib<MODE, 2>(what);
	bad |= reportSib<MODE, 4>(what);
	bad |= reportSib<MODE, 8>(what);
	bad |= reportSib<MODE, 16>(what);
	return bad;
}

template <int MODE>
bool sweep(const char *what) {
	bool bad = false;
	bad |= report<MODE, 0, 0>(what);
	bad |= report<MODE, 0, 1>(what);
	bad |= report<MODE, 0, 4>(what);
	bad |= report<MODE, 4, 0>(what);
	bad |= report<MODE, 4, 1>(what);
	bad |= report<MODE, 4, 2>(what);
	bad |= report<MODE, 4, 4>(what);
	bad |= report<MODE, 4, 8>(what);
	bad |= report<MODE, 8, 0>(what);
	bad |= report<MODE, 8, 4>(what);
	bad |= report<MODE, 8, 8>(what);
	return bad;
}

int main() {
	const bool a = sweep<0>("buffer_store, soffset SGPR");
	const bool b = sweep<1>("buffer_store, soffset 0");
	const bool c = sweep<2>("global_store");
	const bool f = sweep<3>("buffer_store soffset SGPR, LDS ovw");
	const bool g = sweep<4>("global_store, LDS-return ovw");
	const bool d = sweepSib<0>("buffer_store, soffset SGPR");
	const bool e = sweepSib<2>("global_store");
	std::printf("RESULT (VALU overwrite): soffset-SGPR %s, soffset-0 %s, global %s; with an MFMA sibling: soffset-SGPR %s, global %s\n",
	    a ? "AFFECTED" : "clean", b ? "AFFECTED" : "clean", c ? "AFFECTED" : "clean", d ? "AFFECTED" : "clean",
	    e ? "AFFECTED" : "clean");
	std::printf("RESULT (overwrite by an LDS read's return): soffset-SGPR %s, global %s\n", f ? "AFFECTED" : "clean",
	    g ? "AFFECTED" : "clean");
	return (a || b || c || d || e || f || g) ? 1 : 0;
}
