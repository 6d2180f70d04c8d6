// Standalone reproducer #2 (no engine) for the MUBUF-store anomaly of DESIGN.md section 4b, after
// tools/probes/mubuf_sgpr_war.hip showed that an SALU write to the descriptor / soffset SGPRs
// right behind a buffer store is NOT seen by the store (the scalar operands are read at issue).
//
// What is left of the product's failing configuration, reduced to its memory behaviour:
//   * two tensors of 64-byte pixel records in ping-pong, one launch per "layer", 48 launches
//     back to back on one stream (no host synchronisation in between);
//   * a launch = persistent workgroups, TWO per CU (77 KB of LDS each), tiles of 8 rows x 32
//     pixels dealt to the XCDs in contiguous chunks (conv_tower_fp8_kernel's order), so that
//     the halo rows of a tile at a chunk boundary were written, one launch earlier, by a
//     workgroup on ANOTHER XCD (the product's stale tiles sat exactly there: the tile rows
//     above image rows 128 k = the last tiles of an XCD's chunk at the PS2 size);
//   * tile + one-pixel halo ring fetched by LDS-DMA (global_load_lds, or buffer_load ... lds),
//     double-buffered; outputs stored as 16 bytes per lane under a divergent EXEC mask, as
//     `global_store_dwordx4` or as `buffer_store_dwordx4 v, voffset, s[desc], soffset offen`.
// Every record carries its layer count; a layer checks that the centre and the four
// neighbours it read are all at the previous layer's count -- a stale or lost tile shows as an
// error count in the record (and as a wrong layer count at the end).
//
//     hipcc --offload-arch=gfx950 -O3 tools/probes/mubuf_pingpong.hip -o build/mubuf_pingpong
//     build/mubuf_pingpong [rounds]     one line per variant; exit code 1 if any variant failed
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                             \
	do {                                                                     \
		hipError_t e_ = (x);                                                 \
		if (e_ != hipSuccess) {                                              \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
			std::exit(2);                                                    \
		}                                                                    \
	} while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kH = 448, kW = 640;                 // the PS2 geometry
constexpr int kPitch = (kW + 31) / 32 * 32 + 2;   // tower layout: image at (1, 1), zero border
constexpr int kRows = (kH + 7) / 8 * 8 + 2;
constexpr int kTilesX = (kW + 31) / 32, kTilesY = (kH + 7) / 8;
constexpr int kTiles = kTilesX * kTilesY;         // 1120
constexpr int kTileBytes = 10 * 34 * 64;          // tile + halo ring
constexpr int kLds = 2 * kTileBytes + 34 * 1024;  // ~77 KB: two workgroups per CU

struct Params {
	const unsigned char *in;
	unsigned char *out;
	unsigned layer;  // the count this launch writes
};

template <bool MUBUF_LD, bool MUBUF_ST, bool WAIT>
__global__ __launch_bounds__(256, 2) void layer_kernel(Params p) {
	extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
	const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
	const int nwg = gridDim.x, bid = blockIdx.x;
	const int slot = nwg >= 8 ? (bid & 7) * (nwg >> 3) + (bid >> 3) : bid;
	const int fullTiles = kTiles / nwg * nwg;
	auto tileAt = [&](int k) {
		const int base = k * nwg;
		if (base < fullTiles) return base + slot;
		if (fullTiles == 0) return (k == 0 && slot < kTiles) ? slot : -1;
		return (base == fullTiles && base + bid < kTiles) ? base + bid : -1;
	};
	const __amdgpu_buffer_rsrc_t inRsrc =
	    __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char *>(p.in), 0, 0x7ffffff0, 0x00020000);
	const __amdgpu_buffer_rsrc_t outRsrc = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0x7ffffff0, 0x00020000);
	auto stage = [&](int tile, int buf) {
		const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
		const unsigned base = static_cast<unsigned>(((ty * 8) * kPitch + tx * 32) * 64);
		unsigned char *dst = smem + buf * kTileBytes;
#pragma unroll
		for (int k = 0; k < 6; ++k) {
			const int i = wave + 4 * k;
			const int q = i * 16 + (lane >> 2);
			if (i < 22 && q < 340) {
				const int r = q / 34, x = q - r * 34;
				const unsigned off = static_cast<unsigned>((r * kPitch + x) * 64 + (lane & 3) * 16);
				if constexpr (MUBUF_LD) {
					__builtin_amdgcn_raw_ptr_buffer_load_lds(inRsrc, (__attribute__((address_space(3))) void *)(dst + i * 1024), 16,
					    static_cast<int>(off), static_cast<int>(base), 0, 0);
				} else {
					__builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(p.in + base + off),
					    (__attribute__((address_space(3))) void *)(dst + i * 1024), 16, 0, 0);
				}
			}
		}
	};
	int round = 0;
	int tile = tileAt(0);
	if (tile >= 0) stage(tile, 0);
	if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
	__syncthreads();
	int buf = 0;
	for (; tile >= 0; buf ^= 1) {
		const int next = tileAt(++round);
		const int ty = tile / kTilesX, tx = tile - ty * kTilesX;
		if (next >= 0) stage(next, buf ^ 1);
		const unsigned char *t = smem + buf * kTileBytes;
		// thread -> 4 records x chunk: rows r0 .. r0 + 7, pixel px, 16-byte chunk c
		u32x4 res[4];
		int ok[4];
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int cell = j * 256 + tid;       // 1024 cells: (row 0..7, px 0..31, chunk 0..3)
			const int c = cell & 3, px = (cell >> 2) & 31, r = cell >> 7;
			auto rec = [&](int rr, int xx) { return *reinterpret_cast<const u32x4 *>(t + ((rr + 1) * 34 + xx + 1) * 64 + c * 16); };
			const u32x4 ce = rec(r, px), up = rec(r - 1, px), dn = rec(r + 1, px), le = rec(r, px - 1), ri = rec(r, px + 1);
			const int gy = ty * 8 + r, gx = tx * 32 + px;
			// neighbours outside the image are the zero border (count 0 for ever)
			const unsigned want = p.layer - 1;
			unsigned bad = ce.x != want;
			bad += gy > 0 && up.x != want;
			bad += gy + 1 < kH && dn.x != want;
			bad += gx > 0 && le.x != want;
			bad += gx + 1 < kW && ri.x != want;
			res[j] = u32x4{p.layer, ce.y + bad, static_cast<unsigned>(gy * kW + gx), ce.w + (bad ? p.layer : 0u)};
			ok[j] = gy < kH && gx < kW;
		}
		if constexpr (WAIT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
		__syncthreads();
#pragma unroll
		for (int j = 0; j < 4; ++j) {
			const int cell = j * 256 + tid;
			const int c = cell & 3, px = (cell >> 2) & 31, r = cell >> 7;
			const unsigned base = static_cast<unsigned>(((ty * 8 + 1) * kPitch + tx * 32 + 1) * 64);  // uniform
			const unsigned off = static_cast<unsigned>((r * kPitch + px) * 64 + c * 16);
			if (ok[j]) {
				if constexpr (MUBUF_ST) {
					__builtin_amdgcn_raw_buffer_store_b128(res[j], outRsrc, static_cast<int>(off), static_cast<int>(base), 0);
				} else {
					*reinterpret_cast<u32x4 *>(p.out + base + off) = res[j];
				}
			}
		}
		tile = next;
	}
}

template <bool MUBUF_LD, bool MUBUF_ST, bool WAIT>
bool runVariant(const char *name, int grid, int layers, int rounds) {
	const size_t bytes = static_cast<size_t>(kRows) * kPitch * 64;
	unsigned char *a = nullptr, *b = nullptr;
	CHECK(hipMalloc(&a, bytes));
	CHECK(hipMalloc(&b, bytes));
	auto kern = layer_kernel<MUBUF_LD, MUBUF_ST, WAIT>;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
	hipStream_t st;
	CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	std::vector<u32x4> host(bytes / 16);
	long badRecords = 0, badRounds = 0, firstRow = -1;
	for (int rd = 0; rd < rounds; ++rd) {
		CHECK(hipMemsetAsync(a, 0, bytes, st));
		CHECK(hipMemsetAsync(b, 0, bytes, st));
		for (int l = 1; l <= layers; ++l) {
			Params p{(l & 1) ? a : b, (l & 1) ? b : a, static_cast<unsigned>(l)};
			hipLaunchKernelGGL(kern, dim3(grid), dim3(256), kLds, st, p);
		}
		CHECK(hipStreamSynchronize(st));
		CHECK(hipMemcpy(host.data(), (layers & 1) ? b : a, bytes, hipMemcpyDeviceToHost));
		long bad = 0;
		for (int y = 0; y < kH; ++y) {
			for (int x = 0; x < kW; ++x) {
				for (int c = 0; c < 4; ++c) {
					const u32x4 v = host[(static_cast<size_t>(y + 1) * kPitch + x + 1) * 4 + c];
					if (v.x != static_cast<unsigned>(layers) || v.y != 0) {
						++bad;
						if (firstRow < 0) firstRow = y;
					}
				}
			}
		}
		badRecords += bad;
		badRounds += bad != 0;
	}
	std::printf("%-46s grid %4d: %ld of %d rounds bad, %ld bad records%s", name, grid, badRounds, rounds, badRecords,
	    firstRow >= 0 ? "" : "\n");
	if (firstRow >= 0) std::printf(" (first bad image row %ld)\n", firstRow);
	CHECK(hipStreamDestroy(st));
	CHECK(hipFree(a));
	CHECK(hipFree(b));
	return badRounds != 0;
}

int main(int argc, char **argv) {
	const int rounds = argc > 1 ? std::atoi(argv[1]) : 20;
	bool bad = false;
	for (int grid : {512, 256, 1024, 96}) {
		bad |= runVariant<false, false, true>("global_load_lds + global_store (product)", grid, 48, rounds);
		bad |= runVariant<false, true, true>("global_load_lds + buffer_store", grid, 48, rounds);
		bad |= runVariant<true, false, true>("buffer_load lds + global_store", grid, 48, rounds);
		bad |= runVariant<true, true, true>("buffer_load lds + buffer_store", grid, 48, rounds);
		bad |= runVariant<true, true, false>("buffer_load lds + buffer_store, no explicit wait", grid, 48, rounds);
	}
	std::printf(bad ? "RESULT: at least one variant lost or read stale data\n" : "RESULT: every variant clean\n");
	return bad ? 1 : 0;
}
