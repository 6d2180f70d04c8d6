// Probe: round-trip latency of a self-validating 16-byte mailbox slot between two workgroups, by store flavour
// and by placement (same XCD / other XCD).  The resident tower publishes its edge ring with write-through
// (sc1) stores and sweeps with sc1 loads: ~3.5 k cycles from publish to visible.  MI355X_MICROARCH.md says
// sc1 loads are L2-served and that a PLAIN store keeps the line in the XCD's L2 while an sc1 store drops it --
// so between workgroups of ONE XCD a plain-stored slot should travel through L2 only.  How much faster?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/slot_latency.hip -o build/slot_latency && build/slot_latency
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                         \
	do {                                                                 \
		hipError_t e_ = (x);                                             \
		if (e_ != hipSuccess) {                                          \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			std::exit(2);                                                \
		}                                                                \
	} while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int kSc1 = 16;

struct Params {
	unsigned char *mail;   // [2 directions][64 lanes] 16-byte slots, 4 KB apart per direction
	unsigned *xcc;         // [grid] XCC_ID of every workgroup
	unsigned long long *ticks;  // [1] s_memrealtime ticks of workgroup A's loop
	unsigned *error;
	int a, b;              // the two workgroups that play
	int rounds;
	int storeMode;         // 0 sc1 (write-through), 1 plain, 2 sc0
	int lanes;             // active lanes per store / load (1 .. 64)
	int slots;             // 16-byte slots per lane and direction (1 .. 4): payload = lanes x slots x 16 B
};

__device__ inline void storeSlot(const __amdgpu_buffer_rsrc_t r, unsigned off, u32x4 v, int mode) {
	if (mode == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, kSc1);
	else if (mode == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);
	else __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 1);
}

__global__ __launch_bounds__(64) void pingpong(Params p) {
	const int wg = blockIdx.x, lane = threadIdx.x;
	if (lane == 0) {
		unsigned id;
		asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
		p.xcc[wg] = id & 0xf;
	}
	if (wg != p.a && wg != p.b) return;
	const bool isA = wg == p.a;
	const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(p.mail, 0, 2 * 16384, 0x00020000);
	const unsigned mine = (isA ? 0u : 16384u) + lane * 16u;    // where I write
	const unsigned theirs = (isA ? 16384u : 0u) + lane * 16u;  // where I read
	const bool active = lane < p.lanes;
	unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
	for (int k = 1; k <= p.rounds; ++k) {
		const unsigned tag = (unsigned)k;
		if (isA && active) {
			for (int s = 0; s < p.slots; ++s) storeSlot(r, mine + s * 1024, u32x4{tag, tag, tag, tag}, p.storeMode);
		}
		// wait for the partner's slots of round k
		unsigned spins = 0;
		bool ok = !active;
		while (!__all(ok)) {
			ok = true;
			if (active) {
				for (int s = 0; s < p.slots; ++s) {
					const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, theirs + s * 1024, 0, kSc1);
					ok = ok && v[0] == tag && v[1] == tag && v[2] == tag && v[3] == tag;
				}
			}
			if (++spins > 4000000u) {
				if (lane == 0) *p.error = 0x100u + (isA ? 0 : 1);
				return;
			}
		}
		if (!isA && active) {
			for (int s = 0; s < p.slots; ++s) storeSlot(r, mine + s * 1024, u32x4{tag, tag, tag, tag}, p.storeMode);
		}
	}
	if (isA && lane == 0) p.ticks[0] = __builtin_amdgcn_s_memrealtime() - t0;
}

int main() {
	Params p{};
	const int grid = 64;
	CHECK(hipMalloc(&p.mail, 2 * 16384));
	CHECK(hipMalloc(&p.xcc, grid * 4));
	CHECK(hipMalloc(&p.ticks, 8));
	unsigned *herr = nullptr;
	CHECK(hipHostMalloc(&herr, 64, hipHostMallocMapped));
	CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&p.error), herr, 0));
	p.rounds = 2000;
	// who shares an XCD?  (observed: blocks b and b + 8)
	p.a = p.b = -1;
	p.lanes = 1;
	p.slots = 1;
	hipLaunchKernelGGL(pingpong, dim3(grid), dim3(64), 0, 0, p);
	CHECK(hipDeviceSynchronize());
	unsigned xcc[64];
	CHECK(hipMemcpy(xcc, p.xcc, sizeof(xcc), hipMemcpyDeviceToHost));
	std::printf("XCC_ID of blocks 0..15:");
	for (int i = 0; i < 16; ++i) std::printf(" %u", xcc[i]);
	std::printf("\n");
	int same = -1, other = -1;
	for (int i = 1; i < grid; ++i) {
		if (same < 0 && xcc[i] == xcc[0]) same = i;
		if (other < 0 && xcc[i] != xcc[0]) other = i;
	}
	std::printf("block 0 shares its XCD with block %d; block %d is on another one\n", same, other);
	const char *modes[3] = {"sc1 (write-through)", "plain", "sc0"};
	for (int lanes : {1, 64}) {
		for (int slots : {1, 4}) {
			for (int place = 0; place < 2; ++place) {
				for (int mode = 0; mode < 3; ++mode) {
					if (place == 1 && mode != 0) continue;  // (a plain store never reaches another XCD's reader in time)
					p.a = 0;
					p.b = place == 0 ? same : other;
					p.storeMode = mode;
					p.lanes = lanes;
					p.slots = slots;
					double best = 1e30;
					for (int rep = 0; rep < 3; ++rep) {
						CHECK(hipMemset(p.mail, 0, 2 * 16384));
						*herr = 0;
						hipLaunchKernelGGL(pingpong, dim3(grid), dim3(64), 0, 0, p);
						CHECK(hipDeviceSynchronize());
						unsigned long long t;
						CHECK(hipMemcpy(&t, p.ticks, 8, hipMemcpyDeviceToHost));
						if (*herr) {
							std::printf("  TIMEOUT 0x%x\n", *herr);
							break;
						}
						const double us = t / 100.0 / p.rounds;  // s_memrealtime: 100 MHz
						if (us < best) best = us;
					}
					std::printf("%2d lane(s) x %d slot(s), %-10s %-20s round trip %.3f us  (one way %.0f ns)\n", lanes, slots,
					    place == 0 ? "same XCD" : "other XCD", modes[mode], best, best * 500.0);
				}
			}
		}
	}
	return 0;
}
