// Probe: what does a dependent kernel boundary cost as a function of the workgroup SHAPE?
// tools/probes/layer_exchange.hip measured 3.0 us per EMPTY launch of 144-224 workgroups of 512 threads
// with 64 KB of LDS, twice the 1.45 us MI355X_MICROARCH.md quotes for trivial 256-workgroup kernels.
// The flow net is 11 launches of such fat workgroups: is the excess the thread count, the LDS
// allocation, the register allocation, the grid, or the code size?
//   hipcc --offload-arch=gfx950 -O2 tools/probes/launch_shape.hip -o build/launch_shape && build/launch_shape
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

extern __shared__ unsigned char dynSmem[];

// REGS: forces a register allocation of about that many VGPRs (an array kept live through an opaque asm)
template <int THREADS, int REGS>
__global__ __launch_bounds__(THREADS) void shaped(unsigned *sink, int never) {
	float v[REGS];
#pragma unroll
	for (int i = 0; i < REGS; ++i) v[i] = (float)(threadIdx.x + i);
	if (never) {
#pragma unroll
		for (int i = 0; i < REGS; ++i) asm volatile("" : "+v"(v[i]));
		float s = 0;
#pragma unroll
		for (int i = 0; i < REGS; ++i) s += v[i];
		sink[threadIdx.x] = (unsigned)s + dynSmem[threadIdx.x];
	}
}

template <int THREADS, int REGS>
static double run(int grid, int lds, hipStream_t st, unsigned *sink, int launches) {
	auto k = shaped<THREADS, REGS>;
	CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
	hipEvent_t e0, e1;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	double best = 1e30;
	for (int r = 0; r < 4; ++r) {
		CHECK(hipEventRecord(e0, st));
		for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(k, dim3(grid), dim3(THREADS), lds, st, sink, 0);
		CHECK(hipEventRecord(e1, st));
		CHECK(hipStreamSynchronize(st));
		float ms = 0;
		CHECK(hipEventElapsedTime(&ms, e0, e1));
		if (r > 0 && ms < best) best = ms;
	}
	return best * 1e3 / launches;
}

int main() {
	hipStream_t st;
	CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	unsigned *sink;
	CHECK(hipMalloc(&sink, 4096 * 4));
	const int L = 200;
	std::printf("us per launch in a chain of %d dependent EMPTY launches (same stream, eager)\n", L);
	std::printf("%-28s %8s %8s %8s %8s\n", "threads, VGPRs, LDS", "grid 64", "grid 192", "grid 256", "grid 1024");
	const int grids[4] = {64, 192, 256, 1024};
	auto row = [&](const char *name, auto fn) {
		std::printf("%-28s", name);
		for (int g : grids) std::printf(" %8.2f", fn(g));
		std::printf("\n");
	};
	row("64, small, 0", [&](int g) { return run<64, 4>(g, 0, st, sink, L); });
	row("256, small, 0", [&](int g) { return run<256, 4>(g, 0, st, sink, L); });
	row("256, small, 64 KB", [&](int g) { return run<256, 4>(g, 64 * 1024, st, sink, L); });
	row("256, small, 140 KB", [&](int g) { return run<256, 4>(g, 140 * 1024, st, sink, L); });
	row("256, 200 regs, 140 KB", [&](int g) { return run<256, 200>(g, 140 * 1024, st, sink, L); });
	row("512, small, 0", [&](int g) { return run<512, 4>(g, 0, st, sink, L); });
	row("512, small, 64 KB", [&](int g) { return run<512, 4>(g, 64 * 1024, st, sink, L); });
	row("512, small, 140 KB", [&](int g) { return run<512, 4>(g, 140 * 1024, st, sink, L); });
	row("512, 120 regs, 140 KB", [&](int g) { return run<512, 120>(g, 140 * 1024, st, sink, L); });
	row("1024, small, 0", [&](int g) { return run<1024, 4>(g, 0, st, sink, L); });
	// the same chain replayed from a hipGraph (what the engine does)
	{
		hipGraph_t g;
		hipGraphExec_t ge;
		auto k = shaped<512, 4>;
		CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
		for (int l = 0; l < 13; ++l) hipLaunchKernelGGL(k, dim3(192), dim3(512), 140 * 1024, st, sink, 0);
		CHECK(hipStreamEndCapture(st, &g));
		CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
		hipEvent_t e0, e1;
		CHECK(hipEventCreate(&e0));
		CHECK(hipEventCreate(&e1));
		double best = 1e30;
		for (int r = 0; r < 5; ++r) {
			CHECK(hipEventRecord(e0, st));
			for (int i = 0; i < 20; ++i) CHECK(hipGraphLaunch(ge, st));
			CHECK(hipEventRecord(e1, st));
			CHECK(hipStreamSynchronize(st));
			float ms = 0;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (r > 0 && ms < best) best = ms;
		}
		std::printf("hipGraph of 13 x (512 threads, 140 KB, grid 192), 20 replays back to back: %.2f us per kernel node\n",
		    best * 1e3 / (20 * 13));
	}
	return 0;
}
