// Probe: how soon does a kernel start after the host decides, (a) launched the normal way, and
// (b) PRE-SUBMITTED behind hipStreamWaitValue32 and released by one host store to signal memory?
// (Could a synchronous frame's ~10 us submission cost be paid before the frame arrives?)
//   hipcc --offload-arch=gfx950 -O2 tools/probes/armed_launch.hip -o build/armed_launch && build/armed_launch
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CHECK(x)                                                         \
	do {                                                                 \
		hipError_t e_ = (x);                                             \
		if (e_ != hipSuccess) {                                          \
			std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); \
			std::exit(2);                                                \
		}                                                                \
	} while (0)

__global__ void mark(volatile unsigned *flag, unsigned value) {
	if (threadIdx.x == 0) {
		__hip_atomic_store(const_cast<unsigned *>(flag), value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

static double now() {
	return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
	hipStream_t st;
	CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
	unsigned *flag = nullptr;
	CHECK(hipHostMalloc(&flag, 64, hipHostMallocMapped));
	*flag = 0;
	unsigned *dflag = nullptr;
	CHECK(hipHostGetDevicePointer(reinterpret_cast<void **>(&dflag), flag, 0));
	unsigned *sig = nullptr;
	hipError_t e = hipExtMallocWithFlags(reinterpret_cast<void **>(&sig), 8, hipMallocSignalMemory);
	if (e != hipSuccess) {
		std::printf("hipMallocSignalMemory: %s -- no stream wait-value on this platform\n", hipGetErrorString(e));
		return 1;
	}
	*reinterpret_cast<volatile unsigned long long *>(sig) = 0;
	// a 14-kernel graph (the frame's launch count), the last kernel marks completion
	hipGraph_t g;
	hipGraphExec_t ge;
	std::vector<double> normal, armed;
	const int N = 300;
	for (int mode = 0; mode < 2; ++mode) {
		for (int i = 1; i <= N; ++i) {
			const unsigned seq = static_cast<unsigned>(mode * N + i);
			CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
			for (int k = 0; k < 13; ++k) hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, st, dflag + 8, seq);
			hipLaunchKernelGGL(mark, dim3(1), dim3(64), 0, st, dflag, seq);
			CHECK(hipStreamEndCapture(st, &g));
			CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
			double t0;
			if (mode == 0) {
				std::this_thread::sleep_for(std::chrono::microseconds(300));
				t0 = now();
				CHECK(hipGraphLaunch(ge, st));
			} else {
				e = hipStreamWaitValue32(st, sig, seq, hipStreamWaitValueEq, 0xffffffffu);
				if (e != hipSuccess) {
					std::printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e));
					return 1;
				}
				CHECK(hipGraphLaunch(ge, st));
				std::this_thread::sleep_for(std::chrono::microseconds(300));  // the frame "arrives" later
				t0 = now();
				__atomic_store_n(sig, seq, __ATOMIC_RELEASE);
			}
			while (*reinterpret_cast<volatile unsigned *>(flag) != seq) {
			}
			const double t1 = now();
			(mode ? armed : normal).push_back(t1 - t0);
			CHECK(hipStreamSynchronize(st));
			CHECK(hipGraphExecDestroy(ge));
			CHECK(hipGraphDestroy(g));
		}
	}
	std::sort(normal.begin(), normal.end());
	std::sort(armed.begin(), armed.end());
	std::printf("14 empty kernels, decision -> last kernel's host-visible mark:\n");
	std::printf("  hipGraphLaunch at decision time      p10 %.1f  p50 %.1f  p90 %.1f us\n", normal[N / 10], normal[N / 2], normal[N * 9 / 10]);
	std::printf("  pre-submitted, released by a store   p10 %.1f  p50 %.1f  p90 %.1f us\n", armed[N / 10], armed[N / 2], armed[N * 9 / 10]);
	return 0;
}
