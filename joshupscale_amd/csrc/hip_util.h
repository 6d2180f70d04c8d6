// HIP RAII helpers for the engine: device memory that starts zeroed, a stream,
// captured graphs, scoped device selection.  These are the MI355X equivalents
// of the reference's CUDA wrappers (reference core/include/JoshUpscale/core/cuda.h:
// CudaBuffer :61-110 "malloc + memset 0", CudaStream :240-295, CudaGraph(Exec)
// :170-238, DeviceContext :297-308), written against the HIP runtime directly.
#pragma once

#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstddef>
#include <stdexcept>
#include <string>
#include <utility>

namespace ju {

struct HipError : std::runtime_error {
	explicit HipError(hipError_t e, const char *what)
	    : std::runtime_error(std::string(what) + ": " + hipGetErrorName(e) + " (" +
	                         hipGetErrorString(e) + ")")
	    , code(e) {
	}
	hipError_t code;
};

inline void hipCheck(hipError_t e, const char *what) {
	if (e != hipSuccess) {
		throw HipError(e, what);
	}
}
#define JU_HIP(expr) ::ju::hipCheck((expr), #expr)

// Scoped hipSetDevice (restores the caller's device on exit).
class DeviceGuard {
public:
	explicit DeviceGuard(int device) {
		JU_HIP(hipGetDevice(&m_Prev));
		if (m_Prev != device) {
			JU_HIP(hipSetDevice(device));
			m_Switched = true;
		}
	}
	~DeviceGuard() {
		if (m_Switched) {
			(void)hipSetDevice(m_Prev);
		}
	}
	DeviceGuard(const DeviceGuard &) = delete;
	DeviceGuard &operator=(const DeviceGuard &) = delete;

private:
	int m_Prev = 0;
	bool m_Switched = false;
};

// Device allocation, zero-filled on creation (the recurrent state starts at
// zero: reference cuda.h:69-72, scripts/inference/onnx/inference.py:67-70).
class DeviceBuffer {
public:
	DeviceBuffer() = default;
	explicit DeviceBuffer(std::size_t bytes) : m_Bytes(bytes) {
		if (bytes == 0) return;
		JU_HIP(hipMalloc(&m_Ptr, bytes));
		// hipMemset on device memory may return before the fill has run, and the
		// engine's stream is non-blocking (it does not order against the null
		// stream): wait here so no later kernel can race the zero fill.
		hipError_t e = hipMemset(m_Ptr, 0, bytes);
		if (e == hipSuccess) e = hipStreamSynchronize(nullptr);
		if (e != hipSuccess) {
			(void)hipFree(m_Ptr);
			m_Ptr = nullptr;
			throw HipError(e, "hipMemset");
		}
	}
	~DeviceBuffer() {
		if (m_Ptr) (void)hipFree(m_Ptr);
	}
	DeviceBuffer(DeviceBuffer &&o) noexcept : m_Ptr(o.m_Ptr), m_Bytes(o.m_Bytes) {
		o.m_Ptr = nullptr;
		o.m_Bytes = 0;
	}
	DeviceBuffer &operator=(DeviceBuffer &&o) noexcept {
		if (this != &o) {
			if (m_Ptr) (void)hipFree(m_Ptr);
			m_Ptr = o.m_Ptr;
			m_Bytes = o.m_Bytes;
			o.m_Ptr = nullptr;
			o.m_Bytes = 0;
		}
		return *this;
	}
	DeviceBuffer(const DeviceBuffer &) = delete;
	DeviceBuffer &operator=(const DeviceBuffer &) = delete;

	void *get() const { return m_Ptr; }
	template <typename T>
	T *as() const {
		return static_cast<T *>(m_Ptr);
	}
	std::size_t bytes() const { return m_Bytes; }
	void zeroAsync(hipStream_t s) const {
		if (m_Ptr) JU_HIP(hipMemsetAsync(m_Ptr, 0, m_Bytes, s));
	}
	void upload(const void *src, std::size_t n) const {
		if (n > m_Bytes) throw std::out_of_range("DeviceBuffer::upload");
		JU_HIP(hipMemcpy(m_Ptr, src, n, hipMemcpyHostToDevice));
	}

private:
	void *m_Ptr = nullptr;
	std::size_t m_Bytes = 0;
};

// Pinned, device-visible host words (the resident tower's error report): released during
// constructor unwinding like every other member.
class PinnedWords {
public:
	PinnedWords() = default;
	explicit PinnedWords(std::size_t bytes) {
		JU_HIP(hipHostMalloc(&m_Host, bytes, hipHostMallocMapped));
		for (std::size_t i = 0; i < bytes / sizeof(unsigned); ++i) static_cast<unsigned *>(m_Host)[i] = 0;
		void *dev = nullptr;
		const hipError_t e = hipHostGetDevicePointer(&dev, m_Host, 0);
		if (e != hipSuccess) {
			(void)hipHostFree(m_Host);
			m_Host = nullptr;
			throw HipError(e, "hipHostGetDevicePointer");
		}
		m_Dev = static_cast<unsigned *>(dev);
	}
	~PinnedWords() {
		if (m_Host) (void)hipHostFree(m_Host);
	}
	PinnedWords(PinnedWords &&o) noexcept : m_Host(o.m_Host), m_Dev(o.m_Dev) {
		o.m_Host = nullptr;
		o.m_Dev = nullptr;
	}
	PinnedWords &operator=(PinnedWords &&o) noexcept {
		if (this != &o) {
			if (m_Host) (void)hipHostFree(m_Host);
			m_Host = o.m_Host;
			m_Dev = o.m_Dev;
			o.m_Host = nullptr;
			o.m_Dev = nullptr;
		}
		return *this;
	}
	PinnedWords(const PinnedWords &) = delete;
	PinnedWords &operator=(const PinnedWords &) = delete;
	volatile unsigned *host() const { return static_cast<volatile unsigned *>(m_Host); }
	unsigned *device() const { return m_Dev; }

private:
	void *m_Host = nullptr;
	unsigned *m_Dev = nullptr;
};

class Stream {
public:
	Stream() { JU_HIP(hipStreamCreateWithFlags(&m_Stream, hipStreamNonBlocking)); }
	~Stream() {
		if (m_Stream) (void)hipStreamDestroy(m_Stream);
	}
	Stream(const Stream &) = delete;
	Stream &operator=(const Stream &) = delete;
	operator hipStream_t() const { return m_Stream; }  // NOLINT
	void synchronize() const { JU_HIP(hipStreamSynchronize(m_Stream)); }
	// Poll for completion for up to `spinUs` microseconds before blocking: a frame
	// takes well under a millisecond, and the wake-up of a blocking wait costs
	// several microseconds of it.
	void synchronizeSpin(unsigned spinUs) const {
		if (spinUs) {
			const auto t0 = std::chrono::steady_clock::now();
			for (;;) {
				const hipError_t st = hipStreamQuery(m_Stream);
				if (st == hipSuccess) return;
				if (st != hipErrorNotReady) JU_HIP(st);
				if (std::chrono::steady_clock::now() - t0 > std::chrono::microseconds(spinUs)) break;
			}
		}
		JU_HIP(hipStreamSynchronize(m_Stream));
	}

private:
	hipStream_t m_Stream = nullptr;
};

// An instantiated, replayable graph of one frame's kernel sequence.
class GraphExec {
public:
	GraphExec() = default;
	~GraphExec() { reset(); }
	GraphExec(const GraphExec &) = delete;
	GraphExec &operator=(const GraphExec &) = delete;
	GraphExec(GraphExec &&o) noexcept : m_Graph(o.m_Graph), m_Exec(o.m_Exec) {
		o.m_Graph = nullptr;
		o.m_Exec = nullptr;
	}
	GraphExec &operator=(GraphExec &&o) noexcept {
		if (this != &o) {
			reset();
			m_Graph = o.m_Graph;
			m_Exec = o.m_Exec;
			o.m_Graph = nullptr;
			o.m_Exec = nullptr;
		}
		return *this;
	}

	// Capture everything `record` enqueues on `stream`.
	template <typename F>
	static GraphExec capture(hipStream_t stream, F &&record) {
		GraphExec g;
		JU_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
		try {
			record();
		} catch (...) {
			hipGraph_t dead = nullptr;
			(void)hipStreamEndCapture(stream, &dead);
			if (dead) (void)hipGraphDestroy(dead);
			throw;
		}
		JU_HIP(hipStreamEndCapture(stream, &g.m_Graph));
		JU_HIP(hipGraphInstantiate(&g.m_Exec, g.m_Graph, nullptr, nullptr, 0));
		return g;
	}
	void launch(hipStream_t stream) const { JU_HIP(hipGraphLaunch(m_Exec, stream)); }
	bool valid() const { return m_Exec != nullptr; }

private:
	void reset() {
		if (m_Exec) (void)hipGraphExecDestroy(m_Exec);
		if (m_Graph) (void)hipGraphDestroy(m_Graph);
		m_Exec = nullptr;
		m_Graph = nullptr;
	}
	hipGraph_t m_Graph = nullptr;
	hipGraphExec_t m_Exec = nullptr;
};

class Event {
public:
	Event() { JU_HIP(hipEventCreate(&m_Event)); }
	hipEvent_t get() const { return m_Event; }
	~Event() {
		if (m_Event) (void)hipEventDestroy(m_Event);
	}
	Event(const Event &) = delete;
	Event &operator=(const Event &) = delete;
	void record(hipStream_t s) const { JU_HIP(hipEventRecord(m_Event, s)); }
	void synchronize() const { JU_HIP(hipEventSynchronize(m_Event)); }
	static float elapsedMs(const Event &a, const Event &b) {
		float ms = 0.f;
		JU_HIP(hipEventElapsedTime(&ms, a.m_Event, b.m_Event));
		return ms;
	}

private:
	hipEvent_t m_Event = nullptr;
};

}  // namespace ju
