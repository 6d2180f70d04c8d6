// JoshUpscale::core -- C++ plugin surface of the MI355X-native runtime.
//
// Source- and ABI-compatible restatement of the interface that the reference's
// AviSynth and OBS plugins compile against (reference
// core/public/JoshUpscale/core.h:17-96), so that avisynth_plugin/src/main.cc and
// obs_plugin/src/*.cc build and link against this library unchanged.  Everything
// declared here is implemented in joshupscale_amd/csrc/core_api.cpp as a thin
// shim over the C ABI in include/joshupscale_amd.h; errors of the C layer are
// re-raised as C++ exceptions because that is how the reference reports them.
//
// ABI notes (what must not change):
//   * Image is {void*, uint8 enum, ptrdiff_t, size_t, size_t} in this order;
//   * Runtime has a virtual destructor followed by processImage in its vtable
//     and four size_t members read by inline getters in the CALLER's code;
//   * GraphicsResourceImage has a virtual destructor and one Image member;
//   * LogSink has a single virtual call operator.
//
// Differences from the reference, all inside the implementation:
//   * DataLocation::CUDA means "pointer valid on the HIP device" here;
//   * GRAPHICS_RESOURCE images are OpenGL textures registered through HIP-GL
//     interop (getGLImage / getGLDeviceIndex); there are no D3D11 entry points
//     (Linux only);
//   * processImage throws std::invalid_argument on a size mismatch instead of
//     asserting (reference core/src/core.cc:179-182).
#pragma once

#include <cstddef>
#include <cstdint>
#include <filesystem>
#include <string>

#if defined(__GNUC__)
#define JOSHUPSCALE_EXPORT __attribute__((visibility("default")))
#else
#define JOSHUPSCALE_EXPORT
#endif

namespace JoshUpscale {

namespace core {

// ---------------------------------------------------------------- logging --
// (uint8_t-backed, values 0, 1, 2: the plugins pass them through a virtual call)
enum class LogLevel : std::uint8_t {
	INFO,     // 0
	WARNING,  // 1
	ERROR     // 2
};

// Borrowed for the lifetime of the process once installed (OBS installs a
// static instance: reference obs_plugin/src/plugin.cc:93-106).  The only virtual
// function is the call operator.
struct LogSink {
	virtual void operator()(const char *component, LogLevel severity, const std::string &text) = 0;
};

// nullptr restores the default console sink.
JOSHUPSCALE_EXPORT void setLogSink(LogSink *sink);

// ----------------------------------------------------------------- frames --
enum class DataLocation : std::uint8_t {
	CPU,               // host pointer
	CUDA,              // device pointer: a HIP device pointer in this runtime
	GRAPHICS_RESOURCE  // an image from getGLImage (a registered OpenGL texture)
};

// 4 bytes per pixel, byte order B,G,R,X.  `stride` is in bytes and may be
// negative (bottom-up frames); `ptr` addresses the first logical row.  The
// memory is borrowed for the duration of processImage only.  Field order and
// types are ABI: {pointer, 1-byte enum, signed stride, width, height}.
struct Image {
	void *ptr;
	DataLocation location;
	std::ptrdiff_t stride;
	std::size_t width, height;
};

enum class GraphicsResourceImageType : std::uint8_t { INPUT, OUTPUT };

// vtable: the destructor only; one Image member behind it.
struct GraphicsResourceImage {
	virtual ~GraphicsResourceImage() {}
	Image getImage() const { return m_Image; }

protected:
	Image m_Image = {};
};

// The HIP device that drives the calling thread's current OpenGL context
// (reference core/src/core.cc:140-149).
JOSHUPSCALE_EXPORT int getGLDeviceIndex();
// Registers a 2-D BGRX / RGBA8 texture of the current OpenGL context (core.cc:92-138); the
// caller owns the result (delete unregisters it).  processImage maps it for the duration
// of its staging copy only.
JOSHUPSCALE_EXPORT GraphicsResourceImage *getGLImage(std::uint32_t image,
    GraphicsResourceImageType type);

// ---------------------------------------------------------------- runtime --
// One recurrent super-resolution stream.  Not thread-safe; processImage is
// synchronous and advances the recurrent state by one frame.  vtable: destructor,
// then processImage; the four sizes follow the vptr and are read by the inline
// getters compiled into the CALLER.
struct Runtime {
	virtual ~Runtime() {}
	virtual void processImage(const Image &inputImage, const Image &outputImage) = 0;

	std::size_t getInputWidth() const { return m_InputWidth; }
	std::size_t getInputHeight() const { return m_InputHeight; }
	std::size_t getOutputWidth() const { return m_OutputWidth; }
	std::size_t getOutputHeight() const { return m_OutputHeight; }

protected:
	// the reference's member names and order (core.h:84-88): they are ABI, the inline
	// getters above are compiled into the caller
	std::size_t m_InputWidth = 0;
	std::size_t m_InputHeight = 0;
	std::size_t m_OutputWidth = 0;
	std::size_t m_OutputHeight = 0;
};

// Caller owns the result and destroys it with `delete`.  `model` names a .jupw
// container (joshupscale_amd/model_file.py); TensorRT engines are rejected with
// std::invalid_argument.
JOSHUPSCALE_EXPORT Runtime *createRuntime(int deviceId, const std::filesystem::path &modelPath);

// Only valid inside a catch block: formats the exception being handled exactly as the
// reference does (core/src/exception.cc:51-79): "Type: what()", each nested exception
// after "\n  ", no trailing newline, "Unknown error" for a non-std exception.
JOSHUPSCALE_EXPORT std::string getExceptionString();

}  // namespace core

}  // namespace JoshUpscale
