// Graphics-resource frames (DataLocation::GRAPHICS_RESOURCE): what the OBS plugin hands
// over on Linux -- OpenGL textures registered with the compute runtime
// (reference core/src/core.cc:92-149 GLResourceImage / getGLImage / getGLDeviceIndex,
// core/include/JoshUpscale/core/cuda.h:310-349 GraphicsResource map / unmap / getArray,
// core/src/cuda_convert.cc.cu:380-397, 419-436 the array <-> staging-buffer copies,
// obs_plugin/src/filter.cc:242-279, 297-307 the caller).
//
// The engine talks to a small backend interface so that the plumbing (register ->
// map -> array <-> staging copy -> unmap -> unregister, size and format checks) runs in
// the GPU tests on a test double; the product backend is HIP-GL interop
// (hipGraphicsGLRegisterImage & co).  The MI355X boxes are headless (no GL context can
// be created on them), so the HIP-GL backend itself is untested on hardware.
#pragma once

#include <hip/hip_runtime_api.h>

#include <cstddef>
#include <cstdint>

namespace ju {

// A mapped resource's pixel array: RGBA8-like, 4 bytes per pixel.
struct GraphicsArray {
	void *array = nullptr;   // hipArray_t (HIP-GL) or a backend-private handle
	std::size_t width = 0, height = 0;
	bool fourBytes = false;  // 8-bit x 4 unsigned channels
};

class GraphicsBackend {
public:
	virtual ~GraphicsBackend() = default;
	// type: 0 = input (read only), 1 = output (write discard).  Returns the registered
	// resource and the texture's size.
	virtual void *registerImage(std::uint32_t texture, int type, std::size_t *width, std::size_t *height) = 0;
	virtual void unregisterImage(void *resource) = 0;
	virtual GraphicsArray map(void *resource, hipStream_t stream) = 0;
	virtual void unmap(void *resource, hipStream_t stream) = 0;
	virtual void copyFromArray(void *dst, std::size_t dstPitch, const GraphicsArray &src, std::size_t rowBytes,
	    std::size_t rows, hipStream_t stream) = 0;
	virtual void copyToArray(const GraphicsArray &dst, const void *src, std::size_t srcPitch,
	    std::size_t rowBytes, std::size_t rows, hipStream_t stream) = 0;
	virtual int deviceIndex() = 0;
};

// What ju_image::ptr holds for JU_LOC_GRAPHICS_RESOURCE (opaque to the caller, like the
// reference's cudaGraphicsResource_t).
struct GraphicsHandle {
	GraphicsBackend *backend;
	void *resource;
};

// The HIP-GL backend, or the test double once one is installed.
GraphicsBackend &graphicsBackend();

#ifdef JU_TEST_HOOKS
// ---- test double (ju_debug_fake_gl_*, libJoshUpscale_test.so only): "textures" are pitched device buffers ----
void fakeGraphicsDefineTexture(std::uint32_t texture, void *devicePtr, std::size_t pitch, std::size_t width,
    std::size_t height, int bytesPerPixel);
void fakeGraphicsReset();  // back to the HIP-GL backend
void fakeGraphicsCounters(int *registered, int *mapped, int *maps, int *unmaps);
#endif

}  // namespace ju
