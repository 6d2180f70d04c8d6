#include "graphics.h"

#include <dlfcn.h>
#include <hip/hip_gl_interop.h>

#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

#include "hip_util.h"

namespace ju {

namespace {

// ---------------------------------------------------------------------------
// HIP-GL interop (the product backend)
// ---------------------------------------------------------------------------
class HipGlBackend final : public GraphicsBackend {
public:
	void *registerImage(std::uint32_t texture, int type, std::size_t *width, std::size_t *height) override {
		// texture size through the caller's current GL context (core.cc:107-121); libGL is
		// opened here, not linked: only a caller that has a GL context gets this far
		constexpr unsigned kTexture2D = 0x0DE1, kWidth = 0x1000, kHeight = 0x1001;
		struct Gl {
			void (*bindTexture)(unsigned, unsigned) = nullptr;
			unsigned (*getError)() = nullptr;
			void (*getTexLevelParameteriv)(unsigned, int, unsigned, int *) = nullptr;
		};
		static const Gl gl = [] {
			Gl g;
			void *lib = nullptr;
			for (const char *n : {"libGL.so.1", "libOpenGL.so.0", "libGL.so"}) {
				lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
				if (lib) break;
			}
			if (!lib) throw std::runtime_error(std::string("cannot load libGL: ") + dlerror());
			g.bindTexture = reinterpret_cast<decltype(g.bindTexture)>(dlsym(lib, "glBindTexture"));
			g.getError = reinterpret_cast<decltype(g.getError)>(dlsym(lib, "glGetError"));
			g.getTexLevelParameteriv =
			    reinterpret_cast<decltype(g.getTexLevelParameteriv)>(dlsym(lib, "glGetTexLevelParameteriv"));
			if (!g.bindTexture || !g.getError || !g.getTexLevelParameteriv) {
				throw std::runtime_error("libGL lacks glBindTexture / glGetError / glGetTexLevelParameteriv");
			}
			return g;
		}();
		gl.bindTexture(kTexture2D, texture);
		if (const unsigned e = gl.getError()) {
			throw std::runtime_error("Failed to bind texture: " + std::to_string(e));
		}
		int w = 0, h = 0;
		gl.getTexLevelParameteriv(kTexture2D, 0, kWidth, &w);
		gl.getTexLevelParameteriv(kTexture2D, 0, kHeight, &h);
		gl.bindTexture(kTexture2D, 0);
		if (w <= 0 || h <= 0) throw std::runtime_error("texture has no size (no current OpenGL context?)");
		hipGraphicsResource *res = nullptr;
		const unsigned flags = type == 0 ? hipGraphicsRegisterFlagsReadOnly : hipGraphicsRegisterFlagsWriteDiscard;
		JU_HIP(hipGraphicsGLRegisterImage(&res, texture, kTexture2D, flags));
		*width = static_cast<std::size_t>(w);
		*height = static_cast<std::size_t>(h);
		return res;
	}
	void unregisterImage(void *resource) override {
		(void)hipGraphicsUnregisterResource(static_cast<hipGraphicsResource_t>(resource));
	}
	GraphicsArray map(void *resource, hipStream_t stream) override {
		auto res = static_cast<hipGraphicsResource_t>(resource);
		JU_HIP(hipGraphicsMapResources(1, &res, stream));
		GraphicsArray a;
		hipArray_t arr = nullptr;
		hipError_t e = hipGraphicsSubResourceGetMappedArray(&arr, res, 0, 0);
		hipChannelFormatDesc desc{};
		hipExtent ext{};
		if (e == hipSuccess) e = hipArrayGetInfo(&desc, &ext, nullptr, arr);
		if (e != hipSuccess) {
			(void)hipGraphicsUnmapResources(1, &res, stream);
			throw HipError(e, "hipGraphicsSubResourceGetMappedArray");
		}
		a.array = arr;
		a.width = ext.width;
		a.height = ext.height;
		a.fourBytes = desc.f == hipChannelFormatKindUnsigned && desc.x == 8 && desc.y == 8 && desc.z == 8 && desc.w == 8;
		return a;
	}
	void unmap(void *resource, hipStream_t stream) override {
		auto res = static_cast<hipGraphicsResource_t>(resource);
		(void)hipGraphicsUnmapResources(1, &res, stream);
	}
	void copyFromArray(void *dst, std::size_t dstPitch, const GraphicsArray &src, std::size_t rowBytes,
	    std::size_t rows, hipStream_t stream) override {
		JU_HIP(hipMemcpy2DFromArrayAsync(dst, dstPitch, static_cast<hipArray_const_t>(src.array), 0, 0, rowBytes,
		    rows, hipMemcpyDeviceToDevice, stream));
	}
	void copyToArray(const GraphicsArray &dst, const void *src, std::size_t srcPitch, std::size_t rowBytes,
	    std::size_t rows, hipStream_t stream) override {
		JU_HIP(hipMemcpy2DToArrayAsync(static_cast<hipArray_t>(dst.array), 0, 0, src, srcPitch, rowBytes, rows,
		    hipMemcpyDeviceToDevice, stream));
	}
	int deviceIndex() override {
		int device = -1;
		unsigned count = 0;
		JU_HIP(hipGLGetDevices(&count, &device, 1, hipGLDeviceListAll));
		if (count != 1) throw std::runtime_error("Failed to determine HIP device");  // core.cc:145-147
		return device;
	}
};

#ifdef JU_TEST_HOOKS  // libJoshUpscale_test.so only: the product library carries no test double
// ---------------------------------------------------------------------------
// test double: a "texture" is a pitched device buffer the test owns
// ---------------------------------------------------------------------------
class FakeBackend final : public GraphicsBackend {
public:
	struct Texture {
		void *ptr;
		std::size_t pitch, width, height;
		int bpp;
	};
	struct Resource {
		Texture tex;
		int type;
		bool mapped = false;
	};
	std::map<std::uint32_t, Texture> textures;
	int registered = 0, mapped = 0, maps = 0, unmaps = 0;

	void *registerImage(std::uint32_t texture, int type, std::size_t *width, std::size_t *height) override {
		auto it = textures.find(texture);
		if (it == textures.end()) throw std::runtime_error("Failed to bind texture: 1282");  // GL_INVALID_OPERATION
		auto *r = new Resource{it->second, type};
		*width = r->tex.width;
		*height = r->tex.height;
		++registered;
		return r;
	}
	void unregisterImage(void *resource) override {
		delete static_cast<Resource *>(resource);
		--registered;
	}
	GraphicsArray map(void *resource, hipStream_t) override {
		auto *r = static_cast<Resource *>(resource);
		if (r->mapped) throw std::runtime_error("resource is already mapped");
		r->mapped = true;
		++mapped;
		++maps;
		GraphicsArray a;
		a.array = r;
		a.width = r->tex.width;
		a.height = r->tex.height;
		a.fourBytes = r->tex.bpp == 4;
		return a;
	}
	void unmap(void *resource, hipStream_t) override {
		auto *r = static_cast<Resource *>(resource);
		r->mapped = false;
		--mapped;
		++unmaps;
	}
	void copyFromArray(void *dst, std::size_t dstPitch, const GraphicsArray &src, std::size_t rowBytes,
	    std::size_t rows, hipStream_t stream) override {
		auto *r = static_cast<Resource *>(src.array);
		if (!r->mapped) throw std::runtime_error("copy from an unmapped resource");
		JU_HIP(hipMemcpy2DAsync(dst, dstPitch, r->tex.ptr, r->tex.pitch, rowBytes, rows, hipMemcpyDeviceToDevice, stream));
	}
	void copyToArray(const GraphicsArray &dst, const void *src, std::size_t srcPitch, std::size_t rowBytes,
	    std::size_t rows, hipStream_t stream) override {
		auto *r = static_cast<Resource *>(dst.array);
		if (!r->mapped) throw std::runtime_error("copy to an unmapped resource");
		if (r->type != 1) throw std::runtime_error("copy to a read-only (input) resource");
		JU_HIP(hipMemcpy2DAsync(r->tex.ptr, r->tex.pitch, src, srcPitch, rowBytes, rows, hipMemcpyDeviceToDevice, stream));
	}
	int deviceIndex() override {
		int dev = 0;
		JU_HIP(hipGetDevice(&dev));
		return dev;
	}
};
#endif  // JU_TEST_HOOKS

#ifdef JU_TEST_HOOKS
std::mutex g_Mutex;
FakeBackend *g_Fake = nullptr;
#endif

}  // namespace

GraphicsBackend &graphicsBackend() {
#ifdef JU_TEST_HOOKS
	std::lock_guard<std::mutex> lock(g_Mutex);
	if (g_Fake) return *g_Fake;
#endif
	static HipGlBackend real;
	return real;
}

#ifdef JU_TEST_HOOKS
void fakeGraphicsDefineTexture(std::uint32_t texture, void *devicePtr, std::size_t pitch, std::size_t width,
    std::size_t height, int bytesPerPixel) {
	std::lock_guard<std::mutex> lock(g_Mutex);
	if (!g_Fake) g_Fake = new FakeBackend();
	g_Fake->textures[texture] = FakeBackend::Texture{devicePtr, pitch, width, height, bytesPerPixel};
}

void fakeGraphicsReset() {
	std::lock_guard<std::mutex> lock(g_Mutex);
	delete g_Fake;
	g_Fake = nullptr;
}

void fakeGraphicsCounters(int *registered, int *mapped, int *maps, int *unmaps) {
	std::lock_guard<std::mutex> lock(g_Mutex);
	if (registered) *registered = g_Fake ? g_Fake->registered : 0;
	if (mapped) *mapped = g_Fake ? g_Fake->mapped : 0;
	if (maps) *maps = g_Fake ? g_Fake->maps : 0;
	if (unmaps) *unmaps = g_Fake ? g_Fake->unmaps : 0;
}
#endif  // JU_TEST_HOOKS

}  // namespace ju
