// extern "C" entry points of libJoshUpscale.so (include/joshupscale_amd.h).
// Exceptions never cross this boundary: each call is wrapped, the message is
// kept per thread for ju_last_error().

#include "joshupscale_amd.h"
#ifdef JU_TEST_HOOKS  // libJoshUpscale_test.so only (Makefile): the product library exports none of the hooks
#include "joshupscale_amd_test.h"
#endif

#include <cstdio>
#include <fstream>
#include <ios>
#include <new>
#include <stdexcept>
#include <string>
#include <vector>

#include "engine.h"
#include "fp8.h"
#include "graphics.h"
#include "log.h"
#include "model.h"

struct ju_runtime {
	std::unique_ptr<ju::Engine> engine;
};

namespace {

thread_local std::string g_LastError;

int fail(int code, const std::string &msg) {
	g_LastError = msg;
	ju::logMessage(ju::LogLevel::Error, "Core", msg);
	return code;
}

template <typename F>
int guarded(F &&f) {
	try {
		f();
		return JU_OK;
	} catch (const ju::HipError &e) {
		return fail(JU_ERR_DEVICE, std::string("HipException: ") + e.what());
	} catch (const std::invalid_argument &e) {
		return fail(JU_ERR_INVALID_ARGUMENT, std::string("std::invalid_argument: ") + e.what());
	} catch (const std::ios_base::failure &e) {
		return fail(JU_ERR_IO, std::string("std::ios_base::failure: ") + e.what());
	} catch (const std::bad_alloc &e) {
		return fail(JU_ERR_INTERNAL, std::string("std::bad_alloc: ") + e.what());
	} catch (const std::runtime_error &e) {
		return fail(JU_ERR_INTERNAL, std::string("std::runtime_error: ") + e.what());
	} catch (const std::exception &e) {
		return fail(JU_ERR_INTERNAL, std::string("std::exception: ") + e.what());
	} catch (...) {
		return fail(JU_ERR_INTERNAL, "unknown exception");
	}
}

}  // namespace

// exception mapping for the other translation units of the C ABI (comm.cpp)
int juGuarded(void (*fn)(void *), void *ctx) {
	return guarded([&] { fn(ctx); });
}

namespace {

ju::Frame toFrame(const ju_image *img) {
	if (img == nullptr) throw std::invalid_argument("image is NULL");
	if (img->location > JU_LOC_GRAPHICS_RESOURCE) {
		throw std::invalid_argument("image has an unknown location");
	}
	return ju::Frame{img->ptr, static_cast<ju::Location>(img->location), img->stride, img->width,
	    img->height};
}

ju::Engine &engineOf(ju_runtime *rt) {
	if (rt == nullptr || !rt->engine) throw std::invalid_argument("runtime is NULL");
	return *rt->engine;
}

int createFromBytes(int device, const void *bytes, std::size_t size, int dtype, ju_runtime **out) {
	return guarded([&] {
		if (out == nullptr) throw std::invalid_argument("out_runtime is NULL");
		*out = nullptr;
		int count = 0;
		JU_HIP(hipGetDeviceCount(&count));
		if (device < 0 || device >= count) {
			throw std::invalid_argument("device " + std::to_string(device) + " does not exist (" +
			                            std::to_string(count) + " HIP devices visible)");
		}
		ju::DeviceGuard guard(device);  // like cuda::DeviceContext in core/src/core.cc:168
		auto rt = std::make_unique<ju_runtime>();
		rt->engine = std::make_unique<ju::Engine>(device, bytes, size, dtype);
		*out = rt.release();
	});
}

}  // namespace

extern "C" {

int ju_create(int device_id, const char *model_path, ju_runtime **out_runtime) {
	std::vector<char> bytes;
	int rc = guarded([&] {
		if (model_path == nullptr) throw std::invalid_argument("model_path is NULL");
		// whole-file read with failbit/badbit exceptions, as core/src/core.cc:156-167
		std::ifstream f(model_path, std::ifstream::in | std::ifstream::binary | std::ifstream::ate);
		if (!f) throw std::ios_base::failure(std::string("cannot open model file ") + model_path);
		f.exceptions(std::ifstream::badbit | std::ifstream::failbit);
		const auto size = static_cast<std::size_t>(f.tellg());
		bytes.resize(size);
		f.seekg(0);
		f.read(bytes.data(), static_cast<std::streamsize>(size));
	});
	if (rc != JU_OK) return rc;
	return createFromBytes(device_id, bytes.data(), bytes.size(), JU_DTYPE_DEFAULT, out_runtime);
}

int ju_create_from_memory(int device_id, const void *model_bytes, size_t model_size, int dtype,
    ju_runtime **out_runtime) {
	return createFromBytes(device_id, model_bytes, model_size, dtype, out_runtime);
}

int ju_validate_model(const void *model_bytes, size_t model_size) {
	return guarded([&] {
		const ju::ModelFile model(model_bytes, model_size);
		(void)ju::foldModel(model);
	});
}

void ju_destroy(ju_runtime *runtime) {
	delete runtime;
}

int ju_process(ju_runtime *runtime, const ju_image *input, const ju_image *output) {
	return guarded([&] { engineOf(runtime).process(toFrame(input), toFrame(output)); });
}

int ju_process_batch(ju_runtime *runtime, const ju_image *inputs, const ju_image *outputs, int count) {
	return guarded([&] {
		if (count < 0 || (count > 0 && (inputs == nullptr || outputs == nullptr))) {
			throw std::invalid_argument("ju_process_batch: NULL images or a negative count");
		}
		std::vector<ju::Frame> in(static_cast<std::size_t>(count)), out(static_cast<std::size_t>(count));
		for (int i = 0; i < count; ++i) {
			in[i] = toFrame(inputs + i);
			out[i] = toFrame(outputs + i);
		}
		engineOf(runtime).processBatch(in.data(), out.data(), count);
	});
}

int ju_prepare_batch(ju_runtime *runtime, const ju_image *inputs, const ju_image *outputs, int count, int *captured) {
	if (captured) *captured = 0;
	return guarded([&] {
		if (count < 0 || (count > 0 && (inputs == nullptr || outputs == nullptr))) {
			throw std::invalid_argument("ju_prepare_batch: NULL images or a negative count");
		}
		std::vector<ju::Frame> in(static_cast<std::size_t>(count)), out(static_cast<std::size_t>(count));
		for (int i = 0; i < count; ++i) {
			in[i] = toFrame(inputs + i);
			out[i] = toFrame(outputs + i);
		}
		const int n = engineOf(runtime).prepareBatch(in.data(), out.data(), count);
		if (captured) *captured = n;
	});
}

int ju_set_lookahead(ju_runtime *runtime, int frames) {
	return guarded([&] { engineOf(runtime).setLookahead(frames); });
}

int ju_enqueue(ju_runtime *runtime, const ju_image *input, const ju_image *output) {
	return guarded([&] {
		const ju::Frame in = toFrame(input), out = toFrame(output);
		if (in.location != ju::Location::Device || out.location != ju::Location::Device) {
			throw std::invalid_argument("ju_enqueue needs JU_LOC_DEVICE images");
		}
		engineOf(runtime).enqueue(in, out);
	});
}

int ju_prepare_frames(ju_runtime *runtime, const ju_image *input, const ju_image *output, int *captured) {
	if (captured) *captured = 0;
	return guarded([&] {
		const int n = engineOf(runtime).prepareFrames(toFrame(input), toFrame(output));
		if (captured) *captured = n;
	});
}

int ju_synchronize(ju_runtime *runtime) {
	return guarded([&] { engineOf(runtime).synchronize(); });
}

int ju_get_size(const ju_runtime *runtime, size_t *input_width, size_t *input_height,
    size_t *output_width, size_t *output_height) {
	return guarded([&] {
		const ju::FrameSize fs = engineOf(const_cast<ju_runtime *>(runtime)).frameSize();
		if (input_width) *input_width = fs.inputWidth;
		if (input_height) *input_height = fs.inputHeight;
		if (output_width) *output_width = fs.outputWidth;
		if (output_height) *output_height = fs.outputHeight;
	});
}

int ju_reset(ju_runtime *runtime) {
	return guarded([&] { engineOf(runtime).reset(); });
}

const char *ju_last_error(void) {
	return g_LastError.c_str();
}

void ju_set_log_callback(ju_log_callback callback, void *user) {
	ju::setLogCallback(callback, user);
}

int ju_get_gl_device_index(int *out_device) {
	if (out_device) *out_device = -1;
	return guarded([&] {
		if (out_device == nullptr) throw std::invalid_argument("out_device is NULL");
		*out_device = ju::graphicsBackend().deviceIndex();
	});
}

int ju_get_gl_image(uint32_t gl_texture, int type, ju_image *out_image) {
	if (out_image) *out_image = ju_image{};
	return guarded([&] {
		if (out_image == nullptr) throw std::invalid_argument("out_image is NULL");
		if (type != 0 && type != 1) throw std::invalid_argument("image type must be 0 (input) or 1 (output)");
		ju::GraphicsBackend &backend = ju::graphicsBackend();
		std::size_t w = 0, h = 0;
		void *res = backend.registerImage(gl_texture, type, &w, &h);
		out_image->ptr = new ju::GraphicsHandle{&backend, res};
		out_image->location = JU_LOC_GRAPHICS_RESOURCE;
		out_image->stride = 0;
		out_image->width = w;
		out_image->height = h;
	});
}

void ju_release_gl_image(ju_image *image) {
	if (image == nullptr || image->location != JU_LOC_GRAPHICS_RESOURCE || image->ptr == nullptr) return;
	auto *h = static_cast<ju::GraphicsHandle *>(image->ptr);
	try {
		h->backend->unregisterImage(h->resource);
	} catch (...) {
	}
	delete h;
	image->ptr = nullptr;
}

#ifdef JU_TEST_HOOKS
int ju_debug_fake_gl_texture(uint32_t gl_texture, void *device_ptr, size_t pitch, size_t width, size_t height,
    int bytes_per_pixel) {
	return guarded([&] {
		if (device_ptr == nullptr) {
			ju::fakeGraphicsReset();
			return;
		}
		ju::fakeGraphicsDefineTexture(gl_texture, device_ptr, pitch, width, height, bytes_per_pixel);
	});
}

void ju_debug_fake_gl_counters(int *registered, int *mapped, int *maps, int *unmaps) {
	ju::fakeGraphicsCounters(registered, mapped, maps, unmaps);
}
#endif  // JU_TEST_HOOKS

int ju_get_dtype(const ju_runtime *runtime) {
	if (runtime == nullptr || !runtime->engine) return -1;
	return runtime->engine->reportedDtype();
}

#ifdef JU_TEST_HOOKS
int ju_read_tensor(ju_runtime *runtime, const char *name, float *dst, size_t capacity,
    size_t *count) {
	return guarded([&] {
		if (name == nullptr) throw std::invalid_argument("name is NULL");
		const std::size_t n = engineOf(runtime).readTensor(name, dst, capacity);
		if (count) *count = n;
	});
}

int ju_time_steps(ju_runtime *runtime, const char *tag, int iters, double *ms_per_launch,
    int *launches, double *flops) {
	return guarded([&] {
		const std::string t = tag ? tag : "";
		ju::Engine &e = engineOf(runtime);
		const double ms = e.timeSteps(t, iters, launches);
		if (ms_per_launch) *ms_per_launch = ms;
		if (flops) *flops = e.flopsOf(t);
	});
}
#endif  // JU_TEST_HOOKS

int ju_get_stat(const ju_runtime *runtime, const char *key, double *value) {
	return guarded([&] {
		if (key == nullptr || value == nullptr) throw std::invalid_argument("ju_get_stat: null argument");
		*value = engineOf(const_cast<ju_runtime *>(runtime)).stat(key);
	});
}

#ifdef JU_TEST_HOOKS
int ju_debug_set(const char *key, int value) {
	return guarded([&] {
		const std::string k = key ? key : "";
		if (k == "tower_variant") ju::setTowerVariant(value);
		else if (k == "resident_fault") ju::setResidentFault(value);
		else if (k == "tower_fast") ju::setResidentTowerFast(value);
		else if (k == "res_block_plain") ju::setResBlockPlain(value);
		else if (k == "fp8_block_form") ju::setFp8BlockForm(value);
		else throw std::invalid_argument("unknown debug key " + k);
	});
}

int ju_debug_e4m3(const float *values, unsigned char *codes, size_t count) {
	return guarded([&] {
		if (count && (!values || !codes)) throw std::invalid_argument("ju_debug_e4m3: null buffer");
		for (size_t i = 0; i < count; ++i) codes[i] = ju::e4m3FromFloat(values[i]);
	});
}
#endif  // JU_TEST_HOOKS

const char *ju_version(void) {
	#ifdef JU_TEST_HOOKS
	return "joshupscale-amd 0.1 (gfx950, test hooks)";
#else
	return "joshupscale-amd 0.1 (gfx950)";
#endif
}

}  // extern "C"
