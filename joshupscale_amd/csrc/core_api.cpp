// C++ plugin surface (include/JoshUpscale/core.h) implemented over the C ABI.

#include "JoshUpscale/core.h"

#include <cxxabi.h>

#include <cstdlib>
#include <exception>
#include <ios>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <typeinfo>

#include "joshupscale_amd.h"

namespace JoshUpscale {

namespace core {

namespace {

static_assert(sizeof(Image) == sizeof(ju_image), "Image and ju_image must match");
static_assert(offsetof(Image, stride) == offsetof(ju_image, stride), "Image layout");
static_assert(offsetof(Image, height) == offsetof(ju_image, height), "Image layout");

// The C layer flattens exceptions to (code, "Type: message"); turn them back
// into the exception family the reference throws.
[[noreturn]] void raise(int code) {
	const std::string msg = ju_last_error();
	const std::size_t colon = msg.find(": ");
	const std::string what = colon == std::string::npos ? msg : msg.substr(colon + 2);
	switch (code) {
	case JU_ERR_INVALID_ARGUMENT:
		throw std::invalid_argument(what);
	case JU_ERR_IO:
		throw std::ios_base::failure(what);
	default:
		throw std::runtime_error(what);
	}
}

ju_image toC(const Image &img) {
	return ju_image{img.ptr, static_cast<std::uint8_t>(img.location), img.stride, img.width,
	    img.height};
}

struct HipRuntime final : Runtime {
	HipRuntime(int deviceId, const std::filesystem::path &modelPath) {
		const int rc = ju_create(deviceId, modelPath.string().c_str(), &m_Handle);
		if (rc != JU_OK) raise(rc);
		ju_get_size(m_Handle, &m_InputWidth, &m_InputHeight, &m_OutputWidth, &m_OutputHeight);
	}
	~HipRuntime() override {
		ju_destroy(m_Handle);
	}
	void processImage(const Image &inputImage, const Image &outputImage) override {
		const ju_image in = toC(inputImage), out = toC(outputImage);
		const int rc = ju_process(m_Handle, &in, &out);
		if (rc != JU_OK) raise(rc);
	}

private:
	ju_runtime *m_Handle = nullptr;
};

LogSink *g_Sink = nullptr;

void sinkTrampoline(const char *tag, int level, const char *message, void *) {
	if (g_Sink) (*g_Sink)(tag, static_cast<LogLevel>(level), std::string(message));
}

std::string demangled(const char *name) {
	int status = 0;
	std::unique_ptr<char, void (*)(void *)> p(
	    abi::__cxa_demangle(name, nullptr, nullptr, &status), std::free);
	return status == 0 && p ? std::string(p.get()) : std::string(name);
}

// "Type: what()"; a nested exception follows after "\n  " (the same two spaces at every
// depth, no trailing newline): the format of the reference's printException
// (core/src/exception.cc:51-79), which callers log verbatim.
void describeCurrent(std::ostream &os);

void describe(std::ostream &os, const std::exception &e) {
	os << demangled(typeid(e).name()) << ": " << e.what();
	try {
		std::rethrow_if_nested(e);
	} catch (...) {
		os << "\n  ";
		describeCurrent(os);
	}
}

void describeCurrent(std::ostream &os) {
	try {
		throw;  // re-raise the exception currently being handled
	} catch (const std::exception &e) {
		describe(os, e);
	} catch (...) {
		os << "Unknown error";
	}
}

}  // namespace

void setLogSink(LogSink *sink) {
	g_Sink = sink;
	ju_set_log_callback(sink ? &sinkTrampoline : nullptr, nullptr);
}

int getGLDeviceIndex() {
	int dev = -1;
	const int rc = ju_get_gl_device_index(&dev);
	if (rc != JU_OK) raise(rc);
	return dev;
}

namespace {
// reference core/src/core.cc:92-131 (GLResourceImage)
struct GLResourceImage final : GraphicsResourceImage {
	GLResourceImage(std::uint32_t image, GraphicsResourceImageType type) {
		const int rc = ju_get_gl_image(image, static_cast<int>(type), &m_C);
		if (rc != JU_OK) raise(rc);
		m_Image.ptr = m_C.ptr;
		m_Image.location = DataLocation::GRAPHICS_RESOURCE;
		m_Image.stride = 0;
		m_Image.width = m_C.width;
		m_Image.height = m_C.height;
	}
	~GLResourceImage() override { ju_release_gl_image(&m_C); }

private:
	ju_image m_C{};
};
}  // namespace

GraphicsResourceImage *getGLImage(std::uint32_t image, GraphicsResourceImageType type) {
	return new GLResourceImage(image, type);
}

Runtime *createRuntime(int deviceId, const std::filesystem::path &modelPath) {
	return new HipRuntime(deviceId, modelPath);
}

std::string getExceptionString() {
	std::ostringstream ss;
	describeCurrent(ss);
	return ss.str();
}

}  // namespace core

}  // namespace JoshUpscale
