// Compile check of include/JoshUpscale/core.h against every use the reference's two plugins
// make of JoshUpscale::core (test infrastructure; compiled with -fsyntax-only by
// tests/test_cxx_surface.py, never linked or run).  Written from the call sites, not from
// the plugin sources: each block names the reference lines whose USE of the boundary it
// reproduces -- the same expressions on the same types, with the AviSynth / OBS SDK objects
// around them replaced by the plain values they supply.
//
//   avisynth_plugin/src/main.cc:40      std::unique_ptr<core::Runtime> member
//   avisynth_plugin/src/main.cc:57-60   reset(createRuntime(int, const char*)) / getExceptionString()
//   avisynth_plugin/src/main.cc:62-68   inline getters compared with / cast to int
//   avisynth_plugin/src/main.cc:113-121 DataLocation chosen by a switch, value-initialised first
//   avisynth_plugin/src/main.cc:125-142 core::Image from DESIGNATED initialisers (C++20), BYTE* -> void*,
//                                       negated int pitch -> ptrdiff_t stride
//   avisynth_plugin/src/main.cc:144-148 processImage(const Image&, const Image&) in a try / catch (...)
//   obs_plugin/include/JoshUpscale/obs/filter.h:82, 87-88   unique_ptr<Runtime>, unique_ptr<GraphicsResourceImage>
//   obs_plugin/src/filter.cc:64-69      int m_Device = core::getGLDeviceIndex()
//   obs_plugin/src/filter.cc:247-275    reset(core::getGLImage(uint32_t, GraphicsResourceImageType::INPUT / OUTPUT))
//   obs_plugin/src/filter.cc:291-306    reset(createRuntime(int, char*)); getters cast to uint32_t; reset()
//   obs_plugin/src/filter.cc:384-389    processImage(m_InputImage->getImage(), m_OutputImage->getImage())
//   obs_plugin/src/plugin.cc:93-106     a LogSink subclass with the override signature, static, setLogSink(&sink)
//   obs_plugin/src/logging.cc:20-33     LogLevel switched over INFO / WARNING / ERROR
//   obs_plugin/src/logging.cc:44-45     getExceptionString().c_str()
#include <cassert>
#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>
#include <string_view>
#include <type_traits>

#include "JoshUpscale/core.h"

namespace JoshUpscale {

namespace avisynth_like {

using BYTE = unsigned char;

// the three facts an AviSynth frame supplies at main.cc:125-142
struct FrameFacts {
	const BYTE *readPtr;
	BYTE *writePtr;
	int pitch;
	int width;
	int height;
	int deviceType;  // DEV_TYPE_CPU = 1, DEV_TYPE_CUDA = 2 in avisynth.h
};

class Filter {
public:
	Filter(const char *modelPath, int device, int viWidth, int viHeight, std::string *error) {
		try {
			m_Runtime.reset(core::createRuntime(device, modelPath));
		} catch (...) {
			auto exception = core::getExceptionString();
			*error = exception.c_str();
			return;
		}
		if (viWidth != static_cast<int>(m_Runtime->getInputWidth()) ||
		    viHeight != static_cast<int>(m_Runtime->getInputHeight())) {
			*error = "unsupported video size";
		}
		m_Width = static_cast<int>(m_Runtime->getOutputWidth());
		m_Height = static_cast<int>(m_Runtime->getOutputHeight());
	}

	bool getFrame(const FrameFacts &src, const FrameFacts &dst, std::string *error) {
		core::DataLocation location{};
		switch (src.deviceType) {
		case 1:
			location = core::DataLocation::CPU;
			break;
		case 2:
			location = core::DataLocation::CUDA;
			break;
		default:
			*error = "unsupported device";
			return false;
		}
		core::Image inputImage = {
		    .ptr = const_cast<BYTE *>(
		        src.readPtr + ((static_cast<std::ptrdiff_t>(src.height) - 1) * src.pitch)),
		    .location = location,
		    .stride = -src.pitch,
		    .width = static_cast<std::size_t>(src.width),
		    .height = static_cast<std::size_t>(src.height),
		};
		core::Image outputImage = {
		    .ptr = dst.writePtr + ((static_cast<std::ptrdiff_t>(m_Height) - 1) * dst.pitch),
		    .location = location,
		    .stride = -dst.pitch,
		    .width = static_cast<std::size_t>(m_Width),
		    .height = static_cast<std::size_t>(m_Height),
		};
		try {
			m_Runtime->processImage(inputImage, outputImage);
		} catch (...) {
			auto exception = core::getExceptionString();
			*error = exception.c_str();
			return false;
		}
		return true;
	}

private:
	std::unique_ptr<core::Runtime> m_Runtime;
	int m_Width = 0;
	int m_Height = 0;
};

}  // namespace avisynth_like

namespace obs_like {

// logging.cc:20-33: every enumerator is named in a switch
inline int toHostLevel(core::LogLevel level) {
	switch (level) {
	case core::LogLevel::INFO:
		return 300;
	case core::LogLevel::WARNING:
		return 200;
	case core::LogLevel::ERROR:
		return 100;
	}
	return 0;
}

void log(core::LogLevel level, std::string_view format, ...);

// logging.cc:42-48
inline void logException() noexcept {
	try {
		log(core::LogLevel::ERROR, "Exception: %s", core::getExceptionString().c_str());
	} catch (...) {
	}
}

// plugin.cc:93-106
inline void setupLogging() {
	struct LogSink : JoshUpscale::core::LogSink {
		void operator()(const char *tag, JoshUpscale::core::LogLevel logLevel,
		    const std::string &message) override {
			if (logLevel == JoshUpscale::core::LogLevel::ERROR ||
			    logLevel == JoshUpscale::core::LogLevel::WARNING) {
				JoshUpscale::obs_like::log(logLevel, "%s: %s", tag, message.c_str());
			}
		}
	};

	static LogSink logSink;
	JoshUpscale::core::setLogSink(&logSink);
}

class Filter {
public:
	// filter.cc:64-69: the device index is a plain int, negative = unsupported
	bool pickDevice() {
		m_Device = core::getGLDeviceIndex();
		return m_Device >= 0;
	}

	// filter.cc:247-256 (texture object = pointer to the GL name, read as uint32)
	void createInputImage(void *textureObj) {
		auto glTexture = *reinterpret_cast<std::uint32_t *>(textureObj);
		m_InputImage.reset(core::getGLImage(glTexture, core::GraphicsResourceImageType::INPUT));
		assert(m_InputImage);
	}

	// filter.cc:266-272
	void createOutputImage(void *textureObj) {
		auto glTexture = *reinterpret_cast<std::uint32_t *>(textureObj);
		m_OutputImage.reset(core::getGLImage(glTexture, core::GraphicsResourceImageType::OUTPUT));
		assert(m_OutputImage);
	}

	// filter.cc:283-317: (re)load a model; a failure leaves the filter without a runtime
	void initModel(char *modelFile, void *outputTextureObj, std::uint32_t sizes[4]) noexcept {
		try {
			m_Runtime.reset(core::createRuntime(m_Device, modelFile));
			m_InputImage.reset();
			m_OutputImage.reset();
			sizes[0] = static_cast<std::uint32_t>(m_Runtime->getInputWidth());
			sizes[1] = static_cast<std::uint32_t>(m_Runtime->getInputHeight());
			sizes[2] = static_cast<std::uint32_t>(m_Runtime->getOutputWidth());
			sizes[3] = static_cast<std::uint32_t>(m_Runtime->getOutputHeight());
			createOutputImage(outputTextureObj);
			log(core::LogLevel::INFO, "Successfully loaded model: %s", modelFile);
		} catch (...) {
			logException();
			m_Runtime.reset();
		}
	}

	// filter.cc:232-243: getters on a runtime that may be absent
	std::uint32_t getWidth() noexcept {
		if (m_Runtime == nullptr) {
			return 0;
		}
		return static_cast<std::uint32_t>(m_Runtime->getOutputWidth());
	}

	// filter.cc:380-392
	bool processFrame(void *inputTextureObj) noexcept {
		try {
			if (m_InputImage == nullptr) {
				createInputImage(inputTextureObj);
			}
			m_Runtime->processImage(m_InputImage->getImage(), m_OutputImage->getImage());
		} catch (...) {
			logException();
			return false;
		}
		return true;
	}

private:
	int m_Device = -1;
	std::unique_ptr<core::Runtime> m_Runtime = nullptr;
	std::unique_ptr<core::GraphicsResourceImage> m_InputImage = nullptr;
	std::unique_ptr<core::GraphicsResourceImage> m_OutputImage = nullptr;
};

}  // namespace obs_like

// What the callers rely on without writing it down (core/public/JoshUpscale/core.h:21-94).
static_assert(std::is_same_v<std::underlying_type_t<core::LogLevel>, std::uint8_t>);
static_assert(std::is_same_v<std::underlying_type_t<core::DataLocation>, std::uint8_t>);
static_assert(std::is_aggregate_v<core::Image>, "Image is brace-initialised by the AviSynth plugin");
static_assert(std::is_same_v<decltype(core::Image::ptr), void *>);
static_assert(std::is_same_v<decltype(core::Image::stride), std::ptrdiff_t>);
static_assert(std::is_same_v<decltype(core::Image::width), std::size_t>);
static_assert(std::has_virtual_destructor_v<core::Runtime>, "deleted through unique_ptr<Runtime>");
static_assert(std::has_virtual_destructor_v<core::GraphicsResourceImage>);
static_assert(std::is_abstract_v<core::Runtime> && std::is_abstract_v<core::LogSink>);
static_assert(std::is_same_v<decltype(&core::createRuntime), core::Runtime *(*)(int, const std::filesystem::path &)>);
static_assert(std::is_same_v<decltype(&core::getGLImage),
    core::GraphicsResourceImage *(*)(std::uint32_t, core::GraphicsResourceImageType)>);
static_assert(std::is_same_v<decltype(&core::getGLDeviceIndex), int (*)()>);
static_assert(std::is_same_v<decltype(&core::getExceptionString), std::string (*)()>);
static_assert(std::is_same_v<decltype(&core::setLogSink), void (*)(core::LogSink *)>);
static_assert(std::is_same_v<decltype(std::declval<const core::Runtime &>().getInputWidth()), std::size_t>);
static_assert(std::is_same_v<decltype(std::declval<const core::GraphicsResourceImage &>().getImage()), core::Image>);
static_assert(std::is_same_v<decltype(&core::Runtime::processImage),
    void (core::Runtime::*)(const core::Image &, const core::Image &)>);

}  // namespace JoshUpscale
