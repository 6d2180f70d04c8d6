// Stand-in for the reference's two callers, compiled against the C++ plugin surface
// only (include/JoshUpscale/core.h) -- the same code paths avisynth_plugin/src/main.cc
// and obs_plugin/src/filter.cc exercise:
//   * AviSynth: frames n = -16..-1 are the mirrored warm-up (source frame |n|),
//     RGB32 frames are bottom-up, passed with a NEGATIVE stride and ptr = first logical
//     row (reference avisynth_plugin/src/main.cc:41, 93-110, 125-142);
//   * OBS: steady per-tick loop, runtime destroyed and recreated on a model switch
//     (reference obs_plugin/src/filter.cc:146-151, 291, 384-389), errors reported via
//     getExceptionString() inside a catch block, log lines through a LogSink.
// usage: plugin_harness <model.jupw> <frames.raw> <n_frames> <out.raw>
//   frames.raw: n_frames dense top-down BGRX frames; out.raw receives, per call
//   pattern, the last output frame (top-down BGRX) -- compared by the test against
//   the Python binding.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "JoshUpscale/core.h"

namespace core = JoshUpscale::core;

struct CountingSink : core::LogSink {
	int lines = 0;
	void operator()(const char *tag, core::LogLevel level, const std::string &message) override {
		++lines;
		std::fprintf(stderr, "[sink] %s %d %s\n", tag, static_cast<int>(level), message.c_str());
	}
};

int main(int argc, char **argv) {
	if (argc != 5) {
		std::fprintf(stderr, "usage: %s model frames.raw n out.raw\n", argv[0]);
		return 2;
	}
	static CountingSink sink;  // borrowed forever, like OBS's static sink
	core::setLogSink(&sink);
	const int n = std::atoi(argv[3]);
	try {
		std::unique_ptr<core::Runtime> rt(core::createRuntime(0, argv[1]));
		const std::size_t w = rt->getInputWidth(), h = rt->getInputHeight();
		const std::size_t ow = rt->getOutputWidth(), oh = rt->getOutputHeight();
		if (ow != 4 * w || oh != 4 * h) return 3;
		std::vector<std::uint8_t> frames(static_cast<std::size_t>(n) * w * h * 4);
		std::ifstream in(argv[2], std::ios::binary);
		in.read(reinterpret_cast<char *>(frames.data()), static_cast<std::streamsize>(frames.size()));
		if (!in) return 4;
		std::ofstream out(argv[4], std::ios::binary);

		// ---- AviSynth pattern: bottom-up storage, negative stride, mirrored warm-up ----
		std::vector<std::uint8_t> flipped(w * h * 4), result(ow * oh * 4), topdown(ow * oh * 4);
		for (int fn = -16; fn < n; ++fn) {
			const int src = fn < 0 ? (-fn < n ? -fn : n - 1) : fn;
			const std::uint8_t *f = frames.data() + static_cast<std::size_t>(src) * w * h * 4;
			for (std::size_t y = 0; y < h; ++y) {  // store bottom-up
				std::memcpy(flipped.data() + (h - 1 - y) * w * 4, f + y * w * 4, w * 4);
			}
			core::Image inImg{flipped.data() + (h - 1) * w * 4, core::DataLocation::CPU,
			    -static_cast<std::ptrdiff_t>(w * 4), w, h};
			core::Image outImg{result.data() + (oh - 1) * ow * 4, core::DataLocation::CPU,
			    -static_cast<std::ptrdiff_t>(ow * 4), ow, oh};
			rt->processImage(inImg, outImg);
		}
		for (std::size_t y = 0; y < oh; ++y) {
			std::memcpy(topdown.data() + y * ow * 4, result.data() + (oh - 1 - y) * ow * 4, ow * 4);
		}
		out.write(reinterpret_cast<const char *>(topdown.data()), static_cast<std::streamsize>(topdown.size()));

		// ---- OBS pattern: model switch = destroy + recreate (state starts from zero) ----
		rt.reset();
		rt.reset(core::createRuntime(0, argv[1]));
		for (int fn = 0; fn < n; ++fn) {
			core::Image inImg{frames.data() + static_cast<std::size_t>(fn) * w * h * 4,
			    core::DataLocation::CPU, static_cast<std::ptrdiff_t>(w * 4), w, h};
			core::Image outImg{topdown.data(), core::DataLocation::CPU,
			    static_cast<std::ptrdiff_t>(ow * 4), ow, oh};
			rt->processImage(inImg, outImg);
		}
		out.write(reinterpret_cast<const char *>(topdown.data()), static_cast<std::streamsize>(topdown.size()));

		// ---- error path: what the plugins do in their catch blocks ----
		int caught = 0;
		try {
			core::Image bad{frames.data(), core::DataLocation::CPU, static_cast<std::ptrdiff_t>(w * 4), w + 1, h};
			core::Image outImg{topdown.data(), core::DataLocation::CPU,
			    static_cast<std::ptrdiff_t>(ow * 4), ow, oh};
			rt->processImage(bad, outImg);
		} catch (...) {
			const std::string s = core::getExceptionString();
			if (s.find("invalid_argument") != std::string::npos) ++caught;
		}
		try {
			std::unique_ptr<core::Runtime> none(core::createRuntime(0, "/nonexistent/model.trt"));
		} catch (...) {
			const std::string s = core::getExceptionString();
			if (s.find("failure") != std::string::npos || s.find("ios") != std::string::npos) ++caught;
		}
		try {
			core::getGLDeviceIndex();
		} catch (const std::exception &) {
			++caught;
		}
		std::printf("harness ok: %d frames, %d errors caught as exceptions, %d log lines through the sink\n",
		    n, caught, sink.lines);
		return caught == 3 && sink.lines >= 3 ? 0 : 5;
	} catch (...) {
		std::fprintf(stderr, "harness failed: %s\n", core::getExceptionString().c_str());
		return 1;
	}
}
