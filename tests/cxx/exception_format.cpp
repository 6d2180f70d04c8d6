// Compiled by tests/test_cxx_surface.py against include/JoshUpscale/core.h only: the
// exception-string format of the plugin surface must be the reference's
// (core/src/exception.cc:51-79), byte for byte -- callers log it verbatim
// (obs_plugin/src/filter.cc:386-389, avisynth_plugin/src/main.cc:145-148).
#include <JoshUpscale/core.h>

#include <cstdio>
#include <stdexcept>
#include <string>

namespace core = JoshUpscale::core;

struct Probe : core::Runtime {
	void processImage(const core::Image &, const core::Image &) override {}
	Probe() {
		m_InputWidth = 1;   // the reference's member names (core.h:84-88) are part of the surface
		m_InputHeight = 2;
		m_OutputWidth = 3;
		m_OutputHeight = 4;
	}
};

int main() {
	Probe p;
	if (p.getInputWidth() != 1 || p.getInputHeight() != 2 || p.getOutputWidth() != 3 || p.getOutputHeight() != 4) return 2;
	try {
		try {
			try {
				throw std::invalid_argument("innermost");
			} catch (...) {
				std::throw_with_nested(std::runtime_error("middle"));
			}
		} catch (...) {
			std::throw_with_nested(std::logic_error("outer"));
		}
	} catch (...) {
		std::printf("[%s]\n", core::getExceptionString().c_str());
	}
	try {
		throw 42;
	} catch (...) {
		std::printf("[%s]\n", core::getExceptionString().c_str());
	}
	try {
		core::createRuntime(0, "/nonexistent/model.jupw");
	} catch (...) {
		std::printf("[%s]\n", core::getExceptionString().c_str());
	}
	return 0;
}
