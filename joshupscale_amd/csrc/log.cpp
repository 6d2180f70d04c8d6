#include "log.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <mutex>

namespace ju {

namespace {

std::mutex g_LogMutex;
LogCallback g_Callback = nullptr;
void *g_User = nullptr;

void defaultSink(const char *tag, int level, const char *message) {
	static const char *const names[] = {"INFO", "WARNING", "ERROR"};
	using namespace std::chrono;
	const auto now = system_clock::now();
	const std::time_t t = system_clock::to_time_t(now);
	const auto ms = duration_cast<milliseconds>(now.time_since_epoch()).count() % 1000;
	std::tm tmv{};
	localtime_r(&t, &tmv);
	char buf[32];
	std::strftime(buf, sizeof(buf), "%H:%M:%S", &tmv);
	std::fprintf(stderr, "[%s.%03d] [%s] %s: %s\n", buf, static_cast<int>(ms),
	    names[level < 0 || level > 2 ? 0 : level], tag, message);
}

}  // namespace

void setLogCallback(LogCallback cb, void *user) {
	std::lock_guard<std::mutex> lock(g_LogMutex);
	g_Callback = cb;
	g_User = user;
}

void logMessage(LogLevel level, const char *tag, const std::string &message) {
	std::lock_guard<std::mutex> lock(g_LogMutex);
	if (g_Callback) {
		g_Callback(tag, static_cast<int>(level), message.c_str(), g_User);
	} else if (level != LogLevel::Info) {
		defaultSink(tag, static_cast<int>(level), message.c_str());
	} else {
		static const bool verbose = [] {
			const char *v = std::getenv("JU_VERBOSE");
			return v && v[0] == '1';
		}();
		if (verbose) defaultSink(tag, static_cast<int>(level), message.c_str());
	}
}

}  // namespace ju
