// Logging of the runtime: messages go to a swappable sink; the default sink
// prints to stderr with a millisecond timestamp (same convention as the
// reference's core log, reference core/src/logging.cc:51-62 and
// core/public/JoshUpscale/core.h:21-28).
#pragma once

#include <string>

namespace ju {

enum class LogLevel : int { Info = 0, Warning = 1, Error = 2 };

using LogCallback = void (*)(const char *tag, int level, const char *message, void *user);

// nullptr restores the default stderr sink.
void setLogCallback(LogCallback cb, void *user);
void logMessage(LogLevel level, const char *tag, const std::string &message);

}  // namespace ju
