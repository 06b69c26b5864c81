// Logger of the engine, compatible with the reference's environment switches (utils/logger.cc:22-56, utils/logger.h:30-105):
//   FT_LOG_LEVEL = TRACE | DEBUG | INFO | WARNING | ERROR   (default INFO, as a release build of the reference; an unknown name
//                  is reported once and ignored)
//   FT_LOG_FIRST_RANK_ONLY = ON   every device but 0 logs errors only
// Lines go to stderr (stdout belongs to the caller: bench.py prints ONE JSON line there) as "[FT][LEVEL] message".
#pragma once
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace ftcf {

enum LogLevel { LOG_TRACE = 0, LOG_DEBUG = 10, LOG_INFO = 20, LOG_WARNING = 30, LOG_ERROR = 40 };

// the level of this process for `device` (FT_LOG_FIRST_RANK_ONLY decides per device, like the reference's per-device logger)
inline int log_threshold(const int device)
{
    static const int base = [] {
        const char* name = std::getenv("FT_LOG_LEVEL");
        if (!name) {
            return (int)LOG_INFO;
        }
        static const struct {
            const char* n;
            int         l;
        } tab[] = {{"TRACE", LOG_TRACE}, {"DEBUG", LOG_DEBUG}, {"INFO", LOG_INFO}, {"WARNING", LOG_WARNING}, {"ERROR", LOG_ERROR}};
        for (const auto& e : tab) {
            if (std::strcmp(name, e.n) == 0) {
                return e.l;
            }
        }
        std::fprintf(stderr,
                     "[FT][WARNING] Invalid logger level FT_LOG_LEVEL=%s. Ignore the environment variable and use a default "
                     "logging level.\n",
                     name);
        return (int)LOG_INFO;
    }();
    static const bool first_only = [] {
        const char* v = std::getenv("FT_LOG_FIRST_RANK_ONLY");
        return v && std::strcmp(v, "ON") == 0;
    }();
    return (first_only && device != 0) ? (int)LOG_ERROR : base;
}

inline const char* log_level_name(const int level)
{
    return level <= LOG_TRACE ? "TRACE" : level <= LOG_DEBUG ? "DEBUG" : level <= LOG_INFO ? "INFO" : level <= LOG_WARNING ? "WARNING" : "ERROR";
}

#if defined(__GNUC__)
__attribute__((format(printf, 3, 4)))
#endif
inline void log_at(const int level, const int device, const char* fmt, ...)
{
    if (level < log_threshold(device)) {
        return;
    }
    char    buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    std::fprintf(stderr, "[FT][%s] %s\n", log_level_name(level), buf);
}

}  // namespace ftcf

#define FT_LOG_TRACE(dev, ...) ::ftcf::log_at(::ftcf::LOG_TRACE, (dev), __VA_ARGS__)
#define FT_LOG_DEBUG(dev, ...) ::ftcf::log_at(::ftcf::LOG_DEBUG, (dev), __VA_ARGS__)
#define FT_LOG_INFO(dev, ...) ::ftcf::log_at(::ftcf::LOG_INFO, (dev), __VA_ARGS__)
#define FT_LOG_WARNING(dev, ...) ::ftcf::log_at(::ftcf::LOG_WARNING, (dev), __VA_ARGS__)
#define FT_LOG_ERROR(dev, ...) ::ftcf::log_at(::ftcf::LOG_ERROR, (dev), __VA_ARGS__)
