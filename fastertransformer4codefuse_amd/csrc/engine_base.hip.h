// Shared preamble of the engine's translation units (engine.hip, comm.hip, batcher.hip): includes, the error plumbing of the
// C ABI (every entry point runs under guarded(): exceptions become status codes + ftcf_last_error()).
#pragma once
#include <rccl/rccl.h>
#include <roctracer/roctx.h>

#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ftcf.h"
#include "ftcf_common.h"
#include "host_quant.h"
#include "kernels.h"
#include "layers.hip.h"
#include "logger.h"

using namespace ftcf;

// ---------------------------------------------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------------------------------------------
extern thread_local std::string g_last_error;  // (engine.hip)

template<typename F>
static int guarded(F&& f)
{
    try {
        f();
        return FTCF_OK;
    }
    catch (const ftcf::Error& e) {
        g_last_error = e.what();
        FT_LOG_DEBUG(0, "call failed (%d): %s", e.code, e.what());  // (the binding raises it: not an ERROR line of its own)
        return e.code;
    }
    catch (const std::exception& e) {
        g_last_error = e.what();
        FT_LOG_DEBUG(0, "call failed: %s", e.what());
        return FTCF_ERR_INVALID_ARG;
    }
}

static void require_device()
{
    if (ftcf_device_count() <= 0) {
        throw Error(FTCF_ERR_NO_DEVICE,
                    "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback in this library)");
    }
}
