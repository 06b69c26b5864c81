// Host-side engine + C ABI (include/ftcf.h) of the MI355X GPT-NeoX / CodeFuse decoder.
//
// Host classes mirror the reference's layer split (layers are thin: they own no scratch of their own and never
// reMalloc per call -- one arena is planned per request shape):
//   GptNeoX               <- models/gptneox/GptNeoX.cc:386-1052 (generation loop, LM head, dynamic decode, outputs)
//   GptNeoXContextDecoder <- models/gptneox/GptNeoXContextDecoder.cc:223-512 (prefill)
//   GptNeoXDecoder        <- models/gptneox/GptNeoXDecoder.cc:197-389 (one token through L layers)
//   DecoderSelfAttentionLayer / FfnLayer / DynamicDecodeLayer are the launch helpers used by those.
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
static thread_local std::string g_last_error;

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

extern "C" const char* ftcf_last_error(void)
{
    return g_last_error.c_str();
}
extern "C" int ftcf_version(void)
{
    return FTCF_VERSION;
}
extern "C" int ftcf_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

static void require_device()
{
    if (ftcf_device_count() <= 0) {
        throw Error(FTCF_ERR_NO_DEVICE,
                    "no HIP device visible: the MI355X kernels cannot run (there is no CPU fallback in this library)");
    }
}

// ---------------------------------------------------------------------------------------------------------------
// communicator (RCCL over xGMI) -- utils/nccl_utils.cc:56-435, nccl_inherit_utils.cc:25-68
// ---------------------------------------------------------------------------------------------------------------
// A communicator is either an RCCL communicator (one process per GPU, the product) or a member of a LOCAL GROUP: the ranks
// of a tensor-parallel job living in ONE process on ONE device, each driven by its own host thread (ftcf_comm_init_local).
// The local group exists so that the engine's tensor-parallel path -- column / row sharding, the per-layer all-reduce, the
// x / TP residual, the vocabulary split + all-gather + transpose, and the in-kernel exchange of the persistent decode
// kernel -- can be executed and checked against TP = 1 and the oracle on a single-GPU box.  Its collectives are host
// synchronous (stream sync + thread barrier + a summing / copying kernel): slow, deterministic, test infrastructure.
struct LocalGroup {
    int                     world = 0;
    std::mutex              m;
    std::condition_variable cv;
    int                     arrived = 0;
    long                    gen = 0;
    std::vector<void*>      slot;   // per rank: the buffer it brought to the collective in progress
    std::vector<void*>      win;    // per rank: exchange window (device memory), see ftcf_comm::window
    std::vector<size_t>     win_bytes;
    std::vector<const void*> item;  // per rank: an opaque pointer for the group launch of the persistent kernel
    void barrier()
    {
        std::unique_lock<std::mutex> lk(m);
        const long g = gen;
        if (++arrived == world) {
            arrived = 0;
            gen++;
            cv.notify_all();
        }
        else {
            cv.wait(lk, [&] { return gen != g; });
        }
    }
};

struct ftcf_comm {
    ncclComm_t                  comm = nullptr;
    std::shared_ptr<LocalGroup> local;
    // host-exchange communicator (ftcf_comm_init_host_exchange): every exchange is an all-gather of host bytes by the caller
    ftcf_host_allgather_fn      hx = nullptr;
    void*                       hx_user = nullptr;
    std::vector<char>           hx_send, hx_recv;
    int                         world = 1, rank = 0, device = 0;
    void*                       tmp = nullptr;  // local group: result buffer of the emulated all-reduce
    size_t                      tmp_bytes = 0;
    // in-kernel exchange windows of the persistent tensor-parallel decode kernel: win[r] = rank r's window as THIS rank
    // addresses it (own memory for r == rank; a peer mapping -- hipIpc over xGMI -- or, in a local group, the same device)
    std::vector<void*>          win;
    size_t                      win_bytes = 0;
    bool                        win_ok = false, win_tried = false;
    // RCCL-free all-reduce of the prompt phase's messages through the same windows (k_window_allreduce): behind the granule
    // area of the decode exchange lie 16 flags and four message buffers ([call parity][input | reduced], ar_cap bytes each)
    size_t                      ar_flag_off = 0, ar_data_off = 0, ar_cap = 0;
    unsigned                    ar_seq = 0;       // calls so far (every rank calls in the same order)
    int*                        ar_sync = nullptr;  // device: two arrival counters + the sticky give-up word
    bool                        ar_failed = false;
    int                         ar_nb = 0;        // grid of the launches so far (the arrival counters count in its units)
};

#define FTCF_NCCL_CHECK(expr)                                                                                          \
    do {                                                                                                               \
        ncclResult_t _r = (expr);                                                                                      \
        if (_r != ncclSuccess) {                                                                                       \
            throw Error(FTCF_ERR_COMM, std::string("RCCL error ") + ncclGetErrorString(_r) + " (" #expr ")");          \
        }                                                                                                              \
    } while (0)

// ---- host-exchange communicator: all-gather of host bytes through the caller, collectives staged through host memory ----
static void hx_allgather(ftcf_comm* c, const void* send, void* recv, size_t bytes)
{
    if (c->hx(c->hx_user, send, recv, bytes) != 0) {
        throw Error(FTCF_ERR_COMM, "host-exchange communicator: the caller's all-gather failed");
    }
}
static void hx_barrier(ftcf_comm* c)
{
    int              z = 0;
    std::vector<int> all(c->world);
    hx_allgather(c, &z, all.data(), sizeof(int));
}
// min (op 0) / max (op 1) of one int over the ranks
static int hx_reduce_int(ftcf_comm* c, int v, int op)
{
    std::vector<int> all(c->world);
    hx_allgather(c, &v, all.data(), sizeof(int));
    int r = v;
    for (int x : all) {
        r = op ? std::max(r, x) : std::min(r, x);
    }
    return r;
}
// sum of a device buffer over the ranks: fp32 in rank order, rounded once -- the same bits on every rank
static void hx_allreduce(ftcf_comm* c, void* buf, size_t count, bool fp16, hipStream_t s)
{
    const size_t bytes = count * (fp16 ? 2 : 4);
    c->hx_send.resize(bytes);
    c->hx_recv.resize(bytes * c->world);
    FTCF_HIP_CHECK(hipMemcpyAsync(c->hx_send.data(), buf, bytes, hipMemcpyDeviceToHost, s));
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
    hx_allgather(c, c->hx_send.data(), c->hx_recv.data(), bytes);
    if (fp16) {
        const f16* all = reinterpret_cast<const f16*>(c->hx_recv.data());
        f16*       out = reinterpret_cast<f16*>(c->hx_send.data());
#pragma omp parallel for
        for (long i = 0; i < (long)count; i++) {
            float a = 0.f;
            for (int r = 0; r < c->world; r++) {
                a += (float)all[(size_t)r * count + i];
            }
            out[i] = (f16)a;
        }
    }
    else {
        const float* all = reinterpret_cast<const float*>(c->hx_recv.data());
        float*       out = reinterpret_cast<float*>(c->hx_send.data());
#pragma omp parallel for
        for (long i = 0; i < (long)count; i++) {
            float a = 0.f;
            for (int r = 0; r < c->world; r++) {
                a += all[(size_t)r * count + i];
            }
            out[i] = a;
        }
    }
    FTCF_HIP_CHECK(hipMemcpyAsync(buf, c->hx_send.data(), bytes, hipMemcpyHostToDevice, s));
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
}
// in place: rank r's segment lives at offset r of buf
static void hx_allgather_device(ftcf_comm* c, void* buf, size_t count_per_rank, size_t esz, hipStream_t s)
{
    const size_t seg = count_per_rank * esz;
    c->hx_send.resize(seg);
    c->hx_recv.resize(seg * c->world);
    FTCF_HIP_CHECK(hipMemcpyAsync(c->hx_send.data(), (const char*)buf + (size_t)c->rank * seg, seg, hipMemcpyDeviceToHost, s));
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
    hx_allgather(c, c->hx_send.data(), c->hx_recv.data(), seg);
    FTCF_HIP_CHECK(hipMemcpyAsync(buf, c->hx_recv.data(), seg * c->world, hipMemcpyHostToDevice, s));
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
}

// Host wait on a stream that carries RCCL work (utils/nccl_utils.cc:215-272, ftNcclStreamSynchronize): instead of blocking
// in hipStreamSynchronize -- where a dead or hung peer hangs this rank for good -- poll the stream and the communicator's
// asynchronous error state; an asynchronous error or FTCF_COMM_TIMEOUT_S seconds without progress (default 600, 0 = wait for
// ever) aborts the communicator and raises FTCF_ERR_COMM.  Single-rank and local-group communicators block plainly.
static double comm_timeout_s()
{
    static const double t = [] {
        const char* e = getenv("FTCF_COMM_TIMEOUT_S");
        return e ? atof(e) : 600.0;
    }();
    return t;
}
template<typename Query>
static void comm_wait(ftcf_comm* c, Query&& query, const char* what)
{
    const auto   t0 = std::chrono::steady_clock::now();
    const double limit = comm_timeout_s();
    for (long spin = 0;; spin++) {
        const hipError_t e = query();
        if (e == hipSuccess) {
            return;
        }
        if (e != hipErrorNotReady) {
            throw Error(FTCF_ERR_HIP, std::string("HIP error while waiting for ") + what + ": " + hipGetErrorString(e));
        }
        if ((spin & 63) == 63) {
            ncclResult_t async = ncclSuccess;
            FTCF_NCCL_CHECK(ncclCommGetAsyncError(c->comm, &async));
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            const bool   late = limit > 0 && waited > limit;
            if (async != ncclSuccess || late) {
                (void)ncclCommAbort(c->comm);  // the reference does the same and leaves the communicator unusable
                c->comm = nullptr;
                throw Error(FTCF_ERR_COMM,
                            late ? std::string("tensor-parallel peer made no progress for ") + std::to_string((int)waited)
                                       + " s while waiting for " + what + " (FTCF_COMM_TIMEOUT_S); communicator aborted" :
                                   std::string("RCCL asynchronous error ") + ncclGetErrorString(async) + " while waiting for "
                                       + what + "; communicator aborted");
            }
            if (waited > 1e-3) {
                std::this_thread::sleep_for(std::chrono::microseconds(50));
            }
        }
    }
}
static void comm_stream_sync(ftcf_comm* c, hipStream_t s, const char* what = "the engine stream")
{
    if (!c || c->local || c->world == 1 || !c->comm) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        return;
    }
    comm_wait(c, [&] { return hipStreamQuery(s); }, what);
}
static void comm_event_sync(ftcf_comm* c, hipEvent_t ev, const char* what = "a recorded event")
{
    if (!c || c->local || c->world == 1 || !c->comm) {
        FTCF_HIP_CHECK(hipEventSynchronize(ev));
        return;
    }
    comm_wait(c, [&] { return hipEventQuery(ev); }, what);
}

// roctx ranges around the host-side phases (utils/nvtx_utils.cc:59-87: FT_NVTX=ON there, FTCF_ROCTX=ON here; rocprofv3
// --marker-trace shows them).  The per-token kernels inside a replayed hipGraph carry no ranges: the range is the token.
static bool roctx_on()
{
    static const bool on = [] {
        const char* e = getenv("FTCF_ROCTX");
        return e && (std::string(e) == "ON" || std::string(e) == "1");
    }();
    return on;
}
struct Range {
    bool on;
    explicit Range(const char* name): on(roctx_on())
    {
        if (on) {
            roctxRangePushA(name);
        }
    }
    ~Range()
    {
        if (on) {
            roctxRangePop();
        }
    }
    Range(const Range&) = delete;
    Range& operator=(const Range&) = delete;
};

// ---- local group collectives (test infrastructure, see above) ----
__global__ void k_local_allreduce_f16(f16* out, const f16* const* src, int world, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < world; r++) {  // rank order, fp32, rounded once: the same value on every rank
            a += (float)src[r][i];
        }
        out[i] = (f16)a;
    }
}
__global__ void k_local_allreduce_f32(float* out, const float* const* src, int world, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int r = 0; r < world; r++) {
            a += src[r][i];
        }
        out[i] = a;
    }
}

static void local_allreduce(ftcf_comm* c, void* buf, size_t count, bool fp16, hipStream_t s)
{
    LocalGroup& g   = *c->local;
    const size_t esz = fp16 ? 2 : 4;
    const size_t ptr_bytes = sizeof(void*) * (size_t)g.world;
    if (c->tmp_bytes < count * esz + ptr_bytes + 256) {
        if (c->tmp) {
            FTCF_HIP_CHECK(hipFree(c->tmp));
        }
        c->tmp_bytes = count * esz + ptr_bytes + 256;
        FTCF_HIP_CHECK(hipMalloc(&c->tmp, c->tmp_bytes));
    }
    FTCF_HIP_CHECK(hipStreamSynchronize(s));  // my contribution is complete
    g.slot[c->rank] = buf;
    g.barrier();
    char* ptrs = (char*)c->tmp + ((count * esz + 255) & ~(size_t)255);
    FTCF_HIP_CHECK(hipMemcpyAsync(ptrs, g.slot.data(), ptr_bytes, hipMemcpyHostToDevice, s));
    const int blocks = (int)std::min<size_t>(1024, (count + 255) / 256);
    if (fp16) {
        hipLaunchKernelGGL(k_local_allreduce_f16, dim3(blocks), dim3(256), 0, s, (f16*)c->tmp, (const f16* const*)ptrs,
                           g.world, count);
    }
    else {
        hipLaunchKernelGGL(k_local_allreduce_f32, dim3(blocks), dim3(256), 0, s, (float*)c->tmp,
                           (const float* const*)ptrs, g.world, count);
    }
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
    g.barrier();  // every rank has read every buffer: they may be overwritten now
    FTCF_HIP_CHECK(hipMemcpyAsync(buf, c->tmp, count * esz, hipMemcpyDeviceToDevice, s));
}

static void local_allgather(ftcf_comm* c, void* buf, size_t count_per_rank, bool fp16, hipStream_t s)
{
    LocalGroup& g   = *c->local;
    const size_t seg = count_per_rank * (fp16 ? 2 : 4);
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
    g.slot[c->rank] = buf;
    g.barrier();
    for (int r = 0; r < g.world; r++) {  // rank r's segment lives at offset r in ITS buffer (in-place convention)
        if (r != c->rank) {
            FTCF_HIP_CHECK(hipMemcpyAsync((char*)buf + (size_t)r * seg, (const char*)g.slot[r] + (size_t)r * seg, seg,
                                          hipMemcpyDeviceToDevice, s));
        }
    }
    FTCF_HIP_CHECK(hipStreamSynchronize(s));
    g.barrier();
}

// ---- exchange windows of the persistent tensor-parallel decode kernel ----------------------------------------------
// One window per rank, written by every rank from inside its kernel (system-scope granule stores) and polled by the owner:
//   local group : plain device memory, the peers' pointers come through the group;
//   RCCL ranks  : fine-grained device memory exported with hipIpcGetMemHandle, the handles travel through an RCCL
//                 all-gather, peers map them with hipIpcOpenMemHandle (xGMI peer access), and a hand-shake kernel proves
//                 on THIS hardware that a granule stored by a peer's kernel becomes visible to a polling kernel here --
//                 any failure on any rank (agreed on through an all-reduce) leaves win_ok false on EVERY rank and the
//                 engine keeps the RCCL path (per-stage launches + ncclAllReduce per layer).
// Collective: every rank must call it with the same size.
__global__ void k_window_handshake(unsigned long long* const* win, int world, int rank, unsigned tag, int* result,
                                   long long limit_ticks)
{
    // granule [rank] of every rank's window <- {tag, rank}; then wait for every peer's granule in the own window
    const int t = threadIdx.x;
    if (t < world) {
        __hip_atomic_store((__attribute__((address_space(1))) unsigned long long*)(win[t] + rank),
                           ((unsigned long long)tag << 32) | (unsigned)rank, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    bool            ok = true;
    const long long t0 = wall_clock64();
    if (t < world) {
        for (;;) {
            const unsigned long long v = __hip_atomic_load(
                (const __attribute__((address_space(1))) unsigned long long*)(win[rank] + t), __ATOMIC_RELAXED,
                __HIP_MEMORY_SCOPE_SYSTEM);
            if ((unsigned)(v >> 32) == tag && (unsigned)v == (unsigned)t) {
                break;
            }
            if (wall_clock64() - t0 > limit_ticks) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(32);
        }
    }
    if (!ok) {
        atomicExch(result, 0);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Two-shot all-reduce over the peer-mapped exchange windows (the reference's twoShotAllReduceKernel,
// kernels/custom_ar_kernels.cu:202-260, for the messages its one-shot form is too small for): no RCCL call, no host in the
// loop -- one launch per rank.
//   A  copy x into the own window (buffer of this call's parity); the last workgroup to finish tells every peer (flag A)
//   B  when every rank's flag A shows this call: reduce-scatter -- rank r adds chunk r of all ranks' inputs IN RANK ORDER in
//      fp32, rounds once (the sum every rank would compute: the same bits everywhere), writes it to its window's result
//      buffer and to x; the last workgroup tells every peer (flag B)
//   C  when every flag B shows this call: all-gather -- chunk c comes from rank c's result buffer
// Two parities of buffers: a peer can be at most one call behind (it posts its input of call k only after it has finished call
// k - 1, and call k - 1 here needed that input), so the buffers of call k - 2 are free when call k overwrites them.  Flags carry
// the call number (monotone, never reset while the window lives).  Every spin is bounded and reports through a sticky word.
// ---------------------------------------------------------------------------------------------------------------------
struct WinArParams {
    unsigned long long* win[8];  // every rank's window as this rank addresses it
    int                 tp, rank;
    f16*                x;
    size_t              count;  // halves, a multiple of 8 * tp
    size_t              flag_off, data_off, cap;  // bytes
    unsigned            seq;
    int*                sync;  // [0] arrivals of step A, [1] of step B (monotone), [2] give-up word
    long long           limit_ticks;
};

__device__ __forceinline__ bool winar_wait(const WinArParams& p, const size_t slot0, const long long t0)
{
    // lanes 0..tp-1 of every workgroup's first wave poll the tp flags of the OWN window
    bool ok = true;
    if ((int)threadIdx.x < p.tp) {
        const auto* f = (const __attribute__((address_space(1))) unsigned long long*)(reinterpret_cast<char*>(p.win[p.rank]) + p.flag_off)
                        + slot0 + threadIdx.x;
        for (;;) {
            if ((unsigned)__hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == p.seq) {
                break;
            }
            if (wall_clock64() - t0 > p.limit_ticks
                || __hip_atomic_load((__attribute__((address_space(1))) int*)&p.sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                ok = false;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) {
            __hip_atomic_store((__attribute__((address_space(1))) int*)&p.sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    return __syncthreads_and(ok ? 1 : 0) != 0;
}

__global__ __launch_bounds__(256) void k_window_allreduce(const WinArParams p)
{
    typedef unsigned long long   u64;
    typedef __attribute__((address_space(1))) u64 gu64;
    const long long t0   = wall_clock64();
    const int       par  = (int)(p.seq & 1u);
    const size_t    n8   = p.count / 8, c8 = n8 / p.tp;  // 16-byte vectors in all / per chunk
    char*           mine = reinterpret_cast<char*>(p.win[p.rank]);
    u32x4*          X    = reinterpret_cast<u32x4*>(mine + p.data_off + (size_t)par * 2 * p.cap);
    u32x4*          R    = reinterpret_cast<u32x4*>(mine + p.data_off + (size_t)par * 2 * p.cap + p.cap);
    u32x4*          x8   = reinterpret_cast<u32x4*>(p.x);
    const size_t    gsz  = (size_t)gridDim.x * blockDim.x, gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    auto tell = [&](const int which, const int arrivals_slot) {
        // the last workgroup of this rank to arrive stores the call number into slot [rank] of every rank's flag array
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = atomicAdd(&p.sync[arrivals_slot], 1);
            if ((unsigned)(old + 1) == p.seq * gridDim.x) {
                for (int r = 0; r < p.tp; r++) {
                    gu64* f = (gu64*)(reinterpret_cast<char*>(p.win[r]) + p.flag_off) + (size_t)which * 8 + p.rank;
                    __hip_atomic_store(f, (u64)p.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
    };
    // ---- A: the input into the own window ----
    for (size_t i = gid; i < n8; i += gsz) {
        X[i] = x8[i];
    }
    tell(0, 0);
    // ---- B: reduce-scatter of chunk [rank] ----
    if (!winar_wait(p, 0, t0)) {
        return;
    }
    for (size_t i = gid; i < c8; i += gsz) {
        const size_t at = (size_t)p.rank * c8 + i;
        float        acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < p.tp; r++) {  // rank order: the same sum on every rank
            const gu64* src = (const gu64*)(reinterpret_cast<char*>(p.win[r]) + p.data_off + (size_t)par * 2 * p.cap) + at * 2;
            const u64   lo = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const u64   hi = __hip_atomic_load(src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const u32x4 v  = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
            const f16x8 h  = __builtin_bit_cast(f16x8, v);
#pragma unroll
            for (int e = 0; e < 8; e++) {
                acc[e] += (float)h[e];
            }
        }
        f16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) {
            o[e] = (f16)acc[e];
        }
        const u32x4 ov = __builtin_bit_cast(u32x4, o);
        R[at]  = ov;
        x8[at] = ov;
    }
    tell(1, 1);
    // ---- C: all-gather of the other ranks' chunks ----
    if (!winar_wait(p, 8, t0)) {
        return;
    }
    for (int k = 1; k < p.tp; k++) {
        const int   c   = (p.rank + k) % p.tp;  // (every rank starts at another peer)
        const gu64* src = (const gu64*)(reinterpret_cast<char*>(p.win[c]) + p.data_off + (size_t)par * 2 * p.cap + p.cap);
        for (size_t i = gid; i < c8; i += gsz) {
            const size_t at = (size_t)c * c8 + i;
            const u64    lo = __hip_atomic_load(src + at * 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const u64    hi = __hip_atomic_load(src + at * 2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            x8[at]          = u32x4{(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
        }
    }
}

// true when the all-reduce went through the windows (else the caller uses its collective)
static bool window_allreduce(ftcf_comm* c, f16* buf, size_t count, hipStream_t s)
{
    static const int on = getenv("FTCF_TP_WINAR") ? atoi(getenv("FTCF_TP_WINAR")) : 1;
    // (not in a local group: its ranks are streams of ONE process on one device, and streams that share a hardware queue run
    // their kernels one after the other -- a rank's kernel would wait for a peer's that cannot start: 2 s, give up, replay)
    if (!on || c->local || !c->win_ok || c->ar_failed || c->ar_cap == 0 || c->world < 2 || c->world > 8 || count % ((size_t)8 * c->world) != 0
        || count * 2 > c->ar_cap || count * 2 < (size_t)64 * 1024) {
        return false;  // (small messages: the decode path has its own in-kernel exchange; RCCL / the emulation otherwise)
    }
    if (!c->ar_sync) {
        FTCF_HIP_CHECK(hipMalloc((void**)&c->ar_sync, 64));
        FTCF_HIP_CHECK(hipMemsetAsync(c->ar_sync, 0, 64, s));
    }
    // workgroups per rank: every rank's grid must be resident together with its peers' (ranks sharing one device -- the local
    // group, two test processes -- split the compute units) and next to a GEMM on another stream
    static const int nb_env = getenv("FTCF_TP_WINAR_NB") ? atoi(getenv("FTCF_TP_WINAR_NB")) : 0;
    const int        shared = (c->local || c->hx) ? c->world : 1;
    const int        nb     = nb_env > 0 ? nb_env : std::max(8, 128 / shared);
    WinArParams      p{};
    for (int r = 0; r < c->world; r++) {
        p.win[r] = static_cast<unsigned long long*>(c->win[r]);
    }
    p.tp          = c->world;
    p.rank        = c->rank;
    p.x           = buf;
    p.count       = count;
    p.flag_off    = c->ar_flag_off;
    p.data_off    = c->ar_data_off;
    p.cap         = c->ar_cap;
    p.seq         = ++c->ar_seq;
    p.sync        = c->ar_sync;
    p.limit_ticks = (long long)200000000;  // 100 MHz ticks: 2 s
    hipLaunchKernelGGL(k_window_allreduce, dim3(nb), dim3(256), 0, s, p);
    FTCF_HIP_CHECK(hipGetLastError());
    c->ar_nb = nb;
    return true;
}

static void comm_barrier(ftcf_comm* c, hipStream_t s, int* d_scratch)
{
    if (c->local) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        c->local->barrier();
        return;
    }
    if (c->hx) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        hx_barrier(c);
        return;
    }
    FTCF_NCCL_CHECK(ncclAllReduce(d_scratch, d_scratch, 1, ncclInt32, ncclMin, c->comm, s));
    comm_stream_sync(c, s, "a tensor-parallel collective");
}

// all-reduce (min) of a host flag over the communicator
static int comm_agree(ftcf_comm* c, int flag, hipStream_t s, int* d_scratch)
{
    if (c->hx) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        return hx_reduce_int(c, flag, 0);
    }
    FTCF_HIP_CHECK(hipMemcpyAsync(d_scratch, &flag, sizeof(int), hipMemcpyHostToDevice, s));
    FTCF_NCCL_CHECK(ncclAllReduce(d_scratch, d_scratch, 1, ncclInt32, ncclMin, c->comm, s));
    int out = 0;
    FTCF_HIP_CHECK(hipMemcpyAsync(&out, d_scratch, sizeof(int), hipMemcpyDeviceToHost, s));
    comm_stream_sync(c, s, "a tensor-parallel collective");
    return out;
}

// all-reduce (max) of a host int (local group: through the group's slots)
static int comm_max(ftcf_comm* c, int v, hipStream_t s, int* d_scratch)
{
    if (c->local) {
        LocalGroup& g = *c->local;
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        g.slot[c->rank] = reinterpret_cast<void*>((intptr_t)v);
        g.barrier();
        int m = v;
        for (int r = 0; r < g.world; r++) {
            m = std::max(m, (int)(intptr_t)g.slot[r]);
        }
        g.barrier();
        return m;
    }
    if (c->hx) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        return hx_reduce_int(c, v, 1);
    }
    FTCF_HIP_CHECK(hipMemcpyAsync(d_scratch, &v, sizeof(int), hipMemcpyHostToDevice, s));
    FTCF_NCCL_CHECK(ncclAllReduce(d_scratch, d_scratch, 1, ncclInt32, ncclMax, c->comm, s));
    int out = 0;
    FTCF_HIP_CHECK(hipMemcpyAsync(&out, d_scratch, sizeof(int), hipMemcpyDeviceToHost, s));
    comm_stream_sync(c, s, "a tensor-parallel collective");
    return out;
}

static void comm_ensure_window(ftcf_comm* c, size_t bytes, hipStream_t s)
{
    bytes = (bytes + 4095) & ~(size_t)4095;
    if (c->win_ok && c->win_bytes >= bytes) {
        return;
    }
    if (c->win_tried && !c->local) {
        return;  // the RCCL ranks agreed once that the windows do not work here: stay on the collective path
    }
    c->win_tried = true;
    c->win.assign(c->world, nullptr);
    if (c->local) {
        LocalGroup& g = *c->local;
        void*       mine = nullptr;
        FTCF_HIP_CHECK(hipMalloc(&mine, bytes));
        FTCF_HIP_CHECK(hipMemsetAsync(mine, 0, bytes, s));
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        g.barrier();  // nobody is still using the old windows
        if (g.win[c->rank]) {
            (void)hipFree(g.win[c->rank]);
        }
        g.win[c->rank]       = mine;
        g.win_bytes[c->rank] = bytes;
        g.barrier();
        for (int r = 0; r < c->world; r++) {
            c->win[r] = g.win[r];
        }
        g.barrier();
        c->win_bytes = bytes;
        c->win_ok    = true;
        return;
    }
    // ---- RCCL ranks: IPC mapping + hand-shake, with a collective agreement after every step that can fail ----
    struct Rec {
        hipIpcMemHandle_t h;
        int               ok, pad[3];
    };
    int* d_scratch = nullptr;
    FTCF_HIP_CHECK(hipMalloc((void**)&d_scratch, 256));
    FTCF_HIP_CHECK(hipMemsetAsync(d_scratch, 0, 256, s));
    void* mine = nullptr;
    Rec   me{};
    me.ok = 1;
    if (getenv("FTCF_TP_WINDOWS") && atoi(getenv("FTCF_TP_WINDOWS")) == 0) {
        me.ok = 0;
    }
    if (me.ok && hipExtMallocWithFlags(&mine, bytes, hipDeviceMallocFinegrained) != hipSuccess) {
        (void)hipGetLastError();
        mine = nullptr;
        if (hipMalloc(&mine, bytes) != hipSuccess) {
            (void)hipGetLastError();
            mine  = nullptr;
            me.ok = 0;
        }
    }
    if (me.ok && hipIpcGetMemHandle(&me.h, mine) != hipSuccess) {
        (void)hipGetLastError();
        me.ok = 0;
    }
    std::vector<Rec> recs(c->world);
    Rec*             d_recs = nullptr;
    FTCF_HIP_CHECK(hipMalloc((void**)&d_recs, sizeof(Rec) * c->world));
    if (c->hx) {
        FTCF_HIP_CHECK(hipStreamSynchronize(s));
        hx_allgather(c, &me, recs.data(), sizeof(Rec));
    }
    else {
        FTCF_HIP_CHECK(hipMemcpyAsync(d_recs + c->rank, &me, sizeof(Rec), hipMemcpyHostToDevice, s));
        FTCF_NCCL_CHECK(ncclAllGather(d_recs + c->rank, d_recs, sizeof(Rec), ncclChar, c->comm, s));
        FTCF_HIP_CHECK(hipMemcpyAsync(recs.data(), d_recs, sizeof(Rec) * c->world, hipMemcpyDeviceToHost, s));
        comm_stream_sync(c, s, "a tensor-parallel collective");
    }
    int ok = 1;
    for (int r = 0; r < c->world; r++) {
        ok &= recs[r].ok;
    }
    std::vector<void*> opened(c->world, nullptr);
    if (ok) {
        for (int r = 0; r < c->world && ok; r++) {
            if (r == c->rank) {
                c->win[r] = mine;
            }
            else if (hipIpcOpenMemHandle(&opened[r], recs[r].h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
                (void)hipGetLastError();
                opened[r] = nullptr;
                ok        = 0;
            }
            else {
                c->win[r] = opened[r];
            }
        }
    }
    ok = comm_agree(c, ok, s, d_scratch);
    if (ok) {
        // hand-shake on the hardware: zero, barrier, every rank's kernel stores to all and polls its own (2 s bound)
        FTCF_HIP_CHECK(hipMemsetAsync(mine, 0, bytes, s));
        comm_barrier(c, s, d_scratch + 1);
        void** d_win = nullptr;
        int*   d_res = nullptr;
        FTCF_HIP_CHECK(hipMalloc((void**)&d_win, sizeof(void*) * c->world + 64));
        d_res = reinterpret_cast<int*>(reinterpret_cast<char*>(d_win) + sizeof(void*) * c->world);
        const int one = 1;
        FTCF_HIP_CHECK(hipMemcpyAsync(d_win, c->win.data(), sizeof(void*) * c->world, hipMemcpyHostToDevice, s));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_res, &one, sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_window_handshake, dim3(1), dim3(64), 0, s, (unsigned long long* const*)d_win, c->world, c->rank,
                           0x5eedu, d_res, (long long)200000000);  // 100 MHz ticks: 2 s
        int res = 0;
        FTCF_HIP_CHECK(hipMemcpyAsync(&res, d_res, sizeof(int), hipMemcpyDeviceToHost, s));
        comm_stream_sync(c, s, "a tensor-parallel collective");
        (void)hipFree(d_win);
        ok = comm_agree(c, res, s, d_scratch);
        if (ok) {
            FTCF_HIP_CHECK(hipMemsetAsync(mine, 0, bytes, s));
            comm_barrier(c, s, d_scratch + 1);
        }
    }
    (void)hipFree(d_recs);
    (void)hipFree(d_scratch);
    if (!ok) {
        for (int r = 0; r < c->world; r++) {
            if (opened[r]) {
                (void)hipIpcCloseMemHandle(opened[r]);
            }
        }
        if (mine) {
            (void)hipFree(mine);
        }
        c->win.assign(c->world, nullptr);
        c->win_ok = false;
        if (c->rank == 0) {
            FT_LOG_WARNING(0, "tensor-parallel exchange windows unavailable: RCCL all-reduce per layer instead");
        }
        return;
    }
    c->win_bytes = bytes;
    c->win_ok    = true;
}

static std::mutex                                        g_local_mu;
static std::map<std::string, std::weak_ptr<LocalGroup>> g_local_groups;
static long                                              g_local_next = 1;

extern "C" int ftcf_comm_local_unique_id(uint8_t id[FTCF_UNIQUE_ID_BYTES])
{
    return guarded([&] {
        std::lock_guard<std::mutex> lk(g_local_mu);
        memset(id, 0, FTCF_UNIQUE_ID_BYTES);
        snprintf((char*)id, FTCF_UNIQUE_ID_BYTES, "ftcf-local-group-%ld", g_local_next++);
    });
}

extern "C" int ftcf_comm_init_local(const uint8_t id[FTCF_UNIQUE_ID_BYTES], int world_size, int rank, int device,
                                    ftcf_comm_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(out != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, "bad communicator args");
        require_device();
        auto c    = std::make_unique<ftcf_comm>();
        c->world  = world_size;
        c->rank   = rank;
        c->device = device;
        {
            std::lock_guard<std::mutex> lk(g_local_mu);
            const std::string key((const char*)id, strnlen((const char*)id, FTCF_UNIQUE_ID_BYTES));
            FTCF_CHECK_ARG(!key.empty(), "local group id is empty: call ftcf_comm_local_unique_id");
            auto g = g_local_groups[key].lock();
            if (!g) {
                g        = std::make_shared<LocalGroup>();
                g->world = world_size;
                g->slot.assign(world_size, nullptr);
                g->win.assign(world_size, nullptr);
                g->win_bytes.assign(world_size, 0);
                g->item.assign(world_size, nullptr);
                g_local_groups[key] = g;
            }
            FTCF_CHECK_ARG(g->world == world_size, "local group: world size mismatch");
            c->local = g;
        }
        *out = c.release();
    });
}

extern "C" int ftcf_comm_get_unique_id(uint8_t id[FTCF_UNIQUE_ID_BYTES])
{
    return guarded([&] {
        static_assert(sizeof(ncclUniqueId) == FTCF_UNIQUE_ID_BYTES, "unique id size");
        ncclUniqueId uid;
        FTCF_NCCL_CHECK(ncclGetUniqueId(&uid));
        memcpy(id, &uid, sizeof(uid));
    });
}

extern "C" int ftcf_comm_init(const uint8_t id[FTCF_UNIQUE_ID_BYTES], int world_size, int rank, int device,
                              ftcf_comm_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(out != nullptr && world_size >= 1 && rank >= 0 && rank < world_size, "bad communicator args");
        require_device();
        FTCF_HIP_CHECK(hipSetDevice(device));
        auto c    = std::make_unique<ftcf_comm>();
        c->world  = world_size;
        c->rank   = rank;
        c->device = device;
        ncclUniqueId uid;
        memcpy(&uid, id, sizeof(uid));
        FTCF_NCCL_CHECK(ncclCommInitRank(&c->comm, world_size, uid, rank));
        *out = c.release();
    });
}

extern "C" int ftcf_comm_init_host_exchange(int world_size, int rank, int device, ftcf_host_allgather_fn allgather, void* user,
                                            ftcf_comm_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(out != nullptr && allgather != nullptr && world_size >= 1 && rank >= 0 && rank < world_size,
                       "bad communicator args");
        require_device();
        FTCF_HIP_CHECK(hipSetDevice(device));
        auto c     = std::make_unique<ftcf_comm>();
        c->world   = world_size;
        c->rank    = rank;
        c->device  = device;
        c->hx      = allgather;
        c->hx_user = user;
        hx_barrier(c.get());  // every rank is up and the callback works
        *out = c.release();
    });
}

extern "C" int ftcf_comm_destroy(ftcf_comm_t c)
{
    return guarded([&] {
        if (c) {
            if (c->comm) {
                ncclCommDestroy(c->comm);
            }
            if (c->tmp) {
                (void)hipFree(c->tmp);
            }
            if (c->local && c->win_ok && c->rank < (int)c->win.size() && c->win[c->rank]) {
                (void)hipFree(c->win[c->rank]);
            }
            delete c;
        }
    });
}

extern "C" int ftcf_comm_allreduce_sum(ftcf_comm_t c, void* buf, size_t count, ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        FTCF_CHECK_ARG(c && (c->comm || c->local || c->hx), "communicator not initialised");
        if (c->local) {
            local_allreduce(c, buf, count, dtype == FTCF_FP16, (hipStream_t)stream);
            return;
        }
        if (c->hx) {
            hx_allreduce(c, buf, count, dtype == FTCF_FP16, (hipStream_t)stream);
            return;
        }
        FTCF_NCCL_CHECK(ncclAllReduce(buf, buf, count, dtype == FTCF_FP16 ? ncclFloat16 : ncclFloat32, ncclSum,
                                      c->comm, (hipStream_t)stream));
    });
}

extern "C" int ftcf_comm_allgather(ftcf_comm_t c, void* buf, size_t count_per_rank, ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        FTCF_CHECK_ARG(c && (c->comm || c->local || c->hx), "communicator not initialised");
        if (c->local) {
            local_allgather(c, buf, count_per_rank, dtype == FTCF_FP16, (hipStream_t)stream);
            return;
        }
        if (c->hx) {
            hx_allgather_device(c, buf, count_per_rank, dtype == FTCF_FP16 ? 2 : 4, (hipStream_t)stream);
            return;
        }
        const size_t esz = dtype == FTCF_FP16 ? 2 : 4;
        // in place: rank r's data lives at buf + r*count (ftNcclAllGather, nccl_utils.cc:70-82)
        FTCF_NCCL_CHECK(ncclAllGather((const char*)buf + (size_t)c->rank * count_per_rank * esz, buf, count_per_rank,
                                      dtype == FTCF_FP16 ? ncclFloat16 : ncclFloat32, c->comm, (hipStream_t)stream));
    });
}

// ---------------------------------------------------------------------------------------------------------------
// host quantiser entry points (libth_common counterpart)
// ---------------------------------------------------------------------------------------------------------------
extern "C" int ftcf_symmetric_quantize_int8(const void* weight, ftcf_dtype dtype, size_t E, size_t K, size_t N,
                                            int8_t* out_q, void* out_scale)
{
    return guarded([&] { host_symmetric_quantize_int8(weight, (int)dtype, E, K, N, out_q, out_scale); });
}
extern "C" int ftcf_int8_rowmajor_to_tiled(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] { host_int8_rowmajor_to_tiled(q, K, N, out); });
}
extern "C" int ftcf_int8_tiled_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] { host_int8_tiled_to_rowmajor(q, K, N, out); });
}
extern "C" int ftcf_int8_cuda_sm80_to_rowmajor(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(q && out && K % 64 == 0 && N % 2 == 0, "SM80 int8 layout needs K % 64 == 0 and N % 2 == 0");
        host_int8_cuda_sm80_to_rowmajor(q, K, N, out);
    });
}
extern "C" int ftcf_int8_rowmajor_to_cuda_sm80(const int8_t* q, size_t K, size_t N, int8_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(q && out && K % 64 == 0 && N % 2 == 0, "SM80 int8 layout needs K % 64 == 0 and N % 2 == 0");
        host_int8_rowmajor_to_cuda_sm80(q, K, N, out);
    });
}
extern "C" int ftcf_fp16_rowmajor_to_tiled(const void* w, size_t K, size_t N, void* out, void* stream)
{
    return guarded([&] {
        require_device();
        launch_fp16_rowmajor_to_tiled((const f16*)w, K, N, (f16*)out, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// kernel-level entry points
// ---------------------------------------------------------------------------------------------------------------
static float* abi_gemm_workspace(int m, hipStream_t s);
static void gemm_dispatch(const f16* A, const void* W, const f16* scale, const f16* bias, int act, f16* C, int m,
                          int n, int k, bool int8, hipStream_t s, float* smallm_ws = nullptr, size_t smallm_partial = 0,
                          int num_cu = 256, const int* d_step = nullptr, unsigned* smallm_seq = nullptr, float* tiled_ws = nullptr)
{
    if (m <= 4) {
        SplitKParams p{};
        p.x_a = A;
        p.W_a = W;
        p.scale_a = scale;
        p.bias = bias;
        p.out = C;
        p.N = n;
        p.KT_a = k / (int8 ? TILE_K_I8 : TILE_K_F16);
        p.KT_b = 0;
        p.act = act;
        p.tp = 1;
        plan_splitk(p, int8, m, 4);
        launch_gemv_splitk(p, int8, m, EPI_PLAIN, s);
    }
    else if (m <= 16) {
        launch_gemm_smallm(A, W, scale, bias, act, C, smallm_ws, smallm_partial, m, n, k, int8, num_cu, s, d_step,
                           smallm_seq);
    }
    else {
        launch_gemm_tiled(A, W, scale, bias, act, C, m, n, k, int8, s, tiled_ws);
    }
}

// the stand-alone GEMM entry points carry no workspace argument (the reference's runner takes one from its caller:
// fpA_intB_gemm.h gemm(..., workspace_ptr, workspace_bytes)): one split-K workspace per (device, stream) that has called with
// 17..320 rows, kept for the life of the process
static float* abi_gemm_workspace(int m, hipStream_t s)
{
    if (m <= 16 || m > gemm_tiled_splitk_max_m()) {
        return nullptr;  // (only the split-K form of 17..320 rows uses it)
    }
    static std::mutex                                    mu;
    static std::map<std::pair<int, hipStream_t>, float*> ws;
    int                                                  dev = 0;
    FTCF_HIP_CHECK(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    float*&                     p = ws[{dev, s}];
    if (!p) {
        FTCF_HIP_CHECK(hipMalloc(&p, gemm_tiled_workspace_bytes()));
        // only the tickets need zeros (the partial tiles are written before they are read), and on the CALLER's stream: a
        // null-stream memset is not ordered with a launch on a non-blocking stream
        FTCF_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(p) + gemm_tiled_workspace_bytes() - gemm_tiled_ticket_bytes(), 0,
                                      gemm_tiled_ticket_bytes(), s));
    }
    return p;
}

extern "C" int ftcf_fpA_intB_gemm(const void* A, const int8_t* B, const void* scales, const void* bias, ftcf_act act,
                                  void* C, int m, int n, int k, void* stream)
{
    return guarded([&] {
        require_device();
        FTCF_CHECK_ARG(k % 64 == 0 && n % 16 == 0 && m >= 1, "fpA_intB GEMM needs k % 64 == 0, n % 16 == 0");
        gemm_dispatch((const f16*)A, B, (const f16*)scales, (const f16*)bias, (int)act, (f16*)C, m, n, k, true,
                      (hipStream_t)stream, nullptr, 0, 256, nullptr, nullptr, abi_gemm_workspace(m, (hipStream_t)stream));
    });
}
extern "C" int ftcf_fp16_gemm(const void* A, const void* W, const void* bias, ftcf_act act, void* C, int m, int n,
                              int k, void* stream)
{
    return guarded([&] {
        require_device();
        FTCF_CHECK_ARG(k % 64 == 0 && n % 16 == 0 && m >= 1, "fp16 GEMM needs k % 64 == 0, n % 16 == 0");
        gemm_dispatch((const f16*)A, W, nullptr, (const f16*)bias, (int)act, (f16*)C, m, n, k, false,
                      (hipStream_t)stream, nullptr, 0, 256, nullptr, nullptr, abi_gemm_workspace(m, (hipStream_t)stream));
    });
}
static void lm_head_dispatch(const f16* A, const f16* W, float* logits, int m, int n, int k, int ldc, hipStream_t s)
{
    if (m <= 4) {
        launch_lm_head(A, W, logits, m, n, k, ldc, s);
    }
    else {
        launch_gemm_nk_f32out(A, W, logits, m, n, k, ldc, s);
    }
}
extern "C" int ftcf_lm_head(const void* A, const void* W, float* logits, int m, int n, int k, int ldc, void* stream)
{
    return guarded([&] {
        require_device();
        lm_head_dispatch((const f16*)A, (const f16*)W, logits, m, n, k, ldc, (hipStream_t)stream);
    });
}
extern "C" int ftcf_layernorm(const void* x, const void* gamma, const void* beta, void* out, int m, int n, float eps,
                              ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        require_device();
        launch_layernorm(x, gamma, beta, out, m, n, eps, dtype == FTCF_FP16, (hipStream_t)stream);
    });
}
extern "C" int ftcf_add_bias_attn_ffn_residual(void* out, const void* ffn, const void* attn, const void* in,
                                               const void* bias, int m, int n, int tp, int inplace_variant,
                                               ftcf_dtype dtype, void* stream)
{
    return guarded([&] {
        require_device();
        launch_add_bias_attn_ffn_residual(out, ffn, attn, in, bias, m, n, tp, inplace_variant, dtype == FTCF_FP16,
                                          (hipStream_t)stream);
    });
}
extern "C" size_t ftcf_masked_multihead_attention_workspace(int B, int nh, int dh, int s_max)
{
    return mmha_workspace_bytes(B, nh, dh, mmha_pick_nsplit(B, nh, s_max));
}
extern "C" int ftcf_masked_multihead_attention(const void* qkv, const void* qkv_bias, void* k_cache, void* v_cache,
                                               const int* seq_len, const int* pad_count, const uint8_t* masked_tokens,
                                               const uint8_t* finished, int B, int nh, int dh, int rot, int s_max,
                                               int step, void* ctx, void* workspace, size_t workspace_bytes,
                                               void* stream)
{
    return guarded([&] {
        require_device();
        MmhaParams p{};
        p.qkv = (const f16*)qkv;
        p.qkv_bias = (const f16*)qkv_bias;
        p.k_cache = (f16*)k_cache;
        p.v_cache = (f16*)v_cache;
        p.seq_len = seq_len;
        p.pad_count = pad_count;
        p.masked_tokens = masked_tokens;
        p.finished = finished;
        p.d_step = nullptr;
        p.step = step;
        p.B = B;
        p.nh = nh;
        p.dh = dh;
        p.rot = rot;
        p.s_max = s_max;
        p.ctx = (f16*)ctx;
        p.gran = (unsigned long long*)workspace;
        p.layer = 0;
        p.nsplit = mmha_pick_nsplit(B, nh, s_max);
        FTCF_CHECK_ARG(workspace_bytes >= mmha_workspace_bytes(B, nh, dh, p.nsplit), "MMHA workspace too small");
        // the granule tags must not match anything left from an earlier call
        FTCF_HIP_CHECK(hipMemsetAsync(p.gran, 0, mmha_workspace_bytes(B, nh, dh, p.nsplit), (hipStream_t)stream));
        launch_mmha(p, (hipStream_t)stream);
    });
}
extern "C" int ftcf_context_attention(const void* qkv, const void* qkv_bias, const int* input_lengths, void* k_cache,
                                      void* v_cache, int B, int S, int nh, int dh, int rot, int s_max, void* ctx,
                                      void* stream)
{
    return guarded([&] {
        require_device();
        launch_context_attention((const f16*)qkv, (const f16*)qkv_bias, input_lengths, (f16*)k_cache, (f16*)v_cache, B,
                                 S, nh, dh, rot, s_max, (f16*)ctx, (hipStream_t)stream);
    });
}

// ---------------------------------------------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------------------------------------------
namespace {
// (DenseWeight / LayerWeights and the host-side layer units DecoderSelfAttentionLayer, GptContextAttentionLayer, FfnLayer,
// DynamicDecodeLayer: layers.hip.h)

struct DeviceBuffer {
    void*  ptr = nullptr;
    size_t cap = 0;
    void   reserve(size_t bytes)
    {
        if (bytes > cap) {
            if (ptr) {
                FTCF_HIP_CHECK(hipFree(ptr));
                ptr = nullptr;
                cap = 0;
            }
            FTCF_HIP_CHECK(hipMalloc(&ptr, bytes));
            cap = bytes;
        }
    }
    ~DeviceBuffer()
    {
        if (ptr) {
            (void)hipFree(ptr);
        }
    }
};

// carve helper over one arena
struct Carver {
    char*  base;
    size_t off = 0;
    explicit Carver(void* b): base((char*)b) {}
    template<typename T>
    T* take(size_t n)
    {
        off      = (off + 255) & ~(size_t)255;
        T* p     = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += n * sizeof(T);
        return p;
    }
};

enum { KIND_LN_GEMV = 0, KIND_SPLITK = 1, KIND_LM_HEAD = 2, KIND_FUSED = 3, KIND_PERSIST = 4, KIND_SMALLM = 5, KIND_COUNT = 6 };

}  // namespace

__global__ void k_transpose_gathered_logits(float* out, const float* in, int tp, int B, int vl);  // (defined below)

struct ftcf_gptneox {
    ftcf_gptneox_config       cfg{};
    int                       H = 0, nhl = 0, hl = 0, il = 0, L = 0, V = 0, vl = 0, dh = 0;
    bool                      int8 = false;
    bool                      fp32 = false;  // FTGptNeoX<float> (GptNeoXOp.cc:56-70): fp32 weights, activations and K/V; general path only
    // `stream` is the engine's own work stream (capturable, unlike the legacy null stream torch usually hands over);
    // it is ordered after `user_stream` at begin() and drained before forward()/finish() return
    hipStream_t               stream = nullptr, user_stream = nullptr;
    // second stream of the batched decode layer: [QKV -> MMHA -> out-proj] on `stream`, [FFN1 -> FFN2] here (fork / join by
    // events; under capture the branch becomes a parallel branch of the token's hipGraph)
    hipStream_t               side = nullptr;
    hipEvent_t                ev_fork = nullptr, ev_join = nullptr;
    // batched decode GEMMs: 1 = the attention branch and the FFN branch on two streams, 0 = one stream, the independent GEMMs
    // paired per launch.  Default: two streams at tensor_para_size 1 (the launches are bandwidth bound and fill each other's ramps:
    // 13B int8 bs = 16, 4.31 vs 4.66 ms per step), pairs on a tensor-parallel shard (launch-latency bound: TP 8 shard 1.79 vs 2.36
    // ms, TP 4 2.16 vs 2.62, TP 2 3.13 vs 3.28; `bench.py --fake-tp`).  FTCF_DECODE_BRANCHES overrides.
    int                       decode_branches = 1;
    hipEvent_t                ev_user = nullptr;
    hipEvent_t                tok_ev[2] = {nullptr, nullptr};  // per-token events of the pipelined token loop
    bool                      tp_graph = false;
    int                       k1_wpg = 2;  // waves per column group of the QKV launch (0: legacy 4-groups-per-block form)
    std::vector<LayerWeights> layers;
    const f16 *               wte = nullptr, *final_g = nullptr, *final_b = nullptr, *lm_head = nullptr;
    std::vector<void*>        owned;  // tiled fp16 copies (int8_mode == 0)
    void*                     bounce = nullptr;  // FTCF_FP16_RETILE_IN_PLACE: staging of one matrix during create()
    size_t                    bounce_bytes = 0;

    DeviceBuffer arena;
    // decode / state views (valid after plan())
    f16 *x = nullptr, *nrm = nullptr, *nrm2 = nullptr, *qkv = nullptr, *ctx = nullptr, *att = nullptr, *mid = nullptr, *ffn = nullptr;
    f16 *k_cache = nullptr, *v_cache = nullptr;
    f16 *px = nullptr, *pnrm = nullptr, *pnrm2 = nullptr, *pqkv = nullptr, *pctx = nullptr, *patt = nullptr, *pmid = nullptr,
        *pffn = nullptr;
    float *      logits = nullptr, *gather = nullptr, *mmha_ws = nullptr, *rot_table = nullptr;
    unsigned long long* chunk_ws = nullptr;
    int          k3_q = 1;  // chunks per column group of the K3 launch (0: legacy one-workgroup-per-group form)
    void*        samp_ws = nullptr;
    DecodeState* state = nullptr;
    uint8_t *    finished = nullptr, *masked = nullptr;
    int *        seq_len = nullptr, *pad_count = nullptr, *step_ids = nullptr, *d_top_k = nullptr,
        *d_min_length = nullptr;
    float *   cum = nullptr, *d_p_topk = nullptr, *d_p_topp = nullptr, *d_temp = nullptr, *d_rep = nullptr;
    uint64_t *draws = nullptr, *d_seed = nullptr;
    float*    smallm_ws = nullptr;  // split-K partials + tickets of the batched decode GEMM (5..16 rows)
    float*    tiled_ws  = nullptr;  // split-K partial tiles + tickets of the tiled GEMM at 17..320 rows (short prompt phases)
    size_t    smallm_partial = 0;
    unsigned  smallm_seq = 0;       // launch counter: part of the granule tag of its in-launch reduction
    // beam search (beam_width K > 1; rows = batch * K everywhere above)
    int *  tiled_ids = nullptr, *tiled_len = nullptr, *parent_ids = nullptr, *cache_indir = nullptr;
    void*  beam_ws = nullptr;
    float *d_div = nullptr, *d_lenpen = nullptr;
    int*      h_flags = nullptr;  // pinned
    int       nsplit = 1;
    // persistent decode layers (kernels_persist.hip): on whenever the shape is eligible (FTCF_PERSIST=0: per-stage launches)
    int                 persist = 1, persist_tp = 1, persist_nb = 0, persist_cs1 = 12, persist_cs3 = 10;
    int                 num_cu = 0;
    PersistPlan         pplan{};
    PersistLayer*       d_players = nullptr;  // device [L]
    char*               ps_tab = nullptr;     // the plan's run / tile tables, built once per request by a launch over no layers
    bool                ps_tab_ready = false;
    bool                ps_lm_fused = false;  // the LM head runs as the tail of the persistent launch (one GPU, H % 512 == 0)
    unsigned long long *ps_gq = nullptr, *ps_gm = nullptr, *ps_gc = nullptr, *ps_gx = nullptr, *ps_gp = nullptr,
                       *ps_ga = nullptr;
    size_t              ps_slab_n = 0;
    int*                ps_err = nullptr;
    int*                tp_scratch = nullptr;  // device int for the barrier all-reduce of the tensor-parallel windows
    long long*          ps_ts = nullptr;  // FTCF_PERSIST_TS=<file>: in-kernel stamps of the last token
    std::string         ps_ts_file;

    // profiling
    bool               profiling = false;
    ftcf_forward_stats stats{};
    double             kind_ms[KIND_COUNT]{}, kind_bytes[KIND_COUNT]{};
    long               kind_n[KIND_COUNT]{};
    std::vector<std::tuple<hipEvent_t, hipEvent_t, int, double>> pending;
    std::vector<hipEvent_t>                                       event_pool;

    ~ftcf_gptneox()
    {
        for (void* p : owned) {
            (void)hipFree(p);
        }
        if (ses.graph_exec) {
            (void)hipGraphExecDestroy(ses.graph_exec);
        }
        if (ses.graph_exec_n) {
            (void)hipGraphExecDestroy(ses.graph_exec_n);
        }
        if (tp_scratch) {
            (void)hipFree(tp_scratch);
        }
        if (stream) {
            (void)hipStreamDestroy(stream);
            for (int c = 0; c < 2; c++) {
                if (ov_done[c]) {
                    (void)hipEventDestroy(ov_done[c]);
                    (void)hipEventDestroy(ov_red[c]);
                }
            }
            if (side) {
                (void)hipStreamDestroy(side);
                (void)hipEventDestroy(ev_fork);
                (void)hipEventDestroy(ev_join);
            }
        }
        if (ev_user) {
            (void)hipEventDestroy(ev_user);
        }
        for (hipEvent_t e : tok_ev) {
            if (e) {
                (void)hipEventDestroy(e);
            }
        }
        if (h_flags) {
            (void)hipHostFree(h_flags);
        }
        for (auto e : event_pool) {
            (void)hipEventDestroy(e);
        }
    }

    hipEvent_t get_event()
    {
        if (!event_pool.empty()) {
            hipEvent_t e = event_pool.back();
            event_pool.pop_back();
            return e;
        }
        hipEvent_t e;
        FTCF_HIP_CHECK(hipEventCreate(&e));
        return e;
    }

    template<typename F>
    void timed(int kind, double bytes, F&& f, hipStream_t on = nullptr)
    {
        if (!profiling) {
            f();
            return;
        }
        hipEvent_t a = get_event(), b = get_event();
        FTCF_HIP_CHECK(hipEventRecord(a, on ? on : stream));
        f();
        FTCF_HIP_CHECK(hipEventRecord(b, on ? on : stream));
        pending.emplace_back(a, b, kind, bytes);
    }
    void drain_events()
    {
        for (auto& t : pending) {
            float ms = 0.f;
            FTCF_HIP_CHECK(hipEventSynchronize(std::get<1>(t)));
            FTCF_HIP_CHECK(hipEventElapsedTime(&ms, std::get<0>(t), std::get<1>(t)));
            kind_ms[std::get<2>(t)] += ms;
            kind_bytes[std::get<2>(t)] += std::get<3>(t);
            kind_n[std::get<2>(t)] += 1;
            event_pool.push_back(std::get<0>(t));
            event_pool.push_back(std::get<1>(t));
        }
        pending.clear();
    }

    // ---- arena planning: everything a request of shape (B, S, total) needs, carved once ----
    // B = rows of the request (batch * beam_width)
    void plan(int B, int S, int total, int K)
    {
        const int s_max = total;
        nsplit          = mmha_pick_nsplit(B, nhl, s_max);
        for (int pass = 0; pass < 2; pass++) {
            Carver c(pass == 0 ? nullptr : arena.ptr);
            const size_t es    = fp32 ? 2 : 1;  // fp32 engine: the same views hold floats
            const size_t cache = (size_t)L * B * nhl * s_max * dh * es;
            k_cache            = c.take<f16>(cache);
            v_cache            = c.take<f16>(cache);
            x                  = c.take<f16>((size_t)B * H * es);
            nrm                = c.take<f16>((size_t)B * H * es);
            nrm2               = c.take<f16>((size_t)B * H * es);
            qkv                = c.take<f16>((size_t)B * 3 * hl * es);
            ctx                = c.take<f16>((size_t)B * hl * es);
            att                = c.take<f16>((size_t)B * H * es);
            mid                = c.take<f16>((size_t)B * il * es);
            ffn                = c.take<f16>((size_t)B * H * es);
            logits             = c.take<float>((size_t)B * V);
            gather             = c.take<float>((size_t)B * V);
            mmha_ws            = c.take<float>(mmha_workspace_bytes(B, nhl, dh, nsplit) / 4);
            samp_ws            = c.take<char>(sampling_workspace_bytes(B, V));
            rot_table          = c.take<float>((size_t)B * 256);
            chunk_ws           = c.take<unsigned long long>(chunk_workspace_bytes(H, std::min(B, 4), 8) / 8);
            pplan = PersistPlan{};
            // With tensor parallelism the per-layer all-reduce happens INSIDE the persistent launch, through the ranks'
            // exchange windows (persist_device.hip.h ps_tp_exchange); where the windows are not available (peer mapping or
            // hand-shake failed, FTCF_TP_PERSIST=0) the per-stage launches + RCCL all-reduce stay in charge.
            const int  tpn      = cfg.tensor_para_size;
            const bool tp_local = tpn > 1 && cfg.comm && cfg.comm->local;
            // (L <= 255: the hand-off tags carry the layer in their low byte; tpn <= 8: the exchange-window table of the kernel)
            if (persist && !fp32 && K == 1 && B <= 2 && cfg.use_gptj_residual && L <= 255 && tpn <= PERSIST_MAX_TP
                && (tpn == 1 || (persist_tp && cfg.comm && cfg.comm->win_ok))) {
                // (a local group shares ONE device: every rank gets 1 / world of its compute units)
                const int nb = tp_local ? std::max(1, (persist_nb > 0 ? persist_nb : num_cu) / tpn) : persist_nb;
                pplan = persist_plan(B, H, hl, il, nhl, dh, s_max, int8, num_cu, nb, persist_cs1, persist_cs3, tpn == 1);
#ifdef PS_EXPERIMENTS
                // (experiment builds only, `make EXPERIMENTS=1`; FTCF_PERSIST_A4=1 selects it)  Second form of the kernel
                // (persist4_device.hip.h: the attention branch on the control waves under the FFN streams): built, parity green,
                // measured 1-2.5 % SLOWER than the first form at TP = 1 (profiles/r04_notes.md) -- six streaming waves carry a
                // lower rate than eight, and what the removed hand-off window gains is lost there
                static const int a4_max_tp = getenv("FTCF_PERSIST_A4_MAX_TP") ? atoi(getenv("FTCF_PERSIST_A4_MAX_TP")) : 2;
                static const int a4_cs3 = getenv("FTCF_PERSIST4_CS3") ? atoi(getenv("FTCF_PERSIST4_CS3")) : 12;
                if (pplan.ok && B == 1 && tpn <= a4_max_tp) {
                    const PersistPlan p4 = persist_plan4(
                        persist_plan(B, H, hl, il, nhl, dh, s_max, int8, num_cu, nb, persist_cs1, a4_cs3, false), B, H, hl, il, nhl,
                        dh, s_max, int8);
                    if (p4.ok && p4.a4) {
                        pplan = p4;
                    }
                }
#endif
                const bool resident = !pplan.ok ? false
                                      : tp_local ? persist_group_resident(pplan, int8, B, dh, num_cu, tpn)
                                                 : persist_resident(pplan, int8, B, dh, num_cu, tpn);
                if (!resident) {
                    pplan = PersistPlan{};  // not every workgroup would be resident: the hand-offs could never complete
                }
            }
            ps_lm_fused = false;
            if (pplan.ok) {
                ps_slab_n   = (size_t)B * 3 * hl / 2 + (size_t)B * il / 2 + (size_t)B * hl / 2 + (size_t)B * H / 2
                            + (size_t)(H / 16) * (pplan.PA + pplan.PB) * B * 16 + (size_t)B * nhl * pplan.nsplit * (dh + 2);
                ps_gq       = c.take<unsigned long long>(ps_slab_n + 8);
                ps_gm       = ps_gq ? ps_gq + (size_t)B * 3 * hl / 2 : nullptr;
                ps_gc       = ps_gq ? ps_gm + (size_t)B * il / 2 : nullptr;
                ps_gx       = ps_gq ? ps_gc + (size_t)B * hl / 2 : nullptr;
                ps_gp       = ps_gq ? ps_gx + (size_t)B * H / 2 : nullptr;
                ps_ga       = ps_gq ? ps_gp + (size_t)(H / 16) * (pplan.PA + pplan.PB) * B * 16 : nullptr;
                ps_err      = ps_gq ? reinterpret_cast<int*>(ps_gq + ps_slab_n) : nullptr;
                d_players   = c.take<PersistLayer>(L);
                ps_tab      = c.take<char>(persist_table_bytes(pplan) * pplan.NB);
                ps_tab_ready = false;
                static const int lm_env = persist_lm_tail_built() && getenv("FTCF_PERSIST_LM") ? atoi(getenv("FTCF_PERSIST_LM")) : 0;
                ps_lm_fused = lm_env != 0 && tpn == 1 && H % 512 == 0;
                ps_ts       = ps_ts_file.empty() ? nullptr : c.take<long long>((size_t)pplan.NB * L * 128);
            }
            // (the four GEMMs of a layer may be in flight together: one region each)
            // (17..SMALLM_MAX_ROWS rows run the same kernel in chunks of 16 rows: sized for one chunk)
            // (a prompt phase of up to SMALLM_MAX_ROWS tokens in all is HBM bound like a decode step: it takes the same kernel)
            const int  bc         = 16;
            const long prefill_m  = S > 1 ? (long)(B / K) * S : 0;
            const bool decode_ws  = B > STAGE_MAX_ROWS && B <= SMALLM_MAX_ROWS;
            const bool prefill_ws = prefill_m > 4 && prefill_m <= SMALLM_MAX_ROWS;
            smallm_partial = gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8) + gemm_smallm_workspace_bytes(bc, il, H, int8)
                             + gemm_smallm_workspace_bytes(bc, H, hl, int8) + gemm_smallm_workspace_bytes(bc, H, il, int8);
            smallm_ws = (!fp32 && (decode_ws || prefill_ws)) ? c.take<float>((smallm_partial + gemm_smallm_ticket_bytes()) / 4) : nullptr;
            const bool tiled_rows = prefill_m > 16 || (B > 16 && B <= 320);
            tiled_ws              = (!fp32 && tiled_rows) ? c.take<float>(gemm_tiled_workspace_bytes() / 4) : nullptr;
            state              = c.take<DecodeState>(1);
            finished           = c.take<uint8_t>(B);
            masked             = c.take<uint8_t>((size_t)B * s_max);
            seq_len            = c.take<int>(B);
            pad_count          = c.take<int>(B);
            step_ids           = c.take<int>((size_t)total * B);
            d_top_k            = c.take<int>(B);
            d_min_length       = c.take<int>(B);
            cum                = c.take<float>(B);
            d_p_topk           = c.take<float>(B);
            d_p_topp           = c.take<float>(B);
            d_temp             = c.take<float>(B);
            d_rep              = c.take<float>(B);
            draws              = c.take<uint64_t>(B);
            d_seed             = c.take<uint64_t>(B);
            if (K > 1) {
                tiled_ids   = c.take<int>((size_t)B * S);
                tiled_len   = c.take<int>(B);
                parent_ids  = c.take<int>((size_t)total * B);
                cache_indir = c.take<int>((size_t)2 * B * s_max);
                beam_ws     = c.take<char>(beam_workspace_bytes(B / K, K));
                d_div       = c.take<float>(B);
                d_lenpen    = c.take<float>(B);
            }
            if (S > 1) {
                const size_t M = (size_t)(B / K) * S;  // beam search prefills one row per request
                px             = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pnrm           = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pnrm2          = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pqkv           = c.take<f16>(M * 3 * hl * (fp32 ? 2 : 1));
                pctx           = c.take<f16>(M * hl * (fp32 ? 2 : 1));
                patt           = c.take<f16>(M * H * (fp32 ? 2 : 1));
                pmid           = c.take<f16>(M * il * (fp32 ? 2 : 1));
                pffn           = c.take<f16>(M * H * (fp32 ? 2 : 1));
            }
            if (pass == 0) {
                arena.reserve(c.off + 4096);
            }
        }
    }

    // ---- FfnLayer / attention projections over M rows (general path) ----
    // ---- host-side layer units (layers.hip.h), bound to this engine's GEMM dispatch -----------------------------------
    DecoderSelfAttentionLayer self_attention_layer;
    GptContextAttentionLayer  context_attention_layer;
    FfnLayer                  ffn_layer;
    DynamicDecodeLayer        dynamic_decode_layer;
    bool                      layers_bound = false;
    void bind_layers()
    {
        if (layers_bound) {
            return;
        }
        // (row-count dispatch of gemm(); `slot` is unused here: the burst kernel's four workspace regions are only in flight
        // together on the two-stream branch form, which names its regions itself)
        GemmFn g = [this](const f16* A, const DenseWeight& w, const f16* bias, int act, f16* C, int m, int n, int k, hipStream_t s,
                          int) { gemm(A, w, bias, act, C, m, n, k, s); };
        self_attention_layer    = DecoderSelfAttentionLayer{g, H, hl};
        context_attention_layer = GptContextAttentionLayer{g, H, hl, nhl, dh, cfg.rotary_embedding_dim};
        ffn_layer               = FfnLayer{g, H, il};
        layers_bound            = true;
    }

    void gemm(const f16* A, const DenseWeight& w, const f16* bias, int act, f16* C, int m, int n, int k, hipStream_t on = nullptr)
    {
        hipStream_t stream = on ? on : this->stream;  // (the layer units pass the stream they were given)
        // 5..SMALLM_MAX_ROWS rows (batched decode steps off the branch form, short prompt phases): the burst kernel, 16 rows
        // per launch.  13B int8 prefill, ms: 17 tokens 12.5 -> 6.6, 33..48: 16.3 -> 9.6 (above that the tiled GEMM is as fast).
        if (smallm_ws && m > 4 && m <= SMALLM_MAX_ROWS && gemm_smallm_workspace_bytes(16, n, k, int8) <= smallm_partial) {
            for (int r0 = 0; r0 < m; r0 += 16) {
                launch_gemm_smallm(A + (size_t)r0 * k, w.kernel, w.scale, bias, act, C + (size_t)r0 * n, smallm_ws, smallm_partial,
                                   std::min(16, m - r0), n, k, int8, num_cu, stream, &state->step, &smallm_seq);
            }
            return;
        }
        if (m > 16) {
            launch_gemm_tiled(A, w.kernel, w.scale, bias, act, C, m, n, k, int8, stream, tiled_ws);
            return;
        }
        gemm_dispatch(A, w.kernel, w.scale, bias, act, C, m, n, k, int8, stream, nullptr, smallm_partial, num_cu, &state->step,
                      &smallm_seq);
    }

    void allreduce(f16* buf, size_t count, hipStream_t on = nullptr)
    {
        if (cfg.tensor_para_size > 1) {
            Range r("ftcf.allreduce");
            hipStream_t st = on ? on : stream;
            FTCF_CHECK_ARG(cfg.comm && (cfg.comm->comm || cfg.comm->local || cfg.comm->hx), "tensor_para_size > 1 needs a communicator");
            if (window_allreduce(cfg.comm, buf, count, st)) {
                stats.window_allreduces++;
                return;  // through the peer-mapped windows: no RCCL call (prompt-phase messages; k_window_allreduce)
            }
            if (cfg.comm->local) {
                local_allreduce(cfg.comm, buf, count, true, st);
                return;
            }
            if (cfg.comm->hx) {
                hx_allreduce(cfg.comm, buf, count, true, st);
                return;
            }
            FTCF_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclFloat16, ncclSum, cfg.comm->comm, st));
        }
    }

    // vocabulary-split LM head (GptNeoX.cc:888-925): rank r has written its [B, V/TP] slice of `gath` ([TP][B][V/TP] fp32);
    // all-gather it and transpose into out [B, V]
    void allgather_logits(float* gath, float* out, int B, hipStream_t st)
    {
        const int tp = cfg.tensor_para_size;
        float*    mine = gath + (size_t)cfg.tensor_para_rank * B * vl;
        if (cfg.comm->local) {
            local_allgather(cfg.comm, gath, (size_t)B * vl, false, st);
        }
        else if (cfg.comm->hx) {
            hx_allgather_device(cfg.comm, gath, (size_t)B * vl, 4, st);
        }
        else {
            FTCF_NCCL_CHECK(ncclAllGather(mine, gath, (size_t)B * vl, ncclFloat32, cfg.comm->comm, st));
        }
        hipLaunchKernelGGL(k_transpose_gathered_logits, dim3(256), dim3(256), 0, st, out, gath, tp, B, vl);
    }

    // ---------------------------------------------------------------------------------------------------------------
    // fp32 instantiation (kernels_fp32.hip): the arena views (x, nrm, qkv, ..., the caches, the prefill buffers) hold floats
    // ---------------------------------------------------------------------------------------------------------------
    static float*       F(f16* p) { return reinterpret_cast<float*>(p); }
    static const float* F(const f16* p) { return reinterpret_cast<const float*>(p); }
    static const float* F(const void* p) { return reinterpret_cast<const float*>(p); }
    void allreduce32(float* buf, size_t count)
    {
        if (cfg.tensor_para_size > 1) {
            Range r("ftcf.allreduce");
            FTCF_CHECK_ARG(cfg.comm && (cfg.comm->comm || cfg.comm->local || cfg.comm->hx), "tensor_para_size > 1 needs a communicator");
            if (cfg.comm->local) {
                local_allreduce(cfg.comm, buf, count, false, stream);
                return;
            }
            if (cfg.comm->hx) {
                hx_allreduce(cfg.comm, buf, count, false, stream);
                return;
            }
            FTCF_NCCL_CHECK(ncclAllReduce(buf, buf, count, ncclFloat32, ncclSum, cfg.comm->comm, stream));
        }
    }
    // one layer's GEMMs / residual on M rows, shared by the context phase and the decode step
    // (GptNeoXContextDecoder.cc:283-507, GptNeoXDecoder.cc:245-384 with T = float)
    template<typename Attn>
    void layer32(const LayerWeights& w, float* X, float* N1, float* Q, float* C, float* A, float* MID, float* FF, int M,
                 bool first_or_last_inplace_variant, Attn&& attention)
    {
        launch_layernorm(X, w.ln1_g, w.ln1_b, N1, M, H, 1e-5f, false, stream);
        launch32_gemm(N1, F(w.qkv.kernel), nullptr, 0, Q, M, 3 * hl, H, stream);
        attention();
        launch32_gemm(C, F(w.attn_out.kernel), nullptr, 0, A, M, H, hl, stream);
        if (!cfg.use_gptj_residual) {
            // sequential residual (GptNeoXDecoder.cc:313-331,362-367): h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
            allreduce32(A, (size_t)M * H);
            launch32_add_bias_residual(A, X, A, F(w.attn_out.bias), M, H, stream);
            launch_layernorm(A, w.ln2_g, w.ln2_b, N1, M, H, 1e-5f, false, stream);
            launch32_gemm(N1, F(w.ffn1.kernel), F(w.ffn1.bias), 1, MID, M, il, H, stream);
            launch32_gemm(MID, F(w.ffn2.kernel), nullptr, 0, FF, M, H, il, stream);
            allreduce32(FF, (size_t)M * H);
            launch32_add_bias_residual(X, FF, A, F(w.ffn2.bias), M, H, stream);
            return;
        }
        launch_layernorm(X, w.ln2_g, w.ln2_b, N1, M, H, 1e-5f, false, stream);
        launch32_gemm(N1, F(w.ffn1.kernel), F(w.ffn1.bias), 1, MID, M, il, H, stream);
        launch32_gemm(MID, F(w.ffn2.kernel), nullptr, 0, FF, M, H, il, stream);
        launch_add_bias_attn_ffn_residual(X, FF, A, X, w.ffn2.bias, M, H, cfg.tensor_para_size,
                                          first_or_last_inplace_variant ? 0 : 1, false, stream);
        allreduce32(X, (size_t)M * H);
    }
    void context_decoder32(int B, int S, const int* input_lengths, int s_max, int tile)
    {
        Range        r("ftcf.GptNeoXContextDecoder");
        const int    M       = B * S;
        const size_t cache_l = (size_t)B * tile * nhl * s_max * dh;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // (the context decoder's layer_input == layer_output for every layer with padding removal: the fp32-sum variant)
            layer32(w, F(px), F(pnrm), F(pqkv), F(pctx), F(patt), F(pmid), F(pffn), M, false, [&] {
                launch32_context_attention(F(pqkv), F(w.qkv.bias), input_lengths, F(k_cache) + l * cache_l,
                                           F(v_cache) + l * cache_l, B, S, nhl, dh, cfg.rotary_embedding_dim, s_max, F(pctx),
                                           stream, tile);
            });
        }
    }
    void decoder32(int B, int s_max)
    {
        Range        r("ftcf.GptNeoXDecoder");
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        stats.decode_path    = 2;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // layer_input/output alias for 0 < l < L-1 in the reference (GptNeoXDecoder.cc:249-250) -> residual form
            const bool outer = !(l > 0 && l < L - 1);
            layer32(w, F(x), F(nrm), F(qkv), F(ctx), F(att), F(mid), F(ffn), B, outer, [&] {
                Mmha32Params mp{};
                mp.qkv = F(qkv);
                mp.qkv_bias = F(w.qkv.bias);
                mp.k_cache = F(k_cache) + l * cache_l;
                mp.v_cache = F(v_cache) + l * cache_l;
                mp.seq_len = seq_len;
                mp.pad_count = pad_count;
                mp.masked_tokens = masked;
                mp.finished = finished;
                mp.d_step = &state->step;
                mp.B = B;
                mp.nh = nhl;
                mp.dh = dh;
                mp.rot = cfg.rotary_embedding_dim;
                mp.s_max = s_max;
                mp.ctx = F(ctx);
                if (ses.K > 1) {
                    mp.cache_indir   = cache_indir;
                    mp.beam_width    = ses.K;
                    mp.max_input_len = ses.S;
                    mp.indir_plane   = (size_t)B * s_max;
                }
                launch32_mmha(mp, stream);
            });
        }
    }

    // GptNeoXContextDecoder::forward (GptNeoXContextDecoder.cc:283-507), parallel residual only
    // B prompt rows; their K/V go to cache rows b * tile of a cache with B * tile rows (beam search: tile = beam_width)
    // Prompt phase under tensor parallelism with the per-layer all-reduce OVERLAPPED (GptNeoXContextDecoder.cc:462-465 calls
    // ftNcclAllReduceSum on the compute stream, nothing runs under it).  The prompt is cut into two micro-batches -- whole
    // sequences when there are several (their attention is independent), the first and the second half of the tokens of a
    // single sequence (every GEMM / LayerNorm / residual is row wise, and the second half's attention reads the first half's
    // K/V from the cache, where the first half's attention call has put them).  A layer runs micro-batch 0, then 1, on the
    // engine stream; each micro-batch's all-reduce goes to the side stream behind an event, and the NEXT layer's work on that
    // micro-batch waits for it: the reduction of one half runs under the GEMMs of the other.  Same arithmetic per row as
    // context_decoder (the all-reduce sums the same values): results are bit-identical to the un-overlapped path.
    // Chunked prompt phase of ONE sequence (the continuous-batching front end, section 4e of DESIGN.md): the prompt's tokens pass
    // through all layers `prefill_chunk` at a time -- every GEMM / LayerNorm / residual is row wise, a chunk's attention reads
    // the earlier chunks' K/V from the cache -- and `prefill_hook` runs between two chunks (the batcher enqueues one decode step
    // of its running slots there: an admission delays them by one chunk, not by the whole prompt).  The row-wise arithmetic is
    // that of context_decoder; the split-K form of the GEMMs depends on the row count, so results agree to fp16 rounding of
    // the GEMM outputs, not bit for bit.
    int                   prefill_chunk = 0;
    std::function<void()> prefill_hook;
    bool context_decoder_chunked(int S, const int* input_lengths, int s_max)
    {
        if (!prefill_hook || prefill_chunk <= 0 || S <= prefill_chunk || fp32 || cfg.tensor_para_size != 1 || !cfg.use_gptj_residual
            || !residual_dual_ln_supported(H)) {
            return false;
        }
        Range r("ftcf.GptNeoXContextDecoder.chunked");
        bind_layers();
        const size_t cache_l = (size_t)nhl * s_max * dh;
        for (int s0 = 0; s0 < S; s0 += prefill_chunk) {
            const int s1 = std::min(S, s0 + prefill_chunk), m = s1 - s0;
            f16*      X  = px + (size_t)s0 * H;
            for (int l = 0; l < L; l++) {
                const LayerWeights& w = layers[l];
                launch_residual_dual_ln(X, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm + (size_t)s0 * H,
                                        pnrm2 + (size_t)s0 * H, m, H, 1e-5f, stream);
                context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l, 1,
                                                S, s_max, 1, stream, s0, s1);
                ffn_layer.forward(pnrm2 + (size_t)s0 * H, pmid + (size_t)s0 * il, pffn + (size_t)s0 * H, w, m, stream);
                launch_add_bias_attn_ffn_residual(X, pffn + (size_t)s0 * H, patt + (size_t)s0 * H, X, w.ffn2.bias, m, H, 1, 1, true, stream);
            }
            if (s1 < S) {
                prefill_hook();
            }
        }
        return true;
    }

    hipEvent_t ov_done[2] = {nullptr, nullptr}, ov_red[2] = {nullptr, nullptr};
    int        ov_trial = 0;             // auto mode: 0 the next eligible prompt phase runs plain, 1 overlapped, 2 decided
    float      ov_ms[2] = {0.f, 0.f};    // ... what the two trials took (the slowest rank's time)
    bool       ov_ran = false, ov_eligible = false;  // this request: ran overlapped / counts as a trial
    bool context_decoder_overlapped(int B, int S, const int* input_lengths, int s_max)
    {
        // OPT-IN (FTCF_PREFILL_OVERLAP=1; read per request, the tests flip it): on the one GPU the builder has, one rank's shard of
        // the 1024-token prompt phase takes 19.1 -> 27.6 ms (TP 2) / 10.3 -> 18.3 ms (TP 8) in two micro-batches -- GEMMs of 512
        // rows fill the chip worse than GEMMs of 1024 (profiles/r03_faketp_prefill.txt) -- and what the overlap hides (40 all-reduces
        // of 10 MiB over xGMI) cannot be measured without the peers.  Whoever has the node should measure both.
        // Round 4: DECIDED FROM DATA on the node it runs on.  FTCF_PREFILL_OVERLAP = 0 / 1 forces it; unset or "auto" (the
        // default for ranks joined by RCCL, i.e. a real multi-GPU job): the first eligible prompt phase of at least 512 tokens
        // runs plain and is timed, the second one overlapped, every rank learns the slower rank's times (comm_max) and the
        // engine keeps the faster form; ftcf_forward_stats says what ran and what the two trials took.
        const char* ev  = getenv("FTCF_PREFILL_OVERLAP");
        const bool  aut = (!ev || !strcmp(ev, "auto")) && cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->comm && !cfg.comm->local
                         && !cfg.comm->hx && cfg.comm->world > 1;
        const int   env = (ev && strcmp(ev, "auto")) ? atoi(ev) : 0;
        static const bool valu_form = getenv("FTCF_CTX_ATTN_VALU") != nullptr;
        ov_ran      = false;
        ov_eligible = false;
        if ((!env && !aut) || cfg.tensor_para_size == 1 || !cfg.use_gptj_residual || !residual_dual_ln_supported(H) || !side || valu_form) {
            return false;
        }
        if (aut) {
            ov_eligible = (long)B * S >= 512 && (B >= 2 || (S / 2) / 64 * 64 >= 64);
            const bool want = ov_eligible && (ov_trial == 1 || (ov_trial == 2 && ov_ms[1] < ov_ms[0]));
            if (!want) {
                return false;
            }
        }
        // micro-batches: rows [r0[c], r1[c]) of the [B * S] row space; sequences [b0, b1) x tokens [s0, s1)
        int b0[2] = {0, 0}, b1[2] = {B, B}, s0[2] = {0, 0}, s1[2] = {S, S};
        if (B >= 2) {
            b1[0] = b0[1] = B / 2;
        }
        else {
            const int cut = (S / 2) / 64 * 64;
            if (cut < 64) {
                return false;  // too short to be worth two micro-batches
            }
            s1[0] = s0[1] = cut;
        }
        Range r("ftcf.GptNeoXContextDecoder.overlapped");
        ov_ran = true;
        bind_layers();
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        for (int c = 0; c < 2; c++) {
            if (!ov_done[c]) {
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ov_done[c], hipEventDisableTiming));
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ov_red[c], hipEventDisableTiming));
            }
        }
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            for (int c = 0; c < 2; c++) {
                const size_t row0 = (size_t)b0[c] * S + s0[c];
                const int    m    = (B >= 2) ? (b1[c] - b0[c]) * S : s1[c] - s0[c];
                if (l > 0) {
                    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ov_red[c], 0));  // this micro-batch's x has been reduced
                }
                f16* X = px + row0 * H;
                launch_residual_dual_ln(X, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm + row0 * H,
                                        pnrm2 + row0 * H, m, H, 1e-5f, stream);
                if (B >= 2) {  // whole sequences [b0, b1): the attention layer on their rows of every buffer
                    const size_t cb = (size_t)b0[c] * nhl * s_max * dh;
                    context_attention_layer.forward(pnrm + row0 * H, pqkv + row0 * 3 * hl, pctx + row0 * hl, patt + row0 * H, w,
                                                    input_lengths + b0[c], k_cache + l * cache_l + cb, v_cache + l * cache_l + cb,
                                                    b1[c] - b0[c], S, s_max, 1, stream);
                }
                else {  // tokens [s0, s1) of the one sequence: the earlier tokens' K/V are in the cache
                    context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l,
                                                    1, S, s_max, 1, stream, s0[c], s1[c]);
                }
                ffn_layer.forward(pnrm2 + row0 * H, pmid + row0 * il, pffn + row0 * H, w, m, stream);
                launch_add_bias_attn_ffn_residual(X, pffn + row0 * H, patt + row0 * H, X, w.ffn2.bias, m, H, cfg.tensor_para_size,
                                                  1, true, stream);
                FTCF_HIP_CHECK(hipEventRecord(ov_done[c], stream));
                FTCF_HIP_CHECK(hipStreamWaitEvent(side, ov_done[c], 0));
                allreduce(X, (size_t)m * H, side);
                FTCF_HIP_CHECK(hipEventRecord(ov_red[c], side));
            }
        }
        for (int c = 0; c < 2; c++) {
            FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ov_red[c], 0));
        }
        return true;
    }

    void context_decoder(int B, int S, const int* input_lengths, int s_max, int tile)
    {
        if (tile == 1 && B == 1 && context_decoder_chunked(S, input_lengths, s_max)) {
            return;
        }
        if (tile == 1 && context_decoder_overlapped(B, S, input_lengths, s_max)) {
            return;
        }
        Range r("ftcf.GptNeoXContextDecoder");
        bind_layers();
        const int    M       = B * S;
        const size_t cache_l = (size_t)B * tile * nhl * s_max * dh;
        // parallel-residual layers: both LayerNorms in one pass, fused with the previous layer's residual when no collective
        // sits in between (as in the batched decode path)
        const bool dual = cfg.use_gptj_residual && residual_dual_ln_supported(H);
        const bool tp1  = cfg.tensor_para_size == 1;
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            if (!dual) {
                launch_layernorm(px, w.ln1_g, w.ln1_b, pnrm, M, H, 1e-5f, true, stream);
            }
            else if (l == 0 || !tp1) {
                launch_residual_dual_ln(px, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, pnrm, pnrm2,
                                        M, H, 1e-5f, stream);
            }
            context_attention_layer.forward(pnrm, pqkv, pctx, patt, w, input_lengths, k_cache + l * cache_l, v_cache + l * cache_l, B, S,
                                            s_max, tile, stream);
            if (!cfg.use_gptj_residual) {
                // sequential residual (GptNeoXContextDecoder.cc:401-418,463-470): the TensorParallel layers reduce their
                // own outputs; h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
                allreduce(patt, (size_t)M * H);
                launch_add_bias_residual(patt, px, patt, w.attn_out.bias, M, H, stream);
                launch_layernorm(patt, w.ln2_g, w.ln2_b, pnrm, M, H, 1e-5f, true, stream);
                ffn_layer.forward(pnrm, pmid, pffn, w, M, stream);
                allreduce(pffn, (size_t)M * H);
                launch_add_bias_residual(px, pffn, patt, w.ffn2.bias, M, H, stream);
                continue;
            }
            if (!dual) {
                launch_layernorm(px, w.ln2_g, w.ln2_b, pnrm, M, H, 1e-5f, true, stream);
            }
            ffn_layer.forward(dual ? pnrm2 : pnrm, pmid, pffn, w, M, stream);
            // layer_input == layer_output for every layer with padding removal -> fp32-sum variant (:311-322,:445-461)
            if (dual && tp1) {
                const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                launch_residual_dual_ln(px, pffn, patt, w.ffn2.bias, 1, 1, nx ? nx->ln1_g : nullptr, nx ? nx->ln1_b : nullptr,
                                        nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, pnrm, pnrm2, M, H, 1e-5f, stream);
            }
            else {
                launch_add_bias_attn_ffn_residual(px, pffn, patt, px, w.ffn2.bias, M, H, cfg.tensor_para_size, 1, true,
                                                  stream);
            }
            allreduce(px, (size_t)M * H);
        }
    }

    PersistParams persist_params(int B, int s_max)
    {
        PersistParams pp{};
        pp.layers = d_players;
        pp.L = L;
        pp.x_in = x;
        pp.x_out = x;
        pp.gq = ps_gq;
        pp.gm = ps_gm;
        pp.gc = ps_gc;
        pp.gx = ps_gx;
        pp.gp = ps_gp;
        pp.ga = ps_ga;
        pp.err = ps_err;
        pp.H = H;
        pp.Hl = hl;
        pp.Il = il;
        pp.nh = nhl;
        pp.dh = dh;
        pp.rot = cfg.rotary_embedding_dim;
        pp.s_max = s_max;
        pp.B = B;
        pp.tp = cfg.tensor_para_size;
        pp.tp_rank = cfg.tensor_para_rank;
        for (int r = 0; r < PERSIST_MAX_TP; r++) {
            pp.xw[r] = (cfg.tensor_para_size > 1 && cfg.comm && r < (int)cfg.comm->win.size())
                           ? static_cast<unsigned long long*>(cfg.comm->win[r]) : nullptr;
        }
        if (cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->world == 1 && !cfg.comm->win.empty()) {
            // timing aid (bench.py --fake-tp N: ONE rank of a TP = N job without its peers): the rank plays every peer --
            // "slot [rank] of rank r's window" is made to land in slot [r] of its own -- so that the kernel's exchange
            // completes (the sums are meaningless, the work and the waits of a rank are all there)
            for (int r = 0; r < cfg.tensor_para_size && r < PERSIST_MAX_TP; r++) {
                pp.xw[r] = static_cast<unsigned long long*>(cfg.comm->win[0])
                           + (ptrdiff_t)(r - cfg.tensor_para_rank) * ((ptrdiff_t)B * H / 2);
            }
        }
        pp.plan = pplan;
        pp.d_step = &state->step;
        pp.d_stop = &state->all_finished;
        pp.seq_len = seq_len;
        pp.pad_count = pad_count;
        pp.masked_tokens = masked;
        pp.finished = finished;
        pp.rot_table = rot_table;
        pp.eps = 1e-5f;
        pp.ts = ps_ts;
        pp.tab = ps_tab;
        pp.tab_mode = (ps_tab && ps_tab_ready) ? 2 : 0;
        if (ps_lm_fused) {  // final LayerNorm + LM head as the launch's tail (enqueue_step then skips its own launch)
            pp.lm_w      = lm_head;
            pp.lm_g      = final_g;
            pp.lm_b      = final_b;
            pp.lm_logits = logits;
            pp.lm_rows   = V;
            pp.lm_ldc    = V;
        }
        return pp;
    }

    // GptNeoXDecoder::forward (GptNeoXDecoder.cc:245-384)
    void decoder(int B, int s_max)
    {
        Range r("ftcf.GptNeoXDecoder");
        bind_layers();
        const double wbytes  = int8 ? 1.0 : 2.0;
        // (beam search reads K/V through the cache indirection, sequential-residual layers have their own order: general path)
        const bool staged = B <= STAGE_MAX_ROWS && ses.K == 1 && cfg.use_gptj_residual && (dh == 64 || dh == 128);
        stats.decode_path = pplan.ok ? 1 : (staged ? 0 : 2);
        if (!ses.path_logged) {  // once per request
            ses.path_logged = true;
            FT_LOG_DEBUG(cfg.device, "decoder of this request: %s (rows %d, context %d%s)",
                         pplan.ok ? (pplan.a4 ? "persistent layers, second form (attention branch on the control waves)"
                                              : "persistent layers")
                                  : (staged ? "per-stage launches" : "general path (batched GEMMs)"),
                         B, s_max, pplan.ok ? (pplan.uk == 16 ? ", 512 keys per KV split" : ", 256 keys per KV split") : "");
        }
        if (pplan.ok) {
            // all stages of every layer inside persistent launches (kernels_persist.hip); one launch per token when
            // there is no collective between the layers
            PersistParams pp = persist_params(B, s_max);
            // algorithmic bytes of a layer: its four weight matrices + the K/V rows of the current length
            const double layer_bytes = wbytes * ((double)H * 3 * hl + (double)H * il + (double)hl * H + (double)il * H)
                                       + 4.0 * ses.next_step * hl * B;
            pp.l_begin = 0;
            pp.l_end   = L;
            if (cfg.tensor_para_size > 1 && cfg.comm->local) {
                // local group: ONE launch runs every rank (workgroups [r * NB, (r + 1) * NB) = rank r), issued by rank 0
                // between two thread barriers; the other ranks' streams are idle meanwhile
                LocalGroup& g = *cfg.comm->local;
                FTCF_HIP_CHECK(hipStreamSynchronize(stream));  // this rank's inputs (x, rotary table, state) are complete
                g.item[cfg.tensor_para_rank] = &pp;
                g.barrier();
                if (cfg.tensor_para_rank == 0) {
                    PersistGroupParams gp{};
                    for (int r = 0; r < g.world; r++) {
                        gp.p[r] = *static_cast<const PersistParams*>(g.item[r]);
                    }
                    gp.world = g.world;
                    gp.nb    = pplan.NB;
                    launch_decode_persistent_group(gp, int8, stream);
                    FTCF_HIP_CHECK(hipStreamSynchronize(stream));
                }
                g.barrier();
                return;
            }
            timed(KIND_PERSIST, layer_bytes * L + (ps_lm_fused ? 2.0 * V * H : 0.0), [&] { launch_decode_persistent(pp, int8, stream); });
            return;
        }
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            // layer_input/output alias for 0 < l < L-1 in the reference (:249-250) -> which residual form it runs
            const int inplace = (l > 0 && l < L - 1) ? 1 : 0;
            if (staged) {
                // Per-stage launches over row groups of <= 4 rows (the GEMV kernels' register budget).  STAGE_MAX_ROWS > 4
                // would replay every stage per group; measured no faster than the batched GEMM path (the m = 4 forms
                // of these kernels stream at half the m = 1 rate), so larger batches take the small-m GEMM below.
                const int ngrp = (B + 3) / 4;
                for (int stage = 0; stage < 3; stage++) {
                    for (int rg = 0; rg < ngrp; rg++) {
                        const int r0 = rg * 4, M = std::min(4, B - r0);
                        stage_launch(stage, l, w, inplace, B, s_max, r0, M, l + rg * L, ngrp == 1);
                    }
                }
            }
            else {
                // general path: both LayerNorms of the layer come from one pass over x, fused with the previous layer's
                // residual when there is no collective in between
                MmhaParams mp = mmha_params(l, w, B, s_max, 0, B, l);
                if (ses.K > 1) {
                    mp.cache_indir   = cache_indir;
                    mp.beam_width    = ses.K;
                    mp.max_input_len = ses.S;
                    mp.indir_plane   = (size_t)B * s_max;
                }
                if (!cfg.use_gptj_residual) {
                    // sequential residual (GptNeoXDecoder.cc:313-331,362-367): h = attn + bias + x ; x' = ffn(LN2(h)) + bias + h
                    launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, stream);
                    self_attention_layer.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    allreduce(att, (size_t)B * H);
                    launch_add_bias_residual(att, x, att, w.attn_out.bias, B, H, stream);
                    launch_layernorm(att, w.ln2_g, w.ln2_b, nrm, B, H, 1e-5f, true, stream);
                    ffn_layer.forward(nrm, mid, ffn, w, B, stream);
                    allreduce(ffn, (size_t)B * H);
                    launch_add_bias_residual(x, ffn, att, w.ffn2.bias, B, H, stream);
                    continue;
                }
                const bool dual = residual_dual_ln_supported(H);
                const bool tp1  = cfg.tensor_para_size == 1;
                if (!dual) {
                    launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, stream);
                    launch_layernorm(x, w.ln2_g, w.ln2_b, nrm2, B, H, 1e-5f, true, stream);
                }
                else if (l == 0 || !tp1) {
                    launch_residual_dual_ln(x, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm,
                                            nrm2, B, H, 1e-5f, stream);
                }
                if (B <= SMALLM_MAX_ROWS && smallm_ws && decode_branches && side) {
                    // The attention branch [QKV -> MMHA -> out-proj] (78.6 + K/V + 26.2 MB at 13B int8) and the FFN branch
                    // [FFN1 -> FFN2] (2 x 104.9 MB) of a parallel-residual layer are independent: two streams.  Every one
                    // of these launches is a short burst -- the whole matrix requested at once, gone in ~30 us -- whose
                    // ramp-up and drain leave the HBM idle; the other branch's launch fills those gaps.
                    const int    bc = std::min(B, 16);
                    const size_t o_qkv = 0, o_f1 = o_qkv + gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8),
                                 o_out = o_f1 + gemm_smallm_workspace_bytes(bc, il, H, int8),
                                 o_f2  = o_out + gemm_smallm_workspace_bytes(bc, H, hl, int8);
                    auto one = [&](const SmallmDesc& d0, size_t off, hipStream_t s) {
                        // (one launch that keeps the weights in registers and passes the rows 16 at a time through the x tile
                        // was measured: 256 VGPRs, one workgroup per CU -- 8.2 / 8.3 / 13.0 ms at 24 / 32 / 64 rows, i.e.
                        // slower than re-reading the weights per 16 rows except at 64)
                        for (int r0 = 0; r0 < B; r0 += 16) {  // 16 rows per launch (launches of one GEMM are in stream order)
                            SmallmDesc d = d0;
                            d.A          = d0.A + (size_t)r0 * d0.k;
                            d.C          = d0.C + (size_t)r0 * d0.n;
                            const int M  = std::min(16, B - r0);
                            timed(KIND_SMALLM, wbytes * (double)d.n * d.k, [&] {
                                launch_gemm_smallm_group(&d, 1, smallm_ws, smallm_partial, M, int8, s, &state->step, &smallm_seq, off);
                            }, s);
                        }
                    };
                    // the attention layer on the engine stream, the FFN layer on the side stream: the same two layer units, their
                    // GEMMs bound to the burst kernel with one workspace region per GEMM of the layer
                    const size_t offs[4] = {o_qkv, o_f1, o_out, o_f2};
                    GemmFn burst = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int, int n, int k,
                                       hipStream_t s, int slot) { one(SmallmDesc{A, dw.kernel, dw.scale, bias, act, C, n, k}, offs[slot], s); };
                    const DecoderSelfAttentionLayer attn_b{burst, H, hl};
                    const FfnLayer                  ffn_b{burst, H, il};
                    FTCF_HIP_CHECK(hipEventRecord(ev_fork, stream));
                    FTCF_HIP_CHECK(hipStreamWaitEvent(side, ev_fork, 0));
                    attn_b.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    ffn_b.forward(nrm2, mid, ffn, w, B, side);
                    FTCF_HIP_CHECK(hipEventRecord(ev_join, side));
                    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ev_join, 0));
                }
                else if (B <= 16 && smallm_ws) {
                    // independent GEMMs share a launch (a dependent launch costs ~8 us of dispatch latency, most of a layer
                    // at tensor-parallel shard sizes): [QKV, FFN1] -> MMHA -> [out-proj, FFN2]
                    const SmallmDesc p1[2] = {{nrm, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv, 3 * hl, H},
                                              {nrm2, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid, il, H}};
                    timed(KIND_SMALLM, wbytes * H * (3.0 * hl + il),
                          [&] { launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, B, int8, stream, &state->step, &smallm_seq); });
                    launch_mmha(mp, stream);
                    const SmallmDesc p3[2] = {{ctx, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, att, H, hl},
                                              {mid, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffn, H, il}};
                    timed(KIND_SMALLM, wbytes * H * ((double)hl + il),
                          [&] { launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, B, int8, stream, &state->step, &smallm_seq); });
                }
                else {
                    self_attention_layer.forward(nrm, qkv, ctx, att, w, mp, B, stream);
                    ffn_layer.forward(nrm2, mid, ffn, w, B, stream);
                }
                if (dual && tp1) {
                    const LayerWeights* nx = l + 1 < L ? &layers[l + 1] : nullptr;
                    launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, inplace, nx ? nx->ln1_g : nullptr,
                                            nx ? nx->ln1_b : nullptr, nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr,
                                            nrm, nrm2, B, H, 1e-5f, stream);
                }
                else {
                    launch_add_bias_attn_ffn_residual(x, ffn, att, x, w.ffn2.bias, B, H, cfg.tensor_para_size, inplace,
                                                      true, stream);
                }
            }
            allreduce(x, (size_t)B * H);
        }
    }

    // Rows up to which the per-stage GEMV launches run (when the persistent kernel is not eligible).  Measured at 13B int8,
    // TP = 1, ms per step, per-stage vs general path (burst GEMMs): B = 1: 3.34 vs 3.61 (and 1.28 vs 1.57 on a TP = 8
    // shard); B = 2: 4.13 vs 3.85; B = 3: 4.64 vs 3.65; B = 4: 5.78 vs 3.73 -- the m = 2..4 forms of the GEMV kernels stream
    // at a fraction of the m = 1 rate.  FTCF_STAGE_MAX_ROWS (<= 4) overrides, the tests use it to keep those forms covered.
    int STAGE_MAX_ROWS = 1;
    // Rows up to which the batched decode GEMMs (and short prompt phases) run the burst kernel, 16 rows per launch (the weights
    // are then read ceil(B / 16) times).  Above 16 rows the alternative is the tiled GEMM in its split-K form (64-row tiles cut
    // along K, four k-steps of weights and activations in flight): 13B int8, ms per decode step, chunked burst vs split-K tiled:
    // bs = 24: 6.8 vs 6.4; 32: 7.4 vs 6.5; 48: 10.2 vs 7.5; 64: 13.6 vs 8.7 -- and prompt phases of 17 / 33 / 64 tokens
    // 7.0 / 10.2 / 13.8 vs 6.0 / 6.2 / 6.5 ms.  (Round 2's tiled GEMM without the split: 13.8 ms at bs = 24, 19.4 at 64.)
    // FTCF_SMALLM_MAX_ROWS overrides (<= 256; the chunked form stays covered by the tests through it).
    int SMALLM_MAX_ROWS = 16;

    // decoder attention of rows [r0, r0 + M) of the batch (KV cache [L][B][nh][s_max][dh]); `salt` makes the granule
    // tags of every launch of a token distinct
    MmhaParams mmha_params(int l, const LayerWeights& w, int B, int s_max, int r0, int M, int salt)
    {
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        const size_t row_kv  = (size_t)nhl * s_max * dh;
        MmhaParams   mp{};
        mp.qkv = qkv + (size_t)r0 * 3 * hl;
        mp.qkv_bias = w.qkv.bias;
        mp.k_cache = k_cache + l * cache_l + r0 * row_kv;
        mp.v_cache = v_cache + l * cache_l + r0 * row_kv;
        mp.seq_len = seq_len + r0;
        mp.pad_count = pad_count + r0;
        mp.masked_tokens = masked + (size_t)r0 * s_max;
        mp.finished = finished + r0;
        mp.d_step = &state->step;
        mp.rot_table = rot_table + (size_t)r0 * (cfg.rotary_embedding_dim / 2) * 2;
        mp.B = M;
        mp.nh = nhl;
        mp.dh = dh;
        mp.rot = cfg.rotary_embedding_dim;
        mp.s_max = s_max;
        mp.ctx = ctx + (size_t)r0 * hl;
        mp.gran = (unsigned long long*)mmha_ws + (size_t)r0 * nhl * nsplit * (dh + 2);
        mp.layer = salt;
        mp.nsplit = nsplit;
        return mp;
    }

    // One of the three launches of a layer for rows [r0, r0 + M), M <= 4:
    //   0: K1  LN1 -> QKV                                  (78.6 MB/TP int8)
    //   1: K2  MMHA  ||  LN2 -> FFN1 + bias + gelu         (attention hidden under 104.9 MB/TP of streaming)
    //   2: K3  [out-proj U FFN2] -> residual               (131 MB/TP)
    void stage_launch(int stage, int l, const LayerWeights& w, int inplace, int B, int s_max, int r0, int M, int salt,
                      bool time_it)
    {
        const double wbytes = int8 ? 1.0 : 2.0;
        f16*         xr     = x + (size_t)r0 * H;
        auto run = [&](int kind, double bytes, auto&& f) {
            if (time_it) {
                timed(kind, bytes, f);
            }
            else {
                f();
            }
        };
        if (stage == 0) {
            LnGemvParams a{};
            a.x = xr;
            a.gamma0 = w.ln1_g;
            a.beta0 = w.ln1_b;
            a.W0 = w.qkv.kernel;
            a.scale0 = w.qkv.scale;
            a.out0 = qkv + (size_t)r0 * 3 * hl;
            a.K = H;
            a.NT0 = 3 * hl / 16;
            a.NT1 = 0;
            a.blocks0 = (a.NT0 + 3) / 4;
            a.blocks1 = 0;
            a.eps = 1e-5f;
            run(KIND_LN_GEMV, wbytes * H * (3.0 * hl), [&] {
                if (k1_wpg > 0) {
                    launch_ln_gemv_group(a, int8, M, k1_wpg, stream);
                }
                else {
                    launch_ln_gemv(a, int8, M, stream);
                }
            });
        }
        else if (stage == 1) {
            MmhaParams   mp = mmha_params(l, w, B, s_max, r0, M, salt);
            LnGemvParams f{};
            f.x = xr;
            f.gamma1 = w.ln2_g;
            f.beta1 = w.ln2_b;
            f.W1 = w.ffn1.kernel;
            f.scale1 = w.ffn1.scale;
            f.bias1 = w.ffn1.bias;
            f.out1 = mid + (size_t)r0 * il;
            f.K = H;
            f.NT0 = 0;
            f.NT1 = il / 16;
            f.blocks0 = 0;
            f.blocks1 = f.NT1 / 2;  // two column groups per workgroup (NT1 is even: local inter is a multiple of 64)
            f.eps = 1e-5f;
            run(KIND_FUSED, wbytes * H * (double)il, [&] { launch_mmha_ln_gemv(mp, f, int8, M, stream); });
        }
        else {
            const int tk = int8 ? TILE_K_I8 : TILE_K_F16;
            if (k3_q > 0) {
                ChunkParams c{};
                c.x_a = ctx + (size_t)r0 * hl;
                c.x_b = mid + (size_t)r0 * il;
                c.W_a = w.attn_out.kernel;
                c.W_b = w.ffn2.kernel;
                c.scale_a = w.attn_out.scale;
                c.scale_b = w.ffn2.scale;
                c.bias = w.ffn2.bias;
                c.x_in = xr;
                c.out = xr;
                c.N = H;
                c.KT_a = hl / tk;
                c.KT_b = il / tk;
                c.Q = k3_q;
                c.T = (c.KT_a + c.KT_b + c.Q - 1) / c.Q;
                c.tp = cfg.tensor_para_size;
                c.inplace_variant = inplace;
                c.gran = chunk_ws;
                c.d_step = &state->step;
                c.salt = salt;
                run(KIND_SPLITK, wbytes * H * ((double)hl + il), [&] { launch_gemv_chunked(c, int8, M, stream); });
            }
            else {
                SplitKParams c{};
                c.x_a = ctx + (size_t)r0 * hl;
                c.x_b = mid + (size_t)r0 * il;
                c.W_a = w.attn_out.kernel;
                c.W_b = w.ffn2.kernel;
                c.scale_a = w.attn_out.scale;
                c.scale_b = w.ffn2.scale;
                c.bias = w.ffn2.bias;
                c.x_in = xr;
                c.out = xr;
                c.N = H;
                c.KT_a = hl / tk;
                c.KT_b = il / tk;
                c.tp = cfg.tensor_para_size;
                c.inplace_variant = inplace;
                plan_splitk(c, int8, M, 10);
                run(KIND_SPLITK, wbytes * H * ((double)hl + il),
                    [&] { launch_gemv_splitk(c, int8, M, EPI_RESIDUAL, stream); });
            }
        }
    }

    // ---- request session: forward() == begin() + step(all) + finish() ----
    struct Session {
        bool              active = false;
        ftcf_forward_args a{};
        SamplingParams    sp{};
        BeamParams        bp{};
        int               B = 0, S = 0, total = 0, s_max = 0;  // B = rows (batch * beam_width)
        int               K = 1, batch = 0;
        int               next_step = 0;  // host mirror of state->step
        int               steps = 0;
        bool              all_finished = false;
        hipEvent_t        e0 = nullptr, e1 = nullptr;
        hipGraphExec_t    graph_exec = nullptr;
        bool              path_logged = false;  // the decoder of this request has been named in the log (FT_LOG_LEVEL=DEBUG)
        hipGraphExec_t    graph_exec_n = nullptr;  // graph_tokens consecutive tokens in one graph (persistent path, no callback)
    } ses;
    bool use_graph = true;
    // drops whatever an unfinished request left behind: the captured graph holds the OLD arena pointers, shapes and sampling
    // flags, and plan() may free that arena -- replaying it for the next request would corrupt memory silently
    void abandon_session()
    {
        if (!ses.active && !ses.graph_exec) {
            return;
        }
        (void)hipStreamSynchronize(stream);
        if (ses.graph_exec) {
            (void)hipGraphExecDestroy(ses.graph_exec);
            ses.graph_exec = nullptr;
        }
        if (ses.graph_exec_n) {
            (void)hipGraphExecDestroy(ses.graph_exec_n);
            ses.graph_exec_n = nullptr;
        }
        if (ses.e0) {
            event_pool.push_back(ses.e0);
            ses.e0 = nullptr;
        }
        if (ses.e1) {
            event_pool.push_back(ses.e1);
            ses.e1 = nullptr;
        }
        drain_events();
        ses.active = false;
    }
    void begin(const ftcf_forward_args& a);
    void enqueue_step(bool with_decoder);
    int  step(int max_steps);
    void finish();
    bool persist_failed = false;  // the persistent kernel gave up on a hand-off during the last request
    bool winar_failed = false;    // ... or the exchange-window all-reduce of the prompt phase did
    int  persist_fail_once = 0;
    void forward(const ftcf_forward_args& a)
    {
        begin(a);
        step(a.output_len);
        try {
            finish();
        }
        catch (const Error&) {
            if (winar_failed && !persist_failed) {
                winar_failed = false;
                FT_LOG_WARNING(cfg.device, "exchange-window all-reduce gave up: replaying the request with the communicator's own "
                                           "all-reduce (it stays there)");
                begin(a);
                step(a.output_len);
                finish();
                return;
            }
            if (!persist_failed) {
                throw;
            }
            // The persistent kernel's hand-offs need every workgroup resident; the plan checks that, but compute units can
            // still be taken away (another process, a masked CU) after the check.  Its spins are bounded and report through
            // a sticky error word instead of hanging the GPU; the request is then replayed from the start on the
            // per-stage / general path (tensor parallel: every rank takes this branch -- finish() agrees on the error
            // word across the ranks) and the engine stays off the persistent path.
            persist_failed = false;
            persist        = 0;
            FT_LOG_WARNING(cfg.device, "persistent decode kernel gave up on a hand-off: replaying the request on the per-stage "
                                       "path (this engine stays there)");
            begin(a);
            step(a.output_len);
            finish();
        }
    }
};

// transposeAxis01 for the TP logits all-gather: [tp][B][vl] -> [B][V] (GptNeoX.cc:913-924)
__global__ void k_transpose_gathered_logits(float* out, const float* in, int tp, int B, int vl)
{
    const size_t total = (size_t)tp * B * vl;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int    j = (int)(i % vl);
        const size_t t = i / vl;
        const int    b = (int)(t % B), r = (int)(t / B);
        out[(size_t)b * tp * vl + (size_t)r * vl + j] = in[i];
    }
}

template<typename T>
static std::vector<T> broadcast_arg(const T* p, int n, int B, T dflt, const char* name)
{
    std::vector<T> v((size_t)B, dflt);
    if (n > 0) {
        FTCF_CHECK_ARG(p != nullptr, std::string(name) + " pointer is NULL");
        FTCF_CHECK_ARG(n == 1 || n == B, std::string(name) + " must have 1 or batch_size entries");
        for (int b = 0; b < B; b++) {
            v[b] = p[n == 1 ? 0 : b];
        }
    }
    return v;
}

void ftcf_gptneox::begin(const ftcf_forward_args& a)
{
    Range r("ftcf.begin");
    FT_LOG_TRACE(cfg.device, "begin: batch %d x beam %d, max_input_len %d, output_len %d", a.batch_size, a.beam_width, a.max_input_len,
                 a.output_len);
    const int B = a.batch_size * (a.beam_width > 0 ? a.beam_width : 1);  // rows
    const int S = a.max_input_len, out_len = a.output_len;
    FTCF_CHECK_ARG(a.batch_size >= 1 && S >= 1 && out_len >= 1, "batch_size, max_input_len and output_len must be >= 1");
    FTCF_CHECK_ARG(a.input_ids && a.input_lengths && a.output_ids && a.sequence_lengths, "NULL tensor");
    const int K = a.beam_width, batch = a.batch_size;
    FTCF_CHECK_ARG(K >= 1 && K <= BEAM_MAX_K, "beam_width must be in [1, 64]");
    FTCF_HIP_CHECK(hipSetDevice(cfg.device));
    abandon_session();  // (a request left open -- begin / step without finish, or a step that threw -- must not leak its graph)
    stats.window_allreduces = 0;
    // everything the caller enqueued on its stream (input tensors) happens-before the engine's work
    FTCF_HIP_CHECK(hipEventRecord(ev_user, user_stream));
    FTCF_HIP_CHECK(hipStreamWaitEvent(stream, ev_user, 0));
    const int total = S + out_len;  // max_output_seq_len == max_seq_len == max_cache_seq_len (GptNeoX.cc:520-523)
    const int s_max = total;
    ses.K = K;  // (decoder path selection reads it)
    const int  tpn     = cfg.tensor_para_size;
    const bool want_tp = tpn > 1 && persist && persist_tp && K == 1 && B <= 2 && cfg.use_gptj_residual && tpn <= PERSIST_MAX_TP;
    // message buffers of the RCCL-free prompt-phase all-reduce (k_window_allreduce) behind the granules of the decode exchange
    static const int winar_on = getenv("FTCF_TP_WINAR") ? atoi(getenv("FTCF_TP_WINAR")) : 1;
    static const int winar_mb = getenv("FTCF_TP_WINAR_MB") ? atoi(getenv("FTCF_TP_WINAR_MB")) : 16;
    const bool want_ar = tpn > 1 && tpn <= 8 && winar_on && !fp32 && cfg.comm && !cfg.comm->local && !cfg.comm->ar_failed;
    if (want_tp || want_ar) {
        // (collective: every rank sees the same request shape) room for two rows: [tp][2 * H / 2] granules
        // (two planes by layer parity: persist_device.hip.h ps_tp_exchange), then 16 flags, then four message buffers
        const size_t gran = ((size_t)2 * tpn * H * 8 + 4095) & ~(size_t)4095;
        const size_t cap  = want_ar ? (size_t)std::max(1, winar_mb) << 20 : 0;
        const bool   had  = cfg.comm->win_ok && cfg.comm->win_bytes >= gran + 4096 + 4 * cap;
        comm_ensure_window(cfg.comm, gran + 4096 + 4 * cap, stream);
        if (cfg.comm->win_ok && cfg.comm->win_bytes >= gran + 4096 + 4 * cap && cap > 0 && (!had || cfg.comm->ar_cap != cap)) {
            cfg.comm->ar_flag_off = gran;  // (a fresh window is zero: the call numbers start over)
            cfg.comm->ar_data_off = gran + 4096;
            cfg.comm->ar_cap      = cap;
            if (!had) {
                cfg.comm->ar_seq = 0;
                if (cfg.comm->ar_sync) {
                    FTCF_HIP_CHECK(hipMemsetAsync(cfg.comm->ar_sync, 0, 64, stream));
                }
            }
        }
    }
    plan(B, S, total, K);
    if (want_tp && pplan.ok) {
        // granule tags repeat from request to request: every rank's window is zeroed between two barriers -- nobody is
        // still writing into it from the previous request, nobody writes before it is clean
        if (!tp_scratch) {
            FTCF_HIP_CHECK(hipMalloc((void**)&tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(tp_scratch, 0, 256, stream));
        }
        comm_barrier(cfg.comm, stream, tp_scratch);
        FTCF_HIP_CHECK(hipMemsetAsync(cfg.comm->win[cfg.tensor_para_rank], 0, (size_t)2 * tpn * H * 8, stream));
        comm_barrier(cfg.comm, stream, tp_scratch);
    }

    // ---- runtime args: routing of TopKSamplingLayer.cu:27-77 / TopPSamplingLayer.cu:30-110 ----
    auto top_k = broadcast_arg<int>(a.top_k, a.n_top_k, batch, 0, "top_k");
    auto top_p = broadcast_arg<float>(a.top_p, a.n_top_p, batch, 0.f, "top_p");
    auto temp  = broadcast_arg<float>(a.temperature, a.n_temperature, batch, 1.f, "temperature");
    auto rep   = broadcast_arg<float>(a.repetition_penalty, a.n_repetition_penalty, batch, 1.f, "repetition_penalty");
    auto seed  = broadcast_arg<uint64_t>(a.random_seed, a.n_random_seed, batch, 0, "random_seed");
    auto minl  = broadcast_arg<int>(a.min_length, a.n_min_length, batch, 0, "min_length");
    auto divr  = broadcast_arg<float>(a.beam_search_diversity_rate, a.n_beam_search_diversity_rate, batch, 0.f,
                                      "beam_search_diversity_rate");
    auto lenp  = broadcast_arg<float>(a.len_penalty, a.n_len_penalty, batch, 0.f, "len_penalty");
    std::vector<int>   k_eff(batch);
    std::vector<float> p_topk(batch), p_topp(batch);
    bool               temp_all_one = true, rep_all_default = true, any_min = false;
    for (int b = 0; b < batch; b++) {
        int   k = top_k[b];
        float p = top_p[b];
        FTCF_CHECK_ARG(k >= 0, "top_k must be >= 0");
        if (k == 0 && p == 0.0f) {
            k = 1;
        }
        float pk = p;
        if (k > 0 && pk == 0.0f) {
            pk = 1.0f;
        }
        k_eff[b]  = k > 1024 ? 1024 : k;
        p_topk[b] = pk < 0.f ? 0.f : (pk > 1.f ? 1.f : pk);
        p_topp[b] = p < 0.f ? 0.f : (p > 1.f ? 1.f : p);
        temp_all_one &= (temp[b] == 1.0f);
        rep_all_default &= (rep[b] == 1.0f);
        any_min |= (minl[b] > 0);
    }
    FTCF_HIP_CHECK(hipMemcpyAsync(d_top_k, k_eff.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_p_topk, p_topk.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_p_topp, p_topp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_temp, temp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_rep, rep.data(), batch * 4, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_seed, seed.data(), batch * 8, hipMemcpyHostToDevice, stream));
    FTCF_HIP_CHECK(hipMemcpyAsync(d_min_length, minl.data(), batch * 4, hipMemcpyHostToDevice, stream));
    if (K > 1) {
        FTCF_HIP_CHECK(hipMemcpyAsync(d_div, divr.data(), batch * 4, hipMemcpyHostToDevice, stream));
        FTCF_HIP_CHECK(hipMemcpyAsync(d_lenpen, lenp.data(), batch * 4, hipMemcpyHostToDevice, stream));
    }
    comm_stream_sync(cfg.comm, stream);  // the host vectors die at scope exit

    hipEvent_t e0 = get_event(), e1 = get_event();
    FTCF_HIP_CHECK(hipEventRecord(e0, stream));
    FTCF_HIP_CHECK(hipMemsetAsync(mmha_ws, 0, mmha_workspace_bytes(B, nhl, dh, nsplit), stream));
    FTCF_HIP_CHECK(hipMemsetAsync(chunk_ws, 0, chunk_workspace_bytes(H, std::min(B, 4), 8), stream));
    if (tiled_ws) {  // tickets of the split-K tiles (re-armed by every launch; a failed request may leave one drawn)
        FTCF_HIP_CHECK(hipMemsetAsync(reinterpret_cast<char*>(tiled_ws) + gemm_tiled_workspace_bytes() - gemm_tiled_ticket_bytes(), 0,
                                      gemm_tiled_ticket_bytes(), stream));
    }
    if (smallm_ws) {
        // granules of the batched-decode GEMMs' in-launch reduction: their tags repeat from request to request
        FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), stream));
    }
    if (pplan.ok) {
        const size_t cache_l = (size_t)B * nhl * s_max * dh;
        std::vector<PersistLayer> pl(L);
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = layers[l];
            PersistLayer&       r = pl[l];
            r.ln1_g = w.ln1_g;
            r.ln1_b = w.ln1_b;
            r.ln2_g = w.ln2_g;
            r.ln2_b = w.ln2_b;
            r.w_qkv = w.qkv.kernel;
            r.w_ffn1 = w.ffn1.kernel;
            r.w_out = w.attn_out.kernel;
            r.w_ffn2 = w.ffn2.kernel;
            r.s_qkv = w.qkv.scale;
            r.s_ffn1 = w.ffn1.scale;
            r.s_out = w.attn_out.scale;
            r.s_ffn2 = w.ffn2.scale;
            r.b_qkv = w.qkv.bias;
            r.b_ffn1 = w.ffn1.bias;
            r.b_res = w.ffn2.bias;
            r.k_cache = k_cache + l * cache_l;
            r.v_cache = v_cache + l * cache_l;
        }
        FTCF_HIP_CHECK(hipMemcpyAsync(d_players, pl.data(), sizeof(PersistLayer) * L, hipMemcpyHostToDevice, stream));
        FTCF_HIP_CHECK(hipMemsetAsync(ps_gq, 0, (ps_slab_n + 8) * 8, stream));
        ps_tab_ready = false;
        static const int tab_env = getenv("FTCF_PERSIST_TABLES") ? atoi(getenv("FTCF_PERSIST_TABLES")) : 1;
        if (tab_env && ps_tab && !(cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->local)) {
            // the run / tile tables of this plan: one launch over no layers builds and stores them, every token's launch
            // loads them (17 us of table building per launch otherwise)
            PersistParams pp = persist_params(B, s_max);
            pp.l_begin = pp.l_end = 0;
            pp.tab_mode = 1;
            pp.d_stop   = nullptr;  // (the flag still holds the previous request's outcome here)
            launch_decode_persistent(pp, int8, stream);
            ps_tab_ready = true;
        }
        comm_stream_sync(cfg.comm, stream);  // `pl` dies at scope exit
    }
    // beam search: the reference tiles the inputs K times and runs the context phase on all batch * K rows
    // (GptNeoX.cc:560-574, 640-735).  Every beam reads the prompt K/V of beam 0 anyway (the cache indirection starts at 0),
    // so the prompt is prefilled once per request into cache row b * K and its last hidden state is tiled.
    const int* in_ids = a.input_ids;
    const int* in_len = a.input_lengths;
    if (K > 1) {
        launch_tile_inputs(tiled_ids, tiled_len, a.input_ids, a.input_lengths, batch, K, S, stream);
        FTCF_HIP_CHECK(hipMemsetAsync(cache_indir, 0, (size_t)2 * B * s_max * 4, stream));
        FTCF_HIP_CHECK(hipMemsetAsync(parent_ids, 0, (size_t)total * B * 4, stream));
        in_ids = tiled_ids;
        in_len = tiled_len;
    }
    launch_decode_init(finished, seq_len, cum, pad_count, masked, draws, in_len, state, B, S, s_max, stream, K);
    if (S > 1 && K > 1 && fp32) {
        launch32_prompt_embedding(F(px), nullptr, F(wte), a.input_ids, batch, S, H, stream);
        launch_tile_prompt_ids(step_ids, a.input_ids, batch, K, S, stream);
        context_decoder32(batch, S, a.input_lengths, s_max, K);
        launch32_gather_last_token(F(x), F(px), a.input_lengths, batch, S, H, stream, K);
    }
    else if (S > 1 && fp32) {
        launch32_prompt_embedding(F(px), step_ids, F(wte), in_ids, B, S, H, stream);
        context_decoder32(B, S, in_len, s_max, 1);
        launch32_gather_last_token(F(x), F(px), in_len, B, S, H, stream);
    }
    else if (S > 1 && K > 1) {
        launch_prompt_embedding(px, nullptr, wte, a.input_ids, batch, S, H, stream);
        launch_tile_prompt_ids(step_ids, a.input_ids, batch, K, S, stream);
        context_decoder(batch, S, a.input_lengths, s_max, K);
        launch_gather_last_token(x, px, a.input_lengths, batch, S, H, stream, K);
    }
    else if (S > 1) {
        launch_prompt_embedding(px, step_ids, wte, in_ids, B, S, H, stream);
        context_decoder(B, S, in_len, s_max, 1);
        launch_gather_last_token(x, px, in_len, B, S, H, stream);
    }
    else {
        FTCF_HIP_CHECK(hipMemcpyAsync(step_ids, in_ids, (size_t)B * 4, hipMemcpyDeviceToDevice, stream));
    }
    FTCF_HIP_CHECK(hipEventRecord(e1, stream));

    SamplingParams sp{};
    sp.logits = logits;
    sp.B = B;
    sp.V = V;
    sp.max_input_len = S;
    sp.total_len = total;
    sp.end_id = cfg.end_id;
    sp.input_lengths = in_len;
    sp.top_k = d_top_k;
    sp.max_top_k = 1;
    sp.any_top_p = 0;
    for (int i = 0; i < batch; i++) {
        sp.max_top_k = std::max(sp.max_top_k, k_eff[i]);
        sp.any_top_p |= (k_eff[i] == 0);
    }
    sp.top_p_topk = d_p_topk;
    sp.top_p_topp = d_p_topp;
    sp.temperature = d_temp;
    sp.repetition_penalty = a.n_repetition_penalty > 0 ? d_rep : nullptr;
    sp.min_length = any_min ? d_min_length : nullptr;
    sp.random_seed = d_seed;
    sp.draw_counter = draws;
    sp.apply_temperature = temp_all_one ? 0 : 1;
    sp.apply_repetition = (a.n_repetition_penalty > 0 && !rep_all_default) ? 1 : 0;
    sp.stop_words = K > 1 ? nullptr : a.stop_words_list;  // (the beam kernel checks them along the parent chain)
    sp.stop_len = a.stop_words_len;
    sp.optional_last_tokens = a.optional_last_tokens;
    sp.optional_count = a.optional_last_tokens_count;
    sp.return_cum_log_probs = a.return_cum_log_probs ? 1 : 0;
    sp.output_ids = step_ids;
    sp.finished = finished;
    sp.seq_len = seq_len;
    sp.cum_log_probs = cum;
    sp.pad_count = pad_count;
    sp.state = state;
    sp.h_flags = h_flags;
    sp.ws = samp_ws;

    BeamParams bp{};
    if (K > 1) {
        bp.logits = logits;
        bp.B = batch;
        bp.K = K;
        bp.V = V;
        bp.max_input_len = S;
        bp.total_len = total;
        bp.end_id = cfg.end_id;
        bp.s_max = s_max;
        bp.input_lengths = in_len;
        bp.temperature = d_temp;
        bp.repetition_penalty = a.n_repetition_penalty > 0 ? d_rep : nullptr;  // penalty type None otherwise
        bp.diversity_rate = d_div;
        bp.len_penalty = d_lenpen;
        bp.min_length = any_min ? d_min_length : nullptr;
        bp.stop_words = a.stop_words_list;
        bp.stop_len = a.stop_words_len;
        bp.optional_last_tokens = a.optional_last_tokens;
        bp.optional_count = a.optional_last_tokens_count;
        bp.output_ids = step_ids;
        bp.parent_ids = parent_ids;
        bp.finished = finished;
        bp.seq_len = seq_len;
        bp.cum_log_probs = cum;
        bp.cache_indir = cache_indir;
        bp.state = state;
        bp.ws = beam_ws;
    }

    ses.active = true;
    ses.a = a;
    ses.sp = sp;
    ses.bp = bp;
    ses.K = K;
    ses.batch = batch;
    ses.B = B;
    ses.S = S;
    ses.total = total;
    ses.s_max = s_max;
    ses.next_step = S;
    ses.steps = 0;
    ses.path_logged = false;
    ses.all_finished = false;
    ses.e0 = e0;
    ses.e1 = e1;
    stats.decode_ms = 0.f;
}

// enqueues one iteration of the token loop (GptNeoX.cc:776-1048) on `stream` (and the side stream)
void ftcf_gptneox::enqueue_step(bool with_decoder)
{
    const ftcf_forward_args& a = ses.a;
    const int B = ses.B, S = ses.S, s_max = ses.s_max;
    const int tp = cfg.tensor_para_size;
    if (with_decoder && fp32) {
        launch32_step_prologue(F(x), F(wte), step_ids, &state->step, rot_table, pad_count, B, H, cfg.rotary_embedding_dim,
                               stream);
        decoder32(B, s_max);
    }
    else if (with_decoder) {
            launch_step_prologue(x, wte, step_ids, &state->step, rot_table, pad_count, B, H, cfg.rotary_embedding_dim,
                                 stream, &state->all_finished);
            decoder(B, s_max);
        }
        // final LayerNorm (GptNeoX.cc:854-863) is fused into the LM-head GEMV for m <= 4
        const bool fuse_ln = B <= 4 && !fp32;
        const bool lm_done = with_decoder && pplan.ok && ps_lm_fused;  // the persistent launch computed the logits itself
        if (!fuse_ln && !lm_done) {
            launch_layernorm(x, final_g, final_b, nrm, B, H, 1e-5f, !fp32, stream);
        }
        auto lm = [&](const f16* Wrows, float* out, int rows, int ld) {
            if (fp32) {  // Wrows counts f16 elements: the caller's row offset is scaled here
                launch32_lm_head(F(nrm), F(lm_head) + (Wrows - lm_head), out, B, rows, H, ld, stream);
            }
            else if (fuse_ln) {
                launch_lm_head(x, Wrows, out, B, rows, H, ld, stream, final_g, final_b, 1e-5f, &state->all_finished);
            }
            else {
                lm_head_dispatch(nrm, Wrows, out, B, rows, H, ld, stream);
            }
        };
        if (lm_done) {
        }
        else if (tp == 1) {
            timed(KIND_LM_HEAD, 2.0 * V * H, [&] { lm(lm_head, logits, V, V); });
        }
        else {
            // rank r computes rows [r*vl, (r+1)*vl) of the replicated lm_head (GptNeoX.cc:888-925)
            float* mine = gather + (size_t)cfg.tensor_para_rank * B * vl;
            timed(KIND_LM_HEAD, 2.0 * vl * H,
                  [&] { lm(lm_head + (size_t)cfg.tensor_para_rank * vl * H, mine, vl, vl); });
            allgather_logits(gather, logits, B, stream);
        }
        if (a.debug_logits) {
            FTCF_HIP_CHECK(hipMemcpyAsync(a.debug_logits + (size_t)(ses.next_step - S) * B * V, logits,
                                          (size_t)B * V * 4, hipMemcpyDeviceToDevice, stream));
        }
    if (ses.K > 1) {
        dynamic_decode_layer.forward(ses.bp, ses.sp, stream);
    }
    else {
        dynamic_decode_layer.forward(ses.sp, stream);
    }
}

// the token loop of GptNeoX<T>::forward (GptNeoX.cc:776-1048); returns the number of iterations executed
int ftcf_gptneox::step(int max_steps)
{
    Range r("ftcf.step");
    FTCF_CHECK_ARG(ses.active, "no request in flight: call ftcf_gptneox_begin first");
    FTCF_HIP_CHECK(hipSetDevice(cfg.device));
    const ftcf_forward_args& a = ses.a;
    const int B = ses.B, S = ses.S, total = ses.total;
    const int tp = cfg.tensor_para_size;
    std::vector<int> h_tokens(B), h_idx(B), h_seq(B);
    hipEvent_t ea = get_event(), eb = get_event();
    FTCF_HIP_CHECK(hipEventRecord(ea, stream));
    // (read per call, not once per process: the tests switch it between engines)
    const int graph_tokens_cfg = getenv("FTCF_GRAPH_TOKENS") ? atoi(getenv("FTCF_GRAPH_TOKENS")) : 8;
    int  done    = 0;
    bool lagging = false;  // the host has not yet seen the flags of the token launched last
    int  lag_slot = 0;     // launches of the pipelined loop so far (two events alternate)
    while (done < max_steps && ses.next_step < total && !ses.all_finished) {
        const int  step         = ses.next_step;
        const bool with_decoder = !(S > 1 && step == S);
        // with tensor parallelism the step contains RCCL collectives: capturing them is opt-in (FTCF_TP_GRAPH=1) until it
        // has been validated on a multi-GPU node (this round's boxes have one GPU)
        const bool graph_ok     = use_graph && with_decoder && !profiling && (tp == 1 || (tp_graph && !cfg.comm->local && !cfg.comm->hx)) && !a.debug_logits;
        // Several tokens per graph launch (FTCF_GRAPH_TOKENS, default 8): a graph launch costs ~14 us of GPU idle time between
        // two tokens (profiles/r03_notes.md section 6), and every kernel of a persistent-path token returns at once when the
        // device-side "every row has finished" flag is set, so the tokens of a graph behind the request's last one cost a few
        // microseconds each, not a decoder pass.  Persistent decode path, no streaming callback, one GPU.
        const int  graph_tokens = std::max(1, std::min(64, graph_tokens_cfg));
        const bool multi        = graph_ok && graph_tokens > 1 && pplan.ok && ses.K == 1 && !a.callback && tp == 1
                                  && max_steps - done >= graph_tokens && total - step >= graph_tokens;
        int launched = 1;
        if (graph_ok) {
            hipGraphExec_t& ge = multi ? ses.graph_exec_n : ses.graph_exec;
            if (!ge) {
                // capture the regular decode step(s) (all pointers are fixed for the session, the step counter lives
                // on the device) and replay: no per-kernel host launch cost, cross-stream fork/join become edges
                hipGraph_t g = nullptr;
                FTCF_HIP_CHECK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
                try {
                    for (int t = 0; t < (multi ? graph_tokens : 1); t++) {
                        enqueue_step(true);
                    }
                }
                catch (...) {
                    (void)hipStreamEndCapture(stream, &g);
                    if (g) {
                        (void)hipGraphDestroy(g);
                    }
                    throw;
                }
                FTCF_HIP_CHECK(hipStreamEndCapture(stream, &g));
                FTCF_HIP_CHECK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
                FTCF_HIP_CHECK(hipGraphDestroy(g));
            }
            FTCF_HIP_CHECK(hipGraphLaunch(ge, stream));
            launched = multi ? graph_tokens : 1;
        }
        else {
            enqueue_step(with_decoder);
        }
        if (a.debug_logits) {
            // (logits are modified in place by the decode kernels; the tap is taken inside enqueue_step when eager)
        }
        ses.steps += launched;
        ses.next_step += launched;
        done += launched;
        // The reference synchronises once per token here (stop_criteria_kernels.cu:149-156).  Without a streaming
        // callback nothing on the host needs token t before token t+1 is enqueued, so the replayed graph of the next
        // token(s) is launched first and the host only waits for the PREVIOUS launch's event: the GPU never idles for the
        // host round trip (~25 us per token).  `finished` is sticky on the device and the launches behind the last token
        // return at once; the host's counters are set back to the device's when it learns of the end (below).
        if (graph_ok && !a.callback && tp == 1) {  // (tp > 1: every rank must leave the loop at the SAME token -> synchronous)
            hipEvent_t& ev = tok_ev[lag_slot & 1];
            if (!ev) {
                FTCF_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            }
            FTCF_HIP_CHECK(hipEventRecord(ev, stream));
            if (lagging) {
                comm_event_sync(cfg.comm, tok_ev[(lag_slot - 1) & 1]);
                ses.all_finished = h_flags[0] != 0;
            }
            lag_slot++;
            lagging = true;
            continue;
        }
        lagging = false;
        comm_stream_sync(cfg.comm, stream);
        ses.all_finished = h_flags[0] != 0;
        if (a.callback && step + 1 < total && cfg.tensor_para_rank == 0) {
            // pybind_callback_utils.cc:22-103: last token of every row; a row that did not advance reports end_id
            FTCF_HIP_CHECK(hipMemcpy(h_tokens.data(), step_ids + (size_t)step * B, (size_t)B * 4, hipMemcpyDeviceToHost));
            FTCF_HIP_CHECK(hipMemcpy(h_seq.data(), seq_len, (size_t)B * 4, hipMemcpyDeviceToHost));
            for (int b = 0; b < B; b++) {
                h_idx[b] = h_seq[b];
                if (h_seq[b] != step) {
                    h_tokens[b] = cfg.end_id;
                }
            }
            a.callback(h_tokens.data(), h_idx.data(), ses.batch, ses.K, a.callback_user);
        }
    }
    FTCF_HIP_CHECK(hipEventRecord(eb, stream));
    comm_event_sync(cfg.comm, eb, "the prefill");
    if (lagging) {
        ses.all_finished = h_flags[0] != 0;
    }
    if (ses.all_finished && ses.K == 1) {
        // launches behind the token that finished the last row did nothing on the device (every kernel of a token returns at
        // once on the flag, k_decode_finish included: the device's step counter stopped): the host's counters follow the
        // device's, so that `steps_done` is the reference's loop count (GptNeoX.cc:776-1048 leaves its loop at that token)
        const int over = ses.next_step - (h_flags[1] + 1);
        if (over > 0) {
            ses.next_step -= over;
            ses.steps -= over;
            done -= over;
        }
    }
    float ms = 0.f;
    FTCF_HIP_CHECK(hipEventElapsedTime(&ms, ea, eb));
    stats.decode_ms += ms;
    event_pool.push_back(ea);
    event_pool.push_back(eb);
    return done;
}

void ftcf_gptneox::finish()
{
    Range r("ftcf.finish");
    FTCF_CHECK_ARG(ses.active, "no request in flight");
    const ftcf_forward_args& a = ses.a;
    // setOutputTensors (GptNeoX.cc:1090-1181)
    if (ses.K > 1) {
        launch_gather_tree_beam(a.output_ids, a.sequence_lengths, step_ids, parent_ids, seq_len, tiled_len, ses.batch,
                                ses.K, ses.S, ses.total, cfg.end_id, stream);
    }
    else {
        launch_gather_tree(a.output_ids, a.sequence_lengths, step_ids, seq_len, a.input_lengths, ses.B, ses.S,
                           ses.total, cfg.end_id, stream);
    }
    if (a.cum_log_probs) {
        FTCF_HIP_CHECK(hipMemcpyAsync(a.cum_log_probs, cum, (size_t)ses.B * 4, hipMemcpyDeviceToDevice, stream));
    }
    int ps_error = 0, smallm_error = 0;
    if (pplan.ok) {
        FTCF_HIP_CHECK(hipMemcpyAsync(&ps_error, ps_err, sizeof(int), hipMemcpyDeviceToHost, stream));
    }
    if (smallm_ws) {
        FTCF_HIP_CHECK(hipMemcpyAsync(&smallm_error, reinterpret_cast<char*>(smallm_ws) + smallm_partial, sizeof(int),
                                      hipMemcpyDeviceToHost, stream));
    }
    comm_stream_sync(cfg.comm, stream);
    float ms = 0.f;
    FTCF_HIP_CHECK(hipEventElapsedTime(&ms, ses.e0, ses.e1));
    stats.prefill_ms   = ms;
    stats.decode_steps = ses.steps;
    if (ov_eligible && ov_trial < 2) {  // a trial of the auto mode: every rank keeps the slowest rank's time
        const int us   = comm_max(cfg.comm, (int)(ms * 1000.f), stream, tp_scratch);
        ov_ms[ov_trial] = us * 1e-3f;
        ov_trial++;
        if (ov_trial == 2) {
            FT_LOG_INFO(cfg.device, "prompt-phase all-reduce overlap (auto): plain %.2f ms, overlapped %.2f ms -> %s", ov_ms[0], ov_ms[1],
                        ov_ms[1] < ov_ms[0] ? "overlapped from now on" : "plain from now on");
        }
    }
    stats.prefill_overlap        = ov_ran ? 1 : 0;
    stats.prefill_ms_plain       = ov_ms[0];
    stats.prefill_ms_overlapped  = ov_ms[1];
    event_pool.push_back(ses.e0);
    event_pool.push_back(ses.e1);
    ses.e0 = ses.e1 = nullptr;
    ses.active = false;
    if (ses.graph_exec) {
        (void)hipGraphExecDestroy(ses.graph_exec);
        ses.graph_exec = nullptr;
    }
    if (ses.graph_exec_n) {
        (void)hipGraphExecDestroy(ses.graph_exec_n);
        ses.graph_exec_n = nullptr;
    }
    drain_events();
    if (pplan.ok && ps_ts) {
        std::vector<long long> h((size_t)pplan.NB * L * 128);
        FTCF_HIP_CHECK(hipMemcpy(h.data(), ps_ts, h.size() * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(ps_ts_file.c_str(), "wb")) {
            const int hdr[4] = {pplan.NB, L, 8, 16};
            fwrite(hdr, 4, 4, f);
            fwrite(h.data(), 8, h.size(), f);
            fclose(f);
        }
    }
    if (pplan.ok && persist_fail_once) {  // test hook (FTCF_PERSIST_FAIL_ONCE=1): pretend the kernel gave up once
        persist_fail_once = 0;
        ps_error          = 99;
    }
    if (pplan.ok && cfg.tensor_para_size > 1) {
        ps_error = comm_max(cfg.comm, ps_error, stream, tp_scratch);  // every rank learns of any rank's failure
    }
    if (cfg.tensor_para_size > 1 && cfg.comm && cfg.comm->ar_seq > 0 && !cfg.comm->ar_failed) {
        // the window all-reduce's sticky give-up word (a peer that never arrived: its bounded waits ran out), agreed on
        int ar_err = 0;
        FTCF_HIP_CHECK(hipMemcpy(&ar_err, cfg.comm->ar_sync + 2, sizeof(int), hipMemcpyDeviceToHost));
        if (!tp_scratch) {
            FTCF_HIP_CHECK(hipMalloc((void**)&tp_scratch, 256));
            FTCF_HIP_CHECK(hipMemsetAsync(tp_scratch, 0, 256, stream));
        }
        ar_err = comm_max(cfg.comm, ar_err, stream, tp_scratch);
        if (ar_err != 0) {
            cfg.comm->ar_failed = true;  // this communicator keeps RCCL (or the emulation) for the prompt phase from now on
            winar_failed        = true;
            throw Error(-2, "exchange-window all-reduce gave up waiting for a peer (its kernel was not running next to this one)");
        }
    }
    if (smallm_error != 0) {
        throw Error(-2, "batched decode GEMM: a split-K reducer gave up waiting for its sibling workgroups' partial sums");
    }
    if (ps_error != 0) {
        persist_failed = true;
        persist        = 0;  // whoever drives begin / step / finish: the next request is planned off the persistent path
        throw Error(-2, "persistent decode kernel gave up waiting for a hand-off (code " + std::to_string(ps_error)
                            + "): not every workgroup was resident");
    }
}

extern "C" int ftcf_gptneox_create(const ftcf_gptneox_config* cfg, const ftcf_gptneox_weights* w, ftcf_gptneox_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(cfg && w && out, "NULL argument");
        require_device();
        FTCF_CHECK_ARG(cfg->pipeline_para_size == 1, "pipeline_para_size must be 1 (the CodeFuse harness forces it)");
        if (cfg->dtype != FTCF_FP16 && cfg->dtype != FTCF_FP32) {
            throw Error(FTCF_ERR_UNSUPPORTED, "the GPU engine instantiates fp16 and fp32 (as GptNeoXOp.cc:56-105)");
        }
        FTCF_CHECK_ARG(cfg->int8_mode == 0 || cfg->int8_mode == 1, "int8_mode must be 0 or 1");
        if (cfg->dtype == FTCF_FP32 && cfg->int8_mode != 0) {
            throw Error(FTCF_ERR_UNSUPPORTED, "weight-only int8 needs half activations (CutlassFpAIntBGemmRunner<half, uint8_t>)");
        }
        const int tp = cfg->tensor_para_size;
        FTCF_CHECK_ARG(tp >= 1 && cfg->head_num % tp == 0 && cfg->inter_size % tp == 0 && cfg->vocab_size % tp == 0,
                       "head_num, inter_size and vocab_size must be divisible by tensor_para_size");
        const int L = cfg->num_layer;
        FTCF_CHECK_ARG(w->n_weights == 12 * L + 4, "weights must hold 12*L+4 tensors");
        FTCF_CHECK_ARG(L >= 1 && L <= 256, "num_layer must be in 1..256");
        FTCF_HIP_CHECK(hipSetDevice(cfg->device));
        FT_LOG_INFO(cfg->device, "GptNeoX engine on device %d: %d layers, %d heads x %d, inter %d, vocab %d, rotary %d, %s%s, "
                                 "tensor_para %d/%d, %s residual",
                    cfg->device, L, cfg->head_num, cfg->size_per_head, cfg->inter_size, cfg->vocab_size, cfg->rotary_embedding_dim,
                    cfg->dtype == FTCF_FP32 ? "fp32" : "fp16", cfg->int8_mode ? " + int8 weights" : "", cfg->tensor_para_rank, tp,
                    cfg->use_gptj_residual ? "parallel" : "sequential");
        auto e   = std::make_unique<ftcf_gptneox>();
        e->cfg   = *cfg;
        e->L     = L;
        e->dh    = cfg->size_per_head;
        e->H     = cfg->head_num * cfg->size_per_head;
        e->nhl   = cfg->head_num / tp;
        e->hl    = e->nhl * e->dh;
        e->il    = cfg->inter_size / tp;
        e->V     = cfg->vocab_size;
        e->vl    = e->V / tp;
        e->int8  = cfg->int8_mode == 1;
        e->fp32  = cfg->dtype == FTCF_FP32;
        e->user_stream = (hipStream_t)cfg->stream;
        FTCF_HIP_CHECK(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
        FTCF_HIP_CHECK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming));
        FTCF_HIP_CHECK(hipEventCreateWithFlags(&e->ev_user, hipEventDisableTiming));
        // weights still being uploaded / produced on the caller's stream happen-before the re-tiling below
        FTCF_HIP_CHECK(hipEventRecord(e->ev_user, e->user_stream));
        FTCF_HIP_CHECK(hipStreamWaitEvent(e->stream, e->ev_user, 0));
        if (e->fp32) {
            FTCF_CHECK_ARG(e->dh >= 32 && e->dh <= 256 && e->dh % 2 == 0, "size_per_head must be even, 32..256");
        }
        else {
            // the reference's list (DecoderSelfAttentionLayer.cc:280-282).  64 and 128 run every decode path; the other sizes
            // take the general path (one attention launch per layer): the persistent and the fused per-stage kernels, and the
            // paged attention of the batcher, keep their two tuned lane layouts
            FTCF_CHECK_ARG(mmha_head_size_supported(e->dh), "size_per_head must be one of 32, 48, 64, 80, 96, 128, 144, 160, 192, 224, 256");
            FTCF_CHECK_ARG(e->H % 64 == 0 && e->hl % 64 == 0 && e->il % 64 == 0,
                           "hidden, local hidden and local inter sizes must be multiples of 64");
        }
        if (e->int8) {
            FTCF_CHECK_ARG(w->n_int8_weights == 4 * L && w->n_scales == 4 * L, "int8 lists must hold 4*L tensors");
        }
        auto W = [&](int g, int l) { return w->weights[(size_t)g * L + l]; };
        e->layers.resize(L);
        const int H = e->H, hl = e->hl, il = e->il;
        for (int l = 0; l < L; l++) {
            LayerWeights& lw = e->layers[l];
            lw.ln1_b = (const f16*)W(0, l);
            lw.ln1_g = (const f16*)W(1, l);
            lw.qkv.bias = (const f16*)W(3, l);
            lw.attn_out.bias = (const f16*)W(5, l);
            lw.ffn1.bias = (const f16*)W(7, l);
            lw.ffn2.bias = (const f16*)W(9, l);
            lw.ln2_b = (const f16*)W(10, l);
            lw.ln2_g = (const f16*)W(11, l);
            FTCF_CHECK_ARG(lw.ln1_b && lw.ln1_g && lw.ln2_b && lw.ln2_g && lw.qkv.bias && lw.ffn1.bias && lw.ffn2.bias,
                           "missing layernorm / bias tensor");
            FTCF_CHECK_ARG(cfg->use_gptj_residual || lw.attn_out.bias,
                           "use_gptj_residual == 0 needs the attention output bias (weights[5*L + l])");
            if (e->int8) {
                lw.qkv.kernel = w->int8_weights[0 * L + l];
                lw.attn_out.kernel = w->int8_weights[1 * L + l];
                lw.ffn1.kernel = w->int8_weights[2 * L + l];
                lw.ffn2.kernel = w->int8_weights[3 * L + l];
                lw.qkv.scale = (const f16*)w->scales[0 * L + l];
                lw.attn_out.scale = (const f16*)w->scales[1 * L + l];
                lw.ffn1.scale = (const f16*)w->scales[2 * L + l];
                lw.ffn2.scale = (const f16*)w->scales[3 * L + l];
                FTCF_CHECK_ARG(lw.qkv.kernel && lw.attn_out.kernel && lw.ffn1.kernel && lw.ffn2.kernel && lw.qkv.scale
                                   && lw.attn_out.scale && lw.ffn1.scale && lw.ffn2.scale,
                               "missing int8 kernel / scale tensor");
            }
            else if (e->fp32) {
                // row-major [K, N] fp32 kernels are read in place
                lw.qkv.kernel = W(2, l);
                lw.attn_out.kernel = W(4, l);
                lw.ffn1.kernel = W(6, l);
                lw.ffn2.kernel = W(8, l);
                FTCF_CHECK_ARG(lw.qkv.kernel && lw.attn_out.kernel && lw.ffn1.kernel && lw.ffn2.kernel, "missing fp32 kernel tensor");
            }
            else {
                // re-tile the reference-layout [K, N] fp16 kernels once (the binding keeps the originals alive)
                struct {
                    int          g;
                    size_t       K, N;
                    DenseWeight* d;
                } items[4] = {{2, (size_t)H, (size_t)3 * hl, &lw.qkv},
                              {4, (size_t)hl, (size_t)H, &lw.attn_out},
                              {6, (size_t)H, (size_t)il, &lw.ffn1},
                              {8, (size_t)il, (size_t)H, &lw.ffn2}};
                // FTCF_FP16_RETILE_IN_PLACE=1: the caller's row-major kernels are OVERWRITTEN with the tiled image (through one
                // bounce buffer) instead of being kept next to a tiled copy -- the engine then holds one copy of the fp16 weights
                // (26 GB instead of 52 GB at 13B); the caller must not read those tensors as [K, N] matrices afterwards
                const bool in_place = getenv("FTCF_FP16_RETILE_IN_PLACE") && atoi(getenv("FTCF_FP16_RETILE_IN_PLACE")) != 0;
                for (auto& it : items) {
                    const void* src = W(it.g, l);
                    FTCF_CHECK_ARG(src != nullptr, "missing fp16 kernel tensor");
                    const size_t bytes = it.K * it.N * 2;
                    if (in_place) {
                        if (e->bounce_bytes < bytes) {
                            if (e->bounce) {
                                FTCF_HIP_CHECK(hipStreamSynchronize(e->stream));
                                (void)hipFree(e->bounce);
                            }
                            FTCF_HIP_CHECK(hipMalloc(&e->bounce, bytes));
                            e->bounce_bytes = bytes;
                        }
                        launch_fp16_rowmajor_to_tiled((const f16*)src, it.K, it.N, (f16*)e->bounce, e->stream);
                        FTCF_HIP_CHECK(hipMemcpyAsync(const_cast<void*>(src), e->bounce, bytes, hipMemcpyDeviceToDevice, e->stream));
                        it.d->kernel = src;
                        continue;
                    }
                    void* dst = nullptr;
                    FTCF_HIP_CHECK(hipMalloc(&dst, bytes));
                    e->owned.push_back(dst);
                    launch_fp16_rowmajor_to_tiled((const f16*)src, it.K, it.N, (f16*)dst, e->stream);
                    it.d->kernel = dst;
                }
            }
        }
        e->wte     = (const f16*)w->weights[12 * L];
        e->final_g = (const f16*)w->weights[12 * L + 1];  // GptNeoXOp.h:172-173: slot 12L+1 = gamma
        e->final_b = (const f16*)w->weights[12 * L + 2];
        e->lm_head = (const f16*)w->weights[12 * L + 3];
        FTCF_CHECK_ARG(e->wte && e->final_g && e->final_b && e->lm_head, "missing embedding / final layernorm / lm_head");
        e->k3_q = e->fp32 ? 1 : chunk_pick_q(e->H / 16, (e->hl + e->il) / (e->int8 ? TILE_K_I8 : TILE_K_F16));
        if (const char* m = getenv("FTCF_STAGE_MAX_ROWS")) {
            e->STAGE_MAX_ROWS = std::max(0, std::min(4, atoi(m)));
        }
        if (const char* m = getenv("FTCF_SMALLM_MAX_ROWS")) {
            e->SMALLM_MAX_ROWS = std::max(16, std::min(256, atoi(m)));
        }
        if (const char* m = getenv("FTCF_K3_Q")) {
            e->k3_q = atoi(m);
        }
        if (const char* m = getenv("FTCF_K1_WPG")) {
            e->k1_wpg = atoi(m);
        }
        {
            hipDeviceProp_t prop;
            FTCF_HIP_CHECK(hipGetDeviceProperties(&prop, cfg->device));
            e->num_cu = prop.multiProcessorCount;
        }
        if (const char* m = getenv("FTCF_PERSIST")) {
            e->persist = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_FAIL_ONCE")) {
            e->persist_fail_once = atoi(m);
        }
        if (const char* m = getenv("FTCF_TP_PERSIST")) {
            e->persist_tp = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_TS")) {
            e->ps_ts_file = m;
        }
        if (const char* m = getenv("FTCF_PERSIST_NB")) {
            e->persist_nb = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_CS1")) {
            e->persist_cs1 = atoi(m);
        }
        if (const char* m = getenv("FTCF_PERSIST_CS3")) {
            e->persist_cs3 = atoi(m);
        }
        e->decode_branches = cfg->tensor_para_size == 1 ? 1 : 0;
        if (const char* m = getenv("FTCF_DECODE_BRANCHES")) {
            e->decode_branches = atoi(m);
        }
        e->use_graph = cfg->use_hip_graph != 0;
        if (const char* m = getenv("FTCF_TP_GRAPH")) {
            e->tp_graph = atoi(m) != 0;
        }
        if (const char* m = getenv("FTCF_USE_GRAPH")) {
            e->use_graph = atoi(m) != 0;
        }
        FTCF_HIP_CHECK(hipHostMalloc((void**)&e->h_flags, 64, hipHostMallocDefault));
        e->h_flags[0] = e->h_flags[1] = 0;
        FTCF_HIP_CHECK(hipStreamSynchronize(e->stream));
        if (e->bounce) {
            (void)hipFree(e->bounce);
            e->bounce       = nullptr;
            e->bounce_bytes = 0;
        }
        *out = e.release();
    });
}

extern "C" int ftcf_gptneox_forward(ftcf_gptneox_t h, const ftcf_forward_args* args)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && args, "NULL argument");
        h->forward(*args);
    });
}

extern "C" int ftcf_gptneox_begin(ftcf_gptneox_t h, const ftcf_forward_args* args)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && args, "NULL argument");
        h->begin(*args);
    });
}
extern "C" int ftcf_gptneox_step(ftcf_gptneox_t h, int max_steps, int* steps_done)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        const int n = h->step(max_steps);
        if (steps_done) {
            *steps_done = n;
        }
    });
}
extern "C" int ftcf_gptneox_finish(ftcf_gptneox_t h)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        h->finish();
    });
}

extern "C" int ftcf_gptneox_get_stats(ftcf_gptneox_t h, ftcf_forward_stats* s)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h && s, "NULL argument");
        *s = h->stats;
        // dominant weight-streaming kernel kind
        int best = 0;
        for (int k = 1; k < KIND_COUNT; k++) {
            if (h->kind_ms[k] > h->kind_ms[best]) {
                best = k;
            }
        }
        s->gemv_kind     = best;
        s->gemv_ms_sum   = (float)h->kind_ms[best];
        s->gemv_launches = h->kind_n[best];
        s->gemv_bytes    = h->kind_bytes[best];
    });
}

extern "C" int ftcf_gptneox_set_profiling(ftcf_gptneox_t h, int enabled)
{
    return guarded([&] {
        FTCF_CHECK_ARG(h, "NULL argument");
        h->profiling = enabled != 0;
        for (int k = 0; k < KIND_COUNT; k++) {
            h->kind_ms[k] = h->kind_bytes[k] = 0;
            h->kind_n[k]                     = 0;
        }
    });
}

extern "C" int ftcf_gptneox_destroy(ftcf_gptneox_t h)
{
    return guarded([&] { delete h; });
}

// ---------------------------------------------------------------------------------------------------------------
// Continuous batching over a paged K/V cache (no counterpart in the reference, whose serving layer -- the
// Triton backend -- allocates the cache per request, GptNeoX.cc:84-156).
//
// A batcher borrows an engine (weights, streams, kernels) and owns
//   * a K/V POOL of fixed-size pages, [L][page][head][P tokens][dh] fp16, shared by all sequences, with a free list;
//   * `max_batch` SLOTS: page table, length, last token, sampling parameters of the sequence living there;
//   * a queue of waiting requests.
// One iteration (ftcf_batcher_step) = ADMIT waiting requests into free slots while their pages (prompt + max_new_tokens,
// reserved up front: a running sequence never has to be preempted) are available, then ONE decode step for all running
// slots.  Admission runs the prompt through the engine's own context path (ftcf_gptneox_forward with output_len 1: prefill +
// first token sampled by the engine's dynamic decode) and scatters the prompt's K/V into the slot's pages; the decode
// step is the general layer sequence of the engine (§4a: dual LayerNorm, burst / tiled GEMMs, fused residual) with the
// attention replaced by k_mmha_paged, the LM head, and the engine's sampling kernels on per-slot arrays.  A sequence leaves
// when it emits end_id or reaches max_new_tokens; its pages return to the free list at once.
// Scope: parallel-residual models, any tensor_para_size (round 4: one batcher per rank), fp16 / int8 engines, beam_width 1, top-k /
// top-p / temperature sampling (no repetition penalty, stop words or callbacks).
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_batcher_embed(f16* out, const f16* table, const int* tok, int H)
{
    const int  id  = tok[blockIdx.x];
    const f16* src = table + (size_t)id * H;
    f16*       dst = out + (size_t)blockIdx.x * H;
    for (int i = threadIdx.x * 8; i < H; i += blockDim.x * 8) {
        *reinterpret_cast<f16x8*>(dst + i) = *reinterpret_cast<const f16x8*>(src + i);
    }
}
// d_tok[b] = the token the sampling kernels just wrote into slot b's history (time-major [max_len][B], position len[b])
__global__ void k_batcher_last_token(int* tok, const int* hist, const int* len, int B)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        tok[b] = hist[(size_t)len[b] * B + b];
    }
}
__global__ void k_batcher_tick(DecodeState* gemm_state)
{
    gemm_state->step = (gemm_state->step + 1) & 0x7ffff;  // part of the burst GEMMs' granule tags (19 bits)
}

struct ftcf_batcher {
    struct Request {
        long             id;
        std::vector<int> prompt;
        int              max_new, top_k;
        float            top_p, temperature, repetition_penalty = 1.f;
        uint64_t         seed;
        std::vector<std::vector<int>> stop;  // stop sequences (token ids)
    };
    struct Slot {
        bool             active = false;
        long             id = 0;
        int              len = 0, generated = 0, max_new = 0;
        std::vector<int> pages;
        // the request's own token history (prompt + generated) and stop sequences: the stop criterion
        // (stop_criteria_kernels.cu:24-83: finished AFTER the sequence has been emitted) is the scheduler's, on the host,
        // which sees every token anyway
        std::vector<int>              hist;
        std::vector<std::vector<int>> stop;
        float                         repetition_penalty = 1.f;
    };
    ftcf_gptneox* e = nullptr;
    int           max_batch = 0, P = 0, num_pages = 0, max_pages = 0, max_len = 0;
    size_t        pool_layer_elems = 0;
    // device
    f16 *kpool = nullptr, *vpool = nullptr;
    f16 *x = nullptr, *nrm = nullptr, *nrm2 = nullptr, *qkv = nullptr, *ctx = nullptr, *att = nullptr, *mid = nullptr, *ffn = nullptr;
    float*    logits = nullptr;
    float*    gather = nullptr;  // tensor parallel: [TP][max_batch][V / TP] slices of the LM head
    int *     d_pt = nullptr, *d_len = nullptr, *d_tok = nullptr, *d_topk = nullptr, *d_zero = nullptr, *d_prompt = nullptr, *d_plen = nullptr,
        *d_pout = nullptr, *d_pseq = nullptr, *d_pages_tmp = nullptr;
    uint8_t*     d_fin = nullptr;
    float *      d_ptopk = nullptr, *d_ptopp = nullptr, *d_temp = nullptr, *d_cum = nullptr, *d_rep = nullptr;
    int*         d_hist = nullptr;  // [max_len + 1][max_batch] time-major token history of the slots (repetition penalty)
    int *        d_sw = nullptr;    // admission: stop words of the ragged batch, the reference's [n][2][Lw] layout
    uint64_t *   d_seed = nullptr, *d_draws = nullptr;
    DecodeState *d_state = nullptr, *d_gstate = nullptr;
    void*        samp_ws = nullptr;
    float*       smallm_ws = nullptr;
    float*       tiled_ws  = nullptr;  // split-K workspace of the tiled GEMM (decode steps of 17..320 rows)
    // chunked admission: with slots running, a prompt longer than this is prefilled alone, `prefill_chunk` tokens at a time, one
    // decode step of the running slots between two chunks (FTCF_BATCHER_PREFILL_CHUNK; 0 = whole prompts)
    int prefill_chunk = 512;
    size_t       smallm_partial = 0;
    unsigned     smallm_seq = 0;
    long         gemm_steps = 0;
    std::vector<void*> owned;
    // host
    std::vector<Slot>   slots;
    std::deque<Request> waiting;
    std::vector<int>    free_pages;
    long                next_id = 1;
    int                 max_prompt = 0;

    template<typename T>
    T* dmalloc(size_t n, bool zero = true)
    {
        void* p = nullptr;
        FTCF_HIP_CHECK(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
        if (zero) {
            FTCF_HIP_CHECK(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
        }
        owned.push_back(p);
        return reinterpret_cast<T*>(p);
    }
    ~ftcf_batcher()
    {
        for (void* p : owned) {
            (void)hipFree(p);
        }
    }

    void init(ftcf_gptneox* eng, int mb, int page_tokens, int pages, int max_seq_len)
    {
        e = eng;
        FTCF_CHECK_ARG(!e->fp32 && e->cfg.use_gptj_residual && (e->dh == 64 || e->dh == 128),
                       "the batcher serves fp16 / int8 engines with parallel residual and size_per_head 64 / 128");
        FTCF_CHECK_ARG(mb >= 1 && mb <= 64 && page_tokens >= 8 && pages >= 1 && max_seq_len >= 2, "bad batcher geometry");
        FTCF_HIP_CHECK(hipSetDevice(e->cfg.device));
        if (const char* c = getenv("FTCF_BATCHER_PREFILL_CHUNK")) {
            prefill_chunk = std::max(0, atoi(c));
        }
        max_batch = mb;
        P         = page_tokens;
        num_pages = pages;
        max_pages = (max_seq_len + P - 1) / P;
        max_len   = max_pages * P;
        max_prompt = max_seq_len - 1;
        FTCF_CHECK_ARG(mmha_paged_smem_bytes(e->dh, max_pages, max_len) <= 64 * 1024, "max_seq_len too large for the paged attention");
        const int    H = e->H, hl = e->hl, il = e->il, L = e->L, V = e->V;
        const size_t B = (size_t)max_batch;
        pool_layer_elems = (size_t)num_pages * e->nhl * P * e->dh;
        kpool = dmalloc<f16>((size_t)L * pool_layer_elems, false);
        vpool = dmalloc<f16>((size_t)L * pool_layer_elems, false);
        x = dmalloc<f16>(B * H);
        nrm = dmalloc<f16>(B * H);
        nrm2 = dmalloc<f16>(B * H);
        qkv = dmalloc<f16>(B * 3 * hl);
        ctx = dmalloc<f16>(B * hl);
        att = dmalloc<f16>(B * H);
        mid = dmalloc<f16>(B * il);
        ffn = dmalloc<f16>(B * H);
        logits = dmalloc<float>(B * V);
        // tensor parallel (every rank runs its own batcher over its shard, fed the same requests in the same order: the
        // schedulers take identical decisions, the decode step's collectives are the engine's): the LM head's [TP][B][V/TP] slices
        gather = e->cfg.tensor_para_size > 1 ? dmalloc<float>(B * V) : nullptr;
        d_pt = dmalloc<int>(B * max_pages);
        d_len = dmalloc<int>(B);
        d_tok = dmalloc<int>(B);
        d_topk = dmalloc<int>(B);
        d_zero = dmalloc<int>(B);
        d_fin = dmalloc<uint8_t>(B);
        d_ptopk = dmalloc<float>(B);
        d_ptopp = dmalloc<float>(B);
        d_temp = dmalloc<float>(B);
        d_cum = dmalloc<float>(B);
        d_rep = dmalloc<float>(B);
        d_hist = dmalloc<int>((size_t)(max_len + 2) * B);
        d_sw = dmalloc<int>(B * 2 * STOP_LW);
        d_seed = dmalloc<uint64_t>(B);
        d_draws = dmalloc<uint64_t>(B);
        d_state = dmalloc<DecodeState>(1);
        d_gstate = dmalloc<DecodeState>(1);
        d_prompt = dmalloc<int>(B * max_seq_len);
        d_plen = dmalloc<int>(B);
        d_pout = dmalloc<int>(B * (max_seq_len + 1));
        d_pseq = dmalloc<int>(B);
        d_pages_tmp = dmalloc<int>(max_pages);
        samp_ws = dmalloc<char>(sampling_workspace_bytes(max_batch, V), false);
        if (max_batch > 4 && max_batch <= e->SMALLM_MAX_ROWS) {
            const bool i8 = e->int8;
            const int  bc = std::min(max_batch, 16);  // 16 rows per launch
            smallm_partial = gemm_smallm_workspace_bytes(bc, 3 * hl, H, i8) + gemm_smallm_workspace_bytes(bc, il, H, i8)
                             + gemm_smallm_workspace_bytes(bc, H, hl, i8) + gemm_smallm_workspace_bytes(bc, H, il, i8);
            smallm_ws = dmalloc<float>((smallm_partial + gemm_smallm_ticket_bytes()) / 4 + 1);
        }
        if (max_batch > 16 && max_batch <= 320) {
            tiled_ws = dmalloc<float>(gemm_tiled_workspace_bytes() / 4);
        }
        std::vector<uint8_t> fin(max_batch, 1);
        FTCF_HIP_CHECK(hipMemcpy(d_fin, fin.data(), max_batch, hipMemcpyHostToDevice));
        slots.assign(max_batch, Slot{});
        free_pages.resize(num_pages);
        for (int i = 0; i < num_pages; i++) {
            free_pages[i] = num_pages - 1 - i;
        }
    }

    static constexpr int STOP_LW = 64;  // total stop-word tokens per request (the [2][Lw] word list of the admission)
    long submit(const int* ids, int n, int max_new, int top_k, float top_p, float temperature, uint64_t seed,
                float repetition_penalty = 1.f, const int* stop_words = nullptr, int stop_len = 0)
    {
        FTCF_CHECK_ARG(repetition_penalty > 0.f, "repetition_penalty must be positive");
        FTCF_CHECK_ARG(stop_len >= 0 && stop_len <= STOP_LW && (stop_len == 0 || stop_words), "bad stop word list");
        // (the decode step stages total_len = max_len + 2 history entries: the same bound as launch_dynamic_decode's)
        FTCF_CHECK_ARG(((size_t)max_len + 2) * 8 <= 60 * 1024 || repetition_penalty == 1.f,
                       "max_seq_len too large for the repetition-penalty staging buffer");
        FTCF_CHECK_ARG(ids && n >= 1 && max_new >= 1, "empty prompt or max_new_tokens < 1");
        FTCF_CHECK_ARG(n + max_new <= max_len && n <= max_prompt, "prompt + max_new_tokens exceed the batcher's max_seq_len");
        FTCF_CHECK_ARG((n + max_new + P - 1) / P <= num_pages, "the request needs more pages than the pool has");
        FTCF_CHECK_ARG(top_k >= 0 && top_k <= 1024 && top_p >= 0.f && top_p <= 1.f && temperature > 0.f, "bad sampling parameters");
        for (int i = 0; i < n; i++) {
            FTCF_CHECK_ARG(ids[i] >= 0 && ids[i] < e->V, "token id out of range");
        }
        Request r;
        r.id = next_id++;
        r.prompt.assign(ids, ids + n);
        r.max_new = max_new;
        r.top_k = top_k;
        r.top_p = top_p;
        r.temperature = temperature;
        r.seed = seed;
        r.repetition_penalty = repetition_penalty;
        // to_word_list_format (codefuse_example.py:26-53): [0][..] flat ids, [1][..] cumulative end offsets, -1 padded
        for (int i = 0, start = 0; i < stop_len; i++) {
            const int end = stop_words[stop_len + i];
            if (end < 0) {
                break;
            }
            FTCF_CHECK_ARG(end > start && end <= stop_len, "bad stop word offsets");
            r.stop.emplace_back(stop_words + start, stop_words + end);
            start = end;
        }
        waiting.push_back(std::move(r));
        return waiting.back().id;
    }

    struct Event {
        long id;
        int  token, finished;
    };
    std::vector<Event>* hook_ev = nullptr;  // where the decode steps inside a chunked admission put their events
    std::deque<Event>   outbox;             // events of an iteration that did not fit the caller's arrays
    ftcf_token_callback_fn on_token = nullptr;  // called for every event the moment it exists (inside step())
    void*                  on_token_user = nullptr;
    void emit(std::vector<Event>& ev, const Event& x)
    {
        ev.push_back(x);
        if (on_token) {
            on_token(on_token_user, x.id, x.token, x.finished);
        }
    }

    // the first tokens of an admission reach the callback only when the admission is known to have succeeded: a failed one is
    // rolled back and retried, and a streaming consumer must not see its first tokens twice
    void fire(const std::vector<Event>& ev, const size_t from)
    {
        if (on_token) {
            for (size_t i = from; i < ev.size(); i++) {
                on_token(on_token_user, ev[i].id, ev[i].token, ev[i].finished);
            }
        }
    }

    // stop_criteria_kernels.cu:24-83: does the history END with one of the request's stop sequences?
    static bool hits_stop_word(const Slot& s)
    {
        for (const auto& wd : s.stop) {
            if (!wd.empty() && s.hist.size() >= wd.size() && std::equal(wd.begin(), wd.end(), s.hist.end() - wd.size())) {
                return true;
            }
        }
        return false;
    }
    void release(Slot& s)
    {
        for (int pg : s.pages) {
            free_pages.push_back(pg);
        }
        s.pages.clear();
        s.active = false;
    }

    // prompts -> ONE ragged batch through the engine's context path (+ first tokens) -> pages of their slots
    void admit(const std::vector<int>& sis, const std::vector<Request>& rs, std::vector<Event>& ev)
    {
        Range        rg("ftcf.batcher.admit");
        hipStream_t  st = e->stream;
        const int    n  = (int)sis.size();
        int          S  = 0;
        for (const Request& r : rs) {
            S = std::max(S, (int)r.prompt.size());
        }
        std::vector<int>      ids((size_t)n * S, e->cfg.end_id), lens(n), topk(n);
        std::vector<float>    topp(n), temp(n), rep(n);
        bool                  any_stop = false;
        std::vector<int>      sw((size_t)n * 2 * STOP_LW, 0);
        std::vector<uint64_t> seed(n);
        for (int i = 0; i < n; i++) {
            const Request& r = rs[i];
            std::copy(r.prompt.begin(), r.prompt.end(), ids.begin() + (size_t)i * S);
            lens[i] = (int)r.prompt.size();
            topk[i] = r.top_k;
            topp[i] = r.top_p;
            temp[i] = r.temperature;
            rep[i]  = r.repetition_penalty;
            seed[i] = r.seed;
            {  // the request's stop sequences in the reference's word-list layout (the engine samples the FIRST token)
                int* ids_row = sw.data() + (size_t)i * 2 * STOP_LW;
                int* off_row = ids_row + STOP_LW;
                std::fill(off_row, off_row + STOP_LW, -1);
                int pos = 0, k = 0;
                for (const auto& wd : r.stop) {
                    std::copy(wd.begin(), wd.end(), ids_row + pos);
                    pos += (int)wd.size();
                    off_row[k++] = pos;
                    any_stop = true;
                }
            }
            Slot&     s = slots[sis[i]];
            const int need = (lens[i] + r.max_new + P - 1) / P;
            s.pages.clear();
            for (int k = 0; k < need; k++) {
                s.pages.push_back(free_pages.back());
                free_pages.pop_back();
            }
        }
        FTCF_HIP_CHECK(hipMemcpy(d_prompt, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
        FTCF_HIP_CHECK(hipMemcpy(d_plen, lens.data(), (size_t)n * 4, hipMemcpyHostToDevice));
        ftcf_forward_args a{};
        a.input_ids = d_prompt;
        a.input_lengths = d_plen;
        a.batch_size = n;
        a.max_input_len = S;
        a.output_len = 1;
        a.beam_width = 1;
        a.top_k = topk.data();
        a.n_top_k = n;
        a.top_p = topp.data();
        a.n_top_p = n;
        a.temperature = temp.data();
        a.n_temperature = n;
        a.repetition_penalty = rep.data();
        a.n_repetition_penalty = n;
        if (any_stop) {
            FTCF_HIP_CHECK(hipMemcpy(d_sw, sw.data(), sw.size() * 4, hipMemcpyHostToDevice));
            a.stop_words_list = d_sw;
            a.stop_words_len = STOP_LW;
        }
        a.random_seed = seed.data();
        a.n_random_seed = n;
        a.output_ids = d_pout;
        a.sequence_lengths = d_pseq;
        e->forward(a);  // host synchronous: K/V of row i, positions [0, len_i), are in the engine's cache [L][n][nh][S + 1][dh]
        std::vector<int> out((size_t)n * (S + 1));
        FTCF_HIP_CHECK(hipMemcpy(out.data(), d_pout, out.size() * 4, hipMemcpyDeviceToHost));
        const size_t row_kv = (size_t)e->nhl * (S + 1) * e->dh;  // one row of one layer of the engine's cache
        for (int i = 0; i < n; i++) {
            const Request& r  = rs[i];
            const int      si = sis[i], len = lens[i];
            Slot&          s  = slots[si];
            // the engine's output rows are compacted (prompt, then the generated tokens: invokeGatherTree removes the padding)
            const int        first = out[(size_t)i * (S + 1) + len];
            std::vector<int> row(max_pages, 0);
            std::copy(s.pages.begin(), s.pages.end(), row.begin());
            FTCF_HIP_CHECK(hipMemcpyAsync(d_pt + (size_t)si * max_pages, row.data(), (size_t)max_pages * 4, hipMemcpyHostToDevice, st));
            launch_scatter_kv_to_pages(e->k_cache + (size_t)i * row_kv, e->v_cache + (size_t)i * row_kv, kpool, vpool,
                                       d_pt + (size_t)si * max_pages, e->L, e->nhl, e->dh, S + 1, len, P, pool_layer_elems, st,
                                       (size_t)n * row_kv);
            // per-slot state of the decode steps
            const int      keff = (r.top_k == 0 && r.top_p == 0.f) ? 1 : std::min(r.top_k, 1024);  // BaseSamplingLayer: (0, 0) = greedy
            const float    ptk = (r.top_p == 0.f) ? 1.f : r.top_p;
            const uint8_t  zero8 = 0;
            const uint64_t one = 1;
            const float    zf = 0.f;
            FTCF_HIP_CHECK(hipMemcpyAsync(d_len + si, &len, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_tok + si, &first, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_topk + si, &keff, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_ptopk + si, &ptk, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_ptopp + si, &r.top_p, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_temp + si, &r.temperature, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_rep + si, &r.repetition_penalty, 4, hipMemcpyHostToDevice, st));
            // the slot's token history, time-major column si: prompt, then the first token
            s.hist.assign(r.prompt.begin(), r.prompt.end());
            s.hist.push_back(first);
            s.stop = r.stop;
            s.repetition_penalty = r.repetition_penalty;
            FTCF_HIP_CHECK(hipMemcpy2DAsync(d_hist + si, (size_t)max_batch * 4, s.hist.data(), 4, 4, s.hist.size(),
                                            hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_seed + si, &r.seed, 8, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_draws + si, &one, 8, hipMemcpyHostToDevice, st));  // draw 0 went to the first token
            FTCF_HIP_CHECK(hipMemcpyAsync(d_cum + si, &zf, 4, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipMemcpyAsync(d_fin + si, &zero8, 1, hipMemcpyHostToDevice, st));
            FTCF_HIP_CHECK(hipStreamSynchronize(st));  // the host temporaries above die here
            s.active = true;
            s.id = r.id;
            s.len = len;
            s.generated = 1;
            s.max_new = r.max_new;
            const int done = (first == e->cfg.end_id || s.generated >= s.max_new || hits_stop_word(s)) ? 1 : 0;
            ev.push_back(Event{r.id, first, done});  // (the token callback fires when the admission has succeeded: step())
            if (done) {
                const uint8_t one8 = 1;
                FTCF_HIP_CHECK(hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                release(s);
            }
        }
    }

    // one token for every running slot
    void decode(std::vector<Event>& ev)
    {
        Range                  rg("ftcf.batcher.decode");
        hipStream_t            st = e->stream;
        const int              B = max_batch, H = e->H, hl = e->hl, il = e->il, L = e->L, V = e->V;
        const bool             int8 = e->int8;
        const bool             dual = residual_dual_ln_supported(H);
        const int              tp = e->cfg.tensor_para_size;
        const bool             tp1 = tp == 1;  // (tensor parallel: the layer ends with residual + all-reduce, GptNeoXDecoder.cc:357-359)
        hipLaunchKernelGGL(k_batcher_embed, dim3(B), dim3(256), 0, st, x, e->wte, d_tok, H);
        if ((gemm_steps++ & 0x3ffff) == 0 && smallm_ws) {  // the tag space of the burst GEMMs wraps: start it clean
            FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), st));
        }
        hipLaunchKernelGGL(k_batcher_tick, dim3(1), dim3(1), 0, st, d_gstate);
        for (int l = 0; l < L; l++) {
            const LayerWeights& w = e->layers[l];
            if (!dual) {
                launch_layernorm(x, w.ln1_g, w.ln1_b, nrm, B, H, 1e-5f, true, st);
                launch_layernorm(x, w.ln2_g, w.ln2_b, nrm2, B, H, 1e-5f, true, st);
            }
            else if (l == 0 || !tp1) {
                launch_residual_dual_ln(x, nullptr, nullptr, nullptr, 1, 0, w.ln1_g, w.ln1_b, w.ln2_g, w.ln2_b, nrm, nrm2, B, H,
                                        1e-5f, st);
            }
            MmhaPagedParams mp{};
            mp.qkv = qkv;
            mp.qkv_bias = w.qkv.bias;
            mp.kpool = kpool + (size_t)l * pool_layer_elems;
            mp.vpool = vpool + (size_t)l * pool_layer_elems;
            mp.page_table = d_pt;
            mp.len = d_len;
            mp.finished = d_fin;
            mp.B = B;
            mp.nh = e->nhl;
            mp.dh = e->dh;
            mp.rot = e->cfg.rotary_embedding_dim;
            mp.P = P;
            mp.max_pages = max_pages;
            mp.ctx = ctx;
            if (smallm_ws && e->decode_branches && e->side) {
                // the attention branch and the FFN branch on two streams, as the engine's batched decode
                const int    bc = std::min(B, 16);
                const size_t o_qkv = 0, o_f1 = o_qkv + gemm_smallm_workspace_bytes(bc, 3 * hl, H, int8),
                             o_out = o_f1 + gemm_smallm_workspace_bytes(bc, il, H, int8),
                             o_f2  = o_out + gemm_smallm_workspace_bytes(bc, H, hl, int8);
                auto one = [&](const SmallmDesc& d0, size_t off, hipStream_t s2) {
                    for (int r0 = 0; r0 < B; r0 += 16) {
                        SmallmDesc d = d0;
                        d.A          = d0.A + (size_t)r0 * d0.k;
                        d.C          = d0.C + (size_t)r0 * d0.n;
                        launch_gemm_smallm_group(&d, 1, smallm_ws, smallm_partial, std::min(16, B - r0), int8, s2, &d_gstate->step,
                                                 &smallm_seq, off);
                    }
                };
                const size_t offs[4] = {o_qkv, o_f1, o_out, o_f2};
                GemmFn burst = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int, int n, int k,
                                   hipStream_t s2, int slot) { one(SmallmDesc{A, dw.kernel, dw.scale, bias, act, C, n, k}, offs[slot], s2); };
                FTCF_HIP_CHECK(hipEventRecord(e->ev_fork, st));
                FTCF_HIP_CHECK(hipStreamWaitEvent(e->side, e->ev_fork, 0));
                DecoderSelfAttentionLayer{burst, H, hl}.forward_paged(nrm, qkv, ctx, att, w, mp, max_len, B, st);
                FfnLayer{burst, H, il}.forward(nrm2, mid, ffn, w, B, e->side);
                FTCF_HIP_CHECK(hipEventRecord(e->ev_join, e->side));
                FTCF_HIP_CHECK(hipStreamWaitEvent(st, e->ev_join, 0));
            }
            else if (smallm_ws && B <= 16) {
                const SmallmDesc p1[2] = {{nrm, w.qkv.kernel, w.qkv.scale, nullptr, 0, qkv, 3 * hl, H},
                                          {nrm2, w.ffn1.kernel, w.ffn1.scale, w.ffn1.bias, 1, mid, il, H}};
                launch_gemm_smallm_group(p1, 2, smallm_ws, smallm_partial, B, int8, st, &d_gstate->step, &smallm_seq);
                launch_mmha_paged(mp, max_len, st);
                const SmallmDesc p3[2] = {{ctx, w.attn_out.kernel, w.attn_out.scale, nullptr, 0, att, H, hl},
                                          {mid, w.ffn2.kernel, w.ffn2.scale, nullptr, 0, ffn, H, il}};
                launch_gemm_smallm_group(p3, 2, smallm_ws, smallm_partial, B, int8, st, &d_gstate->step, &smallm_seq);
            }
            else {
                GemmFn plain = [&](const f16* A, const DenseWeight& dw, const f16* bias, int act, f16* C, int m, int n, int k,
                                   hipStream_t s2, int) {
                    gemm_dispatch(A, dw.kernel, dw.scale, bias, act, C, m, n, k, int8, s2, nullptr, 0, e->num_cu, nullptr, nullptr,
                                  s2 == st ? tiled_ws : nullptr);
                };
                DecoderSelfAttentionLayer{plain, H, hl}.forward_paged(nrm, qkv, ctx, att, w, mp, max_len, B, st);
                FfnLayer{plain, H, il}.forward(nrm2, mid, ffn, w, B, st);
            }
            // (every slot's hidden state is recomputed from its token each step: the residual never aliases across steps,
            // so the fp32-sum variant of the context decoder applies to all layers)
            if (dual && tp1) {
                const LayerWeights* nx = l + 1 < L ? &e->layers[l + 1] : nullptr;
                launch_residual_dual_ln(x, ffn, att, w.ffn2.bias, 1, (l > 0 && l < L - 1) ? 1 : 0, nx ? nx->ln1_g : nullptr,
                                        nx ? nx->ln1_b : nullptr, nx ? nx->ln2_g : nullptr, nx ? nx->ln2_b : nullptr, nrm, nrm2, B, H,
                                        1e-5f, st);
            }
            else {
                launch_add_bias_attn_ffn_residual(x, ffn, att, x, w.ffn2.bias, B, H, tp, (l > 0 && l < L - 1) ? 1 : 0, true, st);
                e->allreduce(x, (size_t)B * H, st);
            }
        }
        {
            // LM head; tensor parallel: rank r computes rows [r V/TP, (r+1) V/TP) of the replicated lm_head into its slice of
            // `gather`, all-gather + transpose (GptNeoX.cc:888-925, as the engine's own step)
            const int    rows = tp1 ? V : e->vl;
            const f16*   Wr   = tp1 ? e->lm_head : e->lm_head + (size_t)e->cfg.tensor_para_rank * e->vl * H;
            float*       out  = tp1 ? logits : gather + (size_t)e->cfg.tensor_para_rank * B * e->vl;
            if (B <= 4) {
                launch_lm_head(x, Wr, out, B, rows, H, rows, st, e->final_g, e->final_b, 1e-5f);
            }
            else {
                launch_layernorm(x, e->final_g, e->final_b, nrm, B, H, 1e-5f, true, st);
                lm_head_dispatch(nrm, Wr, out, B, rows, H, rows, st);
            }
            if (!tp1) {
                e->allgather_logits(gather, logits, B, st);
            }
        }
        SamplingParams sp{};
        sp.logits = logits;
        sp.B = B;
        sp.V = V;
        sp.max_input_len = 0;
        sp.end_id = e->cfg.end_id;
        sp.input_lengths = d_zero;
        sp.top_k = d_topk;
        sp.top_p_topk = d_ptopk;
        sp.top_p_topp = d_ptopp;
        sp.temperature = d_temp;
        sp.random_seed = d_seed;
        sp.draw_counter = d_draws;
        sp.apply_temperature = host_any_temperature ? 1 : 0;
        sp.apply_repetition = host_any_repetition ? 1 : 0;  // (BaseSamplingLayer.cc:283-313: skipped when every row has 1.0)
        sp.repetition_penalty = d_rep;
        sp.return_cum_log_probs = 1;
        // every slot has its own step: its history is column b of the time-major d_hist, positions [0, len[b]]; the sampled
        // token goes to position len[b] + 1 (the penalty of sampling_penalty_kernels.cu:367-425 reads the whole history)
        sp.output_ids = d_hist;
        sp.row_len = d_len;
        sp.total_len = max_len + 2;
        sp.finished = d_fin;
        sp.seq_len = d_len;     // + 1 per sampled token: the slot's length
        sp.cum_log_probs = d_cum;
        sp.pad_count = d_zero;
        sp.state = d_state;     // step stays 0
        sp.ws = samp_ws;
        sp.max_top_k = host_max_top_k;
        sp.any_top_p = host_any_top_p;
        DynamicDecodeLayer{}.forward(sp, st, false);  // (the stop / length criteria are the scheduler's: no finish step)
        hipLaunchKernelGGL(k_batcher_last_token, dim3(1), dim3(64), 0, st, d_tok, d_hist, d_len, B);
        std::vector<int>     tok(B);
        std::vector<uint8_t> fin(B);
        int                  gemm_err = 0;
        FTCF_HIP_CHECK(hipMemcpyAsync(tok.data(), d_tok, (size_t)B * 4, hipMemcpyDeviceToHost, st));
        FTCF_HIP_CHECK(hipMemcpyAsync(fin.data(), d_fin, (size_t)B, hipMemcpyDeviceToHost, st));
        if (smallm_ws) {  // sticky flag of the burst GEMMs' in-launch split-K reduction (the engine's finish() reads its own)
            FTCF_HIP_CHECK(hipMemcpyAsync(&gemm_err, reinterpret_cast<char*>(smallm_ws) + smallm_partial, sizeof(int),
                                          hipMemcpyDeviceToHost, st));
        }
        FTCF_HIP_CHECK(hipStreamSynchronize(st));
        if (gemm_err != 0) {
            // the tokens of this step are not to be trusted: nothing is reported, the slots keep their state (lengths and
            // draw counters advanced on the device: the requests cannot be resumed exactly), the flag is cleared for the caller's
            // next attempt
            FTCF_HIP_CHECK(hipMemsetAsync(smallm_ws, 0, smallm_partial + gemm_smallm_ticket_bytes(), st));
            FTCF_HIP_CHECK(hipStreamSynchronize(st));
            throw Error(-2, "batcher decode: a split-K reducer of the batched GEMM gave up waiting for its sibling workgroups");
        }
        for (int si = 0; si < B; si++) {
            Slot& s = slots[si];
            if (!s.active) {
                continue;
            }
            s.len += 1;
            s.generated += 1;
            s.hist.push_back(tok[si]);
            const int done = (fin[si] || s.generated >= s.max_new || hits_stop_word(s)) ? 1 : 0;
            emit(ev, Event{s.id, tok[si], done});
            if (done) {
                if (!fin[si]) {
                    const uint8_t one8 = 1;
                    FTCF_HIP_CHECK(hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                }
                release(s);
            }
        }
    }

    int  host_max_top_k = 1, host_any_top_p = 0;
    bool host_any_temperature = false, host_any_repetition = false;
    std::vector<int>   slot_topk;
    std::vector<float> slot_temp;

    void step(std::vector<Event>& ev)
    {
        FTCF_HIP_CHECK(hipSetDevice(e->cfg.device));
        if (slot_topk.empty()) {
            slot_topk.assign(max_batch, 1);
            slot_temp.assign(max_batch, 1.f);
        }
        // the running slots first (a request admitted in this iteration has its first token already)
        bool any = false;
        for (const Slot& s : slots) {
            any |= s.active;
        }
        if (any) {
            decode(ev);
        }
        // admissions: as many of the queue's head requests as there are free slots and pages, prefilled as ONE ragged batch
        std::vector<int>     sis;
        std::vector<Request> rs;
        int                  pages_left = (int)free_pages.size();
        for (int si = 0; si < max_batch && !waiting.empty(); si++) {
            if (slots[si].active) {
                continue;
            }
            const Request& r    = waiting.front();
            const int      need = ((int)r.prompt.size() + r.max_new + P - 1) / P;
            if (pages_left < need) {
                break;  // FIFO: nobody overtakes the head of the queue
            }
            pages_left -= need;
            const int keff = (r.top_k == 0 && r.top_p == 0.f) ? 1 : std::min(r.top_k, 1024);
            slot_topk[si]  = keff;
            slot_temp[si]  = r.temperature;
            sis.push_back(si);
            rs.push_back(std::move(waiting.front()));
            waiting.pop_front();
        }
        bool any_long = false;
        for (const Request& r : rs) {
            any_long |= prefill_chunk > 0 && (int)r.prompt.size() > prefill_chunk;
        }
        if (!sis.empty() && any && any_long) {
            // slots are running and a long prompt arrives: one request at a time, its prompt phase in chunks with a decode step of
            // the running slots after every chunk (events of those steps are final whatever happens to the admission)
            while (!sis.empty()) {
                const std::vector<int>     one_si{sis.front()};
                const std::vector<Request> one_r{rs.front()};
                std::vector<Event>         own, between;
                bool                       running = false;
                for (const Slot& s : slots) {
                    running |= s.active;
                }
                try {
                    if (running) {
                        hook_ev          = &between;
                        e->prefill_chunk = prefill_chunk;
                        e->prefill_hook  = [this] {
                            bool live = false;
                            for (const Slot& s : slots) {
                                live |= s.active;
                            }
                            if (live) {
                                refresh_host_flags();
                                decode(*hook_ev);
                            }
                        };
                    }
                    admit(one_si, one_r, own);
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                }
                catch (...) {
                    e->prefill_hook = nullptr;
                    hook_ev         = nullptr;
                    (void)hipDeviceSynchronize();
                    (void)hipGetLastError();
                    ev.insert(ev.end(), between.begin(), between.end());
                    for (const int si : sis) {  // this request and the ones not yet admitted go back to the queue's head
                        release(slots[si]);
                        const uint8_t one8 = 1;
                        (void)hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice);
                    }
                    for (size_t i = rs.size(); i-- > 0;) {
                        waiting.push_front(std::move(rs[i]));
                    }
                    throw;
                }
                ev.insert(ev.end(), between.begin(), between.end());
                ev.insert(ev.end(), own.begin(), own.end());
                fire(own, 0);
                sis.erase(sis.begin());
                rs.erase(rs.begin());
            }
        }
        if (!sis.empty()) {
            const size_t ev0 = ev.size();
            try {
                admit(sis, rs, ev);
            }
            catch (...) {
                // the whole admission is rolled back: pages to the pool, slots free, the requests back at the head of the
                // queue in their order, their events dropped (the caller sees the exception, not half an admission)
                (void)hipDeviceSynchronize();
                (void)hipGetLastError();
                for (const int si : sis) {
                    release(slots[si]);
                    const uint8_t one8 = 1;
                    (void)hipMemcpy(d_fin + si, &one8, 1, hipMemcpyHostToDevice);
                }
                for (size_t i = rs.size(); i-- > 0;) {
                    waiting.push_front(std::move(rs[i]));
                }
                ev.resize(ev0);
                throw;
            }
            fire(ev, ev0);
        }
        refresh_host_flags();
    }
    // host view of the running slots' sampling parameters (kernel selection and LDS sizing of the sampling kernels): after every
    // admission, and before a decode step that runs inside an admission (a request admitted a moment ago is already running)
    void refresh_host_flags()
    {
        host_max_top_k = 1;
        host_any_top_p = 0;
        host_any_temperature = false;
        host_any_repetition = false;
        for (int si = 0; si < max_batch; si++) {
            if (slots[si].active) {
                host_any_repetition |= (slots[si].repetition_penalty != 1.f);
                host_max_top_k = std::max(host_max_top_k, slot_topk[si]);
                host_any_top_p |= (slot_topk[si] == 0);
                host_any_temperature |= (slot_temp[si] != 1.f);
            }
        }
    }
};

extern "C" int ftcf_batcher_create(ftcf_gptneox_t engine, int max_batch, int page_tokens, int num_pages, int max_seq_len,
                                   ftcf_batcher_t* out)
{
    return guarded([&] {
        FTCF_CHECK_ARG(engine && out, "NULL argument");
        require_device();
        auto b = std::make_unique<ftcf_batcher>();
        b->init(engine, max_batch, page_tokens, num_pages, max_seq_len);
        *out = b.release();
    });
}
extern "C" int ftcf_batcher_submit(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k,
                                   float top_p, float temperature, unsigned long long seed, long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit(prompt_ids, prompt_len, max_new_tokens, top_k, top_p, temperature, (uint64_t)seed);
    });
}
extern "C" int ftcf_batcher_submit_ex(ftcf_batcher_t b, const int* prompt_ids, int prompt_len, int max_new_tokens, int top_k,
                                      float top_p, float temperature, float repetition_penalty, unsigned long long seed,
                                      const int* stop_words, int stop_len, long* request_id)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_id, "NULL argument");
        *request_id = b->submit(prompt_ids, prompt_len, max_new_tokens, top_k, top_p, temperature, (uint64_t)seed,
                                repetition_penalty, stop_words, stop_len);
    });
}
extern "C" int ftcf_batcher_step(ftcf_batcher_t b, long* request_ids, int* tokens, int* finished, int capacity, int* n_events)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b && request_ids && tokens && finished && n_events, "NULL argument");
        FTCF_CHECK_ARG(capacity >= 2 * b->max_batch, "event arrays must hold 2 * max_batch entries");
        if (b->outbox.empty()) {  // (else: the rest of the previous iteration's events first)
            std::vector<ftcf_batcher::Event> ev;
            try {
                b->step(ev);
            }
            catch (...) {
                b->outbox.insert(b->outbox.end(), ev.begin(), ev.end());  // tokens of decode steps that did run are not lost
                throw;
            }
            b->outbox.insert(b->outbox.end(), ev.begin(), ev.end());
        }
        int n = 0;
        for (; n < capacity && !b->outbox.empty(); n++) {
            request_ids[n] = b->outbox.front().id;
            tokens[n]      = b->outbox.front().token;
            finished[n]    = b->outbox.front().finished;
            b->outbox.pop_front();
        }
        *n_events = n;
    });
}
extern "C" int ftcf_batcher_set_token_callback(ftcf_batcher_t b, ftcf_token_callback_fn fn, void* user)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        b->on_token      = fn;
        b->on_token_user = user;
    });
}
extern "C" int ftcf_batcher_status(ftcf_batcher_t b, int* waiting, int* running, int* free_pages)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        int run = 0;
        for (const auto& s : b->slots) {
            run += s.active ? 1 : 0;
        }
        if (waiting) {
            *waiting = (int)b->waiting.size();
        }
        if (running) {
            *running = run + (b->outbox.empty() ? 0 : 1);  // (events still to be fetched keep the batcher "busy")
        }
        if (free_pages) {
            *free_pages = (int)b->free_pages.size();
        }
    });
}
extern "C" int ftcf_batcher_cancel(ftcf_batcher_t b, long request_id, int* found)
{
    return guarded([&] {
        FTCF_CHECK_ARG(b, "NULL argument");
        int hit = 0;
        for (auto it = b->waiting.begin(); it != b->waiting.end(); ++it) {
            if (it->id == request_id) {
                b->waiting.erase(it);
                hit = 1;
                break;
            }
        }
        for (int si = 0; si < b->max_batch && !hit; si++) {
            ftcf_batcher::Slot& s = b->slots[si];
            if (s.active && s.id == request_id) {
                FTCF_HIP_CHECK(hipSetDevice(b->e->cfg.device));
                const uint8_t one8 = 1;
                FTCF_HIP_CHECK(hipMemcpy(b->d_fin + si, &one8, 1, hipMemcpyHostToDevice));
                b->release(s);
                hit = 1;
            }
        }
        if (found) {
            *found = hit;
        }
    });
}
extern "C" int ftcf_batcher_destroy(ftcf_batcher_t b)
{
    return guarded([&] { delete b; });
}
