// Tensor-parallel communicator of the engine (comm.hip): the structures, the host-side waits and the roctx ranges that every
// translation unit uses inline, and the entry points of comm.hip.
#pragma once
#include "engine_base.hip.h"
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

// ---- comm.hip ----
void local_allreduce(ftcf_comm* c, void* buf, size_t count, bool fp16, hipStream_t s);
void local_allgather(ftcf_comm* c, void* buf, size_t count_per_rank, bool fp16, hipStream_t s);
bool window_allreduce(ftcf_comm* c, f16* buf, size_t count, hipStream_t s);
void comm_barrier(ftcf_comm* c, hipStream_t s, int* d_scratch);
int comm_agree(ftcf_comm* c, int flag, hipStream_t s, int* d_scratch);
int comm_max(ftcf_comm* c, int v, hipStream_t s, int* d_scratch);
void comm_ensure_window(ftcf_comm* c, size_t bytes, hipStream_t s);
