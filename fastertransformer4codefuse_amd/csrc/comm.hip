// Communicator translation unit (split out of engine.hip in round 4): the local group's emulated collectives, the
// peer-mapped exchange windows (IPC mapping, hand-shake, the two-shot all-reduce kernel), agreements / barriers, and the
// `ftcf_comm_*` entry points of include/ftcf.h (utils/nccl_utils.cc:56-435, th_op/gptneox/utils/nccl_inherit_utils.cc:25-68,
// kernels/custom_ar_kernels.cu:139-260).
#include "comm.hip.h"


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

void local_allreduce(ftcf_comm* c, void* buf, size_t count, bool fp16, hipStream_t s)
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

void local_allgather(ftcf_comm* c, void* buf, size_t count_per_rank, bool fp16, hipStream_t s)
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
    const bool all_ok = __syncthreads_and(ok ? 1 : 0) != 0;
    // acquire at system scope: nothing of this workgroup reads the peers' buffers out of a line cached before their flags
    // showed this call (the writers release with __threadfence_system() before they store the flag)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    return all_ok;
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
bool window_allreduce(ftcf_comm* c, f16* buf, size_t count, hipStream_t s)
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
    const int        shared = (c->local || c->hx) ? c->world : 1;
    const int        nb     = std::max(8, 128 / shared);
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

void comm_barrier(ftcf_comm* c, hipStream_t s, int* d_scratch)
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
int comm_agree(ftcf_comm* c, int flag, hipStream_t s, int* d_scratch)
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
int comm_max(ftcf_comm* c, int v, hipStream_t s, int* d_scratch)
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

void comm_ensure_window(ftcf_comm* c, size_t bytes, hipStream_t s)
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
