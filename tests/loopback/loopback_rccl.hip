// loopback_rccl.hip -- TEST INFRASTRUCTURE, not product: the ten nccl* entry points pqps_exchange resolves with dlsym
// (csrc/pqps_hip.hip:load_rccl: ncclGetUniqueId, ncclCommInitRank, ncclAllGather, ncclAllReduce, ncclSend, ncclRecv,
// ncclGroupStart, ncclGroupEnd, ncclCommDestroy, ncclGetErrorString; + the optional ncclCommAbort) between RANKS THAT
// ARE THREADS OF ONE PROCESS, on one GPU or several.  It exists so that the world > 1 code of the exchange -- sizes
// all-gather, displacements, the held-back send / recv group, ragged capacities, the COUNT all-reduce -- executes on
// the one-GPU boxes of this pool (tests/test_gpu_loopback_exchange.py); RCCL itself is not imitated beyond the
// stream semantics the exchange relies on:
//   * a call enqueues its work on the caller's stream and returns; the data moves when ALL ranks' streams have
//     reached their matching call (events), and a rank's stream goes on only when its peers have finished reading its
//     send buffer;
//   * calls block on the host until every rank has made the matching call (RCCL may too) -- every rank needs a thread
//     of its own;
//   * ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd form one operation; the k-th receive from peer p matches
//     p's k-th send to this rank and the counts must agree;
//   * ncclCommAbort wakes every waiting rank with an error.
// A STALL can be injected (tests of the exchange's bounded waits): LOOPBACK_STALL="<rank>:<op>" makes that rank's op number
// <op> (0-based, counted per communicator) never finish ON THE DEVICE -- a kernel spinning on a host flag sits in the
// rank's stream behind the op's copies, and the peers' streams wait for that rank as they would for a stuck RCCL kernel --
// until ncclCommAbort raises the flag (or 60 s pass).  Host calls return as usual: what hangs is the stream.
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <hip/hip_runtime.h>

namespace {

enum { kOk = 0, kUnhandledHip = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
constexpr int kMaxRanks = 32;

struct P2P { bool send; int peer; const void *src; void *dst; size_t bytes; };

struct Desc {
    int kind;                       // 0 all-gather, 1 all-reduce (sum, u64), 2 group of send / recv
    const void *send;
    void *recv;
    size_t bytes;                   // per rank (all-gather) / in all (all-reduce)
    std::vector<P2P> p2p;
};

struct World {
    volatile uint32_t *abort_flag_host = nullptr;   // mapped host word the stall kernel polls
    uint32_t *abort_flag_dev = nullptr;
    int stall_rank = -1;
    uint64_t stall_op = 0;
    int nranks = 0, joined = 0, left = 0;
    std::mutex m;
    std::condition_variable cv;
    uint64_t generation = 0;
    int arrived = 0;
    bool aborted = false;
    Desc desc[2][kMaxRanks];
    hipEvent_t ready[2][kMaxRanks] = {}, done[2][kMaxRanks] = {};
    int device[kMaxRanks] = {};
};

struct Comm {
    World *w;
    int rank, device;
    uint64_t ops = 0;
    uint64_t *tmp = nullptr;        // all-reduce: the ranks' contributions side by side
    size_t tmp_bytes = 0;
};

std::mutex g_lock;
std::map<std::string, World *> g_worlds;
uint64_t g_next_id = 1;

thread_local int t_group = 0;
thread_local Comm *t_group_comm = nullptr;
thread_local std::vector<P2P> t_pending;

size_t type_bytes(int dtype) {
    switch (dtype) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 0;
    }
}

// every rank arrives; the last one to arrive opens the gate
int barrier(World *w) {
    std::unique_lock<std::mutex> lk(w->m);
    if (w->aborted) return kInternalError;
    const uint64_t gen = w->generation;
    if (++w->arrived == w->nranks) { w->arrived = 0; w->generation++; w->cv.notify_all(); return kOk; }
    w->cv.wait(lk, [&] { return w->generation != gen || w->aborted; });
    return w->aborted ? kInternalError : kOk;
}

// the injected stall: one lane polls the flag (bounded: 60 s of the 100 MHz wall clock)
__global__ void stall_kernel(const uint32_t *flag) {
    const uint64_t deadline = wall_clock64() + 6000000000ull;
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == 0u && wall_clock64() < deadline) __builtin_amdgcn_s_sleep(64);
}

__global__ void sum_u64_kernel(const uint64_t *parts, int n, size_t count, uint64_t *out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    uint64_t s = 0;
    for (int r = 0; r < n; r++) s += parts[(size_t)r * count + i];
    out[i] = s;
}

int copy_async(void *dst, int dst_dev, const void *src, int src_dev, size_t bytes, hipStream_t s) {
    if (bytes == 0) return kOk;
    const hipError_t e = dst_dev == src_dev ? hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s)
                                            : hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, s);
    return e == hipSuccess ? kOk : kUnhandledHip;
}

// One operation of the communicator, called by every rank with its own descriptor.
int run_op(Comm *c, Desc &&mine, hipStream_t s) {
    World *w = c->w;
    const int par = (int)(c->ops++ & 1), me = c->rank, n = w->nranks;
    if (hipSetDevice(c->device) != hipSuccess) return kUnhandledHip;
    // A: everything this rank's stream held before the call (its send buffer is final, its receive buffer free)
    if (hipEventRecord(w->ready[par][me], s) != hipSuccess) return kUnhandledHip;
    { std::lock_guard<std::mutex> lk(w->m); w->desc[par][me] = std::move(mine); }
    int rc = barrier(w);
    if (rc) return rc;
    const Desc &d = w->desc[par][me];
    for (int p = 0; p < n; p++)
        if (p != me && hipStreamWaitEvent(s, w->ready[par][p], 0) != hipSuccess) return kUnhandledHip;
    // B: pull from the peers' send buffers
    if (d.kind == 0) {
        for (int p = 0; p < n && !rc; p++) {
            const Desc &pd = w->desc[par][p];
            if (pd.kind != 0 || pd.bytes != d.bytes) { rc = kInvalidUsage; break; }
            rc = copy_async((char *)d.recv + (size_t)p * d.bytes, c->device, pd.send, w->device[p], d.bytes, s);
        }
    } else if (d.kind == 1) {
        if (d.bytes * (size_t)n > c->tmp_bytes) {
            if (c->tmp) (void)hipFree(c->tmp);
            c->tmp_bytes = d.bytes * (size_t)n;
            if (hipMalloc((void **)&c->tmp, c->tmp_bytes) != hipSuccess) return kUnhandledHip;
        }
        for (int p = 0; p < n && !rc; p++) {
            const Desc &pd = w->desc[par][p];
            if (pd.kind != 1 || pd.bytes != d.bytes) { rc = kInvalidUsage; break; }
            rc = copy_async((char *)c->tmp + (size_t)p * d.bytes, c->device, pd.send, w->device[p], d.bytes, s);
        }
        if (!rc) {
            const size_t count = d.bytes / 8;
            hipLaunchKernelGGL(sum_u64_kernel, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, s, c->tmp, n, count, (uint64_t *)d.recv);
            if (hipGetLastError() != hipSuccess) rc = kUnhandledHip;
        }
    } else {
        int seen[kMaxRanks] = {};                                       // receives taken from each peer so far
        for (const P2P &op : d.p2p) {
            if (op.send) continue;
            const Desc &pd = w->desc[par][op.peer];
            if (pd.kind != 2) { rc = kInvalidUsage; break; }
            int k = 0;
            const P2P *match = nullptr;
            for (const P2P &ps : pd.p2p)
                if (ps.send && ps.peer == me && k++ == seen[op.peer]) { match = &ps; break; }
            seen[op.peer]++;
            if (!match || match->bytes != op.bytes) {
                fprintf(stderr, "loopback_rccl: rank %d receives %zu bytes from rank %d, which sends %zu\n", me, op.bytes, op.peer, match ? match->bytes : (size_t)0);
                rc = kInvalidUsage;
                break;
            }
            rc = copy_async(op.dst, c->device, match->src, w->device[op.peer], op.bytes, s);
            if (rc) break;
        }
        // every send must have a taker: a send nobody receives would hang real RCCL
        for (const P2P &op : d.p2p) {
            if (!op.send || rc) continue;
            const Desc &pd = w->desc[par][op.peer];
            bool taken = false;
            for (const P2P &pr : pd.p2p) if (!pr.send && pr.peer == me) taken = true;
            if (pd.kind != 2 || !taken) { fprintf(stderr, "loopback_rccl: rank %d sends to rank %d, which does not receive\n", me, op.peer); rc = kInvalidUsage; }
        }
    }
    if (!rc && w->stall_rank == me && c->ops - 1 == w->stall_op && w->abort_flag_dev) {
        hipLaunchKernelGGL(stall_kernel, dim3(1), dim3(1), 0, s, w->abort_flag_dev);     // this op never finishes on this rank's stream
        if (hipGetLastError() != hipSuccess) rc = kUnhandledHip;
    }
    if (hipEventRecord(w->done[par][me], s) != hipSuccess && !rc) rc = kUnhandledHip;
    if (rc) { std::lock_guard<std::mutex> lk(w->m); w->aborted = true; w->cv.notify_all(); return rc; }
    rc = barrier(w);
    if (rc) return rc;
    // C: this rank's send buffer is its own again once every peer has read it
    for (int p = 0; p < n; p++)
        if (p != me && hipStreamWaitEvent(s, w->done[par][p], 0) != hipSuccess) return kUnhandledHip;
    return kOk;
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return kInvalidArgument;
    std::lock_guard<std::mutex> lk(g_lock);
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "loopback-%llu", (unsigned long long)g_next_id++);
    return kOk;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kInvalidArgument;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return kUnhandledHip;
    World *w = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_lock);
        const std::string key(id.internal, strnlen(id.internal, sizeof id.internal));
        auto it = g_worlds.find(key);
        if (it == g_worlds.end()) {
            w = new World(); w->nranks = nranks; g_worlds[key] = w;
            if (const char *st = getenv("LOOPBACK_STALL")) {
                int r = -1; unsigned long long op = 0;
                if (sscanf(st, "%d:%llu", &r, &op) == 2 && r >= 0 && r < nranks) {
                    void *h = nullptr, *d = nullptr;
                    if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer(&d, h, 0) == hipSuccess) {
                        memset(h, 0, 64);
                        w->abort_flag_host = (volatile uint32_t *)h; w->abort_flag_dev = (uint32_t *)d;
                        w->stall_rank = r; w->stall_op = op;
                    }
                }
            }
        } else w = it->second;
    }
    if (w->nranks != nranks) return kInvalidArgument;
    for (int par = 0; par < 2; par++) {
        if (hipEventCreateWithFlags(&w->ready[par][rank], hipEventDisableTiming) != hipSuccess) return kUnhandledHip;
        if (hipEventCreateWithFlags(&w->done[par][rank], hipEventDisableTiming) != hipSuccess) return kUnhandledHip;
    }
    w->device[rank] = device;
    Comm *c = new Comm();
    c->w = w; c->rank = rank; c->device = device;
    const int rc = barrier(w);                                          // returns once every rank of the world has joined
    if (rc) { delete c; return rc; }
    *comm = c;
    return kOk;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return kOk;
    (void)hipSetDevice(c->device);
    if (c->tmp) (void)hipFree(c->tmp);
    delete c;                                                           // (the world and its events live as long as the process: tests)
    return kOk;
}

int ncclCommAbort(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return kOk;
    { std::lock_guard<std::mutex> lk(c->w->m); c->w->aborted = true; if (c->w->abort_flag_host) *c->w->abort_flag_host = 1u; c->w->cv.notify_all(); }
    return ncclCommDestroy(comm);
}

const char *ncclGetErrorString(int rc) {
    switch (rc) {
    case kOk: return "no error";
    case kUnhandledHip: return "unhandled HIP error (loopback)";
    case kInternalError: return "internal error: the communicator was aborted (loopback)";
    case kInvalidArgument: return "invalid argument (loopback)";
    case kInvalidUsage: return "invalid usage: the ranks' calls do not match (loopback)";
    default: return "error (loopback)";
    }
}

int ncclGroupStart(void) { t_group++; return kOk; }

int ncclGroupEnd(void) {
    if (t_group <= 0) return kInvalidUsage;
    if (--t_group > 0) return kOk;
    Comm *c = t_group_comm;
    t_group_comm = nullptr;
    // (a group without calls is not an operation.  The exchange always has one send and / or receive per peer unless
    //  NOBODY has a match -- in which case no rank has any, because sends and receives are paired.)
    if (!c) return kOk;
    Desc d;
    d.kind = 2; d.send = nullptr; d.recv = nullptr; d.bytes = 0;
    d.p2p.swap(t_pending);
    hipStream_t s = (hipStream_t)d.p2p.front().dst;                     // smuggled below
    d.p2p.erase(d.p2p.begin());
    return run_op(c, std::move(d), s);
}

static int p2p_call(bool send, const void *src, void *dst, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t tb = type_bytes(dtype);
    if (!c || !tb || peer < 0 || peer >= c->w->nranks || peer == c->rank) return kInvalidArgument;
    const bool lone = t_group == 0;
    if (lone) t_group++;
    if (t_group_comm && t_group_comm != c) return kInvalidUsage;        // one communicator per group is all the exchange needs
    if (!t_group_comm) { t_group_comm = c; t_pending.clear(); t_pending.push_back(P2P{false, -1, nullptr, (void *)s, 0}); }   // slot 0 carries the stream
    t_pending.push_back(P2P{send, peer, src, dst, count * tb});
    return lone ? ncclGroupEnd() : kOk;
}

int ncclSend(const void *send, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    return p2p_call(true, send, nullptr, count, dtype, peer, comm, s);
}

int ncclRecv(void *recv, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    return p2p_call(false, nullptr, recv, count, dtype, peer, comm, s);
}

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t tb = type_bytes(dtype);
    if (!c || !tb || t_group) return kInvalidUsage;
    Desc d;
    d.kind = 0; d.send = send; d.recv = recv; d.bytes = count * tb;
    return run_op(c, std::move(d), s);
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    if (!c || t_group) return kInvalidUsage;
    if (dtype != 5 || op != 0) return kInvalidArgument;                 // ncclUint64, ncclSum: what COUNT(*) uses
    Desc d;
    d.kind = 1; d.send = send; d.recv = recv; d.bytes = count * 8;
    return run_op(c, std::move(d), s);
}

}  // extern "C"
