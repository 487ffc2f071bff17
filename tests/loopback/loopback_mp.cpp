// loopback_mp.cpp -- TEST INFRASTRUCTURE, not product: the nccl* entry points pqps_exchange resolves with dlsym, between
// ranks that are PROCESSES sharing one GPU box (any devices, also the same one), for rehearsing `bench.py --gpus N` and the
// rank engines (initializeEngineSyntheticRankHIP + hipEngineJoinRanksHIP) as the driver launches them -- torchrun, one
// process per rank -- on this pool's one-GPU boxes, where RCCL itself refuses two ranks on one device.
// (tests/loopback/loopback_rccl.hip is the stand-in with RCCL's STREAM semantics between threads; this one trades that for
// process boundaries: every call synchronises its stream and moves the data through POSIX shared memory on the host.)
//   * every collective call blocks on the host until all ranks have made the matching call (RCCL may too);
//   * ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd form one operation; the k-th receive from peer p matches p's
//     k-th send to this rank, byte counts must agree;
//   * ncclCommAbort makes every waiting rank return an error.
// Shared memory: /pqps_lbmp_<id>: a header, per rank a descriptor table and an outbox (LOOPBACK_MP_OUTBOX_MB, default 256).
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <hip/hip_runtime.h>

namespace {

enum { kOk = 0, kUnhandledHip = 1, kSystemError = 2, kInternalError = 3, kInvalidArgument = 4, kInvalidUsage = 5 };
constexpr int kMaxRanks = 16, kMaxP2P = 64;

struct P2PDesc { int send, peer; uint64_t offset, bytes; };
struct RankArea {
    std::atomic<uint64_t> arrived;                   // barrier: generation this rank has reached
    int kind;                                        // 0 all-gather, 1 all-reduce, 2 group
    uint64_t bytes;                                  // all-gather / all-reduce: this rank's contribution in its outbox at 0
    int n_p2p;
    P2PDesc p2p[kMaxP2P];
};
struct Header {
    std::atomic<uint32_t> magic;
    std::atomic<int> joined;
    std::atomic<int> aborted;
    int nranks;
    uint64_t outbox_bytes;
    RankArea rank[kMaxRanks];
};

struct Comm {
    Header *h;
    char *base;
    size_t total;
    int rank, nranks, device;
    uint64_t gen;
    std::string name;
    std::vector<char> bounce;
};

thread_local int t_group = 0;
thread_local Comm *t_comm = nullptr;
thread_local hipStream_t t_stream = nullptr;
struct Pending { bool send; int peer; const void *src; void *dst; size_t bytes; };
thread_local std::vector<Pending> t_pending;

char *outbox(Comm *c, int r) { return c->base + sizeof(Header) + (size_t)r * c->h->outbox_bytes; }

size_t type_bytes(int dtype) {
    switch (dtype) {
    case 0: case 1: return 1;
    case 2: case 3: case 7: return 4;
    case 4: case 5: case 8: return 8;
    case 6: case 9: return 2;
    default: return 0;
    }
}

// all ranks reach generation `gen` (bounded: 120 s, or the abort flag)
int barrier(Comm *c) {
    c->gen++;
    c->h->rank[c->rank].arrived.store(c->gen, std::memory_order_release);
    struct timespec t0;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (;;) {
        bool all = true;
        for (int r = 0; r < c->nranks; r++) if (c->h->rank[r].arrived.load(std::memory_order_acquire) < c->gen) { all = false; break; }
        if (all) return kOk;
        if (c->h->aborted.load()) return kInternalError;
        struct timespec t1;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if (t1.tv_sec - t0.tv_sec > 120) { c->h->aborted.store(1); return kInternalError; }
        usleep(50);
    }
}

int run_op(Comm *c, int kind, const void *send, void *recv, size_t bytes, std::vector<Pending> &p2p, hipStream_t s) {
    if (hipSetDevice(c->device) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return kUnhandledHip;
    RankArea &me = c->h->rank[c->rank];
    char *out = outbox(c, c->rank);
    me.kind = kind;
    me.bytes = bytes;
    me.n_p2p = 0;
    if (kind != 2) {
        if (bytes > c->h->outbox_bytes) return kInvalidArgument;
        if (bytes && hipMemcpy(out, send, bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledHip;
    } else {
        uint64_t at = 0;
        for (const Pending &p : p2p) {
            if (me.n_p2p >= kMaxP2P) return kInvalidUsage;
            P2PDesc &d = me.p2p[me.n_p2p++];
            d.send = p.send; d.peer = p.peer; d.bytes = p.bytes; d.offset = at;
            if (p.send) {
                if (at + p.bytes > c->h->outbox_bytes) { fprintf(stderr, "loopback_mp: outbox too small (LOOPBACK_MP_OUTBOX_MB)\n"); return kInvalidArgument; }
                if (p.bytes && hipMemcpy(out + at, p.src, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return kUnhandledHip;
                at += (p.bytes + 63) & ~(uint64_t)63;
            }
        }
    }
    int rc = barrier(c);
    if (rc) return rc;
    for (int r = 0; r < c->nranks; r++) if (c->h->rank[r].kind != kind) { fprintf(stderr, "loopback_mp: ranks %d and %d are in different calls\n", c->rank, r); rc = kInvalidUsage; }
    if (!rc && kind == 0) {
        for (int r = 0; r < c->nranks && !rc; r++) {
            if (c->h->rank[r].bytes != bytes) { rc = kInvalidUsage; break; }
            if (bytes && hipMemcpy((char *)recv + (size_t)r * bytes, outbox(c, r), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = kUnhandledHip;
        }
    } else if (!rc && kind == 1) {
        std::vector<uint64_t> sum(bytes / 8, 0);
        for (int r = 0; r < c->nranks; r++) {
            if (c->h->rank[r].bytes != bytes) { rc = kInvalidUsage; break; }
            const uint64_t *v = (const uint64_t *)outbox(c, r);
            for (size_t i = 0; i < sum.size(); i++) sum[i] += v[i];
        }
        if (!rc && bytes && hipMemcpy(recv, sum.data(), bytes, hipMemcpyHostToDevice) != hipSuccess) rc = kUnhandledHip;
    } else if (!rc) {
        int seen[kMaxRanks] = {};
        for (const Pending &p : p2p) {
            if (p.send) continue;
            const RankArea &pa = c->h->rank[p.peer];
            const P2PDesc *match = nullptr;
            int k = 0;
            for (int i = 0; i < pa.n_p2p; i++)
                if (pa.p2p[i].send && pa.p2p[i].peer == c->rank && k++ == seen[p.peer]) { match = &pa.p2p[i]; break; }
            seen[p.peer]++;
            if (!match || match->bytes != p.bytes) {
                fprintf(stderr, "loopback_mp: rank %d receives %zu bytes from rank %d, which sends %llu\n", c->rank, p.bytes, p.peer, match ? (unsigned long long)match->bytes : 0ull);
                rc = kInvalidUsage;
                break;
            }
            if (p.bytes && hipMemcpy(p.dst, outbox(c, p.peer) + match->offset, p.bytes, hipMemcpyHostToDevice) != hipSuccess) { rc = kUnhandledHip; break; }
        }
        for (const Pending &p : p2p) {                                   // every send must have a taker (real RCCL would hang)
            if (!p.send || rc) continue;
            const RankArea &pa = c->h->rank[p.peer];
            bool taken = false;
            for (int i = 0; i < pa.n_p2p; i++) if (!pa.p2p[i].send && pa.p2p[i].peer == c->rank) taken = true;
            if (!taken) { fprintf(stderr, "loopback_mp: rank %d sends to rank %d, which does not receive\n", c->rank, p.peer); rc = kInvalidUsage; }
        }
    }
    if (rc) { c->h->aborted.store(1); return rc; }
    return barrier(c);                                                   // the outboxes are free again
}

}  // namespace

extern "C" {

typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return kInvalidArgument;
    memset(id, 0, sizeof *id);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf(id->internal, sizeof id->internal, "pqps_lbmp_%d_%lld_%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
    return kOk;
}

int ncclCommInitRank(void **comm, int nranks, ncclUniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return kInvalidArgument;
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return kUnhandledHip;
    const char *mb = getenv("LOOPBACK_MP_OUTBOX_MB");
    const uint64_t outbox_bytes = (uint64_t)(mb ? atoi(mb) : 256) << 20;
    const size_t total = sizeof(Header) + (size_t)nranks * outbox_bytes;
    std::string name = std::string("/") + std::string(id.internal, strnlen(id.internal, sizeof id.internal));
    bool creator = false;
    int fd = shm_open(name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0) { creator = true; if (ftruncate(fd, (off_t)total) != 0) { close(fd); return kSystemError; } }
    else {
        for (int tries = 0; tries < 6000 && fd < 0; tries++) { fd = shm_open(name.c_str(), O_RDWR, 0600); if (fd < 0) usleep(10000); }
        if (fd < 0) return kSystemError;
        struct stat st;
        for (int tries = 0; tries < 6000; tries++) { if (fstat(fd, &st) == 0 && (size_t)st.st_size >= total) break; usleep(10000); }
    }
    void *p = mmap(nullptr, total, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return kSystemError;
    Header *h = (Header *)p;
    if (creator) {
        h->nranks = nranks;
        h->outbox_bytes = outbox_bytes;
        h->joined.store(0);
        h->aborted.store(0);
        for (int r = 0; r < kMaxRanks; r++) h->rank[r].arrived.store(0);
        h->magic.store(0x4C424D50u, std::memory_order_release);
    } else {
        for (int tries = 0; tries < 6000 && h->magic.load(std::memory_order_acquire) != 0x4C424D50u; tries++) usleep(10000);
        if (h->magic.load() != 0x4C424D50u || h->nranks != nranks) { munmap(p, total); return kInvalidArgument; }
    }
    Comm *c = new Comm();
    c->h = h; c->base = (char *)p; c->total = total; c->rank = rank; c->nranks = nranks; c->device = device; c->gen = 0; c->name = name;
    h->joined.fetch_add(1);
    const int rc = barrier(c);                                           // returns once every rank of the world has joined
    if (rc) { munmap(p, total); delete c; return rc; }
    if (rank == 0) shm_unlink(name.c_str());                             // (everybody has it mapped: the name can go)
    *comm = c;
    return kOk;
}

int ncclCommDestroy(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return kOk;
    munmap(c->base, c->total);
    delete c;
    return kOk;
}

int ncclCommAbort(void *comm) {
    Comm *c = (Comm *)comm;
    if (!c) return kOk;
    c->h->aborted.store(1);
    return ncclCommDestroy(comm);
}

const char *ncclGetErrorString(int rc) {
    switch (rc) {
    case kOk: return "no error";
    case kUnhandledHip: return "unhandled HIP error (loopback_mp)";
    case kSystemError: return "system error: shared memory (loopback_mp)";
    case kInternalError: return "internal error: the communicator was aborted or a rank never arrived (loopback_mp)";
    case kInvalidArgument: return "invalid argument (loopback_mp)";
    case kInvalidUsage: return "invalid usage: the ranks' calls do not match (loopback_mp)";
    default: return "error (loopback_mp)";
    }
}

int ncclGroupStart(void) { t_group++; return kOk; }

int ncclGroupEnd(void) {
    if (t_group <= 0) return kInvalidUsage;
    if (--t_group > 0) return kOk;
    Comm *c = t_comm;
    t_comm = nullptr;
    if (!c) return kOk;                                                  // a group without calls is not an operation
    std::vector<Pending> p2p;
    p2p.swap(t_pending);
    return run_op(c, 2, nullptr, nullptr, 0, p2p, t_stream);
}

static int p2p_call(bool send, const void *src, void *dst, size_t count, int dtype, int peer, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t tb = type_bytes(dtype);
    if (!c || !tb || peer < 0 || peer >= c->nranks || peer == c->rank) return kInvalidArgument;
    const bool lone = t_group == 0;
    if (lone) t_group++;
    if (t_comm && t_comm != c) return kInvalidUsage;
    if (!t_comm) { t_comm = c; t_pending.clear(); t_stream = s; }
    t_pending.push_back(Pending{send, peer, src, dst, count * tb});
    return lone ? ncclGroupEnd() : kOk;
}

int ncclSend(const void *send, size_t count, int dtype, int peer, void *comm, hipStream_t s) { return p2p_call(true, send, nullptr, count, dtype, peer, comm, s); }
int ncclRecv(void *recv, size_t count, int dtype, int peer, void *comm, hipStream_t s) { return p2p_call(false, nullptr, recv, count, dtype, peer, comm, s); }

int ncclAllGather(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    const size_t tb = type_bytes(dtype);
    if (!c || !tb || t_group) return kInvalidUsage;
    std::vector<Pending> none;
    return run_op(c, 0, send, recv, count * tb, none, s);
}

int ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s) {
    Comm *c = (Comm *)comm;
    if (!c || t_group) return kInvalidUsage;
    if (dtype != 5 || op != 0) return kInvalidArgument;                  // ncclUint64, ncclSum: what COUNT(*) uses
    std::vector<Pending> none;
    return run_op(c, 1, send, recv, count * 8, none, s);
}

}  // extern "C"
