// pqps_hip.hip -- gfx950 (MI355X, CDNA4) kernels + the C-ABI shim of include/pqps_hip.h.
//
// The ONLY translation unit compiled by hipcc (filter_kernels.hpp is included here).
// No CUDA-compat headers, no dual paths: wave = 64 lanes, written for CDNA4 directly.
// No MFMA anywhere: this is integer compare + compaction, bound by HBM reads.
// The filter launch (scan tiles + expanders in one grid) is described in filter_kernels.hpp.

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <utility>
#include <dlfcn.h>
#include <time.h>

#include <atomic>
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "pqps_hip.h"
#include "filter_kernels.hpp"
#include "radix_sort.hpp"

namespace {

// ---- index build / probe ---------------------------------------------------
template <typename K>
__global__ void reverse_gather_kernel(const K *col, uint64_t n, K *keys_rev, uint32_t *rows_rev) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const uint64_t r = n - 1 - i;           // descending row order in, stable sort keeps it
        keys_rev[i] = col[r];
        rows_rev[i] = (uint32_t)r;
    }
}

template <typename K>
__device__ __forceinline__ bool key_less(K a, K b) { return a < b; }

// One wave, 64-ary search: every round the 64 lanes test the last key of 64 equal chunks of the
// remaining interval, so 100 M keys take 5-6 dependent loads instead of 27 (both bounds at once).
//   range[0] = first position with key >= lo, range[1] = first position with key > hi
template <typename K>
__global__ __launch_bounds__(64) void probe_kernel(const K *keys, uint64_t n, K lo, K hi, uint64_t *range, uint64_t *out_count, uint64_t *claim) {
    const uint64_t lane = threadIdx.x;
    uint64_t l0 = 0, r0 = n, l1 = 0, r1 = n;                   // invariant: answer in [l, r]
    while (l0 < r0 || l1 < r1) {                               // uniform
        const uint64_t len0 = r0 - l0, len1 = r1 - l1;
        const uint64_t c0 = (len0 + 63) / 64, c1 = (len1 + 63) / 64;
        const uint64_t p0 = l0 + (lane + 1) * c0 - 1, p1 = l1 + (lane + 1) * c1 - 1;
        bool t0 = true, t1 = true;                             // positions >= r count as "true"
        if (len0 && p0 < r0) t0 = !(keys[p0] < lo);
        if (len1 && p1 < r1) t1 = hi < keys[p1];
        const uint64_t b0 = __ballot(t0), b1 = __ballot(t1);
        if (len0) {
            if (!b0) l0 = r0;
            else { const uint64_t f = (uint64_t)__builtin_ctzll(b0), q = l0 + (f + 1) * c0 - 1; l0 += f * c0; r0 = q < r0 ? q : r0; }
        }
        if (len1) {
            if (!b1) l1 = r1;
            else { const uint64_t f = (uint64_t)__builtin_ctzll(b1), q = l1 + (f + 1) * c1 - 1; l1 += f * c1; r1 = q < r1 ? q : r1; }
        }
    }
    if (lane == 0) {
        range[0] = l0;
        range[1] = l1 < l0 ? l0 : l1;
        if (claim) {                                               // pqps_index_select's copy: the probe's rows go to out[base ...), all of them
            const uint64_t n = l1 > l0 ? l1 - l0 : 0, base = *out_count;
            claim[0] = base;
            *out_count = base + n;
        }
    }
}

// The rows of a probed range, appended: out[base + i] = perm[begin + i] + id_base (what the gather filter leaves when every
// candidate passes).  Grid-stride: the range is known on the device only.
__global__ __launch_bounds__(256) void append_range_kernel(const uint32_t *perm, const uint64_t *range, const uint64_t *claim, uint32_t id_base,
                                                           uint32_t *out_ids, uint64_t out_cap) {
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));     // (the range begins anywhere: 4-byte aligned accesses)
    const uint64_t begin = range[0], base = claim[0];
    uint64_t n = range[1] > begin ? range[1] - begin : 0;
    if (base >= out_cap) return;
    if (n > out_cap - base) n = out_cap - base;                  // a result that does not fit is cut off (the count says what there was)
    const uint32_t *src = perm + begin;
    uint32_t *dst = out_ids + base;
    for (uint64_t i = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (uint64_t)gridDim.x * 1024) {
        if (i + 4 <= n) {
            u32x4_a4 v = *(const u32x4_a4 *)(src + i);
            v += id_base;
            __builtin_nontemporal_store(v, (u32x4_a4 *)(dst + i));
        } else {
            for (uint64_t j = i; j < n; j++) dst[j] = src[j] + id_base;
        }
    }
}

// ---- synthetic generator ---------------------------------------------------
__host__ __device__ inline uint64_t synth_mix(uint64_t seed, uint64_t row, uint64_t k) {
    uint64_t z = seed + (row + 1) * 0x9E3779B97F4A7C15ull + k * 0xD1B54A32D192ED03ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

struct SynthRow {
    uint64_t command_id;
    int32_t exit_code, user_id, risk_level;
    uint8_t sudo_used, shell_code, host_code, base_code;
    uint16_t user_code;
};

// Marginals of SURVEY.md App. B (reference generator + measured 50k sample),
// as 32-bit fixed-point thresholds so host and device agree bit for bit.
__host__ __device__ inline SynthRow synth_row(uint64_t seed, uint64_t row,
                                              const uint32_t *cdf, const uint8_t *shell) {
    // cumulative P(risk <= r): .568 .865 .956 .990 1
    const uint32_t risk_cdf[5] = { 2439541424u, 3715147407u, 4105988735u, 4252017623u, 4294967295u };
    // P(exit != 0 | risk): .03 .06 .10 .16 .22
    const uint32_t fail_p[5] = { 128849019u, 257698038u, 429496730u, 687194767u, 944892805u };
    // P(sudo | risk): 0 .0065 .265 .966 .990
    const uint32_t sudo_p[5] = { 0u, 27917287u, 1138166333u, 4148938407u, 4252017623u };
    const int32_t fail_codes[5] = { 1, 2, 126, 127, 130 };
    SynthRow o;
    o.command_id = row;
    const uint64_t h0 = synth_mix(seed, row, 0), h1 = synth_mix(seed, row, 1), h2 = synth_mix(seed, row, 2);
    // user: first index whose cumulative threshold is >= x
    const uint32_t x = (uint32_t)(h0 >> 32);
    uint32_t l = 0, r = PQPS_SYNTH_USERS - 1;
    while (l < r) { const uint32_t m = (l + r) / 2; if (cdf[m] < x) l = m + 1; else r = m; }
    o.user_code = (uint16_t)l;
    o.user_id = 1000 + (int32_t)l;
    o.shell_code = shell[l];
    const uint32_t y = (uint32_t)(h1 >> 32);
    int risk = 0;
    while (risk < 4 && y > risk_cdf[risk]) risk++;
    o.risk_level = risk + 1;
    const uint32_t f = (uint32_t)h1;
    o.exit_code = (f < fail_p[risk]) ? fail_codes[(uint32_t)(h2 & 0xFFFF) % 5u] : 0;
    o.sudo_used = ((uint32_t)(h2 >> 32) < sudo_p[risk]) ? 1 : 0;
    o.host_code = (uint8_t)((h2 >> 16) & 15u);
    o.base_code = (uint8_t)(((h2 >> 20) & 0xFFFu) % 111u);
    return o;
}

__host__ __device__ inline void synth_store(const pqps_synth_cols &c, uint64_t i, const SynthRow &o) {
    if (c.command_id) c.command_id[i] = o.command_id;
    if (c.exit_code) c.exit_code[i] = o.exit_code;
    if (c.user_id) c.user_id[i] = o.user_id;
    if (c.risk_level) c.risk_level[i] = o.risk_level;
    if (c.sudo_used) c.sudo_used[i] = o.sudo_used;
    if (c.shell_code) c.shell_code[i] = o.shell_code;
    if (c.user_code) c.user_code[i] = o.user_code;
    if (c.host_code) c.host_code[i] = o.host_code;
    if (c.base_code) c.base_code[i] = o.base_code;
}

__global__ void synth_kernel(uint64_t seed, uint64_t row0, uint64_t n,
                             const uint32_t *cdf, const uint8_t *shell, pqps_synth_cols c) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        synth_store(c, i, synth_row(seed, row0 + i, cdf, shell));
}

// ---- all-gatherv tail: per-rank slots -> one contiguous ID list ------------------------------
// A slot = [u64 match count][u64 reserved][u32 IDs ...] exactly as one rank's filter left it
// (pqps_filter_scan wrote the count and the IDs into it), `slot_stride` u32 apart -- what ONE
// equal-size all-gather delivers.  merged = rank-order concatenation of the ID lists, the layout
// MPI_Allgatherv produces from recvCounts / displs (engine/mpi/executeEngine-mpi.c:753-765).
constexpr uint32_t kSlotHeaderWords = 4;         // u32 words in front of the IDs

__global__ __launch_bounds__(256) void merge_slots_kernel(const uint32_t *slots, uint32_t world, uint64_t slot_stride,
                                                          uint32_t *merged, uint64_t merged_cap, uint64_t *total_out) {
    const uint32_t r = blockIdx.y;
    const uint64_t seg_cap = slot_stride - kSlotHeaderWords;
    uint64_t displ = 0, total = 0, raw = 0;
    for (uint32_t i = 0; i < world; i++) {
        const uint64_t reported = *(const uint64_t *)(slots + (uint64_t)i * slot_stride);
        const uint64_t c = reported < seg_cap ? reported : seg_cap;
        if (i < r) displ += c;
        total += c;
        raw += reported;
    }
    const uint64_t mine = *(const uint64_t *)(slots + (uint64_t)r * slot_stride);
    const uint64_t cnt = mine < seg_cap ? mine : seg_cap;
    const uint32_t *src = slots + (uint64_t)r * slot_stride + kSlotHeaderWords;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x)
        if (displ + i < merged_cap) merged[displ + i] = src[i];
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0 && total_out) {
        total_out[0] = total;          // IDs actually merged
        total_out[1] = raw;            // IDs the ranks reported (> total means a slot overflowed)
    }
}

// ---- all-gatherv payload in compact form ---------------------------------------------------------------
// A shard's ascending ID list is sent as the LOW 16 BITS of every row number (relative to the shard's first row) plus,
// per 65 536-row group of the shard, the position in the list where the group's rows begin: 2 bytes per match + 4 bytes
// per group instead of 4 bytes per match -- the xGMI links carry half the bytes for any answer denser than two matches per
// 64 K rows (engine/mpi/executeEngine-mpi.c:765 ships `int`s; the receiving GPU rebuilds exactly those).  The sender decides
// (header word 3) from its own count; the receiver sees the same header through the sizes all-gather (mpi:753).
//   wire = [u32 goff[groups + 1], padded to 16 bytes][u16 low[n]]        goff[g] = first entry with row >= g * 65536, goff[groups] = n
constexpr uint32_t kWireGroupRows = 1u << 16;
constexpr uint64_t kWireHeaderWords = 4;                            // u64 per rank in the sizes all-gather: reported count, rows, id_base, format

__host__ __device__ inline uint64_t wire_groups(uint64_t n_rows) { return (n_rows + kWireGroupRows - 1) / kWireGroupRows; }
__host__ __device__ inline uint64_t wire_goff_bytes(uint64_t n_rows) { return ((wire_groups(n_rows) + 1) * 4 + 15) & ~15ull; }
__host__ __device__ inline uint64_t wire_bytes(uint64_t n_rows, uint64_t n) { return wire_goff_bytes(n_rows) + ((n * 2 + 3) & ~3ull); }
// ... and only a list of some size is worth the two extra launches (pack here, rebuild there): below kWireMinIds IDs (128 KB as
// u32: latency, not bytes, is what such a payload costs) it travels as it is.  PQPS_WIRE_MIN_IDS: the tests set 0.
constexpr uint64_t kWireMinIds = 32768;
__host__ __device__ inline bool wire_pays(uint64_t n_rows, uint64_t n, uint64_t min_ids) { return n >= min_ids && wire_bytes(n_rows, n) < n * 4; }

// slot = [u64 reported count][u64][u32 IDs ...] as the filter left it; hdr = this rank's 4 words of the sizes all-gather.
// eager_ids != 0: hdr is the head of this rank's block of an EAGER all-gather (below) -- a list of up to eager_ids IDs goes into the
// block behind the header as it is, and the query needs no second step if every rank's list fits.
__global__ __launch_bounds__(256) void wire_pack_kernel(const uint32_t *slot, uint64_t cap, uint64_t n_rows, uint32_t id_base, int enabled,
                                                        uint64_t min_ids, uint64_t *hdr, uint8_t *wire, uint64_t eager_ids, uint32_t *eager_dst) {
    const uint64_t reported = *(const uint64_t *)slot;
    const uint64_t n = reported < cap ? reported : cap;            // a slot that overflowed holds (and sends) `cap` IDs
    const bool eager = eager_ids && reported <= eager_ids;
    const bool compact = !eager && enabled && wire_pays(n_rows, n, min_ids);
    if (blockIdx.x == 0 && threadIdx.x == 0) { hdr[0] = reported; hdr[1] = n_rows; hdr[2] = id_base; hdr[3] = compact ? 1u : 0u; }
    if (eager) {
        const uint32_t *ids = slot + kSlotHeaderWords;
        for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) eager_dst[i] = ids[i];
        return;
    }
    if (!compact) return;
    const uint32_t *ids = slot + kSlotHeaderWords;
    uint32_t *goff = (uint32_t *)wire;
    uint16_t *low = (uint16_t *)(wire + wire_goff_bytes(n_rows));
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = tid; i < n; i += nthreads) low[i] = (uint16_t)(ids[i] - id_base);
    const uint64_t groups = wire_groups(n_rows);
    for (uint64_t g = tid; g <= groups; g += nthreads) {           // lower bound of the group's first row in the ascending list
        const uint64_t first = g * kWireGroupRows;
        uint64_t lo = 0, hi = n;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) / 2;
            if ((uint64_t)(ids[mid] - id_base) < first) lo = mid + 1; else hi = mid;
        }
        goff[g] = (uint32_t)lo;
    }
}

// out[i] = id_base + g * 65536 + low[i] for the entries i of group g (a workgroup walks groups, its threads a group's entries)
__global__ __launch_bounds__(256) void wire_expand_kernel(const uint8_t *wire, uint64_t n_rows, uint32_t id_base, uint32_t *out) {
    const uint32_t *goff = (const uint32_t *)wire;
    const uint16_t *low = (const uint16_t *)(wire + wire_goff_bytes(n_rows));
    const uint64_t groups = wire_groups(n_rows);
    for (uint64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const uint32_t b = goff[g], e = goff[g + 1], base = id_base + (uint32_t)(g * kWireGroupRows);
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) out[i] = base + (uint32_t)low[i];
    }
}

// the same for every peer's payload of one query in ONE launch (blockIdx.y = peer): seven launches of a few microseconds each
// would sit in the exchange stream behind one another for an 8-rank world
constexpr uint32_t kWireManyPeers = 16;
struct WireExpandMany {
    uint32_t n;
    struct { const uint8_t *wire; uint64_t rows; uint32_t *out; uint32_t id_base; uint32_t pad; } p[kWireManyPeers];
};
__global__ __launch_bounds__(256) void wire_expand_many_kernel(const WireExpandMany m) {
    const auto &d = m.p[blockIdx.y];
    const uint32_t *goff = (const uint32_t *)d.wire;
    const uint16_t *low = (const uint16_t *)(d.wire + wire_goff_bytes(d.rows));
    const uint64_t groups = wire_groups(d.rows);
    for (uint64_t g = blockIdx.x; g < groups; g += gridDim.x) {
        const uint32_t b = goff[g], e = goff[g + 1], base = d.id_base + (uint32_t)(g * kWireGroupRows);
        for (uint32_t i = b + threadIdx.x; i < e; i += blockDim.x) d.out[i] = base + (uint32_t)low[i];
    }
}

// ---- small answers in ONE collective --------------------------------------------------------------------------------------
// The sizes all-gather of a query carries, behind each rank's 32-byte header, room for eager_ids IDs: a rank whose list fits puts it
// there.  If EVERY rank's list fits (S1 at 125 M rows per rank: 8 400 IDs) the gathered blocks hold the whole answer -- this kernel
// moves each rank's IDs to its displacement (prefix of the gathered counts, mpi:758-762, computed here by every workgroup) and the
// query is done without the host having seen a size: no send / recv group (14 calls and tens of microseconds of host time at
// 8 ranks), no second wait.  If one rank's list does not fit, this kernel only lays the headers out for the host and the query takes
// the two-step path.   gathered = [world][block_bytes]: [u64 count, rows, id_base, form][u32 ids[eager_ids]]
constexpr uint64_t kEagerIdsDefault = 16384;
__global__ __launch_bounds__(256) void eager_unpack_kernel(const uint8_t *gathered, uint32_t world, uint64_t block_bytes, uint64_t eager_ids,
                                                           const uint64_t *caps, uint32_t *merged, uint64_t *sizes) {
    __shared__ uint64_t s_before[256];
    __shared__ uint32_t s_big[256];
    const uint32_t r = blockIdx.y;
    uint64_t before = 0;
    uint32_t big = 0;
    for (uint32_t q = threadIdx.x; q < world; q += blockDim.x) {
        const uint64_t *h = (const uint64_t *)(gathered + (uint64_t)q * block_bytes);
        const uint64_t c = h[0], held = c < caps[q] ? c : caps[q];
        if (c > eager_ids) big = 1;
        if (q < r) before += held;
        if (r == 0 && blockIdx.x == 0) for (uint32_t w = 0; w < kWireHeaderWords; w++) sizes[(uint64_t)q * kWireHeaderWords + w] = h[w];
    }
    s_before[threadIdx.x] = before;
    s_big[threadIdx.x] = big;
    __syncthreads();
    for (uint32_t step = blockDim.x / 2; step; step >>= 1) {
        if (threadIdx.x < step) { s_before[threadIdx.x] += s_before[threadIdx.x + step]; s_big[threadIdx.x] |= s_big[threadIdx.x + step]; }
        __syncthreads();
    }
    if (s_big[0]) return;                                            // somebody's list is elsewhere: the two-step path moves them all
    const uint64_t *h = (const uint64_t *)(gathered + (uint64_t)r * block_bytes);
    const uint64_t k = h[0] < caps[r] ? h[0] : caps[r];
    const uint32_t *src = (const uint32_t *)(h + kWireHeaderWords);
    uint32_t *dst = merged + s_before[0];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < k; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}

// ---- DELETE support: keep-list gather ------------------------------------------------------------
// dst[i] = src[keep[i]]; keep is ascending, so a wave's reads fall into a few neighbouring lines.
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const T *__restrict__ src, const uint32_t *__restrict__ keep,
                                                          uint64_t n, T *__restrict__ dst) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        dst[i] = src[keep[i]];
}

// ---- index mode across shards -----------------------------------------------------------------------
// keys[i] = order-preserving u64 image of col[ids[i] - id_base] (signed i32 keys are biased so that
// unsigned order = signed order); the count is read on the device (a slot header).
template <typename T, bool SIGNED>
__global__ __launch_bounds__(256) void gather_keys_kernel(const T *__restrict__ col, const uint32_t *__restrict__ ids,
                                                          const uint64_t *count, uint64_t capacity, uint32_t id_base,
                                                          uint64_t *__restrict__ keys) {
    uint64_t n = *count;
    if (n > capacity) n = capacity;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const T v = col[ids[i] - id_base];
        keys[i] = SIGNED ? (uint64_t)((uint32_t)v ^ 0x80000000u) : (uint64_t)v;
    }
}

// projection: out[i] = col[ids[i] - id_base] for the result list of a query (count read on the device)
template <typename T>
__global__ __launch_bounds__(256) void project_kernel(const T *__restrict__ col, const uint32_t *__restrict__ ids,
                                                      const uint64_t *count, uint64_t capacity, uint32_t id_base, T *__restrict__ out) {
    uint64_t n = *count;
    if (n > capacity) n = capacity;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        out[i] = col[ids[i] - id_base];
}

// checksums of an ID list: out[0] += sum ids[i], out[1] += sum ids[i] * (2 i + 1)   (mod 2^64; out zeroed by the caller)
__global__ __launch_bounds__(256) void ids_checksum_kernel(const uint32_t *__restrict__ ids, uint64_t n, uint64_t *out) {
    uint64_t s0 = 0, s1 = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const uint64_t v = ids[i];
        s0 += v;
        s1 += v * (2ull * i + 1ull);
    }
    s0 = wave_sum_u64(s0);
    s1 = wave_sum_u64(s1);
    if ((threadIdx.x & 63) == 0 && (s0 | s1)) {
        atomicAdd((unsigned long long *)&out[0], (unsigned long long)s0);
        atomicAdd((unsigned long long *)&out[1], (unsigned long long)s1);
    }
}

// rank-order compaction of [count | ids] slots together with their parallel key slots
__global__ __launch_bounds__(256) void compact_index_slots_kernel(const uint32_t *slots, const uint64_t *key_slots, uint32_t world,
                                                                  uint64_t slot_stride, uint32_t *ids_out, uint64_t *keys_out,
                                                                  uint64_t capacity, uint64_t *total_out) {
    const uint32_t r = blockIdx.y;
    const uint64_t seg_cap = slot_stride - 4;                       // kSlotHeaderWords
    uint64_t displ = 0, total = 0, raw = 0;
    for (uint32_t i = 0; i < world; i++) {
        const uint64_t reported = *(const uint64_t *)(slots + (uint64_t)i * slot_stride);
        const uint64_t c = reported < seg_cap ? reported : seg_cap;
        if (i < r) displ += c;
        total += c;
        raw += reported;
    }
    const uint64_t mine = *(const uint64_t *)(slots + (uint64_t)r * slot_stride);
    const uint64_t cnt = mine < seg_cap ? mine : seg_cap;
    const uint32_t *src = slots + (uint64_t)r * slot_stride + 4;
    const uint64_t *ksrc = key_slots + (uint64_t)r * seg_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x)
        if (displ + i < capacity) { ids_out[displ + i] = src[i]; keys_out[displ + i] = ksrc[i]; }
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0 && total_out) { total_out[0] = total; total_out[1] = raw; }
}

__global__ __launch_bounds__(256) void negate_iota_kernel(const uint32_t *ids, uint64_t n, uint32_t *neg, uint32_t *pos) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { neg[i] = ~ids[i]; pos[i] = (uint32_t)i; }
}

__global__ __launch_bounds__(256) void permute_pairs_kernel(const uint32_t *order, uint64_t n, const uint64_t *keys_in,
                                                            const uint32_t *ids_in, uint64_t *keys_out, uint32_t *ids_out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { const uint32_t p = order[i]; keys_out[i] = keys_in[p]; ids_out[i] = ids_in[p]; }
}

// ---- INSERT support: shift dictionary codes at or above a new value's rank -----------------
template <typename T>
__global__ void bump_codes_kernel(T *codes, uint64_t n, uint32_t threshold) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
        const T c = codes[i];
        if ((uint32_t)c >= threshold) codes[i] = (T)(c + 1);
    }
}

// Loads this library's code object (several MB, ~8 ms) when a context is created -- an engine does
// that on a background thread beside its CSV parse -- instead of inside the first query.
__global__ void warm_kernel() {}

// ---------------------------------------------------------------------------
// host side of the shim
// ---------------------------------------------------------------------------
thread_local char g_err[512] = "";
thread_local char g_kernel[200] = "";       // the instantiation the calling thread's last filter call launched (pqps_last_kernel)

// host time spent waiting for a ring slot to become free again (as opposed to time inside runtime calls)
uint64_t now_ns() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (uint64_t)ts.tv_sec * 1000000000ull + (uint64_t)ts.tv_nsec;
}

// Tuning switches (A/B runs of scripts/) exist in development builds only (-DPQPS_TUNING: `make libpqps_hip_dev.so`); the
// default build reads the deployment settings and the switches the test-suite forces kernel variants with (scripts/README.md
// lists which is which).
inline const char *tuning_env(const char *name) {
#ifdef PQPS_TUNING
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

uint64_t wire_min_ids() {
    const char *e = getenv("PQPS_WIRE_MIN_IDS");       // read per call: one process may hold test cases with and without it
    return e ? strtoull(e, nullptr, 10) : (uint64_t)kWireMinIds;
}

int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                   \
    do {                                                                                \
        hipError_t e_ = (expr);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(PQPS_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

}  // namespace

constexpr int kMaxTimedLaunches = 4096;
constexpr int kEvalBlocksPerCUMax = 1024; // grid cap of K1 per CU (grid-stride beyond it)

struct pqps_ctx {
    int device;
    int compute_units;
    hipStream_t stream;
    // filter scratch, grown on demand
    uint64_t scratch_steps;     // capacity in steps of 1024 rows
    uint16_t *masks;            // [steps][64] match words (steps that left a bit mask)
    uint16_t *slots, *slots_hi; // [steps][kSlotWords] each: 16-bit entries 0 .. 63 / 64 .. 127 of the steps with at most kListIds matches
    uint32_t *counts;           // [steps] step counts; all zero between queries
    uint64_t *tiny;             // [steps] tiny words: the entries of the steps with 1 - 3 matches (epoch-tagged)
    uint16_t *lists;            // [list_steps][1024] 16-bit row lists of the fuller steps (ID scans; allocated with the first one)
    uint64_t list_steps;
    bool lists_refused;         // the list area did not fit into device memory: bit masks for every step from then on
    // hand-off words of the ID-output launch (filter_kernels.hpp): tagged with the query's epoch, so nothing but
    // the two counters of ctl needs zeroing -- and those are a ping-pong pair, zeroed by the query before
    uint64_t *gsum, *ssum;      // [hand_groups], [hand_groups / 64 + 1]
    uint32_t *deferred;         // [hand_groups]
    uint32_t *ctl;              // [2][kCtlWords]
    uint64_t hand_groups;       // capacity
    uint32_t epoch;             // of the last ID query; 1 .. 65535, then the tagged arrays are zeroed and it starts over
    int parity;                 // ctl half the next ID query uses
    std::atomic<bool> needs_reset;   // a launch failed or a wait ran out: tagged words, ctl and epoch start over before the next ID query
                                // (set by whoever awaits the failed launch, possibly while another thread issues on this context)
    uint64_t *base_slot;        // gather: first output slot of the running query
    uint64_t *partials;         // workgroup totals of the scan (COUNT / FLAGS modes)
    uint32_t *status_host;      // [kStatusWords] mapped host words: a launch whose bounded waits ran out stores its epoch in word epoch % kStatusWords
    uint32_t *status_dev;       // their device address
    uint64_t *check_dev;        // [2] pqps_ids_checksum
    // per-context overrides of the launch parameters (pqps_ctx_set_option: A/B runs inside ONE process, where the physical
    // placement of the table is the same for every variant); -1 = the default
    long opt_list16, opt_list16_min, opt_list16_min_u8, opt_list_max, opt_list_max_u8, opt_tiny_max, opt_expand_lag, opt_sum_lag, opt_tune;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    // optional per-launch timing (bench.py roofline)
    bool timing;
    int timed;                  // launches recorded since the last reset
    hipEvent_t *ev_start, *ev_eval, *ev_stop;
    bool *stop_is_eval;         // ID output is one launch: its stop event is the end of the query
};

namespace {

// Every entry point that launches or copies comes through here: one process may hold contexts on several
// devices (the engine's PQPS_DEVICES shards), and kernels / allocations go to the calling thread's current device.
hipStream_t pick_stream(pqps_ctx *ctx, void *stream) {
    (void)hipSetDevice(ctx->device);
    return stream ? (hipStream_t)stream : ctx->stream;
}

void free_scratch(pqps_ctx *ctx) {
    void *all[] = {ctx->masks, ctx->slots, ctx->slots_hi, ctx->counts, ctx->tiny, ctx->gsum, ctx->ssum, ctx->deferred, ctx->ctl, ctx->base_slot, ctx->partials, ctx->lists};
    for (void *p : all) if (p) (void)hipFree(p);
    ctx->lists = nullptr; ctx->list_steps = 0;
    ctx->masks = nullptr; ctx->slots = nullptr; ctx->slots_hi = nullptr; ctx->counts = nullptr; ctx->tiny = nullptr; ctx->gsum = nullptr; ctx->ssum = nullptr; ctx->deferred = nullptr;
    ctx->ctl = nullptr; ctx->base_slot = nullptr; ctx->partials = nullptr;
    ctx->scratch_steps = 0;
}

// Epoch 0 = "never written": what every tagged word holds after this.
int zero_tagged_words(pqps_ctx *ctx, hipStream_t s) {
    HIP_TRY(hipMemsetAsync(ctx->counts, 0, ctx->scratch_steps * sizeof(uint32_t), s));
    HIP_TRY(hipMemsetAsync(ctx->tiny, 0, ctx->scratch_steps * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(ctx->gsum, 0, ctx->hand_groups * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(ctx->ssum, 0, (ctx->hand_groups / kSuperGroups + 1) * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(ctx->deferred, 0, ctx->hand_groups * sizeof(uint32_t), s));
    return PQPS_OK;
}

// PQPS_EPOCH_START (tests): the epoch a fresh scratch starts from, e.g. 65530 to reach the wrap within a few queries.
uint32_t first_epoch() {
    static const char *env = getenv("PQPS_EPOCH_START");
    const unsigned long v = env ? strtoul(env, nullptr, 10) : 0ul;
    return v < 0xFFFFul ? (uint32_t)v : 0u;
}

int ensure_scratch(pqps_ctx *ctx, uint64_t steps) {
    if (ctx->scratch_steps >= steps && ctx->masks) return PQPS_OK;
    if (ctx->masks) { HIP_TRY(hipDeviceSynchronize()); free_scratch(ctx); }
    const uint64_t cap = steps + steps / 4 + 64;
    const uint64_t groups = (cap + kGroupSteps - 1) / kGroupSteps;
    HIP_TRY(hipMalloc((void **)&ctx->masks, cap * 64 * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&ctx->slots, cap * kSlotWords * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&ctx->slots_hi, cap * kSlotWords * sizeof(uint16_t)));
    HIP_TRY(hipMalloc((void **)&ctx->counts, cap * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->tiny, cap * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&ctx->gsum, groups * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&ctx->ssum, (groups / kSuperGroups + 1) * sizeof(uint64_t)));
    HIP_TRY(hipMalloc((void **)&ctx->deferred, groups * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->ctl, 2 * kCtlWords * sizeof(uint32_t)));
    HIP_TRY(hipMalloc((void **)&ctx->base_slot, 64));
    HIP_TRY(hipMalloc((void **)&ctx->partials, kPartialSlots * sizeof(uint64_t)));
    ctx->scratch_steps = cap;
    ctx->hand_groups = groups;
    int rc = zero_tagged_words(ctx, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemset(ctx->ctl, 0, 2 * kCtlWords * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ctx->partials, 0, kPartialSlots * sizeof(uint64_t)));
    HIP_TRY(hipDeviceSynchronize());
    ctx->parity = 0;
    ctx->epoch = first_epoch();
    ctx->needs_reset.store(false);
    return PQPS_OK;
}

// The list area of ID scans: 2 KB per step (2 bytes per row of the largest table scanned so far on this context), taken
// when the first ID scan comes along.  If the device cannot spare it the scans go on with bit masks (slower for answers
// of more than a tenth of the rows, same results).  PQPS_LIST16=0: never (tests, A/B runs).
uint16_t *ensure_lists(pqps_ctx *ctx, uint64_t steps) {
    static const bool enabled = [] { const char *e = getenv("PQPS_LIST16"); return !e || atoi(e) != 0; }();
    if (!enabled || ctx->lists_refused) return nullptr;
    if (ctx->lists && ctx->list_steps >= steps) return ctx->lists;
    if (ctx->lists) { if (hipDeviceSynchronize() != hipSuccess) return nullptr; (void)hipFree(ctx->lists); ctx->lists = nullptr; ctx->list_steps = 0; }
    const uint64_t cap = steps + steps / 4 + 64;
    if (hipMalloc((void **)&ctx->lists, cap * kStepRows * sizeof(uint16_t)) != hipSuccess) {
        (void)hipGetLastError();
        ctx->lists = nullptr;
        ctx->lists_refused = true;
        return nullptr;
    }
    ctx->list_steps = cap;
    return ctx->lists;
}

// Back to "never written" on the stream the next query runs on: after a failed launch (the ctl half it would have
// zeroed for its successor still holds tickets) or after a kernel reported that a wait ran out.
int reset_handoff(pqps_ctx *ctx, hipStream_t s) {
    int rc = zero_tagged_words(ctx, s);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctx->ctl, 0, 2 * kCtlWords * sizeof(uint32_t), s));
    ctx->parity = 0;
    ctx->epoch = 0;
    return PQPS_OK;
}

// The status words (mapped host memory) name the launches that gave up, by epoch.  A failure is reported ONCE -- its word
// is cleared and the hand-off words start over before the next ID query -- so the context stays usable after the failed
// query (include/executeEngine-hip.h promises that to the engine's callers).
// take_status: any launch of the context (callers that have waited for everything on it: pqps_ctx_sync, the syncs of a
// query stream / an exchange).  take_status_of: the launches with epochs lo .. hi only (pqps_qstream_wait: the slot's own
// launches -- another slot's query may have run on the same lane and is reported to whoever awaits THAT slot).
int take_status(pqps_ctx *ctx, const char *who) {
    uint32_t st = 0;
    for (uint32_t w = 0; w < kStatusWords; w++) {
        const uint32_t v = ((volatile uint32_t *)ctx->status_host)[w];
        if (v) { st = v; ((volatile uint32_t *)ctx->status_host)[w] = 0; }
    }
    if (st == 0) return PQPS_OK;
    ctx->needs_reset.store(true);
    return fail(PQPS_EHIP, "an ID-output launch gave up waiting for its scan tiles (epoch %u): results of this %s are incomplete", st, who);
}

int take_status_of(pqps_ctx *ctx, uint32_t lo, uint32_t hi, const char *who) {
    if (lo == 0 && hi == 0) return PQPS_OK;                      // no ID launch (COUNT, an empty table)
    if (hi < lo || hi - lo >= kStatusWords) return take_status(ctx, who);      // epochs wrapped or started over in between: any
    uint32_t st = 0;
    for (uint32_t e = lo; e <= hi; e++) {
        volatile uint32_t *w = (volatile uint32_t *)ctx->status_host + (e & (kStatusWords - 1u));
        if (*w == e) { st = e; *w = 0; }
    }
    if (st == 0) return PQPS_OK;
    ctx->needs_reset.store(true);
    return fail(PQPS_EHIP, "an ID-output launch gave up waiting for its scan tiles (epoch %u): results of this %s are incomplete", st, who);
}

int check_pred(const pqps_column *cols, uint32_t n_cols, const pqps_predicate *pred) {
    if (!pred) return fail(PQPS_EINVAL, "predicate is NULL");
    if (n_cols > PQPS_MAX_COLUMNS) return fail(PQPS_EINVAL, "too many columns: %u", n_cols);
    if (n_cols && !cols) return fail(PQPS_EINVAL, "column array is NULL");
    if (pred->n_leaves > PQPS_MAX_LEAVES) return fail(PQPS_EINVAL, "too many leaves: %u", pred->n_leaves);
    if (pred->n_columns != n_cols) return fail(PQPS_EINVAL, "predicate uses %u columns, call passes %u", pred->n_columns, n_cols);
    for (uint32_t c = 0; c < n_cols; c++) {
        const uint32_t w = cols[c].width;
        if (w != 1 && w != 2 && w != 4 && w != 8) return fail(PQPS_EINVAL, "column %u: width %u not in {1,2,4,8}", c, w);
        if (!cols[c].data) return fail(PQPS_EINVAL, "column %u: NULL data", c);
        if (((uintptr_t)cols[c].data & 15u) != 0) return fail(PQPS_EINVAL, "column %u: data not 16-byte aligned", c);
    }
    uint32_t prev = 0;
    for (uint32_t k = 0; k < pred->n_leaves; k++) {
        if (pred->leaf[k].column >= n_cols) return fail(PQPS_EINVAL, "leaf %u: column %u out of range", k, pred->leaf[k].column);
        if (pred->leaf[k].column < prev) return fail(PQPS_EINVAL, "leaves must be sorted by column");
        prev = pred->leaf[k].column;
        if (pred->leaf[k].negate > 1) return fail(PQPS_EINVAL, "leaf %u: negate must be 0/1", k);
    }
    if (pred->n_leaves > PQPS_TT_LEAVES) {
        for (uint32_t s = 0; s < pred->n_leaves; s++) {
            const uint8_t t = pred->on_true[s], f = pred->on_false[s];
            if (pred->order[s] >= pred->n_leaves) return fail(PQPS_EINVAL, "step %u: bad leaf slot", s);
            if ((t < PQPS_ACCEPT && (t <= s || t >= pred->n_leaves)) || (f < PQPS_ACCEPT && (f <= s || f >= pred->n_leaves)))
                return fail(PQPS_EINVAL, "step %u: jump targets must point forward", s);
        }
    }
    return PQPS_OK;
}

void fill_args(EvalArgs &a, const pqps_column *cols, uint32_t n_cols, const pqps_predicate *pred) {
    memset(&a, 0, sizeof a);
    a.n_cols = n_cols;
    a.n_leaves = pred->n_leaves;
    a.truth = pred->truth;
    for (uint32_t c = 0; c < n_cols; c++) {
        a.col[c] = cols[c].data;
        a.width_log2[c] = cols[c].width == 1 ? 0 : cols[c].width == 2 ? 1 : cols[c].width == 4 ? 2 : 3;
    }
    uint32_t k = 0;
    for (uint32_t c = 0; c <= n_cols; c++) {
        while (k < pred->n_leaves && pred->leaf[k].column < c) k++;
        a.leaf_begin[c] = (uint8_t)k;
    }
    a.leaf_begin[n_cols] = (uint8_t)pred->n_leaves;
    for (uint32_t i = 0; i < pred->n_leaves; i++) {
        a.lo[i] = pred->leaf[i].lo;
        a.span[i] = pred->leaf[i].span;
        if (pred->leaf[i].negate) a.negmask |= 1u << i;
        a.on_true[i] = pred->on_true[i];
        a.on_false[i] = pred->on_false[i];
        a.order[i] = pred->order[i];
    }
    // Chain form: the truth table has exactly one true row  (AND of leaves, each required to be the
    // bit of that row) or exactly one false row (the negation of such an AND, i.e. an OR form).
    const uint32_t n = pred->n_leaves;
    if (n >= 1 && n <= PQPS_TT_LEAVES) {
        const uint32_t rows = 1u << n;
        const uint64_t all = rows >= 64 ? ~0ull : ((1ull << rows) - 1ull);
        const uint64_t tt = pred->truth & all;
        uint64_t single = 0;
        if (__builtin_popcountll(tt) == 1) { a.chain = 1; single = tt; }
        else if (__builtin_popcountll(tt) == (int)rows - 1) { a.chain = 2; single = ~tt & all; }
        if (a.chain) {
            const uint32_t e = (uint32_t)__builtin_ctzll(single);   // leaf k must evaluate to bit k of e
            a.chain_want = (e ^ a.negmask) & (rows >= 64 ? 0x3Fu : ((1u << n) - 1u));   // ... as a raw window hit
        }
    }
}

// ---- K1 dispatch: width-specialised instantiations ----------------------------------------
typedef void (*eval_fn)(const EvalArgs);

// every non-increasing (W0, W1, W2) from {8,4,2,1}, W = 0 marks an unused slot
#ifdef PQPS_DEV_SHAPES   /* development builds: the bench shapes only (compiles in a fraction of the time) */
#define PQPS_FOR_EACH_SHAPE(X) X(4,0,0) X(2,0,0) X(1,0,0) X(4,1,0) X(2,1,0) X(4,4,4)
#else
#define PQPS_FOR_EACH_SHAPE(X) \
    X(8,0,0) X(4,0,0) X(2,0,0) X(1,0,0) \
    X(8,8,0) X(8,4,0) X(8,2,0) X(8,1,0) X(4,4,0) X(4,2,0) X(4,1,0) X(2,2,0) X(2,1,0) X(1,1,0) \
    X(8,8,8) X(8,8,4) X(8,8,2) X(8,8,1) X(8,4,4) X(8,4,2) X(8,4,1) X(8,2,2) X(8,2,1) X(8,1,1) \
    X(4,4,4) X(4,4,2) X(4,4,1) X(4,2,2) X(4,2,1) X(4,1,1) X(2,2,2) X(2,2,1) X(2,1,1) X(1,1,1)
#endif

// chain kernels exist with 1 step per iteration and (narrow shapes) with several, and in three evaluator variants:
// EV 0 ballots (every shape), EV 1 one comparison on one column on the vector unit (single-column shapes), EV 2 a chain
// on the vector unit (shapes of up to 8 bytes per row whose widest column has 4)
constexpr uint64_t kInterleaveFromGroups = 8192;                    // from this many groups (537 M rows) on the expanders run among the scan tiles, see expand_lag()
constexpr uint64_t kListAreaBelowGroups = 4096;                     // below this many groups (268 M rows) ID scans have a list area, and the S1 shape takes its vector-unit chain kernel
constexpr bool valu_chain_shape(int a, int b, int c) { return a <= 4 && a + b + c <= 8; }
// ... and where it is the default.  A/B runs on one box (us per launch, ballots / vector unit): S1 = (2,1,0) as ID list at
// 100 M rows 59.4 / 57.4, at 1 G rows 496 / 496, as COUNT(*) at 1 G rows 438 / 468; Q_B = (4,1,0) as ID list 103 / 113 (its
// tile path spills at 64 VGPRs), at 1 G rows 901 / 959.  So: the u16 + u8 shape, ID output, below the size from which the
// expanders run among the tiles.
inline bool valu_chain_default(uint32_t w0, uint32_t w1, uint32_t w2, int mode, uint64_t n_rows) {
    return mode == MODE_IDS && w0 == 2 && w1 == 1 && w2 == 0 && n_rows < kListAreaBelowGroups * (uint64_t)kGroupSteps * kStepRows;
}

template <int MODE, int A, int B, int C, int S, bool NT>
eval_fn chain_variant(int ev) {
    if constexpr (B == 0 && C == 0) {
        if (ev == 1) return eval_chain_kernel<MODE, A, B, C, S, NT, 1>;
    }
    if constexpr (valu_chain_shape(A, B, C)) {
        if (ev == 2) return eval_chain_kernel<MODE, A, B, C, S, NT, 2>;
    }
    return eval_chain_kernel<MODE, A, B, C, S, NT, 0>;
}

template <int MODE, bool NT>
eval_fn find_spec_nt(uint32_t w0, uint32_t w1, uint32_t w2, bool chain, bool multi_step, int vc) {
#define X(A, B, C) if (w0 == A && w1 == B && w2 == C) return !chain ? eval_spec_kernel<MODE, A, B, C, NT> \
        : (multi_step ? chain_variant<MODE, A, B, C, chain_steps(A, B, C), NT>(vc) : chain_variant<MODE, A, B, C, 1, NT>(vc));
    PQPS_FOR_EACH_SHAPE(X)
#undef X
    return nullptr;
}

// A scan is "streaming" once the columns it reads outgrow the Infinity Cache (256 MB on MI355X) by
// a margin: it then uses `nt` loads and a one-shot grid; below that a repeated scan finds part of
// the table cached and plain loads win.
constexpr uint64_t kStreamingFootprint = 256ull << 20;

bool is_streaming(uint64_t footprint) {
    static const char *force = getenv("PQPS_NT_LOADS");
    return force ? atoi(force) != 0 : footprint > kStreamingFootprint;
}

void set_streaming(EvalArgs &a, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows) {
    uint64_t row_bytes = 0;
    for (uint32_t c = 0; c < n_cols; c++) row_bytes += cols[c].width;
    a.streaming = is_streaming(n_rows * row_bytes) ? 1u : 0u;
}

template <int MODE>
eval_fn find_spec(uint32_t w0, uint32_t w1, uint32_t w2, bool chain, bool multi_step, bool nt, int vc) {
    return nt ? find_spec_nt<MODE, true>(w0, w1, w2, chain, multi_step, vc) : find_spec_nt<MODE, false>(w0, w1, w2, chain, multi_step, vc);
}

// `a` must already carry the chain classification of fill_args(); sets a.streaming.
template <int MODE>
eval_fn pick_eval(const pqps_column *cols, uint32_t n_cols, const pqps_predicate *pred, EvalArgs &a, uint64_t n_rows) {
    set_streaming(a, cols, n_cols, n_rows);
    if (n_cols >= 1 && n_cols <= 3 && pred->n_leaves >= 1 && pred->n_leaves <= PQPS_TT_LEAVES) {
        const uint32_t w0 = cols[0].width, w1 = n_cols > 1 ? cols[1].width : 0, w2 = n_cols > 2 ? cols[2].width : 0;
        // several steps per iteration only where chain_steps() says so (a lone 1-byte column)
        static const char *force = tuning_env("PQPS_CHAIN_MULTI");
        const bool multi = force ? atoi(force) != 0 : true;
        // one comparison on one column (EV 1), or -- where measured faster -- a chain over narrow columns (EV 2), on the
        // vector unit: see RawStep::one_leaf / valu_leaf
        static const char *valu_env = tuning_env("PQPS_VALU_CHAIN");                 // tuning runs: 0 / 1 = ballots / vector unit for every eligible chain
        const bool one = pred->n_leaves == 1 && n_cols == 1;
        const bool valu = valu_chain_shape((int)w0, (int)w1, (int)w2) && (valu_env ? atoi(valu_env) != 0 : valu_chain_default(w0, w1, w2, MODE, n_rows));
        const int vc = a.chain == 0 ? 0 : (one ? 1 : (valu ? 2 : 0));
        a.valu_chain = vc ? 1u : 0u;
        if (eval_fn f = find_spec<MODE>(w0, w1, w2, a.chain != 0, multi, a.streaming != 0, vc)) {   // nullptr unless widths are non-increasing
            if (a.chain != 0 && multi) a.steps_per_iter = (uint32_t)chain_steps((int)w0, (int)w1, (int)w2);
            const char *mode = MODE == MODE_IDS ? "MODE_IDS" : MODE == MODE_COUNT ? "MODE_COUNT" : "MODE_FLAGS";
            if (a.chain != 0)
                snprintf(g_kernel, sizeof g_kernel, "eval_chain_kernel<%s, W0=%u, W1=%u, W2=%u, S=%u, NT=%s, EV=%d>", mode, w0, w1, w2,
                         a.steps_per_iter ? a.steps_per_iter : 1u, a.streaming ? "true" : "false", vc);
            else
                snprintf(g_kernel, sizeof g_kernel, "eval_spec_kernel<%s, W0=%u, W1=%u, W2=%u, NT=%s>", mode, w0, w1, w2, a.streaming ? "true" : "false");
            return f;
        }
    }
    snprintf(g_kernel, sizeof g_kernel, "eval_generic_kernel<%s, GATHER=false, NT=%s>", MODE == MODE_IDS ? "MODE_IDS" : MODE == MODE_COUNT ? "MODE_COUNT" : "MODE_FLAGS",
             a.streaming ? "true" : "false");
    return a.streaming ? eval_generic_kernel<MODE, false, true> : eval_generic_kernel<MODE, false, false>;
}

// Workgroups of K1.  Measured (fraction of 8 TB/s, S1 / Q_A / Q_C at 0.1 - 1 G rows): waves that
// grid-stride through many iterations drift apart and the chip-wide access window loses its
// locality -- 60 iterations per wave 0.72, 15 -> 0.77, 4 -> 0.82, 1 -> 0.84 at 1 G rows.  So a
// streaming scan is launched one-shot (a workgroup = 4 consecutive steps, dealt out in address order
// by the dispatcher; the grid-stride loop only engages beyond 1024 workgroups per CU), while a scan
// small enough to find part of its columns in the Infinity Cache does best with ~1.5 iterations
// (0.82 against 0.78 one-shot at 100 M rows x 3 B).
uint32_t eval_grid(pqps_ctx *ctx, uint64_t steps, bool streaming, uint32_t steps_per_iter) {
    const uint64_t per_wg = (uint64_t)kWaves * (steps_per_iter ? steps_per_iter : 1);   // steps one workgroup takes per iteration
    const uint64_t want = (steps + per_wg - 1) / per_wg;
    uint64_t cap = streaming ? (uint64_t)ctx->compute_units * kEvalBlocksPerCUMax : want * 2 / 3 + 1;
    static const char *env = tuning_env("PQPS_K1_BLOCKS_PER_CU");    // tuning runs
    if (env && atoi(env) >= 1 && atoi(env) <= kEvalBlocksPerCUMax) cap = (uint64_t)ctx->compute_units * (uint64_t)atoi(env);
    static const char *it_env = tuning_env("PQPS_K1_ITERS");          // tuning runs: iterations per wave
    if (it_env && atof(it_env) > 0) {
        cap = (uint64_t)((double)want / atof(it_env)) + 1;
        const uint64_t hard = (uint64_t)ctx->compute_units * kEvalBlocksPerCUMax;
        if (cap > hard) cap = hard;
    }
    const uint64_t g = want < cap ? want : cap;
    return (uint32_t)(g ? g : 1);
}

// Placement in the grid.  The chip has ~8 workgroups per CU in flight, i.e. `base` groups, and a round of loads
// takes about as long as `base` groups take to pass.  A group (and, from the same count words, a supergroup) is
// summed up by a tile `base` groups behind it -- its count words are in memory by then -- and expanded by a wave
// 2 * base groups behind it -- the sums in front of it are in memory by then.  Speed only: everybody checks what it
// finds, and the expander waits if it must.
uint32_t expand_lag_base(const pqps_ctx *ctx, uint32_t tiles_per_group) {
    return (uint32_t)ctx->compute_units * 10u / (tiles_per_group + 1u);
}

uint32_t expand_sum_lag(const pqps_ctx *ctx, uint32_t tiles_per_group) {
    static const char *env = getenv("PQPS_SUM_LAG");                // huge: no tile sums anything up (tests: every expander does it itself)
    if (env) return (uint32_t)strtoul(env, nullptr, 10);
    // Twice the groups in flight: a tile that looks `base` groups back finds some straggler of those 16 tiles still
    // running more often than not (stamps: expanders needed 3.2 looks on average and 5.6 us from start to settled
    // with `base`, 1.5 looks and 2.9 us with twice that; S1 58.0 -> 57.5 us, Q_A 82.5 -> 80.4, Q_B 107.8 -> 103.8 at
    // 100 M rows).  The groups behind that horizon at the end of the table are summed up by their own expanders.
    return 2u * expand_lag_base(ctx, tiles_per_group);
}

// Expanders among the scan tiles hide their work under the scan, but hold wave slots the scan could use and leave
// a pipeline of ~2 * base groups to drain at the end; expanders behind the last tile cost the scan nothing but
// run after it.  Measured (S1 / Q_A / Q_B, whole query): at 100 M rows (1.5 k groups) all behind wins (56 / 86 /
// 105 us against 58 / 96 / 126), at 300 M rows the two are level, at 1 G rows among the tiles wins by 7 - 10 %.

uint32_t expand_lag(const pqps_ctx *ctx, uint32_t tiles_per_group, uint64_t groups) {
    static const char *env = getenv("PQPS_EXPAND_LAG");
    if (env) return (uint32_t)strtoul(env, nullptr, 10);
    // (round 4, with the copied entries: at 530 M rows all behind the last tile is level for S1 / Q_A / Q_B and 2 - 8 % faster for the wide, the
    // narrow and the dense shapes; at 700 M rows among the tiles wins S1 and Q_B by 2 %, at 1 G rows S1 by 7 %, Q_A / Q_B by 3 %)
    if (groups < kInterleaveFromGroups) return 0x7FFFFFFFu;         // all behind the last tile
    // behind the sums it needs: at 1 G rows (sum lag / expander lag in groups) 150 / 300: Q_A 686 us, Q_B 918;
    // 300 / 600: 699, 938; 300 / 450: 669, 903
    return expand_sum_lag(ctx, tiles_per_group) + expand_lag_base(ctx, tiles_per_group);
}

uint32_t expand_spin_limit() {
    static const char *env = getenv("PQPS_EXPAND_SPIN_LIMIT");      // 0: an expander that is early gives up at once (tests the recovery pass)
    return env ? (uint32_t)strtoul(env, nullptr, 10) : (1u << 16);
}

// The filter.  `rows` = scan rows or the gather upper bound.
//   COUNT / FLAGS: the scan kernel + a one-workgroup reduction of the workgroup totals.
//   ID output:     ONE launch -- scan tiles and, `lag` groups behind them in the grid, the expander
//                  workgroups that turn match words into row IDs (filter_kernels.hpp).
constexpr uint64_t kGatherGridGroups = 32;           // 512 tiles (resident all at once, 2 workgroups per CU) + 512 expander workgroups behind them

int run_filter(pqps_ctx *ctx, eval_fn k1, EvalArgs &a, uint64_t rows, int mode, bool gather,
               uint32_t id_base, uint32_t *out_ids, uint64_t out_cap, uint64_t *out_count, hipStream_t s,
               hipEvent_t *done_io = nullptr) {
    // `done_io` (optional): on return *done_io is an event that becomes ready when the last kernel of this query has
    // finished.  It rides on that kernel's own dispatch packet: a separate hipEventRecord would put a barrier
    // packet behind it and cost the NEXT query on this stream ~7 us of idle queue.  The caller's event is used
    // unless the context records timings -- then the recorder's own stop event of this launch is handed back (one
    // event per dispatch packet).
    hipEvent_t done = done_io ? *done_io : nullptr;
    const uint64_t steps = (rows + kStepRows - 1) / kStepRows;
    int rc = ensure_scratch(ctx, steps);
    if (rc) return rc;
    uint64_t groups = (steps + kGroupSteps - 1) / kGroupSteps;
    // gather: `rows` is only an upper bound (the probed range lives on the device).  The grid is sized for at most
    // kGatherGridGroups groups (4 M candidates); the kernel's workgroups loop if the range turns out wider.
    if (gather && mode == MODE_IDS && groups > kGatherGridGroups) groups = kGatherGridGroups;
    a.masks = ctx->masks;
    a.slots = ctx->slots;
    a.slots_hi = ctx->slots_hi;
    a.counts = ctx->counts;
    a.tiny = ctx->tiny;
    a.partials = ctx->partials;
    const bool timed = ctx->timing && ctx->timed < kMaxTimedLaunches;
    if (mode != MODE_IDS) {
        const uint32_t grid = eval_grid(ctx, steps, a.streaming != 0, a.steps_per_iter);
        if (timed) hipExtLaunchKernelGGL(k1, dim3(grid), dim3(kBlock), 0, s, ctx->ev_start[ctx->timed], ctx->ev_eval[ctx->timed], 0, a);
        else hipLaunchKernelGGL(k1, dim3(grid), dim3(kBlock), 0, s, a);
        HIP_TRY(hipGetLastError());
        if (timed) done = ctx->ev_stop[ctx->timed];              // the end of the query = the end of the reduction
        if (done) hipExtLaunchKernelGGL(reduce_totals_kernel, dim3(1), dim3(kBlock), 0, s, nullptr, done, 0, ctx->partials, out_count);
        else hipLaunchKernelGGL(reduce_totals_kernel, dim3(1), dim3(kBlock), 0, s, ctx->partials, out_count);
        HIP_TRY(hipGetLastError());
        if (timed) { ctx->stop_is_eval[ctx->timed] = false; ctx->timed++; }
        if (done_io) *done_io = done;
        return PQPS_OK;
    }
    if (groups == 0) {
        // no rows at all: the count is the base (0, or unchanged when appending)
        if (!gather) HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint64_t), s));
        if (done) HIP_TRY(hipEventRecord(done, s));
        return PQPS_OK;
    }
    if (ctx->needs_reset.exchange(false)) { rc = reset_handoff(ctx, s); if (rc) { ctx->needs_reset.store(true); return rc; } }
    if (ctx->epoch >= 0xFFFFu) {                                 // the tags are about to repeat: start over from "never written"
        rc = zero_tagged_words(ctx, s);
        if (rc) return rc;
        ctx->epoch = 0;
    }
    // epoch and ctl half are committed only once the launch is in the queue: a launch that failed never zeroed the
    // other half, and the next query must not build on it
    a.epoch = ctx->epoch + 1u;
    const int half = ctx->parity;
    a.gsum = ctx->gsum; a.ssum = ctx->ssum; a.deferred = ctx->deferred;
    a.ctl = ctx->ctl + half * kCtlWords;
    a.zctl = ctx->ctl + (half ^ 1) * kCtlWords;
    a.base_slot = ctx->base_slot;
    // The list area serves the smaller launches whose expanders run BEHIND the last tile (below kListAreaBelowGroups groups): there the
    // expansion is the launch's tail and a copy is what it should be (`risk_level > 1` at 100 M rows 161 -> 130 us, `> 2`
    // 101 -> 91; few-percent answers level).  Among the tiles -- 268 M rows and more -- the expansion hides under the scan
    // whatever form the matches were left in, and the lists only add traffic and a longer tile: same-process A/B at 1 G rows
    // on two boxes, lists / none: S1 500 / 493 and 501 / 484 us, Q_A 800 / 753 and 747 / 743, Q_B 1080 / 999 and 987 / 967,
    // `risk_level > 2` 1051 / 929 and 921 / 893, `> 1` 1475 / 1254 (round 3 had compared processes, whose tables lie elsewhere).
    const bool lists_wanted = ctx->opt_list16 >= 0 ? ctx->opt_list16 != 0 : groups < kListAreaBelowGroups;
    a.lists = (gather || !lists_wanted) ? nullptr : ensure_lists(ctx, steps);
    {
        // Measured at 100 M rows (S1 / Q_A / Q_B / risk_level > 2, us per query): from 103 matches on 58.7 / 91.7 / 108.6 / 91.7, from 33
        // 57.8 / 90.6 / 104.6 / 91.8, from 9 58.3 / 88.8 / 104.4 / 91.5, every non-empty step 60.7 / 88.7 / 104.2 / 91.6 (round 3; the steps below
        // the threshold then left 10-bit lists in their slots, now 16-bit entries or tiny words -- up to list_max matches a step
        // never comes here: step_form).  1-byte columns: their tiles have no instruction slots to spare (a lone u8
        // column with 7 % matches: 47.5 us with bit masks, 56 - 58 with lists).
        static const uint32_t from = [] { const char *e = tuning_env("PQPS_LIST16_MIN"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 8u; }();
        static const uint32_t from_u8 = [] { const char *e = tuning_env("PQPS_LIST16_MIN_U8"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 1024u; }();
        const uint32_t f0 = ctx->opt_list16_min >= 0 ? (uint32_t)ctx->opt_list16_min : from;
        const uint32_t f1 = ctx->opt_list16_min_u8 >= 0 ? (uint32_t)ctx->opt_list16_min_u8 : from_u8;
        a.list16_min = f0 < 1024u ? f0 : 1024u;
        a.list16_min_u8 = f1 < 1024u ? f1 : 1024u;
        // 16-bit entries inside the step's slot up to this many matches (0: never); 1-byte columns: their tiles have no
        // instruction slots to spare for ranking
        // PQPS_LIST_MAX (tests): 0 = no step leaves entries in its slot (bit masks / the list area for every step), 128 = up to a full slot
        static const long env_max = [] { const char *e = getenv("PQPS_LIST_MAX"); return e ? strtol(e, nullptr, 0) : -1l; }();
        const uint32_t m0 = ctx->opt_list_max >= 0 ? (uint32_t)ctx->opt_list_max : (env_max >= 0 ? (uint32_t)env_max : kListDefault);
        // PQPS_TINY_MAX (tests): 0 = no step leaves its entries in a tiny word
        static const long env_tiny = [] { const char *e = getenv("PQPS_TINY_MAX"); return e ? strtol(e, nullptr, 0) : -1l; }();
        const uint32_t t0 = ctx->opt_tiny_max >= 0 ? (uint32_t)ctx->opt_tiny_max : (env_tiny >= 0 ? (uint32_t)env_tiny : kTinyIds);
        a.tiny_max = t0 < kTinyIds ? t0 : kTinyIds;
        // ... not where the widest predicate column is one byte wide (as for the entries in the slots: a step is 1 KB of input there, and
        // an answer of a 1-byte predicate is rarely sparse; the second load of every look cost a lone u8 column 2.5 % at 1 G rows)
        bool narrow = a.n_cols > 0;
        for (uint32_t c = 0; c < a.n_cols; c++) narrow = narrow && a.width_log2[c] == 0;
        if (narrow && ctx->opt_tiny_max < 0 && env_tiny < 0) a.tiny_max = 0;
        a.list_max = m0 < kListIds ? m0 : kListIds;
        a.list_max_u8 = 0;                                           // (the kernels of 1-byte predicates carry neither form: step_form)
    }
    a.status = ctx->status_dev;
    a.out_ids = out_ids; a.out_cap = out_cap; a.out_count = out_count;
    a.id_base = id_base;
    a.accumulate = gather ? 1u : 0u;
    const uint32_t tiles_per_group = (uint32_t)kGroupSteps / ((uint32_t)kWaves * (a.steps_per_iter ? a.steps_per_iter : 1u));
    a.lag = gather ? 0x7FFFFFFFu : (ctx->opt_expand_lag >= 0 ? (uint32_t)ctx->opt_expand_lag : expand_lag(ctx, tiles_per_group, groups));   // (a looping gather grid has all its expanders behind its tiles)
    a.grid_groups = (uint32_t)groups;
    a.sum_lag = ctx->opt_sum_lag >= 0 ? (uint32_t)ctx->opt_sum_lag : expand_sum_lag(ctx, tiles_per_group);
    if (a.sum_lag == 0) a.sum_lag = 1;                           // a tile never sums up its own group
    if (a.sum_lag > 0x3FFFFFFFu) a.sum_lag = 0x3FFFFFFFu;         // (2 * sum_lag is computed in 32 bits)
    a.spin_limit = expand_spin_limit();
    static const uint32_t tune = [] { const char *e = tuning_env("PQPS_TUNE"); return e ? (uint32_t)strtoul(e, nullptr, 0) : 0u; }();
    a.tune = ctx->opt_tune >= 0 ? (uint32_t)ctx->opt_tune : tune;
    // gather: [tiles of `groups` groups][kGatherParts expander workgroups per group]; scan: quads of tiles with their expander slot, then the trailing groups
    const uint64_t main_blocks = gather ? groups * tiles_per_group : ((groups + 3) / 4) * (4ull * tiles_per_group + 1);
    const uint64_t lag = gather ? kGatherParts * groups : trailing_groups((uint32_t)groups, a.lag);
    if (main_blocks + lag > 0x7FFFFFFFull) return fail(PQPS_EINVAL, "scan of %llu rows needs more workgroups than one launch holds", (unsigned long long)rows);
    const uint64_t slack = 0;
#ifdef PQPS_STAMPS
    static uint64_t *stamp_buf = nullptr;
    static size_t stamp_words = 0;
    const uint64_t n_tiles = groups * tiles_per_group;
    const size_t want_words = 4 + groups * 8 + n_tiles;
    if (want_words > stamp_words) { if (stamp_buf) (void)hipFree(stamp_buf); HIP_TRY(hipMalloc((void **)&stamp_buf, want_words * 8)); stamp_words = want_words; }
    HIP_TRY(hipMemsetAsync(stamp_buf, 0, want_words * 8, s));
    a.stamps = stamp_buf;
    a.stamp_groups = groups;
    struct StampDump { uint64_t *buf; size_t words; uint64_t groups, tpg, lag; hipStream_t s; ~StampDump() {
        const char *path = getenv("PQPS_STAMPS_FILE");
        if (!path) return;
        (void)hipStreamSynchronize(s);
        uint64_t *host = (uint64_t *)malloc(words * 8);
        (void)hipMemcpy(host, buf, words * 8, hipMemcpyDeviceToHost);
        host[0] = groups; host[1] = tpg; host[2] = lag;         // (host[3]: the device's stamp of the launch's first instruction)
        if (FILE *f = fopen(path, "wb")) { fwrite(host, 8, words, f); fclose(f); }
        free(host);
    } } stamp_dump{stamp_buf, want_words, groups, tiles_per_group, a.lag, s};
#endif
    hipEvent_t stop = timed ? ctx->ev_eval[ctx->timed] : done;
    if (timed || done) hipExtLaunchKernelGGL(k1, dim3((uint32_t)(main_blocks + lag + slack)), dim3(kBlock), 0, s, timed ? ctx->ev_start[ctx->timed] : nullptr, stop, 0, a);
    else hipLaunchKernelGGL(k1, dim3((uint32_t)(main_blocks + lag + slack)), dim3(kBlock), 0, s, a);
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) {
            ctx->needs_reset.store(true);
            return fail(PQPS_EHIP, "filter launch failed: %s", hipGetErrorString(le));
        }
    }
    ctx->epoch = a.epoch;
    ctx->parity = half ^ 1;
    if (timed) {
        ctx->stop_is_eval[ctx->timed] = true;
        ctx->timed++;
    }
    if (done_io) *done_io = stop;
    return PQPS_OK;
}

}  // namespace

// Two queries in flight (pqps_qstream, pqps_exchange) need their two HIP streams on different hardware queues.
// The runtime's pool is 4 queues for all streams of the process by default; ask for 8 unless the host has
// chosen a number.  Only effective if this library is loaded before the HIP runtime initialises -- a host that
// starts HIP first (torch) sets GPU_MAX_HW_QUEUES itself (bench.py does).
__attribute__((constructor)) static void pqps_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "8", 0); }

extern "C" {

const char *pqps_last_error(void) { return g_err; }
const char *pqps_last_kernel(void) { return g_kernel; }

int pqps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

}  // extern "C"

namespace { int create_ctx(int device, bool lane, pqps_ctx **out); }

extern "C" {

int pqps_ctx_create(int device, pqps_ctx **out) { return create_ctx(device, false, out); }

}  // extern "C"

namespace {

// `lane`: the context is one of the two scan lanes of a query stream / an exchange.  Its HIP stream is created at
// the highest priority: the runtime keeps a separate pool of hardware queues per priority, so the two lanes get
// two queues of their own whatever else the process has created (torch alone creates dozens of streams; two lanes
// that end up on one hardware queue run their scans one after the other and the overlap is gone).  Both lanes
// have the SAME priority: neither is favoured.
int create_ctx(int device, bool lane, pqps_ctx **out) {
    if (!out) return fail(PQPS_EINVAL, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(PQPS_ENODEVICE, "no HIP device visible (%s): the HIP engine has no CPU fallback",
                    e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(PQPS_EINVAL, "device %d out of range (0..%d)", device, n - 1);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    pqps_ctx *ctx = new (std::nothrow) pqps_ctx();
    if (!ctx) return fail(PQPS_ENOMEM, "out of host memory");
    ctx->device = device;
    ctx->compute_units = prop.multiProcessorCount;
    ctx->scratch_steps = 0;
    ctx->masks = nullptr; ctx->slots = nullptr; ctx->slots_hi = nullptr; ctx->counts = nullptr; ctx->tiny = nullptr; ctx->gsum = nullptr; ctx->ssum = nullptr; ctx->deferred = nullptr;
    ctx->ctl = nullptr; ctx->base_slot = nullptr; ctx->partials = nullptr;
    ctx->lists = nullptr; ctx->list_steps = 0; ctx->lists_refused = false;
    ctx->status_host = nullptr; ctx->status_dev = nullptr;
    ctx->parity = 0; ctx->epoch = 0; ctx->hand_groups = 0; ctx->needs_reset.store(false);
    ctx->check_dev = nullptr;
    ctx->opt_list16 = ctx->opt_list16_min = ctx->opt_list16_min_u8 = ctx->opt_list_max = ctx->opt_list_max_u8 = ctx->opt_tiny_max = ctx->opt_expand_lag = ctx->opt_sum_lag = ctx->opt_tune = -1;
    ctx->sort_tmp = nullptr;
    ctx->sort_tmp_bytes = 0;
    ctx->timing = false;
    ctx->timed = 0;
    ctx->ev_start = ctx->ev_eval = ctx->ev_stop = nullptr;
    ctx->stop_is_eval = nullptr;
    hipError_t se;
    const char *prio = tuning_env("PQPS_STREAM_PRIORITY");           // experiments: the context's own stream at a given priority
    if (prio || lane) {
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        int p = prio ? atoi(prio) : greatest;
        if (p < greatest) p = greatest;
        if (p > least) p = least;
        se = hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, p);
    } else {
        se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    }
    if (se != hipSuccess) { delete ctx; return fail(PQPS_EHIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
    // sticky status word in mapped host memory: a kernel whose recovery pass gave up sets it, pqps_ctx_sync reports it
    se = hipHostMalloc((void **)&ctx->status_host, kStatusWords * sizeof(uint32_t), hipHostMallocMapped);
    if (se == hipSuccess) { memset(ctx->status_host, 0, kStatusWords * sizeof(uint32_t)); se = hipHostGetDevicePointer((void **)&ctx->status_dev, ctx->status_host, 0); }
    if (se != hipSuccess) { (void)hipStreamDestroy(ctx->stream); delete ctx; return fail(PQPS_EHIP, "status word: %s", hipGetErrorString(se)); }
    hipLaunchKernelGGL(warm_kernel, dim3(1), dim3(1), 0, ctx->stream);
    se = hipGetLastError();
    if (se == hipSuccess) se = hipStreamSynchronize(ctx->stream);
    if (se != hipSuccess) { (void)hipHostFree(ctx->status_host); (void)hipStreamDestroy(ctx->stream); delete ctx; return fail(PQPS_EHIP, "first kernel launch: %s", hipGetErrorString(se)); }
    *out = ctx;
    return PQPS_OK;
}

}  // namespace

extern "C" {

int pqps_ctx_reserve(pqps_ctx *ctx, uint64_t n_rows) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_scratch(ctx, (n_rows + kStepRows - 1) / kStepRows);
    if (rc) return rc;
    // the runtime's own fill / copy kernels are loaded on first use too (~5 ms): use them once here
    HIP_TRY(hipMemsetAsync(ctx->base_slot, 0, 16, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->base_slot + 4, ctx->base_slot, 16, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    // ... and so are its staging buffers for copies to pageable host memory (an ID list download)
    size_t warm = (size_t)ctx->scratch_steps * 64 * sizeof(uint16_t);
    if (warm > ((size_t)8 << 20)) warm = (size_t)8 << 20;
    if (void *host = malloc(warm)) {
        hipError_t e = hipMemcpy(host, ctx->masks, warm, hipMemcpyDeviceToHost);
        free(host);
        if (e != hipSuccess) return fail(PQPS_EHIP, "staging warm-up: %s", hipGetErrorString(e));
    }
    return PQPS_OK;
}

void pqps_ctx_destroy(pqps_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_scratch(ctx);
    if (ctx->sort_tmp) (void)hipFree(ctx->sort_tmp);
    if (ctx->ev_start) {
        for (int i = 0; i < kMaxTimedLaunches; i++) {
            (void)hipEventDestroy(ctx->ev_start[i]); (void)hipEventDestroy(ctx->ev_eval[i]); (void)hipEventDestroy(ctx->ev_stop[i]);
        }
        delete[] ctx->ev_start;
        delete[] ctx->ev_eval;
        delete[] ctx->ev_stop;
        delete[] ctx->stop_is_eval;
    }
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    if (ctx->check_dev) (void)hipFree(ctx->check_dev);
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pqps_ctx_set_timing(pqps_ctx *ctx, int enable) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    if (enable && !ctx->ev_start) {
        ctx->ev_start = new (std::nothrow) hipEvent_t[kMaxTimedLaunches];
        ctx->ev_eval = new (std::nothrow) hipEvent_t[kMaxTimedLaunches];
        ctx->ev_stop = new (std::nothrow) hipEvent_t[kMaxTimedLaunches];
        ctx->stop_is_eval = new (std::nothrow) bool[kMaxTimedLaunches]();
        if (!ctx->ev_start || !ctx->ev_eval || !ctx->ev_stop || !ctx->stop_is_eval) return fail(PQPS_ENOMEM, "out of host memory");
        for (int i = 0; i < kMaxTimedLaunches; i++) {
            HIP_TRY(hipEventCreate(&ctx->ev_start[i]));
            HIP_TRY(hipEventCreate(&ctx->ev_eval[i]));
            HIP_TRY(hipEventCreate(&ctx->ev_stop[i]));
        }
    }
    ctx->timing = enable != 0;
    ctx->timed = 0;
    return PQPS_OK;
}

int pqps_ctx_kernel_time(pqps_ctx *ctx, double *eval_ms, double *total_ms, int *launches) {
    if (!ctx || !eval_ms || !total_ms || !launches) return fail(PQPS_EINVAL, "NULL argument");
    double sum_eval = 0.0, sum_total = 0.0;
    for (int i = 0; i < ctx->timed; i++) {
        hipEvent_t stop = ctx->stop_is_eval[i] ? ctx->ev_eval[i] : ctx->ev_stop[i];
        HIP_TRY(hipEventSynchronize(stop));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_eval[i]));
        sum_eval += ms;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[i], stop));
        sum_total += ms;
    }
    *eval_ms = sum_eval;
    *total_ms = sum_total;
    *launches = ctx->timed;
    ctx->timed = 0;
    return PQPS_OK;
}

// Launch parameters of this context's ID queries, overriding the defaults (value < 0: back to the default).  For A/B runs
// inside one process (scripts/ab_libs.py) and for tests that force a kernel variant on one context only.
int pqps_ctx_set_option(pqps_ctx *ctx, const char *name, long value) {
    if (!ctx || !name) return fail(PQPS_EINVAL, "NULL argument");
    struct { const char *name; long *slot; } opts[] = {
        {"list16", &ctx->opt_list16}, {"list16_min", &ctx->opt_list16_min}, {"list16_min_u8", &ctx->opt_list16_min_u8},
        {"list_max", &ctx->opt_list_max}, {"tiny_max", &ctx->opt_tiny_max},
        {"expand_lag", &ctx->opt_expand_lag}, {"sum_lag", &ctx->opt_sum_lag}, {"tune", &ctx->opt_tune},
    };
    for (auto &o : opts)
        if (strcmp(o.name, name) == 0) { *o.slot = value < 0 ? -1 : value; return PQPS_OK; }
    return fail(PQPS_EINVAL, "unknown option '%s'", name);
}

int pqps_ctx_sync(pqps_ctx *ctx, void *stream) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(pick_stream(ctx, stream)));
    return take_status(ctx, "context");
}

int pqps_device_info(pqps_ctx *ctx, char *name64, int *compute_units, uint64_t *hbm_bytes) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    if (name64) { snprintf(name64, 64, "%s (%s)", prop.name, prop.gcnArchName); }
    if (compute_units) *compute_units = prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (uint64_t)prop.totalGlobalMem;
    return PQPS_OK;
}

int pqps_malloc(pqps_ctx *ctx, size_t bytes, void **dptr) {
    if (!ctx || !dptr) return fail(PQPS_EINVAL, "ctx/dptr is NULL");
    HIP_TRY(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dptr, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(PQPS_ENOMEM, "hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
    return PQPS_OK;
}

int pqps_free(pqps_ctx *ctx, void *dptr) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    if (dptr) HIP_TRY(hipFree(dptr));
    return PQPS_OK;
}

int pqps_malloc_mapped(pqps_ctx *ctx, size_t bytes, void **host_ptr, void **dev_ptr) {
    if (!ctx || !host_ptr || !dev_ptr) return fail(PQPS_EINVAL, "NULL argument");
    HIP_TRY(hipSetDevice(ctx->device));
    void *h = nullptr, *d = nullptr;
    hipError_t e = hipHostMalloc(&h, bytes ? bytes : 64, hipHostMallocMapped);
    if (e != hipSuccess) return fail(PQPS_ENOMEM, "hipHostMalloc(%zu): %s", bytes, hipGetErrorString(e));
    e = hipHostGetDevicePointer(&d, h, 0);
    if (e != hipSuccess) { (void)hipHostFree(h); return fail(PQPS_EHIP, "hipHostGetDevicePointer: %s", hipGetErrorString(e)); }
    memset(h, 0, bytes ? bytes : 64);
    *host_ptr = h;
    *dev_ptr = d;
    return PQPS_OK;
}

int pqps_free_mapped(pqps_ctx *ctx, void *host_ptr) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    if (host_ptr) HIP_TRY(hipHostFree(host_ptr));
    return PQPS_OK;
}

int pqps_memset(pqps_ctx *ctx, void *dptr, int value, size_t bytes, void *stream) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    HIP_TRY(hipMemsetAsync(dptr, value, bytes, pick_stream(ctx, stream)));
    return PQPS_OK;
}

int pqps_upload(pqps_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    hipStream_t s = pick_stream(ctx, stream);
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PQPS_OK;
}

int pqps_download(pqps_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    hipStream_t s = pick_stream(ctx, stream);
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PQPS_OK;
}

int pqps_filter_scan(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                     uint64_t n_rows, uint32_t id_base, const pqps_predicate *pred,
                     uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count) return fail(PQPS_EINVAL, "ctx/out_count is NULL");
    if (!out_ids && out_capacity) return fail(PQPS_EINVAL, "out_ids is NULL");
    if (n_rows > 0xFFFFFFFFull || (uint64_t)id_base + n_rows > 0x100000000ull)
        return fail(PQPS_EINVAL, "row IDs are u32: id_base + n_rows must be <= 2^32");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    return run_filter(ctx, pick_eval<MODE_IDS>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_IDS, false,
                      id_base, out_ids, out_capacity, out_count, pick_stream(ctx, stream));
}

int pqps_filter_count(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count) return fail(PQPS_EINVAL, "ctx/out_count is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    return run_filter(ctx, pick_eval<MODE_COUNT>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_COUNT, false,
                      0, nullptr, 0, out_count, s);
}

int pqps_filter_flags(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred,
                      uint8_t *out_flags, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count || !out_flags) return fail(PQPS_EINVAL, "ctx/out_flags/out_count is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    a.out_flags = out_flags;
    set_streaming(a, cols, n_cols, n_rows);                      // load policy + grid of a scan that outgrows the Infinity Cache
    return run_filter(ctx, a.streaming ? eval_generic_kernel<MODE_FLAGS, false, true> : eval_generic_kernel<MODE_FLAGS, false, false>, a, n_rows, MODE_FLAGS, false,
                      0, nullptr, 0, out_count, s);
}

}  // extern "C"

namespace {

// `key_col` / `keys` (pqps_index_select): the candidates are an index's rows -- the indexed column is read from its sorted keys
int gather_filter(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                  const uint32_t *cand, const uint64_t *range, uint64_t max_candidates,
                  uint32_t id_base, const pqps_predicate *pred,
                  uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream, const void *key_col, const void *keys) {
    if (!ctx || !out_count || !cand || !range) return fail(PQPS_EINVAL, "ctx/cand/range/out_count is NULL");
    if (!out_ids && out_capacity) return fail(PQPS_EINVAL, "out_ids is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = max_candidates;
    a.cand = cand;
    a.range = range;
    a.key_col = keys ? key_col : nullptr;
    a.keys = keys;
    snprintf(g_kernel, sizeof g_kernel, "eval_generic_kernel<MODE_IDS, GATHER=true, NT=false>");
    return run_filter(ctx, eval_generic_kernel<MODE_IDS, true>, a, max_candidates, MODE_IDS, true,
                      id_base, out_ids, out_capacity, out_count, pick_stream(ctx, stream));
}

}  // namespace

extern "C" {

int pqps_filter_gather(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                       const uint32_t *cand, const uint64_t *range, uint64_t max_candidates,
                       uint32_t id_base, const pqps_predicate *pred,
                       uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream) {
    return gather_filter(ctx, cols, n_cols, cand, range, max_candidates, id_base, pred, out_ids, out_capacity, out_count, stream, nullptr, nullptr);
}

}  // extern "C"

namespace {

template <typename K, bool SIGNED>
int index_build_t(pqps_ctx *ctx, const void *col, uint64_t n, uint32_t *perm, void *sorted_keys, hipStream_t s) {
    if (n == 0) return PQPS_OK;
    K *keys_rev = nullptr;
    uint32_t *rows_rev = nullptr;
    pqps_sort::Workspace w;
    HIP_TRY(hipMalloc((void **)&keys_rev, n * sizeof(K)));
    hipError_t e = hipMalloc((void **)&rows_rev, n * sizeof(uint32_t));
    if (e == hipSuccess) e = pqps_sort::workspace_alloc(w, n);
    bool in_a = true;
    if (e == hipSuccess) {
        const uint32_t blocks = (uint32_t)((n + 255) / 256);
        hipLaunchKernelGGL((reverse_gather_kernel<K>), dim3(blocks), dim3(256), 0, s, (const K *)col, n, keys_rev, rows_rev);
        // stable LSD radix sort: equal keys keep the descending-row input order
        e = pqps_sort::sort_pairs<K, SIGNED>(w, keys_rev, rows_rev, (K *)sorted_keys, perm, n, sizeof(K) * 8,
                                             ctx->compute_units, s, &in_a);
        if (e == hipSuccess && in_a) {                        // an even number of passes left the result in the scratch pair
            e = hipMemcpyAsync(sorted_keys, keys_rev, n * sizeof(K), hipMemcpyDeviceToDevice, s);
            if (e == hipSuccess) e = hipMemcpyAsync(perm, rows_rev, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        }
    }
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(keys_rev);
    (void)hipFree(rows_rev);
    pqps_sort::workspace_free(w);
    if (e != hipSuccess) return fail(PQPS_EHIP, "index sort: %s", hipGetErrorString(e));
    if (e2 != hipSuccess) return fail(PQPS_EHIP, "index build sync: %s", hipGetErrorString(e2));
    return PQPS_OK;
}

}  // namespace

extern "C" {

int pqps_index_build(pqps_ctx *ctx, const pqps_column *col, uint64_t n_rows, int key_kind,
                     uint32_t *perm, void *sorted_keys, void *stream) {
    if (!ctx || !col || !perm || !sorted_keys) return fail(PQPS_EINVAL, "NULL argument");
    if (n_rows > 0xFFFFFFFFull) return fail(PQPS_EINVAL, "row IDs are u32");
    hipStream_t s = pick_stream(ctx, stream);
    if (key_kind == 1) {
        if (col->width != 4) return fail(PQPS_EINVAL, "signed keys must be 4 bytes wide");
        return index_build_t<int32_t, true>(ctx, col->data, n_rows, perm, sorted_keys, s);
    }
    switch (col->width) {
    case 1: return index_build_t<uint8_t, false>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 2: return index_build_t<uint16_t, false>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 4: return index_build_t<uint32_t, false>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 8: return index_build_t<uint64_t, false>(ctx, col->data, n_rows, perm, sorted_keys, s);
    default: return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", col->width);
    }
}

}  // extern "C"

namespace {

int launch_probe(const void *sorted_keys, uint32_t width, int key_kind, uint64_t n_rows, uint64_t key_lo, uint64_t key_hi,
                 uint64_t *range, uint64_t *out_count, uint64_t *claim, hipStream_t s) {
    if (key_kind == 1) {
        if (width != 4) return fail(PQPS_EINVAL, "signed keys must be 4 bytes wide");
        hipLaunchKernelGGL((probe_kernel<int32_t>), dim3(1), dim3(64), 0, s, (const int32_t *)sorted_keys, n_rows,
                           (int32_t)(uint32_t)key_lo, (int32_t)(uint32_t)key_hi, range, out_count, claim);
    } else if (width == 1) {
        hipLaunchKernelGGL((probe_kernel<uint8_t>), dim3(1), dim3(64), 0, s, (const uint8_t *)sorted_keys, n_rows,
                           (uint8_t)key_lo, (uint8_t)key_hi, range, out_count, claim);
    } else if (width == 2) {
        hipLaunchKernelGGL((probe_kernel<uint16_t>), dim3(1), dim3(64), 0, s, (const uint16_t *)sorted_keys, n_rows,
                           (uint16_t)key_lo, (uint16_t)key_hi, range, out_count, claim);
    } else if (width == 4) {
        hipLaunchKernelGGL((probe_kernel<uint32_t>), dim3(1), dim3(64), 0, s, (const uint32_t *)sorted_keys, n_rows,
                           (uint32_t)key_lo, (uint32_t)key_hi, range, out_count, claim);
    } else if (width == 8) {
        hipLaunchKernelGGL((probe_kernel<uint64_t>), dim3(1), dim3(64), 0, s, (const uint64_t *)sorted_keys, n_rows,
                           key_lo, key_hi, range, out_count, claim);
    } else {
        return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", width);
    }
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

// Is `pred` nothing but the probed comparison -- ONE leaf on the indexed column whose window is the probe's, accepted
// when it holds?  Then every row the probe finds passes: key in [lo, hi] in the key's order <=> (key - lo) <= hi - lo in
// w-bit arithmetic (two's complement for the signed keys), which is the leaf's own test.
bool probe_implies_predicate(const pqps_column *cols, uint32_t n_cols, const pqps_column *index_column, int key_kind,
                             uint64_t key_lo, uint64_t key_hi, const pqps_predicate *pred) {
    static const bool enabled = [] { const char *e = getenv("PQPS_INDEX_COPY"); return !e || atoi(e) != 0; }();   // 0 (tests): always evaluate
    if (!enabled || !pred || pred->n_leaves != 1) return false;
    const pqps_leaf &lf = pred->leaf[0];
    if (lf.column >= n_cols || cols[lf.column].data != index_column->data || cols[lf.column].width != index_column->width) return false;
    const uint32_t w = index_column->width;
    const uint64_t mask = w >= 8 ? ~0ull : ((1ull << (8 * w)) - 1ull);
    if (key_kind == 1) { if ((int32_t)(uint32_t)key_lo > (int32_t)(uint32_t)key_hi) return false; }
    else if ((key_lo & mask) > (key_hi & mask)) return false;      // an empty window: left to the filter
    if (((lf.lo ^ key_lo) & mask) != 0 || ((lf.span ^ (key_hi - key_lo)) & mask) != 0) return false;
    return ((pred->truth >> (1u ^ (lf.negate & 1u))) & 1ull) != 0;     // candidates: raw window hit = 1
}

}  // namespace

extern "C" {

int pqps_index_probe(pqps_ctx *ctx, const void *sorted_keys, uint32_t width, int key_kind,
                     uint64_t n_rows, uint64_t key_lo, uint64_t key_hi, uint64_t *range, void *stream) {
    if (!ctx || !range || (!sorted_keys && n_rows)) return fail(PQPS_EINVAL, "NULL argument");
    return launch_probe(sorted_keys, width, key_kind, n_rows, key_lo, key_hi, range, nullptr, nullptr, pick_stream(ctx, stream));
}

int pqps_index_select(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols, const pqps_column *index_column,
                      const uint32_t *perm, const void *sorted_keys, int key_kind, uint64_t n_rows,
                      uint64_t key_lo, uint64_t key_hi, uint32_t id_base, const pqps_predicate *pred,
                      uint64_t *range, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream) {
    if (!ctx || !range || !index_column || !perm || !out_count || (!sorted_keys && n_rows)) return fail(PQPS_EINVAL, "NULL argument");
    if (!out_ids && out_capacity) return fail(PQPS_EINVAL, "out_ids is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    if (!probe_implies_predicate(cols, n_cols, index_column, key_kind, key_lo, key_hi, pred)) {
        rc = launch_probe(sorted_keys, index_column->width, key_kind, n_rows, key_lo, key_hi, range, nullptr, nullptr, s);
        if (rc) return rc;
        return gather_filter(ctx, cols, n_cols, perm, range, n_rows, id_base, pred, out_ids, out_capacity, out_count, stream, index_column->data, sorted_keys);
    }
    rc = ensure_scratch(ctx, 1);                                 // (base_slot: where the probe leaves the first output slot of its rows)
    if (rc) return rc;
    rc = launch_probe(sorted_keys, index_column->width, key_kind, n_rows, key_lo, key_hi, range, out_count, ctx->base_slot + 1, s);
    if (rc) return rc;
    // a workgroup per 16 K rows of the table (16 rows per thread and turn), at most 8 per CU: a narrow probe does not pay for a grid it cannot use
    uint64_t blocks = (n_rows + 16383) / 16384;
    const uint64_t most = (uint64_t)ctx->compute_units * 8;
    if (blocks > most) blocks = most;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(append_range_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, perm, range, ctx->base_slot + 1, id_base, out_ids, out_capacity);
    HIP_TRY(hipGetLastError());
    snprintf(g_kernel, sizeof g_kernel, "append_range_kernel (the probe's rows copied: the predicate is the probed comparison)");
    return PQPS_OK;
}

void pqps_partition(uint64_t n_rows, int world, int rank, uint64_t *start, uint64_t *count) {
    const uint64_t base = n_rows / (uint64_t)world, rem = n_rows % (uint64_t)world;
    if ((uint64_t)rank < rem) { *count = base + 1; *start = (uint64_t)rank * (base + 1); }
    else { *count = base; *start = rem * (base + 1) + ((uint64_t)rank - rem) * base; }
}

// Per-user tables of the synthetic schema: lognormal(0,1) activity weights as
// a 32-bit cumulative table, and one shell per user (bash .7 / zsh .2 / fish
// .05 / sh .05 -> rank in {"bash","fish","sh","zsh"}).
void pqps_synth_user_tables(uint64_t seed, uint32_t *cdf_host, uint8_t *shell_host) {
    static double w[PQPS_SYNTH_USERS];
    double total = 0.0;
    for (int i = 0; i < PQPS_SYNTH_USERS; i++) {
        const uint64_t a = synth_mix(seed ^ 0xA5A5A5A5ull, (uint64_t)i, 7), b = synth_mix(seed ^ 0x5A5A5A5Aull, (uint64_t)i, 8);
        const double u1 = ((double)(a >> 11) + 1.0) / 9007199254740993.0;
        const double u2 = (double)(b >> 11) / 9007199254740992.0;
        const double z = sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
        w[i] = exp(z);
        total += w[i];
        const uint32_t sdraw = (uint32_t)(synth_mix(seed, (uint64_t)i, 9) >> 32);
        // bash < .7 ; zsh < .9 ; fish < .95 ; sh
        shell_host[i] = sdraw < 3006477107u ? 0 : sdraw < 3865470566u ? 3 : sdraw < 4080218931u ? 1 : 2;
    }
    double run = 0.0;
    for (int i = 0; i < PQPS_SYNTH_USERS; i++) {
        run += w[i];
        double c = run / total * 4294967295.0;
        if (c > 4294967295.0) c = 4294967295.0;
        cdf_host[i] = (uint32_t)c;
    }
    cdf_host[PQPS_SYNTH_USERS - 1] = 0xFFFFFFFFu;
}

int pqps_synth_generate(pqps_ctx *ctx, uint64_t seed, uint64_t row0, uint64_t n,
                        const uint32_t *user_cdf_dev, const uint8_t *user_shell_dev,
                        const pqps_synth_cols *out, void *stream) {
    if (!ctx || !out || !user_cdf_dev || !user_shell_dev) return fail(PQPS_EINVAL, "NULL argument");
    if (n == 0) return PQPS_OK;
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t want = (n + 255) / 256;
    const uint32_t blocks = (uint32_t)(want < 65536 ? want : 65536);
    hipLaunchKernelGGL(synth_kernel, dim3(blocks), dim3(256), 0, s, seed, row0, n, user_cdf_dev, user_shell_dev, *out);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

void pqps_synth_generate_host(uint64_t seed, uint64_t row0, uint64_t n,
                              const uint32_t *user_cdf, const uint8_t *user_shell,
                              const pqps_synth_cols *out) {
    for (uint64_t i = 0; i < n; i++) synth_store(*out, i, synth_row(seed, row0 + i, user_cdf, user_shell));
}

int pqps_bump_codes(pqps_ctx *ctx, void *codes, uint32_t width, uint64_t n_rows, uint32_t threshold, void *stream) {
    if (!ctx || (!codes && n_rows)) return fail(PQPS_EINVAL, "NULL argument");
    if (n_rows == 0) return PQPS_OK;
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t want = (n_rows + 255) / 256;
    const uint32_t blocks = (uint32_t)(want < 8192 ? want : 8192);
    if (width == 1) hipLaunchKernelGGL((bump_codes_kernel<uint8_t>), dim3(blocks), dim3(256), 0, s, (uint8_t *)codes, n_rows, threshold);
    else if (width == 2) hipLaunchKernelGGL((bump_codes_kernel<uint16_t>), dim3(blocks), dim3(256), 0, s, (uint16_t *)codes, n_rows, threshold);
    else if (width == 4) hipLaunchKernelGGL((bump_codes_kernel<uint32_t>), dim3(blocks), dim3(256), 0, s, (uint32_t *)codes, n_rows, threshold);
    else return fail(PQPS_EINVAL, "code width %u not in {1,2,4}", width);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_compact_rows(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                      const uint8_t *delete_flags, uint64_t *kept_out, void *stream) {
    if (!ctx || !cols || !delete_flags || !kept_out) return fail(PQPS_EINVAL, "NULL argument");
    if (n_cols == 0 || n_cols > PQPS_MAX_COLUMNS) return fail(PQPS_EINVAL, "n_cols %u out of range", n_cols);
    if (n_rows > 0xFFFFFFFFull) return fail(PQPS_EINVAL, "row IDs are u32");
    for (uint32_t c = 0; c < n_cols; c++) {
        const uint32_t w = cols[c].width;
        if (!cols[c].data || (w != 1 && w != 2 && w != 4 && w != 8)) return fail(PQPS_EINVAL, "column %u: bad width / NULL data", c);
    }
    *kept_out = n_rows;
    if (n_rows == 0) return PQPS_OK;
    hipStream_t s = pick_stream(ctx, stream);
    // keep list = ascending IDs of the rows whose flag is 0: the ordinary scan over a one-column table
    pqps_column fcol = {delete_flags, 1, 0};
    pqps_predicate keep_pred;
    memset(&keep_pred, 0, sizeof keep_pred);
    keep_pred.n_leaves = 1; keep_pred.n_columns = 1; keep_pred.truth = 0x2;
    keep_pred.leaf[0].column = 0; keep_pred.leaf[0].lo = 0; keep_pred.leaf[0].span = 0;
    keep_pred.on_true[0] = PQPS_ACCEPT; keep_pred.on_false[0] = PQPS_REJECT; keep_pred.order[0] = 0;
    uint32_t *keep = nullptr;
    uint64_t *count_dev = nullptr;
    void *tmp = nullptr;
    HIP_TRY(hipMalloc((void **)&keep, n_rows * sizeof(uint32_t)));
    hipError_t e = hipMalloc((void **)&count_dev, 64);
    if (e == hipSuccess) e = hipMalloc(&tmp, n_rows * 8);
    int rc = e == hipSuccess ? PQPS_OK : fail(PQPS_EHIP, "compaction scratch: %s", hipGetErrorString(e));
    uint64_t kept = 0;
    if (!rc) rc = pqps_filter_scan(ctx, &fcol, 1, n_rows, 0, &keep_pred, keep, n_rows, count_dev, (void *)s);
    if (!rc) {
        e = hipMemcpyAsync(&kept, count_dev, sizeof kept, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) rc = fail(PQPS_EHIP, "keep count: %s", hipGetErrorString(e));
    }
    if (!rc && kept < n_rows && kept > 0) {
        const uint32_t grid = (uint32_t)((kept + 255) / 256 < (uint64_t)ctx->compute_units * 16 ? (kept + 255) / 256 : (uint64_t)ctx->compute_units * 16);
        for (uint32_t c = 0; c < n_cols && !rc; c++) {
            void *data = const_cast<void *>(cols[c].data);
            switch (cols[c].width) {
            case 1: hipLaunchKernelGGL(gather_rows_kernel<uint8_t>, dim3(grid), dim3(256), 0, s, (const uint8_t *)data, keep, kept, (uint8_t *)tmp); break;
            case 2: hipLaunchKernelGGL(gather_rows_kernel<uint16_t>, dim3(grid), dim3(256), 0, s, (const uint16_t *)data, keep, kept, (uint16_t *)tmp); break;
            case 4: hipLaunchKernelGGL(gather_rows_kernel<uint32_t>, dim3(grid), dim3(256), 0, s, (const uint32_t *)data, keep, kept, (uint32_t *)tmp); break;
            default: hipLaunchKernelGGL(gather_rows_kernel<uint64_t>, dim3(grid), dim3(256), 0, s, (const uint64_t *)data, keep, kept, (uint64_t *)tmp); break;
            }
            e = hipGetLastError();
            if (e == hipSuccess) e = hipMemcpyAsync(data, tmp, kept * cols[c].width, hipMemcpyDeviceToDevice, s);
            if (e != hipSuccess) rc = fail(PQPS_EHIP, "column %u compaction: %s", c, hipGetErrorString(e));
        }
        if (!rc && (e = hipStreamSynchronize(s)) != hipSuccess) rc = fail(PQPS_EHIP, "compaction: %s", hipGetErrorString(e));
    }
    (void)hipFree(keep); (void)hipFree(count_dev); (void)hipFree(tmp);
    if (!rc) *kept_out = kept;
    return rc;
}

int pqps_project_column(pqps_ctx *ctx, const pqps_column *col, const uint32_t *ids, const uint64_t *count_dev,
                        uint64_t capacity, uint32_t id_base, void *out, void *stream) {
    if (!ctx || !col || !col->data || !ids || !count_dev || !out) return fail(PQPS_EINVAL, "NULL argument");
    hipStream_t s = pick_stream(ctx, stream);
    uint64_t blocks = (capacity + 255) / 256;
    if (blocks > (uint64_t)ctx->compute_units * 16) blocks = (uint64_t)ctx->compute_units * 16;
    if (blocks == 0) blocks = 1;
    const dim3 g((uint32_t)blocks), b(256);
    switch (col->width) {
    case 1: hipLaunchKernelGGL((project_kernel<uint8_t>), g, b, 0, s, (const uint8_t *)col->data, ids, count_dev, capacity, id_base, (uint8_t *)out); break;
    case 2: hipLaunchKernelGGL((project_kernel<uint16_t>), g, b, 0, s, (const uint16_t *)col->data, ids, count_dev, capacity, id_base, (uint16_t *)out); break;
    case 4: hipLaunchKernelGGL((project_kernel<uint32_t>), g, b, 0, s, (const uint32_t *)col->data, ids, count_dev, capacity, id_base, (uint32_t *)out); break;
    case 8: hipLaunchKernelGGL((project_kernel<uint64_t>), g, b, 0, s, (const uint64_t *)col->data, ids, count_dev, capacity, id_base, (uint64_t *)out); break;
    default: return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", col->width);
    }
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_gather_keys(pqps_ctx *ctx, const pqps_column *col, int key_kind, const uint32_t *ids, const uint64_t *count_dev,
                     uint64_t capacity, uint32_t id_base, uint64_t *keys_out, void *stream) {
    if (!ctx || !col || !col->data || !ids || !count_dev || !keys_out) return fail(PQPS_EINVAL, "NULL argument");
    if (key_kind == 1 && col->width != 4) return fail(PQPS_EINVAL, "signed keys must be 4 bytes wide");
    hipStream_t s = pick_stream(ctx, stream);
    uint64_t blocks = (capacity + 255) / 256;
    if (blocks > (uint64_t)ctx->compute_units * 8) blocks = (uint64_t)ctx->compute_units * 8;
    if (blocks == 0) blocks = 1;
    const dim3 g((uint32_t)blocks), b(256);
    if (key_kind == 1) hipLaunchKernelGGL((gather_keys_kernel<int32_t, true>), g, b, 0, s, (const int32_t *)col->data, ids, count_dev, capacity, id_base, keys_out);
    else switch (col->width) {
    case 1: hipLaunchKernelGGL((gather_keys_kernel<uint8_t, false>), g, b, 0, s, (const uint8_t *)col->data, ids, count_dev, capacity, id_base, keys_out); break;
    case 2: hipLaunchKernelGGL((gather_keys_kernel<uint16_t, false>), g, b, 0, s, (const uint16_t *)col->data, ids, count_dev, capacity, id_base, keys_out); break;
    case 4: hipLaunchKernelGGL((gather_keys_kernel<uint32_t, false>), g, b, 0, s, (const uint32_t *)col->data, ids, count_dev, capacity, id_base, keys_out); break;
    case 8: hipLaunchKernelGGL((gather_keys_kernel<uint64_t, false>), g, b, 0, s, (const uint64_t *)col->data, ids, count_dev, capacity, id_base, keys_out); break;
    default: return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", col->width);
    }
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_merge_index_slots(pqps_ctx *ctx, const uint32_t *slots, const uint64_t *key_slots, uint32_t world,
                           uint64_t slot_stride, uint32_t *merged, uint64_t merged_capacity, uint64_t *totals, void *stream) {
    if (!ctx || !slots || !key_slots || !merged || !totals) return fail(PQPS_EINVAL, "NULL argument");
    if (world == 0 || world > 1024) return fail(PQPS_EINVAL, "world %u out of range", world);
    if (slot_stride <= kSlotHeaderWords || (slot_stride & 1u)) return fail(PQPS_EINVAL, "slot stride %llu too small or odd", (unsigned long long)slot_stride);
    if (((uintptr_t)slots & 7u) != 0) return fail(PQPS_EINVAL, "slots must be 8-byte aligned");
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t seg_cap = slot_stride - kSlotHeaderWords, cap = (uint64_t)world * seg_cap;
    uint32_t *ids_a = nullptr, *ids_b = nullptr;
    uint64_t *keys_a = nullptr, *keys_b = nullptr;
    void *tmp = nullptr;
    int rc = PQPS_OK;
    hipError_t e = hipMalloc((void **)&ids_a, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&ids_b, cap * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void **)&keys_a, cap * sizeof(uint64_t));
    if (e == hipSuccess) e = hipMalloc((void **)&keys_b, cap * sizeof(uint64_t));
    uint64_t host_totals[2] = {0, 0};
    if (e == hipSuccess) {
        uint64_t bx = (slot_stride + 255) / 256;
        if (bx > 1024) bx = 1024;
        hipLaunchKernelGGL(compact_index_slots_kernel, dim3((uint32_t)bx, world), dim3(256), 0, s, slots, key_slots, world,
                           slot_stride, ids_a, keys_a, cap, totals);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host_totals, totals, sizeof host_totals, hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
    }
    const uint64_t n = host_totals[0] < merged_capacity ? host_totals[0] : merged_capacity;
    if (e == hipSuccess && host_totals[0] > merged_capacity)
        rc = fail(PQPS_EOVERFLOW, "merged capacity %llu too small for %llu IDs", (unsigned long long)merged_capacity, (unsigned long long)host_totals[0]);
    if (e == hipSuccess && rc == PQPS_OK && n > 0) {
        // global leaf order = (key ascending, row descending): two stable sorts, minor criterion first.
        // pass 1 sorts by ~id (ascending ~id = descending id) carrying a permutation, pass 2 by key.
        pqps_sort::Workspace w;
        uint32_t *nid_a = nullptr, *nid_b = nullptr, *pos_a = nullptr, *pos_b = nullptr;
        e = pqps_sort::workspace_alloc(w, n);
        if (e == hipSuccess) e = hipMalloc((void **)&nid_a, n * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&nid_b, n * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&pos_a, n * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&pos_b, n * 4);
        if (e == hipSuccess) {
            const uint32_t blocks = (uint32_t)((n + 255) / 256);
            hipLaunchKernelGGL(negate_iota_kernel, dim3(blocks), dim3(256), 0, s, ids_a, n, nid_a, pos_a);
            bool in_a = true;
            e = pqps_sort::sort_pairs<uint32_t, false>(w, nid_a, pos_a, nid_b, pos_b, n, 32, ctx->compute_units, s, &in_a);
            const uint32_t *order = in_a ? pos_a : pos_b;         // positions of the compacted list, by descending id
            if (e == hipSuccess) {
                hipLaunchKernelGGL(permute_pairs_kernel, dim3(blocks), dim3(256), 0, s, order, n, keys_a, ids_a, keys_b, ids_b);
                bool in_b = true;                                 // (keys_b, ids_b) is this sort's "a" pair
                e = pqps_sort::sort_pairs<uint64_t, false>(w, keys_b, ids_b, keys_a, ids_a, n, 64, ctx->compute_units, s, &in_b);
                if (e == hipSuccess)
                    e = hipMemcpyAsync(merged, in_b ? ids_b : ids_a, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
            }
            if (e == hipSuccess) e = hipStreamSynchronize(s);
        }
        (void)hipFree(nid_a); (void)hipFree(nid_b); (void)hipFree(pos_a); (void)hipFree(pos_b);
        pqps_sort::workspace_free(w);
    }
    (void)hipFree(ids_a); (void)hipFree(ids_b); (void)hipFree(keys_a); (void)hipFree(keys_b); (void)hipFree(tmp);
    if (e != hipSuccess) return fail(PQPS_EHIP, "index slot merge: %s", hipGetErrorString(e));
    return rc;
}

int pqps_merge_slots(pqps_ctx *ctx, const uint32_t *slots, uint32_t world, uint64_t slot_stride,
                     uint32_t *merged, uint64_t merged_capacity, uint64_t *totals, void *stream) {
    if (!ctx || !slots || !merged) return fail(PQPS_EINVAL, "NULL argument");
    if (world == 0 || world > 1024) return fail(PQPS_EINVAL, "world %u out of range", world);
    if (slot_stride <= kSlotHeaderWords || (slot_stride & 1u)) return fail(PQPS_EINVAL, "slot stride %llu too small or odd", (unsigned long long)slot_stride);
    if (((uintptr_t)slots & 7u) != 0) return fail(PQPS_EINVAL, "slots must be 8-byte aligned");
    hipStream_t s = pick_stream(ctx, stream);
    uint64_t bx = (slot_stride + 255) / 256;
    if (bx > 1024) bx = 1024;
    hipLaunchKernelGGL(merge_slots_kernel, dim3((uint32_t)bx, world), dim3(256), 0, s,
                       slots, world, slot_stride, merged, merged_capacity, totals);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

// ---- multi-GPU SELECT: shard scan + all-gatherv of the matching row IDs over RCCL -----------------
// The exchange step of engine/mpi/executeEngine-mpi.c:703-768 with its own shape kept: MPI_Allgather of the
// per-rank sizes (:753), displacements = exclusive prefix (:758-762), MPI_Allgatherv of the payload (:765).
// RCCL has no all-gatherv: the sizes travel in an 8-byte-per-rank ncclAllGather, the payload as one group of
// ncclSend / ncclRecv of exactly count[r] IDs to / from every peer, landing at its displacement -- no
// padding on the wire, no compaction pass, and no slot that could be too small.  The sizes have to reach the
// host between the two (they are arguments of the send / recv calls); that round trip is hidden by finishing
// query k's exchange only after query k+1's scan has been enqueued (see pqps_exchange_select).
// RCCL is resolved at run time from the library the caller names (the process's torch build ships one;
// /opt/rocm/lib/librccl.so is the system one), so the single-GPU engine does not link against it.
}  // extern "C"

namespace {

struct RcclApi {
    void *dl;
    int (*GetUniqueId)(void *id);
    int (*CommInitRank)(void **comm, int nranks, pqps_rccl_id id, int rank);
    int (*AllGather)(const void *send, void *recv, size_t count, int dtype, void *comm, hipStream_t s);
    int (*AllReduce)(const void *send, void *recv, size_t count, int dtype, int op, void *comm, hipStream_t s);
    int (*Send)(const void *send, size_t count, int dtype, int peer, void *comm, hipStream_t s);
    int (*Recv)(void *recv, size_t count, int dtype, int peer, void *comm, hipStream_t s);
    int (*GroupStart)(void);
    int (*GroupEnd)(void);
    int (*CommDestroy)(void *comm);
    int (*CommAbort)(void *comm);        // optional
    const char *(*GetErrorString)(int rc);
};

constexpr int kRcclUint32 = 3;     // ncclUint32 (rccl.h: ncclDataType_t)
constexpr int kRcclUint64 = 5;     // ncclUint64
constexpr int kRcclSum = 0;        // ncclSum (ncclRedOp_t)

int load_rccl(const char *path, RcclApi *api) {
    if (!path || !*path) return fail(PQPS_EINVAL, "RCCL library path is empty");
    api->dl = dlopen(path, RTLD_NOW | RTLD_LOCAL);
    if (!api->dl) return fail(PQPS_EHIP, "dlopen(%s): %s", path, dlerror());
    struct { const char *name; void **slot; } syms[] = {
        {"ncclGetUniqueId", (void **)&api->GetUniqueId},   {"ncclCommInitRank", (void **)&api->CommInitRank},
        {"ncclAllGather", (void **)&api->AllGather},       {"ncclCommDestroy", (void **)&api->CommDestroy},
        {"ncclAllReduce", (void **)&api->AllReduce},       {"ncclSend", (void **)&api->Send},
        {"ncclRecv", (void **)&api->Recv},                 {"ncclGroupStart", (void **)&api->GroupStart},
        {"ncclGroupEnd", (void **)&api->GroupEnd},
        {"ncclGetErrorString", (void **)&api->GetErrorString},
    };
    for (auto &sy : syms) {
        *sy.slot = dlsym(api->dl, sy.name);
        if (!*sy.slot) return fail(PQPS_EHIP, "%s: symbol %s not found", path, sy.name);
    }
    *(void **)&api->CommAbort = dlsym(api->dl, "ncclCommAbort");
    return PQPS_OK;
}

enum : uint8_t { kSlotIdle = 0, kSlotSizesInFlight = 1, kSlotDone = 2, kSlotCount = 3 };
constexpr uint32_t kExchangeLanes = 2;      // scans in flight, each whole on a stream of its own (as in pqps_qstream)
constexpr int kRcclUint8 = 1;               // ncclUint8: the compact payload travels as bytes

}  // namespace

extern "C" {

struct pqps_exchange {
    pqps_ctx *ctx;
    RcclApi rccl;
    void *comm;
    uint32_t world, rank, ring;
    uint64_t cap, stride;            // IDs this rank's slot holds; u32 words per slot (header + IDs)
    uint64_t *caps;                  // [world] every rank's `cap` (they may differ: shards differ by a row)
    hipStream_t stream;              // the exchange stream
    uint32_t *local;                 // [ring][stride]   [u64 count][u64 reserved][IDs] of this rank
    uint64_t *hdr_dev;               // [ring][4]        this rank's words of the sizes all-gather: reported count, rows, id_base, format
    uint64_t *sizes_dev;             // [ring][world][4] gathered headers
    uint64_t *sizes_host;            // [ring][world][4] ... on the host (pinned)
    uint64_t *sizes_host_dev;        //                  ... as the device addresses it (the eager unpack kernel writes the headers there itself)
    uint8_t **wire_out;              // [ring]           this rank's payload in compact form (wire_pack_kernel), grown to the rows of a call
    uint64_t *wire_out_cap;
    uint8_t **wire_in;               // [ring]           the peers' compact payloads as received, grown to what a query needs
    uint64_t *wire_in_cap;
    uint32_t **merged;               // [ring]           the gathered list, grown to what a query needs
    uint64_t *merged_cap;
    uint64_t *totals;                // [ring][2]        device: COUNT(*) result
    uint64_t *totals_host;           // [ring][2]        merged / reported IDs of a SELECT slot
    hipEvent_t *scan_done, *k1_done, *sizes_done, *merge_done;
    hipEvent_t joined;               // what the caller's stream held when the queries began
    hipEvent_t fence;                // pqps_exchange_sync: the end of a stream, awaited with a bound
    bool ordered;                    // the lanes already wait for the caller's stream
    pqps_ctx **child;                // [kExchangeLanes] scratch + HIP stream of the scans in flight (see pqps_qstream)
    uint8_t *state;
    uint64_t *issued;                // [ring] call number that last used the slot
    uint64_t calls;
    uint64_t wait_ns;                // host time spent waiting for a slot to come free
    uint64_t sizes_wait_ns;          // host time spent waiting for the sizes of a query
    bool compact;                    // answers that gain from it travel in compact form (PQPS_EXCHANGE_COMPACT=0: always u32)
    uint64_t wire_bytes_in, u32_bytes_in;    // payload this rank has received: as it travelled / as u32 IDs would have
    uint64_t eager_want, eager_ids;  // IDs a rank's block of the sizes all-gather has room for (PQPS_EXCHANGE_EAGER_IDS; 0: none), as agreed at connect
    uint64_t eager_block;            // bytes of such a block: header + eager_ids IDs
    uint8_t *eager_out, *eager_in;   // [ring][block] this rank's block, [ring][world][block] the gathered ones
    uint64_t *caps_dev;              // [world] `caps` for the unpack kernel
    bool eager_next;                 // the next SELECT gathers eager blocks: the last answer whose sizes were read fitted (every rank sees the same)
    bool *eager_slot;                // [ring] the slot's query did
    uint64_t eager_queries, select_queries;   // SELECTs finished in the sizes all-gather alone / SELECTs finished
    double timeout_s;                // bound of every host wait of the exchange (PQPS_EXCHANGE_TIMEOUT_S, default 30; 0: none)
    bool dead;                       // a wait ran out (or a rank could not receive): the communicator is aborted, every call fails
};

int pqps_exchange_unique_id(const char *rccl_library, pqps_rccl_id *id) {
    if (!id) return fail(PQPS_EINVAL, "id is NULL");
    RcclApi api{};
    int rc = load_rccl(rccl_library, &api);
    if (rc) return rc;
    int nrc = api.GetUniqueId(id);
    if (nrc) return fail(PQPS_EHIP, "ncclGetUniqueId: %s", api.GetErrorString(nrc));
    return PQPS_OK;
}

int pqps_exchange_destroy(pqps_exchange *x) {
    if (!x) return PQPS_OK;
    (void)hipSetDevice(x->ctx->device);
    // a dead exchange may still have a collective of the aborted communicator in its stream: no unbounded wait for it
    if (x->stream && !x->dead) (void)hipStreamSynchronize(x->stream);
    if (x->comm) { if (x->dead && x->rccl.CommAbort) (void)x->rccl.CommAbort(x->comm); else (void)x->rccl.CommDestroy(x->comm); x->comm = nullptr; }
    if (x->stream && x->dead) (void)hipStreamSynchronize(x->stream);     // (the abort has ended what was stuck)
    for (uint32_t i = 0; i < x->ring; i++) {
        if (x->scan_done && x->scan_done[i]) (void)hipEventDestroy(x->scan_done[i]);
        if (x->k1_done && x->k1_done[i]) (void)hipEventDestroy(x->k1_done[i]);
        if (x->sizes_done && x->sizes_done[i]) (void)hipEventDestroy(x->sizes_done[i]);
        if (x->merge_done && x->merge_done[i]) (void)hipEventDestroy(x->merge_done[i]);
        if (x->merged && x->merged[i]) (void)hipFree(x->merged[i]);
        if (x->wire_out && x->wire_out[i]) (void)hipFree(x->wire_out[i]);
        if (x->wire_in && x->wire_in[i]) (void)hipFree(x->wire_in[i]);
    }
    for (uint32_t i = 0; i < kExchangeLanes; i++)
        if (x->child && x->child[i]) { (void)hipStreamSynchronize(x->child[i]->stream); pqps_ctx_destroy(x->child[i]); }
    if (x->joined) (void)hipEventDestroy(x->joined);
    if (x->fence) (void)hipEventDestroy(x->fence);
    delete[] x->scan_done; delete[] x->k1_done; delete[] x->sizes_done; delete[] x->merge_done; delete[] x->child;
    delete[] x->state; delete[] x->issued; delete[] x->merged; delete[] x->merged_cap; delete[] x->totals_host; delete[] x->caps;
    delete[] x->wire_out; delete[] x->wire_out_cap; delete[] x->wire_in; delete[] x->wire_in_cap;
    if (x->local) (void)hipFree(x->local);
    if (x->eager_out) (void)hipFree(x->eager_out);
    if (x->eager_in) (void)hipFree(x->eager_in);
    if (x->caps_dev) (void)hipFree(x->caps_dev);
    delete[] x->eager_slot;
    if (x->hdr_dev) (void)hipFree(x->hdr_dev);
    if (x->sizes_dev) (void)hipFree(x->sizes_dev);
    if (x->sizes_host) (void)hipHostFree(x->sizes_host);
    if (x->totals) (void)hipFree(x->totals);
    if (x->stream) (void)hipStreamDestroy(x->stream);
    delete x;
    return PQPS_OK;
}

int pqps_exchange_prepare(pqps_ctx *ctx, const char *rccl_library, uint32_t world, uint32_t rank,
                          uint64_t slot_capacity, uint32_t ring, pqps_exchange **out) {
    if (!ctx || !out) return fail(PQPS_EINVAL, "NULL argument");
    if (world == 0 || world > 1024 || rank >= world) return fail(PQPS_EINVAL, "rank %u / world %u out of range", rank, world);
    if (ring == 0 || ring > 64) return fail(PQPS_EINVAL, "ring %u out of range (1..64)", ring);
    if (slot_capacity == 0 || slot_capacity > 0xFFFFFFFFull) return fail(PQPS_EINVAL, "slot capacity out of range");
    pqps_exchange *x = new (std::nothrow) pqps_exchange();
    if (!x) return fail(PQPS_ENOMEM, "out of host memory");
    x->ctx = ctx; x->world = world; x->rank = rank; x->ring = ring;
    x->cap = (slot_capacity + 1) & ~1ull;                      // keeps every slot 8-byte aligned
    x->stride = x->cap + kSlotHeaderWords;
    { const char *e = getenv("PQPS_EXCHANGE_COMPACT"); x->compact = !e || atoi(e) != 0; }
    { const char *e = getenv("PQPS_EXCHANGE_TIMEOUT_S"); x->timeout_s = e ? atof(e) : 30.0; if (x->timeout_s < 0) x->timeout_s = 0; }
    { const char *e = getenv("PQPS_EXCHANGE_EAGER_IDS"); x->eager_want = e ? strtoull(e, nullptr, 10) : kEagerIdsDefault; }
    x->eager_slot = new bool[ring]();
    int rc = load_rccl(rccl_library, &x->rccl);
    if (rc) { pqps_exchange_destroy(x); return rc; }
#define X_TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { pqps_exchange_destroy(x); \
        return fail(PQPS_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); } } while (0)
    X_TRY(hipSetDevice(ctx->device));
    X_TRY(hipStreamCreateWithFlags(&x->stream, hipStreamNonBlocking));
    X_TRY(hipMalloc((void **)&x->local, (size_t)ring * x->stride * 4));
    X_TRY(hipMalloc((void **)&x->hdr_dev, (size_t)ring * kWireHeaderWords * sizeof(uint64_t)));
    X_TRY(hipMalloc((void **)&x->sizes_dev, (size_t)ring * world * kWireHeaderWords * sizeof(uint64_t)));
    X_TRY(hipHostMalloc((void **)&x->sizes_host, (size_t)ring * world * kWireHeaderWords * sizeof(uint64_t), hipHostMallocDefault));
    X_TRY(hipMalloc((void **)&x->totals, (size_t)ring * 2 * sizeof(uint64_t)));
    // (on the exchange's own stream and awaited below: a fill on the null stream is not ordered against this non-blocking stream,
    //  and the first thing connect puts on it is this rank's capacity INTO `local` -- found by the process-loopback rehearsal,
    //  where one rank of three announced a capacity of 0)
    X_TRY(hipMemsetAsync(x->local, 0, (size_t)ring * x->stride * 4, x->stream));
    X_TRY(hipMemsetAsync(x->hdr_dev, 0, (size_t)ring * kWireHeaderWords * sizeof(uint64_t), x->stream));
    X_TRY(hipMemsetAsync(x->totals, 0, (size_t)ring * 2 * sizeof(uint64_t), x->stream));
    X_TRY(hipStreamSynchronize(x->stream));
    memset(x->sizes_host, 0, (size_t)ring * world * kWireHeaderWords * sizeof(uint64_t));
    X_TRY(hipHostGetDevicePointer((void **)&x->sizes_host_dev, x->sizes_host, 0));
    x->scan_done = new hipEvent_t[ring](); x->k1_done = new hipEvent_t[ring](); x->sizes_done = new hipEvent_t[ring]();
    x->merge_done = new hipEvent_t[ring]();
    x->child = new pqps_ctx *[kExchangeLanes](); x->state = new uint8_t[ring](); x->issued = new uint64_t[ring]();
    X_TRY(hipEventCreateWithFlags(&x->joined, hipEventDisableTiming));
    X_TRY(hipEventCreateWithFlags(&x->fence, hipEventDisableTiming));
    for (uint32_t i = 0; i < kExchangeLanes; i++)
        if (create_ctx(ctx->device, true, &x->child[i]) != PQPS_OK) { pqps_exchange_destroy(x); return PQPS_EHIP; }
    x->merged = new uint32_t *[ring](); x->merged_cap = new uint64_t[ring](); x->totals_host = new uint64_t[2 * (size_t)ring]();
    x->wire_out = new uint8_t *[ring](); x->wire_out_cap = new uint64_t[ring](); x->wire_in = new uint8_t *[ring](); x->wire_in_cap = new uint64_t[ring]();
    x->caps = new uint64_t[world]();
    for (uint32_t i = 0; i < ring; i++) {
        X_TRY(hipEventCreateWithFlags(&x->scan_done[i], hipEventDisableTiming));
        X_TRY(hipEventCreateWithFlags(&x->k1_done[i], hipEventDisableTiming));
        X_TRY(hipEventCreateWithFlags(&x->sizes_done[i], hipEventDisableTiming));
        X_TRY(hipEventCreateWithFlags(&x->merge_done[i], hipEventDisableTiming));
        // a first allocation for the gathered list; a query that needs more grows it (never too small)
        x->merged_cap[i] = x->cap < ((uint64_t)1 << 20) ? x->cap : ((uint64_t)1 << 20);
        X_TRY(hipMalloc((void **)&x->merged[i], x->merged_cap[i] * 4));
    }
#undef X_TRY
    *out = x;
    return PQPS_OK;
}

}  // extern "C"

namespace {

// Every host wait of the exchange has a deadline.  A collective whose peer never arrives would otherwise hold the process
// -- one that has touched the GPU and cannot be replaced -- for ever: on expiry the communicator is aborted (which ends the
// peers' matching calls with an error, and whatever of this rank's is stuck in the stream), the exchange is marked dead
// and every later call fails at once; the host falls back to its other exchange path (merge.py) or tears down.
int exchange_died(pqps_exchange *x, const char *what) {
    x->dead = true;
    if (x->comm && x->rccl.CommAbort) { (void)x->rccl.CommAbort(x->comm); x->comm = nullptr; }
    return fail(PQPS_ETIMEOUT, "exchange: %s did not finish within %.1f s (a peer is missing or stalled); communicator aborted", what, x->timeout_s);
}

int exchange_wait(pqps_exchange *x, hipEvent_t ev, const char *what) {
    hipError_t e = hipEventQuery(ev);
    if (e == hipSuccess) return PQPS_OK;
    if (e != hipErrorNotReady) return fail(PQPS_EHIP, "exchange: %s: %s", what, hipGetErrorString(e));
    if (x->timeout_s <= 0.0) { HIP_TRY(hipEventSynchronize(ev)); return PQPS_OK; }
    const uint64_t deadline = now_ns() + (uint64_t)(x->timeout_s * 1e9);
    uint32_t spins = 0;
    for (;;) {
        e = hipEventQuery(ev);
        if (e == hipSuccess) return PQPS_OK;
        if (e != hipErrorNotReady) return fail(PQPS_EHIP, "exchange: %s: %s", what, hipGetErrorString(e));
        if (now_ns() > deadline) return exchange_died(x, what);
        if (++spins > 2000) { struct timespec ts = {0, 20000}; nanosleep(&ts, nullptr); }      // (first ~100 us: a tight poll)
    }
}

int exchange_wait_stream(pqps_exchange *x, hipStream_t s, const char *what) {
    HIP_TRY(hipEventRecord(x->fence, s));
    return exchange_wait(x, x->fence, what);
}

#define X_ALIVE(x) do { if ((x)->dead) return fail(PQPS_ETIMEOUT, "exchange is dead: an earlier wait ran out and the communicator was aborted"); } while (0)

// room for `bytes` in one of the slot's wire buffers (the stream is idle for this slot: it was claimed)
int wire_room(pqps_exchange *x, uint8_t **buf, uint64_t *cap, uint64_t bytes) {
    if (bytes <= *cap) return PQPS_OK;
    if (*buf) { int rc = exchange_wait_stream(x, x->stream, "growing a payload buffer"); if (rc) return rc; (void)hipFree(*buf); *buf = nullptr; *cap = 0; }
    const uint64_t want = bytes + bytes / 8 + 4096;
    hipError_t e = hipMalloc((void **)buf, want);
    if (e != hipSuccess) { (void)hipGetLastError(); *buf = nullptr; return fail(PQPS_ENOMEM, "payload buffer of %llu bytes: %s", (unsigned long long)want, hipGetErrorString(e)); }
    *cap = want;
    return PQPS_OK;
}

// Second half of a SELECT slot: the sizes are on the host, the payload moves.  Every rank comes through here
// for the same slots in the same order (the order of its pqps_exchange_* calls).
int exchange_payload(pqps_exchange *x, uint32_t slot) {
    if (x->state[slot] != kSlotSizesInFlight) return PQPS_OK;
    const uint64_t t0 = now_ns();
    int rc = exchange_wait(x, x->sizes_done[slot], "the sizes all-gather");
    x->sizes_wait_ns += now_ns() - t0;
    if (rc) return rc;
    const uint64_t *hdr = x->sizes_host + (uint64_t)slot * x->world * kWireHeaderWords;      // [rank][count, rows, id_base, format]
    auto held = [&](uint32_t r) { const uint64_t c = hdr[r * kWireHeaderWords]; return c < x->caps[r] ? c : x->caps[r]; };   // a rank whose own slot overflowed sends what it holds
    uint64_t total = 0, reported = 0, staged = 0;
    for (uint32_t r = 0; r < x->world; r++) {                   // mpi:758-762
        reported += hdr[r * kWireHeaderWords];
        total += held(r);
        if (r != x->rank && hdr[r * kWireHeaderWords + 3]) staged += (wire_bytes(hdr[r * kWireHeaderWords + 1], held(r)) + 15) & ~15ull;
    }
    x->select_queries++;
    if (x->eager_ids) {                                          // what the next query gathers: every rank reads the same headers at the same point of its call sequence
        bool fits = true;
        for (uint32_t r = 0; r < x->world; r++) if (hdr[r * kWireHeaderWords] > x->eager_ids) fits = false;
        x->eager_next = fits;
        if (fits && x->eager_slot[slot]) {                       // the gathered blocks held the whole answer, and it is in place (eager_unpack_kernel)
            for (uint32_t r = 0; r < x->world; r++)
                if (r != x->rank) { x->wire_bytes_in += held(r) * 4; x->u32_bytes_in += held(r) * 4; }
            x->eager_queries++;
            x->totals_host[2 * slot] = total;
            x->totals_host[2 * slot + 1] = reported;
            x->state[slot] = kSlotDone;                          // (merge_done was recorded with the sizes: nothing of this query is left to enqueue)
            return PQPS_OK;
        }
    }
    int grow = PQPS_OK;
    if (total > x->merged_cap[slot]) {
        // the consumer of this slot's previous result is done with it (the slot was handed out again)
        rc = exchange_wait_stream(x, x->stream, "growing the gathered list");
        if (rc) return rc;
        (void)hipFree(x->merged[slot]);
        x->merged[slot] = nullptr;
        x->merged_cap[slot] = 0;
        uint64_t want = total + total / 4 + 4096;
        hipError_t e = hipMalloc((void **)&x->merged[slot], want * 4);
        if (e != hipSuccess) { (void)hipGetLastError(); want = total; e = hipMalloc((void **)&x->merged[slot], want * 4); }   // exactly what the payload needs
        if (e != hipSuccess) { x->merged[slot] = nullptr; (void)hipGetLastError(); grow = fail(PQPS_ENOMEM, "gathered ID list of %llu entries: %s", (unsigned long long)want, hipGetErrorString(e)); }
        else x->merged_cap[slot] = want;
    }
    if (grow == PQPS_OK && staged) grow = wire_room(x, &x->wire_in[slot], &x->wire_in_cap[slot], staged);
    if (grow != PQPS_OK) {
        // This rank cannot receive.  Its peers are about to enter (or already sit in) the same send / recv group and
        // would wait for this rank's half of it forever: abort the communicator, which ends their calls with an
        // error instead.  The exchange is dead after this; the caller tears it down.
        char why[256];
        snprintf(why, sizeof why, "%s", g_err);
        x->dead = true;
        if (x->world > 1 && x->rccl.CommAbort && x->comm) { (void)x->rccl.CommAbort(x->comm); x->comm = nullptr; }
        x->state[slot] = kSlotIdle;
        return fail(PQPS_ENOMEM, "%s (communicator aborted)", why);
    }
    const uint32_t *mine = x->local + (uint64_t)slot * x->stride + kSlotHeaderWords;
    uint32_t *merged = x->merged[slot];
    int nrc = 0;
    // this rank's own part first: a failure here must not leave an opened group behind
    uint64_t displ = 0;
    for (uint32_t r = 0; r < x->rank; r++) displ += held(r);
    const uint64_t own = held(x->rank);
    const bool own_compact = hdr[x->rank * kWireHeaderWords + 3] != 0;
    const uint64_t own_wire = own_compact ? wire_bytes(hdr[x->rank * kWireHeaderWords + 1], own) : own * 4;
    if (own) HIP_TRY(hipMemcpyAsync(merged + displ, mine, own * 4, hipMemcpyDeviceToDevice, x->stream));
    if (x->world > 1) nrc = x->rccl.GroupStart();
    displ = 0;
    uint64_t at = 0;
    for (uint32_t r = 0; r < x->world && !nrc; r++) {           // mpi:765, as point-to-point pairs
        const uint64_t k = held(r);
        if (r != x->rank) {
            if (own) nrc = own_compact ? x->rccl.Send(x->wire_out[slot], (size_t)own_wire, kRcclUint8, (int)r, x->comm, x->stream)
                                       : x->rccl.Send(mine, (size_t)own, kRcclUint32, (int)r, x->comm, x->stream);
            if (!nrc && k) {
                if (hdr[r * kWireHeaderWords + 3]) {
                    const uint64_t wb = wire_bytes(hdr[r * kWireHeaderWords + 1], k);
                    nrc = x->rccl.Recv(x->wire_in[slot] + at, (size_t)wb, kRcclUint8, (int)r, x->comm, x->stream);
                    at += (wb + 15) & ~15ull;
                    x->wire_bytes_in += wb;
                } else {
                    nrc = x->rccl.Recv(merged + displ, (size_t)k, kRcclUint32, (int)r, x->comm, x->stream);
                    x->wire_bytes_in += k * 4;
                }
                x->u32_bytes_in += k * 4;
            }
        }
        displ += k;
    }
    if (x->world > 1) { const int end = x->rccl.GroupEnd(); if (!nrc) nrc = end; }
    if (nrc) return fail(PQPS_EHIP, "ncclSend / ncclRecv: %s", x->rccl.GetErrorString(nrc));
    // the peers' compact payloads become row IDs at their displacements: one launch for up to kWireManyPeers of them
    displ = 0; at = 0;
    WireExpandMany many;
    many.n = 0;
    uint64_t most_groups = 0;
    auto flush = [&]() -> int {
        if (many.n == 0) return PQPS_OK;
        const uint64_t max_blocks = (uint64_t)x->ctx->compute_units * 8 / many.n + 1;
        const uint64_t bx = most_groups < max_blocks ? most_groups : max_blocks;
        hipLaunchKernelGGL(wire_expand_many_kernel, dim3((uint32_t)(bx ? bx : 1), many.n), dim3(256), 0, x->stream, many);
        HIP_TRY(hipGetLastError());
        many.n = 0;
        most_groups = 0;
        return PQPS_OK;
    };
    for (uint32_t r = 0; r < x->world; r++) {
        const uint64_t k = held(r);
        if (r != x->rank && k && hdr[r * kWireHeaderWords + 3]) {
            const uint64_t rows = hdr[r * kWireHeaderWords + 1], groups = wire_groups(rows);
            auto &d = many.p[many.n++];
            d.wire = x->wire_in[slot] + at; d.rows = rows; d.out = merged + displ; d.id_base = (uint32_t)hdr[r * kWireHeaderWords + 2]; d.pad = 0;
            if (groups > most_groups) most_groups = groups;
            at += (wire_bytes(rows, k) + 15) & ~15ull;
            if (many.n == kWireManyPeers) { const int frc = flush(); if (frc) return frc; }
        }
        displ += k;
    }
    { const int frc = flush(); if (frc) return frc; }
    x->totals_host[2 * slot] = total;
    x->totals_host[2 * slot + 1] = reported;
    HIP_TRY(hipEventRecord(x->merge_done[slot], x->stream));
    x->state[slot] = kSlotDone;
    return PQPS_OK;
}

// Payload phase of every slot whose sizes are in flight and that was issued before call number `before`,
// oldest first.
int exchange_finish_older(pqps_exchange *x, uint64_t before) {
    for (;;) {
        int pick = -1;
        for (uint32_t i = 0; i < x->ring; i++)
            if (x->state[i] == kSlotSizesInFlight && x->issued[i] < before && (pick < 0 || x->issued[i] < x->issued[pick])) pick = (int)i;
        if (pick < 0) return PQPS_OK;
        int rc = exchange_payload(x, (uint32_t)pick);
        if (rc) return rc;
    }
}

// The slot is free again once whatever last used it has finished (a host wait, normally long satisfied).
int exchange_claim(pqps_exchange *x, uint32_t slot) {
    if (x->state[slot] == kSlotSizesInFlight) { int rc = exchange_payload(x, slot); if (rc) return rc; }
    if (x->state[slot] != kSlotIdle) {
        const uint64_t t0 = now_ns();
        const int rc = exchange_wait(x, x->merge_done[slot], "the payload of an earlier query");
        x->wait_ns += now_ns() - t0;
        if (rc) return rc;
    }
    x->state[slot] = kSlotIdle;
    return PQPS_OK;
}

}  // namespace

extern "C" {

int pqps_exchange_connect(pqps_exchange *x, const pqps_rccl_id *id) {
    if (!x || !id) return fail(PQPS_EINVAL, "NULL argument");
    if (x->comm) return fail(PQPS_EINVAL, "exchange is connected already");
    (void)hipSetDevice(x->ctx->device);
    int nrc = x->rccl.CommInitRank(&x->comm, (int)x->world, *id, (int)x->rank);
    if (nrc) {
        x->comm = nullptr;
        return fail(PQPS_EHIP, "ncclCommInitRank(world %u, rank %u): %s", x->world, x->rank, x->rccl.GetErrorString(nrc));
    }
    // every rank's slot capacity, once: a rank whose own slot overflowed sends what it holds, and its peers
    // have to size their receives the same way
    x->caps[x->rank] = x->cap;
    if (x->world > 1) {
        // (with it: the room for eager IDs every rank asks for -- the smallest wins, so ranks started with different settings still agree)
        uint64_t *mine = x->sizes_dev, *all = x->sizes_host;      // slot 0's buffers, not in use yet
        const uint64_t hello[2] = {x->cap, x->eager_want};
        HIP_TRY(hipMemcpyAsync(x->local, hello, sizeof hello, hipMemcpyHostToDevice, x->stream));
        nrc = x->rccl.AllGather(x->local, mine, 2, kRcclUint64, x->comm, x->stream);
        if (nrc) return fail(PQPS_EHIP, "ncclAllGather: %s", x->rccl.GetErrorString(nrc));
        HIP_TRY(hipMemcpyAsync(all, mine, (size_t)x->world * 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
        HIP_TRY(hipMemsetAsync(x->local, 0, sizeof hello, x->stream));
        int rc = exchange_wait_stream(x, x->stream, "the capacities all-gather");
        if (rc) return rc;
        uint64_t eager = ~0ull;
        for (uint32_t r = 0; r < x->world; r++) {
            x->caps[r] = all[2 * r];
            if (all[2 * r + 1] < eager) eager = all[2 * r + 1];
            if (x->caps[r] < eager) eager = x->caps[r];           // (a block never holds more than the smallest slot does)
        }
        memset(all, 0, (size_t)x->world * 2 * sizeof(uint64_t));
        const uint64_t most = ((uint64_t)1 << 20) / x->world;     // the gathered blocks of a query: 4 MB at most
        if (eager > most) eager = most;
        eager &= ~1ull;
        x->eager_ids = eager >= 256 ? eager : 0;
        if (x->eager_ids) {
            x->eager_block = kWireHeaderWords * sizeof(uint64_t) + x->eager_ids * 4;
            HIP_TRY(hipMalloc((void **)&x->eager_out, (size_t)x->ring * x->eager_block));
            HIP_TRY(hipMalloc((void **)&x->eager_in, (size_t)x->ring * x->world * x->eager_block));
            HIP_TRY(hipMalloc((void **)&x->caps_dev, (size_t)x->world * sizeof(uint64_t)));
            HIP_TRY(hipMemcpyAsync(x->caps_dev, x->caps, (size_t)x->world * sizeof(uint64_t), hipMemcpyHostToDevice, x->stream));
            HIP_TRY(hipMemsetAsync(x->eager_out, 0, (size_t)x->ring * x->eager_block, x->stream));
            rc = exchange_wait_stream(x, x->stream, "setting up the eager blocks");
            if (rc) return rc;
            for (uint32_t i = 0; i < x->ring; i++) {              // the gathered list of a slot holds every rank's eager IDs without growing
                if (x->merged_cap[i] >= x->world * x->eager_ids) continue;
                (void)hipFree(x->merged[i]);
                x->merged[i] = nullptr;
                x->merged_cap[i] = 0;
                HIP_TRY(hipMalloc((void **)&x->merged[i], (size_t)x->world * x->eager_ids * 4));
                x->merged_cap[i] = x->world * x->eager_ids;
            }
            x->eager_next = true;
        }
    }
    return PQPS_OK;
}

int pqps_exchange_create(pqps_ctx *ctx, const char *rccl_library, const pqps_rccl_id *id, uint32_t world, uint32_t rank,
                         uint64_t slot_capacity, uint32_t ring, pqps_exchange **out) {
    if (!id || !out) return fail(PQPS_EINVAL, "NULL argument");
    pqps_exchange *x = nullptr;
    int rc = pqps_exchange_prepare(ctx, rccl_library, world, rank, slot_capacity, ring, &x);
    if (rc) return rc;
    rc = pqps_exchange_connect(x, id);
    if (rc) { pqps_exchange_destroy(x); return rc; }
    *out = x;
    return PQPS_OK;
}

int pqps_exchange_select(pqps_exchange *x, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                         const pqps_predicate *pred, uint32_t slot, void *scan_stream) {
    if (!x) return fail(PQPS_EINVAL, "exchange is NULL");
    X_ALIVE(x);
    if (!x->comm) return fail(PQPS_EINVAL, "exchange is not connected");
    if (slot >= x->ring) return fail(PQPS_EINVAL, "slot %u >= ring %u", slot, x->ring);
    if (n_rows > 0xFFFFFFFFull || (uint64_t)id_base + n_rows > 0x100000000ull)
        return fail(PQPS_EINVAL, "row IDs are u32: id_base + n_rows must be <= 2^32");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    rc = exchange_claim(x, slot);
    if (rc) return rc;
    // First the payload of the query `hold` + 1 calls back.  With hold = 2 (a ring of 5 or more) nothing here waits
    // for a scan that can still be running while the two lanes are busy: the sizes needed on the host belong to
    // the scan three calls back, and the slot claimed above was released by a send / recv group that sits in the
    // exchange stream behind the sizes of a scan at least three calls back -- so this query's scan reaches its lane
    // while the lane's previous scan is still running.  A shorter ring holds less back (4: one query, 3 or
    // fewer: none) and the call then waits for a running scan: the lane idles for a copy, a host wake-up and the
    // enqueue (61 - 66 us per S1 query instead of 55, world of 1).  The order of RCCL calls follows from the call
    // sequence alone, so it is the same on every rank.
    const uint64_t hold = x->ring >= 5 ? 2 : (x->ring == 4 ? 1 : 0);
    rc = exchange_finish_older(x, x->calls + 1 - hold);
    if (rc) return rc;
    uint32_t *local = x->local + (uint64_t)slot * x->stride;
    uint64_t *hdr_dev = x->hdr_dev + (uint64_t)slot * kWireHeaderWords;
    uint64_t *sizes_dev = x->sizes_dev + (uint64_t)slot * x->world * kWireHeaderWords;
    uint64_t *sizes_host = x->sizes_host + (uint64_t)slot * x->world * kWireHeaderWords;
    const bool compact = x->compact && x->world > 1;
    if (compact) {                                               // room for this call's payload in compact form
        rc = wire_room(x, &x->wire_out[slot], &x->wire_out_cap[slot], wire_bytes(n_rows, x->cap < n_rows ? x->cap : n_rows));
        if (rc) return rc;
    }
    hipStream_t scan = pick_stream(x->ctx, scan_stream);
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    // the scan whole on one of two lanes (a stream and a scratch of its own: the tail of one query's launch is
    // filled by the scan tiles of the next, see pqps_qstream); everything after it on the exchange stream, behind
    // the launch's own completion event.
    // (While the context records timings, the scan runs on the caller's stream with the context's own
    // scratch, so that the recorded events mean what pqps_ctx_kernel_time documents.)
    const bool timed = x->ctx->timing;
    if (timed) {
        hipEvent_t ev = x->scan_done[slot];
        rc = run_filter(x->ctx, pick_eval<MODE_IDS>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_IDS, false, id_base,
                        local + kSlotHeaderWords, x->cap, (uint64_t *)local, scan, &ev);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(x->stream, ev, 0));
    } else {
        if (!x->ordered) {                                       // what the caller's stream holds (the table, ...) comes first
            HIP_TRY(hipEventRecord(x->joined, scan));
            for (uint32_t i = 0; i < kExchangeLanes; i++) HIP_TRY(hipStreamWaitEvent(x->child[i]->stream, x->joined, 0));
            x->ordered = true;
        }
        pqps_ctx *lane = x->child[x->calls % kExchangeLanes];
        hipEvent_t ev = x->k1_done[slot];
        rc = run_filter(lane, pick_eval<MODE_IDS>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_IDS, false, id_base,
                        local + kSlotHeaderWords, x->cap, (uint64_t *)local, lane->stream, &ev);
        if (rc) return rc;
        HIP_TRY(hipStreamWaitEvent(x->stream, ev, 0));
    }
    // sizes: mpi:753.  They are needed on the host (send / recv counts): a 32-byte-per-rank all-gather -- reported count,
    // the shard's rows and first row (what a receiver needs to rebuild IDs from the compact form), the form the sender
    // chose -- then a copy into pinned memory behind it.  The pack kernel writes this rank's words (and the compact payload).
    x->eager_slot[slot] = false;
    if (x->world > 1) {
        const uint64_t n_max = x->cap < n_rows ? x->cap : n_rows;
        const bool eager = x->eager_ids && x->eager_next;
        uint64_t blocks = compact ? (n_max + 2047) / 2048 : 1;
        if (eager && blocks < 8) blocks = 8;
        const uint64_t max_blocks = (uint64_t)x->ctx->compute_units * 8;
        if (blocks > max_blocks) blocks = max_blocks;
        if (blocks == 0) blocks = 1;
        uint8_t *block_out = eager ? x->eager_out + (uint64_t)slot * x->eager_block : nullptr;
        uint8_t *blocks_in = eager ? x->eager_in + (uint64_t)slot * x->world * x->eager_block : nullptr;
        hipLaunchKernelGGL(wire_pack_kernel, dim3((uint32_t)blocks), dim3(256), 0, x->stream, local, x->cap, n_rows, id_base, compact ? 1 : 0,
                           wire_min_ids(), eager ? (uint64_t *)block_out : hdr_dev, x->wire_out[slot], eager ? x->eager_ids : (uint64_t)0,
                           eager ? (uint32_t *)(block_out + kWireHeaderWords * sizeof(uint64_t)) : nullptr);
        HIP_TRY(hipGetLastError());
        int nrc;
        if (eager) {
            // one collective for the sizes AND, where every rank's list fits its block, the answer (eager_unpack_kernel)
            nrc = x->rccl.AllGather(block_out, blocks_in, (size_t)(x->eager_block / sizeof(uint64_t)), kRcclUint64, x->comm, x->stream);
            if (nrc) return fail(PQPS_EHIP, "ncclAllGather: %s", x->rccl.GetErrorString(nrc));
            // (the headers go straight into the pinned host words: no copy launch behind the kernel)
            hipLaunchKernelGGL(eager_unpack_kernel, dim3(4, x->world), dim3(256), 0, x->stream, blocks_in, x->world, x->eager_block, x->eager_ids,
                               x->caps_dev, x->merged[slot], x->sizes_host_dev + (uint64_t)slot * x->world * kWireHeaderWords);
            HIP_TRY(hipGetLastError());
            x->eager_slot[slot] = true;
        } else {
            nrc = x->rccl.AllGather(hdr_dev, sizes_dev, kWireHeaderWords, kRcclUint64, x->comm, x->stream);
            if (nrc) return fail(PQPS_EHIP, "ncclAllGather: %s", x->rccl.GetErrorString(nrc));
            HIP_TRY(hipMemcpyAsync(sizes_host, sizes_dev, (size_t)x->world * kWireHeaderWords * sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));
        }
    } else {
        HIP_TRY(hipMemcpyAsync(sizes_host, local, sizeof(uint64_t), hipMemcpyDeviceToHost, x->stream));    // (words 1 - 3 stay 0: nobody to tell)
    }
    HIP_TRY(hipEventRecord(x->sizes_done[slot], x->stream));
    // (an eager query may be complete at this point of the stream; if it is not, its payload step records the event again)
    if (x->eager_slot[slot]) HIP_TRY(hipEventRecord(x->merge_done[slot], x->stream));
    x->state[slot] = kSlotSizesInFlight;
    x->issued[slot] = ++x->calls;
    return PQPS_OK;
}

int pqps_exchange_count(pqps_exchange *x, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                        const pqps_predicate *pred, uint32_t slot, void *scan_stream) {
    if (!x) return fail(PQPS_EINVAL, "exchange is NULL");
    X_ALIVE(x);
    if (!x->comm) return fail(PQPS_EINVAL, "exchange is not connected");
    if (slot >= x->ring) return fail(PQPS_EINVAL, "slot %u >= ring %u", slot, x->ring);
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    rc = exchange_claim(x, slot);
    if (rc) return rc;
    rc = exchange_finish_older(x, x->calls + 1);                  // collectives stay in call order on every rank
    if (rc) return rc;
    uint32_t *local = x->local + (uint64_t)slot * x->stride;
    uint64_t *totals = x->totals + 2 * (uint64_t)slot;
    hipStream_t scan = pick_stream(x->ctx, scan_stream);
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    hipEvent_t ev = x->scan_done[slot];
    if (x->ctx->timing) {
        rc = run_filter(x->ctx, pick_eval<MODE_COUNT>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_COUNT, false, 0,
                        nullptr, 0, (uint64_t *)local, scan, &ev);
    } else {                                                     // on one of the two scan lanes, as in pqps_exchange_select
        if (!x->ordered) {
            HIP_TRY(hipEventRecord(x->joined, scan));
            for (uint32_t i = 0; i < kExchangeLanes; i++) HIP_TRY(hipStreamWaitEvent(x->child[i]->stream, x->joined, 0));
            x->ordered = true;
        }
        pqps_ctx *lane = x->child[x->calls % kExchangeLanes];
        rc = run_filter(lane, pick_eval<MODE_COUNT>(cols, n_cols, pred, a, n_rows), a, n_rows, MODE_COUNT, false, 0,
                        nullptr, 0, (uint64_t *)local, lane->stream, &ev);
    }
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(x->stream, ev, 0));
    HIP_TRY(hipMemsetAsync(totals + 1, 0, sizeof(uint64_t), x->stream));
    int nrc = x->rccl.AllReduce(local, totals, 1, kRcclUint64, kRcclSum, x->comm, x->stream);      // mpi:745
    if (nrc) return fail(PQPS_EHIP, "ncclAllReduce: %s", x->rccl.GetErrorString(nrc));
    HIP_TRY(hipEventRecord(x->merge_done[slot], x->stream));
    x->state[slot] = kSlotCount;
    x->issued[slot] = ++x->calls;
    return PQPS_OK;
}

int pqps_exchange_result(pqps_exchange *x, uint32_t slot, const uint32_t **merged_dev, uint64_t *local_count,
                         uint64_t totals[2]) {
    if (!x || !totals) return fail(PQPS_EINVAL, "NULL argument");
    X_ALIVE(x);
    if (slot >= x->ring || x->state[slot] == kSlotIdle) return fail(PQPS_EINVAL, "slot %u holds no result", slot);
    int rc = exchange_finish_older(x, x->issued[slot] + 1);       // up to and including this slot, in call order
    if (rc) return rc;
    rc = exchange_wait(x, x->merge_done[slot], "the payload");
    if (rc) return rc;
    if (x->state[slot] == kSlotCount) {
        HIP_TRY(hipMemcpy(totals, x->totals + 2 * (uint64_t)slot, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (local_count) HIP_TRY(hipMemcpy(local_count, x->local + (uint64_t)slot * x->stride, sizeof(uint64_t), hipMemcpyDeviceToHost));
        if (merged_dev) *merged_dev = nullptr;
        return PQPS_OK;
    }
    totals[0] = x->totals_host[2 * slot];
    totals[1] = x->totals_host[2 * slot + 1];
    if (local_count) *local_count = x->sizes_host[((uint64_t)slot * x->world + x->rank) * kWireHeaderWords];
    if (merged_dev) *merged_dev = x->merged[slot];
    if (totals[1] > totals[0])
        return fail(PQPS_EOVERFLOW, "exchange slot overflow: %llu IDs reported, capacity %llu per rank",
                    (unsigned long long)totals[1], (unsigned long long)x->cap);
    return PQPS_OK;
}

uint64_t pqps_exchange_wait_ns(pqps_exchange *x, int reset) {
    if (!x) return 0;
    const uint64_t w = x->wait_ns + x->sizes_wait_ns;
    if (reset) x->wait_ns = x->sizes_wait_ns = 0;
    return w;
}

// The compact wire form for a host that drives the exchange itself (merge.py over torch.distributed): the same two kernels.
uint64_t pqps_wire_bytes(uint64_t n_rows, uint64_t n_ids) { return wire_bytes(n_rows, n_ids); }
int pqps_wire_pays(uint64_t n_rows, uint64_t n_ids) { return wire_pays(n_rows, n_ids, wire_min_ids()) ? 1 : 0; }

int pqps_wire_pack(pqps_ctx *ctx, const uint32_t *slot, uint64_t capacity, uint64_t n_rows, uint32_t id_base, int enabled,
                   uint64_t *header_dev, void *wire, void *stream) {
    if (!ctx || !slot || !header_dev || (enabled && !wire)) return fail(PQPS_EINVAL, "NULL argument");
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t n_max = capacity < n_rows ? capacity : n_rows;
    uint64_t blocks = enabled ? (n_max + 2047) / 2048 : 1;
    const uint64_t max_blocks = (uint64_t)ctx->compute_units * 8;
    if (blocks > max_blocks) blocks = max_blocks;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(wire_pack_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, slot, capacity, n_rows, id_base, enabled ? 1 : 0, wire_min_ids(), header_dev, (uint8_t *)wire, (uint64_t)0, (uint32_t *)nullptr);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_wire_expand(pqps_ctx *ctx, const void *wire, uint64_t n_rows, uint32_t id_base, uint32_t *out_ids, void *stream) {
    if (!ctx || !wire || !out_ids) return fail(PQPS_EINVAL, "NULL argument");
    hipStream_t s = pick_stream(ctx, stream);
    const uint64_t groups = wire_groups(n_rows), max_blocks = (uint64_t)ctx->compute_units * 8;
    if (groups == 0) return PQPS_OK;
    hipLaunchKernelGGL(wire_expand_kernel, dim3((uint32_t)(groups < max_blocks ? groups : max_blocks)), dim3(256), 0, s, (const uint8_t *)wire, n_rows, id_base, out_ids);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

void pqps_exchange_wire_bytes(pqps_exchange *x, uint64_t out[2], int reset) {
    if (!x || !out) return;
    out[0] = x->wire_bytes_in;
    out[1] = x->u32_bytes_in;
    if (reset) x->wire_bytes_in = x->u32_bytes_in = 0;
}

void pqps_exchange_eager(pqps_exchange *x, uint64_t out[3], int reset) {
    if (!x || !out) return;
    out[0] = x->eager_queries;
    out[1] = x->select_queries;
    out[2] = x->eager_ids;
    if (reset) x->eager_queries = x->select_queries = 0;
}

int pqps_exchange_sync(pqps_exchange *x) {
    if (!x) return fail(PQPS_EINVAL, "exchange is NULL");
    X_ALIVE(x);
    int rc = exchange_finish_older(x, x->calls + 1);
    if (rc) return rc;
    (void)hipSetDevice(x->ctx->device);
    for (uint32_t i = 0; i < kExchangeLanes; i++) {
        rc = exchange_wait_stream(x, x->child[i]->stream, "a scan lane");
        if (rc) return rc;
        const int st = take_status(x->child[i], "exchange");
        if (st) return st;
    }
    rc = exchange_wait_stream(x, x->stream, "the exchange stream");
    if (rc) return rc;
    x->ordered = false;                                          // the caller may have put new work on its stream meanwhile
    return PQPS_OK;
}

// ---- a stream of queries on one GPU: two queries in flight, each whole on a HIP stream of its own --------
// A query's launch ends with a tail the chip is mostly idle in (the last scan tiles drain, the expanders behind
// them wait for sums and hand out the last IDs: 7 - 25 us of a 100 M-row query) and begins with a ramp.  With
// the next query already queued on ANOTHER stream the dispatcher fills the slots the tail leaves free with that
// query's scan tiles.  Measured at 100 M rows, per query: S1 60 -> 52.7 us, Q_A 90 -> 73.5, Q_B 107 -> 90.8, a
// lone u8 column 46 -> 35.1; a third or fourth stream adds nothing (two scans then run side by side for most
// of their time and their access windows interleave).  Each lane has its own scratch; results go to `depth` SLOTS
// (the caller's output buffers: slot k is free again when the query that last used it has finished).
//
// One lane for large tables: from kInterleaveFromGroups groups on (537 M rows) the expanders run among the scan
// tiles, the tail is a few percent of the launch, and two launches side by side cost more than the overlap gains
// (1 G rows: Q_A 736 us per query with two lanes against 690 one at a time, Q_B 960 against 940).  Such queries all
// go to lane 0, i.e. back to back on one stream; the slots still let the host run ahead.
struct pqps_qstream {
    pqps_ctx *ctx;
    uint32_t depth, lanes;
    uint64_t seq;
    pqps_ctx **child;                // [lanes] scratch + HIP stream of the queries in flight
    hipEvent_t *done;                // [depth] the slots' own events
    hipEvent_t *ready;               // [depth] what to wait for: the slot's own event, or the timing recorder's stop event of its launch
    pqps_ctx **ran_on;               // [depth] lane (or the parent context) the slot's query ran on
    uint32_t *ep_lo, *ep_hi;         // [depth] epochs of the slot's ID launches on that context (0, 0: none): whose status words are the slot's
    hipEvent_t joined;               // what the caller's stream held when the stream of queries began
    bool *used;
    bool ordered;                    // the lanes already wait for the caller's stream
    uint64_t wait_ns;                // host time spent waiting for an output buffer to come free
    std::atomic<int> dense;          // the last answers held a quarter of the rows or more: ID queries on one lane (pqps_qstream_hint_answer)
};

int pqps_qstream_destroy(pqps_qstream *q) {
    if (!q) return PQPS_OK;
    for (uint32_t i = 0; i < q->lanes; i++)
        if (q->child && q->child[i]) { (void)hipStreamSynchronize(q->child[i]->stream); pqps_ctx_destroy(q->child[i]); }
    for (uint32_t i = 0; i < q->depth; i++)
        if (q->done && q->done[i]) (void)hipEventDestroy(q->done[i]);
    if (q->joined) (void)hipEventDestroy(q->joined);
    delete[] q->done; delete[] q->ready; delete[] q->ran_on; delete[] q->child; delete[] q->used; delete[] q->ep_lo; delete[] q->ep_hi;
    delete q;
    return PQPS_OK;
}

int pqps_qstream_create(pqps_ctx *ctx, uint32_t depth, pqps_qstream **out) {
    if (!ctx || !out) return fail(PQPS_EINVAL, "NULL argument");
    if (depth < 1 || depth > 64) return fail(PQPS_EINVAL, "depth %u out of range (1..64)", depth);
    pqps_qstream *q = new (std::nothrow) pqps_qstream();
    if (!q) return fail(PQPS_ENOMEM, "out of host memory");
    q->ctx = ctx; q->depth = depth;
    static const int lanes_env = [] { const char *e = tuning_env("PQPS_QSTREAM_LANES"); return e ? atoi(e) : 0; }();
    q->lanes = lanes_env >= 1 && lanes_env <= 8 ? (uint32_t)lanes_env : 2u;
    if (q->lanes > depth) q->lanes = depth;
    q->child = new pqps_ctx *[q->lanes](); q->done = new hipEvent_t[depth](); q->ready = new hipEvent_t[depth]();
    q->ran_on = new pqps_ctx *[depth](); q->used = new bool[depth](); q->ep_lo = new uint32_t[depth](); q->ep_hi = new uint32_t[depth]();
    hipError_t e = hipSetDevice(ctx->device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&q->joined, hipEventDisableTiming);
    for (uint32_t i = 0; i < depth && e == hipSuccess; i++) e = hipEventCreateWithFlags(&q->done[i], hipEventDisableTiming);
    for (uint32_t i = 0; i < q->lanes && e == hipSuccess; i++)
        if (create_ctx(ctx->device, true, &q->child[i]) != PQPS_OK) { pqps_qstream_destroy(q); return PQPS_EHIP; }
    if (e != hipSuccess) { pqps_qstream_destroy(q); return fail(PQPS_EHIP, "query stream: %s", hipGetErrorString(e)); }
    *out = q;
    return PQPS_OK;
}

}  // extern "C"

namespace {

bool one_lane_table(uint64_t n_rows) {
    static const char *env = tuning_env("PQPS_QSTREAM_ONE_LANE_ROWS");      // tuning runs
    const uint64_t from = env ? strtoull(env, nullptr, 10) : kInterleaveFromGroups * (uint64_t)kGroupSteps * kStepRows;
    return n_rows >= from;
}

// The slot is the caller's again (a host wait for the query that last used it, normally long satisfied), the lanes
// come after what the caller's stream holds, and the query gets its lane.
int qstream_begin(pqps_qstream *q, uint32_t slot, uint64_t n_rows, hipStream_t caller, pqps_ctx **lane, bool ids = false) {
    if (slot >= q->depth) return fail(PQPS_EINVAL, "slot %u >= depth %u", slot, q->depth);
    if (q->used[slot]) { const uint64_t t0 = now_ns(); HIP_TRY(hipEventSynchronize(q->ready[slot])); q->wait_ns += now_ns() - t0; }
    if (!q->ordered) {                                           // what the caller's stream holds (the table, ...) comes first
        HIP_TRY(hipEventRecord(q->joined, caller));
        for (uint32_t i = 0; i < q->lanes; i++) HIP_TRY(hipStreamWaitEvent(q->child[i]->stream, q->joined, 0));
        q->ordered = true;
    }
    const bool one = one_lane_table(n_rows) || (ids && q->dense.load(std::memory_order_relaxed) != 0);
    *lane = q->child[one ? 0u : (uint32_t)(q->seq % q->lanes)];
    return PQPS_OK;
}

int qstream_issue(pqps_qstream *q, uint32_t slot, int mode, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                  const pqps_predicate *pred, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *scan_stream) {
    if (!q || !out_count) return fail(PQPS_EINVAL, "qstream/out_count is NULL");
    if (mode == MODE_IDS) {
        if (!out_ids && out_capacity) return fail(PQPS_EINVAL, "out_ids is NULL");
        if (n_rows > 0xFFFFFFFFull || (uint64_t)id_base + n_rows > 0x100000000ull)
            return fail(PQPS_EINVAL, "row IDs are u32: id_base + n_rows must be <= 2^32");
    }
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    EvalArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    hipStream_t caller = pick_stream(q->ctx, scan_stream);
    const eval_fn k1 = mode == MODE_IDS ? pick_eval<MODE_IDS>(cols, n_cols, pred, a, n_rows) : pick_eval<MODE_COUNT>(cols, n_cols, pred, a, n_rows);
    pqps_ctx *c = nullptr;
    hipStream_t s = caller;
    if (q->ctx->timing) {
        // While the PARENT context records timings the query runs whole on the caller's stream with the context's
        // own scratch, one at a time, so that the recorded events mean what pqps_ctx_kernel_time documents.
        if (slot >= q->depth) return fail(PQPS_EINVAL, "slot %u >= depth %u", slot, q->depth);
        c = q->ctx;
    } else {
        rc = qstream_begin(q, slot, n_rows, caller, &c, mode == MODE_IDS);
        if (rc) return rc;
        s = c->stream;
    }
    hipEvent_t ev = q->done[slot];
    const uint32_t before = c->epoch;
    rc = run_filter(c, k1, a, n_rows, mode, false, id_base, out_ids, out_capacity, out_count, s, &ev);
    if (rc) return rc;
    q->ready[slot] = ev;
    q->ran_on[slot] = c;
    // the launch's epoch (run_filter commits it once the launch is in the queue; COUNT and an empty table launch no ID kernel)
    q->ep_lo[slot] = q->ep_hi[slot] = (mode == MODE_IDS && n_rows != 0 && (c->epoch != before || before == 0)) ? c->epoch : 0u;
    q->used[slot] = true;
    q->seq++;
    return PQPS_OK;
}

}  // namespace

extern "C" {

int pqps_qstream_scan_slot(pqps_qstream *q, uint32_t slot, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                           const pqps_predicate *pred, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *scan_stream) {
    return qstream_issue(q, slot, MODE_IDS, cols, n_cols, n_rows, id_base, pred, out_ids, out_capacity, out_count, scan_stream);
}

int pqps_qstream_count_slot(pqps_qstream *q, uint32_t slot, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                            const pqps_predicate *pred, uint64_t *out_count, void *scan_stream) {
    return qstream_issue(q, slot, MODE_COUNT, cols, n_cols, n_rows, 0, pred, nullptr, 0, out_count, scan_stream);
}

int pqps_qstream_scan(pqps_qstream *q, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                      const pqps_predicate *pred, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count,
                      void *scan_stream) {
    if (!q) return fail(PQPS_EINVAL, "qstream is NULL");
    return qstream_issue(q, (uint32_t)(q->seq % q->depth), MODE_IDS, cols, n_cols, n_rows, id_base, pred, out_ids, out_capacity, out_count, scan_stream);
}

// COUNT(*) through the same two lanes (the scan + its one-workgroup reduction whole on a lane: one query's ramp and
// drain under the other's scan -- S1 58 -> 48 us, Q_A 67 -> 58, a lone u8 column 22 -> 15.5 us per query at 100 M rows).
int pqps_qstream_count(pqps_qstream *q, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                       const pqps_predicate *pred, uint64_t *out_count, void *scan_stream) {
    if (!q) return fail(PQPS_EINVAL, "qstream is NULL");
    return qstream_issue(q, (uint32_t)(q->seq % q->depth), MODE_COUNT, cols, n_cols, n_rows, 0, pred, nullptr, 0, out_count, scan_stream);
}

// A query that is more than one filter call (index probes + gather filters, flag passes before the last pass): the
// caller gets the lane the slot's query runs on, issues its calls there, and marks the end.
int pqps_qstream_lane(pqps_qstream *q, uint32_t slot, uint64_t n_rows, void *scan_stream, pqps_ctx **lane_ctx, void **lane_stream) {
    if (!q || !lane_ctx || !lane_stream) return fail(PQPS_EINVAL, "NULL argument");
    pqps_ctx *c = nullptr;
    int rc = qstream_begin(q, slot, n_rows, pick_stream(q->ctx, scan_stream), &c);
    if (rc) return rc;
    q->ran_on[slot] = c;
    q->ep_lo[slot] = c->epoch;                                   // (pqps_qstream_mark turns this into the range of the calls in between)
    q->ep_hi[slot] = 0;
    q->seq++;
    *lane_ctx = c;
    *lane_stream = (void *)c->stream;
    return PQPS_OK;
}

void pqps_qstream_hint_answer(pqps_qstream *q, uint64_t matches, uint64_t n_rows) {
    if (q) q->dense.store(n_rows != 0 && matches >= n_rows / 4 ? 1 : 0, std::memory_order_relaxed);
}

int pqps_qstream_mark(pqps_qstream *q, uint32_t slot) {
    if (!q || slot >= q->depth || !q->ran_on[slot]) return fail(PQPS_EINVAL, "slot %u has no query", slot);
    (void)hipSetDevice(q->ctx->device);
    HIP_TRY(hipEventRecord(q->done[slot], q->ran_on[slot]->stream));
    q->ready[slot] = q->done[slot];
    {   // the ID launches the caller issued on the lane since pqps_qstream_lane: epochs (first, now]
        const uint32_t first = q->ep_lo[slot], now = q->ran_on[slot]->epoch;
        if (now == first) { q->ep_lo[slot] = q->ep_hi[slot] = 0; }
        else if (now > first) { q->ep_lo[slot] = first + 1u; q->ep_hi[slot] = now; }
        else { q->ep_lo[slot] = 1u; q->ep_hi[slot] = 0xFFFFFFFFu; }       // the epochs started over in between: any word
    }
    q->used[slot] = true;
    return PQPS_OK;
}

// Host wait for ONE slot's query (a stream synchronise would also wait for the queries issued after it).  May be
// called from another thread than the one that issues.
int pqps_qstream_wait(pqps_qstream *q, uint32_t slot) {
    if (!q || slot >= q->depth) return fail(PQPS_EINVAL, "slot out of range");
    if (!q->used[slot]) return PQPS_OK;
    (void)hipSetDevice(q->ctx->device);
    HIP_TRY(hipEventSynchronize(q->ready[slot]));
    if (q->ran_on[slot] == q->ctx) return take_status(q->ctx, "query");       // (timing mode: one query at a time on the parent context)
    return take_status_of(q->ran_on[slot], q->ep_lo[slot], q->ep_hi[slot], "query");
}

// Test hook: marks the slot's (last) ID launch as one that gave up, the way the kernel would -- what pqps_qstream_wait must
// then report for THIS slot and for no other that ran on the same lane.
int pqps_qstream_test_fail_slot(pqps_qstream *q, uint32_t slot) {
    if (!q || slot >= q->depth || !q->used[slot] || q->ep_hi[slot] == 0 || q->ep_hi[slot] == 0xFFFFFFFFu) return fail(PQPS_EINVAL, "slot %u has no ID launch", slot);
    const uint32_t e = q->ep_hi[slot];
    ((volatile uint32_t *)q->ran_on[slot]->status_host)[e & (kStatusWords - 1u)] = e;
    return PQPS_OK;
}

// Reserves the lanes' scratch (hand-off words, slots, the list area of ID scans) for tables of up to n_rows rows now instead
// of inside the first queries: an engine does this when it builds its table.
int pqps_qstream_reserve(pqps_qstream *q, uint64_t n_rows) {
    if (!q) return fail(PQPS_EINVAL, "qstream is NULL");
    HIP_TRY(hipSetDevice(q->ctx->device));
    const uint64_t steps = (n_rows + kStepRows - 1) / kStepRows;
    for (uint32_t i = 0; i < q->lanes; i++) {
        const int rc = ensure_scratch(q->child[i], steps);
        if (rc) return rc;
        if ((steps + kGroupSteps - 1) / kGroupSteps < kListAreaBelowGroups)
            (void)ensure_lists(q->child[i], steps);              // (larger tables scan without lists; refused: bit masks, same results)
    }
    return PQPS_OK;
}

uint64_t pqps_qstream_wait_ns(pqps_qstream *q, int reset) {
    if (!q) return 0;
    const uint64_t w = q->wait_ns;
    if (reset) q->wait_ns = 0;
    return w;
}

int pqps_qstream_sync(pqps_qstream *q) {
    if (!q) return fail(PQPS_EINVAL, "qstream is NULL");
    (void)hipSetDevice(q->ctx->device);
    for (uint32_t i = 0; i < q->lanes; i++) {
        HIP_TRY(hipStreamSynchronize(q->child[i]->stream));
        const int st = take_status(q->child[i], "query stream");
        if (st) return st;
    }
    q->ordered = false;                                          // the caller may have put new work on its stream meanwhile
    return PQPS_OK;
}

// Per-launch timing of the queries AS THEY RUN IN THE STREAM (two in flight): the lanes' own recorders.  The
// events ride on the dispatch packets, so recording does not change how the launches overlap.
int pqps_qstream_set_timing(pqps_qstream *q, int enable) {
    if (!q) return fail(PQPS_EINVAL, "qstream is NULL");
    for (uint32_t i = 0; i < q->lanes; i++) { const int rc = pqps_ctx_set_timing(q->child[i], enable); if (rc) return rc; }
    return PQPS_OK;
}

int pqps_qstream_kernel_time(pqps_qstream *q, double *eval_ms, double *total_ms, int *launches) {
    if (!q || !eval_ms || !total_ms || !launches) return fail(PQPS_EINVAL, "NULL argument");
    *eval_ms = 0.0; *total_ms = 0.0; *launches = 0;
    for (uint32_t i = 0; i < q->lanes; i++) {
        double e = 0.0, t = 0.0; int k = 0;
        const int rc = pqps_ctx_kernel_time(q->child[i], &e, &t, &k);
        if (rc) return rc;
        *eval_ms += e; *total_ms += t; *launches += k;
    }
    return PQPS_OK;
}

// Peer copy between two contexts' devices (or inside one device), asynchronous on `stream` of the DESTINATION
// context (NULL: its own).  What the one-process engine gathers its shards' results with (xGMI DMA; the one-process
// counterpart of the send / recv pairs of pqps_exchange).
int pqps_copy_peer(pqps_ctx *dst_ctx, void *dst, pqps_ctx *src_ctx, const void *src, size_t bytes, void *stream) {
    if (!dst_ctx || !src_ctx || (bytes && (!dst || !src))) return fail(PQPS_EINVAL, "NULL argument");
    if (bytes == 0) return PQPS_OK;
    hipStream_t s = pick_stream(dst_ctx, stream);
    if (dst_ctx->device == src_ctx->device) HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, s));
    else HIP_TRY(hipMemcpyPeerAsync(dst, dst_ctx->device, src, src_ctx->device, bytes, s));
    return PQPS_OK;
}

int pqps_ctx_device(pqps_ctx *ctx) { return ctx ? ctx->device : -1; }

int pqps_ids_checksum(pqps_ctx *ctx, const uint32_t *ids, uint64_t count, uint64_t out[2], void *stream) {
    if (!ctx || !out || (count && !ids)) return fail(PQPS_EINVAL, "NULL argument");
    hipStream_t s = pick_stream(ctx, stream);
    if (!ctx->check_dev) HIP_TRY(hipMalloc((void **)&ctx->check_dev, 2 * sizeof(uint64_t)));
    HIP_TRY(hipMemsetAsync(ctx->check_dev, 0, 2 * sizeof(uint64_t), s));
    if (count) {
        uint64_t blocks = (count + 1023) / 1024;
        const uint64_t cap = (uint64_t)ctx->compute_units * 8;
        if (blocks > cap) blocks = cap;
        hipLaunchKernelGGL(ids_checksum_kernel, dim3((uint32_t)blocks), dim3(256), 0, s, ids, count, ctx->check_dev);
        HIP_TRY(hipGetLastError());
    }
    HIP_TRY(hipMemcpyAsync(out, ctx->check_dev, 2 * sizeof(uint64_t), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
    return PQPS_OK;
}

}  // extern "C"
