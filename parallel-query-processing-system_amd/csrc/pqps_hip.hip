// pqps_hip.hip -- gfx950 (MI355X, CDNA4) kernels + the C-ABI shim of include/pqps_hip.h.
//
// The ONLY translation unit compiled by hipcc.  No CUDA-compat headers, no
// dual paths: wave = 64 lanes, written for CDNA4 directly.
//
// Hot kernel: filter_kernel<MODE, GATHER>
//   replaces linearSearchRecords + evaluateWhereClause + checkCondition + CMP_*
//   of the reference (engine/serial/executeEngine-serial.c:854-878, :292-316,
//   :251-289, :18-123) with one single-pass, order-preserving, HBM-bound scan:
//
//   * columns are separate device arrays (SoA); a 256-thread workgroup owns a
//     tile of 4096 rows; wave w owns 1024 contiguous rows as 4 chunks of 256;
//     lane l owns rows 4l..4l+3 of each chunk => every 4-byte column is read
//     with one fully coalesced global_load_dwordx4 per chunk (1 KiB / wave
//     instruction), 1-byte columns with a dword, 2-byte with dwordx2, 8-byte
//     with two dwordx4.  Each predicate column is read exactly once.
//   * predicate operands (window lo/span per leaf) are staged in LDS once per
//     workgroup; every leaf is the unsigned window test ((x - lo) <= span) ^ neg;
//     the boolean tree is a 64-entry truth table (<= 6 leaves) or a jump table.
//   * stream compaction: per chunk three __ballot()s of the per-lane match
//     count bits + mbcnt give the exclusive lane prefix; wave totals meet in
//     LDS; tile totals are chained across workgroups by a decoupled look-back
//     over 8-byte {flag,value} status words (relaxed agent-scope atomics: the
//     value IS the flag, so no separate payload / fence is needed); row IDs are
//     written in ascending order exactly once.
//   * persistent grid: G = min(tiles, CUs * BLOCKS_PER_CU) workgroups, tile t
//     handled by block t % G in increasing order.  All G blocks are co-resident
//     (BLOCKS_PER_CU is half of what the register/LDS budget admits), so every
//     predecessor a look-back waits for is running; spins are bounded and set
//     an error word instead of hanging.
//
// No MFMA anywhere: this is integer compare + compaction, bound by HBM reads.

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>

#include <hip/hip_runtime.h>
#include <rocprim/device/device_radix_sort.hpp>   // index build only (stable LSD sort)

#include "pqps_hip.h"

namespace {

constexpr int kBlock = 256;                 // threads per workgroup (4 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kChunksPerWave = 4;           // 4 x 256 rows per wave per tile
constexpr int kRowsPerLane = 4;             // consecutive rows per lane per chunk
constexpr int kChunkRows = 64 * kRowsPerLane;                       // 256
constexpr int kTileRows = kWaves * kChunksPerWave * kChunkRows;     // 4096
static_assert(kTileRows == PQPS_TILE_ROWS, "tile size is part of the ABI");
constexpr int kBlocksPerCU = 4;

enum Mode { MODE_IDS = 0, MODE_COUNT = 1, MODE_FLAGS = 2 };

// status word of one tile: [63:62] flag, [61:0] value
constexpr uint64_t kFlagAgg = 1ull << 62;      // value = matches of this tile
constexpr uint64_t kFlagPrefix = 2ull << 62;   // value = matches of tiles 0..this
constexpr uint64_t kValueMask = (1ull << 62) - 1;
constexpr uint32_t kSpinLimit = 1u << 22;      // bounded look-back spin

// scratch header (8 x u64) in front of the status array
enum { HDR_ERROR = 0, HDR_TOTAL = 1, HDR_WORDS = 8 };

struct FilterArgs {
    const void *col[PQPS_MAX_COLUMNS];
    uint64_t lo[PQPS_MAX_LEAVES];
    uint64_t span[PQPS_MAX_LEAVES];
    uint64_t truth;
    uint64_t n_rows;            // scan: rows; gather: upper bound only (range read on device)
    uint64_t out_cap;
    uint32_t *out_ids;
    uint8_t *out_flags;
    uint64_t *out_count;        // device
    uint64_t *scratch;          // header + status[]
    const uint32_t *cand;       // gather: candidate row numbers
    const uint64_t *range;      // gather: [begin, end) into cand, device
    uint32_t id_base;
    uint32_t n_cols;
    uint32_t n_leaves;
    uint32_t negmask;
    uint8_t width_log2[PQPS_MAX_COLUMNS];
    uint8_t leaf_begin[PQPS_MAX_COLUMNS + 1];   // leaves of column c: [leaf_begin[c], leaf_begin[c+1])
    uint8_t on_true[PQPS_MAX_LEAVES];
    uint8_t on_false[PQPS_MAX_LEAVES];
    uint8_t order[PQPS_MAX_LEAVES];
};

__device__ __forceinline__ uint32_t lane_id() { return __lane_id(); }

// exclusive count of set bits of `mask` below this lane
__device__ __forceinline__ uint32_t mbcnt(uint64_t mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ uint64_t status_load(const uint64_t *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void status_store(uint64_t *p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// One value of a column at an arbitrary row (partial tiles, gather mode).
__device__ __forceinline__ uint64_t load_one(const void *base, int wlog2, uint64_t row) {
    switch (wlog2) {
    case 0: return ((const uint8_t *)base)[row];
    case 1: return ((const uint16_t *)base)[row];
    case 2: return ((const uint32_t *)base)[row];
    default: return ((const uint64_t *)base)[row];
    }
}

// Decoupled look-back executed by wave 0 of the workgroup that owns `tile`.
// Returns the number of matches in tiles [0, tile).  `total` = matches of this tile.
__device__ __forceinline__ uint64_t lookback(uint64_t *scratch, uint64_t tile, uint64_t total) {
    uint64_t *status = scratch + HDR_WORDS;
    const uint32_t lane = lane_id();
    if (tile == 0) {
        if (lane == 0) status_store(&status[0], kFlagPrefix | total);
        return 0;
    }
    if (lane == 0) status_store(&status[tile], kFlagAgg | total);
    uint64_t excl = 0;
    int64_t look = (int64_t)tile - 1;
    uint32_t spins = 0;
    while (true) {
        const int64_t t = look - (int64_t)lane;
        // tiles before 0: a virtual "prefix = 0"
        const uint64_t s = (t >= 0) ? status_load(&status[t]) : kFlagPrefix;
        const uint32_t flag = (uint32_t)(s >> 62);
        const uint64_t ready = __ballot(flag != 0);
        const uint64_t isprefix = __ballot(flag == 2);
        const int p = isprefix ? __builtin_ctzll(isprefix) : 64;     // nearest inclusive prefix
        const uint64_t need = (p >= 63) ? ~0ull : ((2ull << p) - 1);
        if ((ready & need) != need) {
            if (++spins > kSpinLimit) {                                // never hang the GPU
                if (lane == 0) atomicExch((unsigned long long *)&scratch[HDR_ERROR], 1ull);
                return excl;
            }
            __builtin_amdgcn_s_sleep(2);
            continue;
        }
        const uint64_t v = ((int)lane <= p) ? (s & kValueMask) : 0;
        excl += wave_sum_u64(v);
        if (p != 64) break;
        look -= 64;
    }
    if (lane == 0) status_store(&status[tile], kFlagPrefix | (excl + total));
    return excl;
}

template <int MODE, bool GATHER>
__global__ __launch_bounds__(kBlock, kBlocksPerCU)
void filter_kernel(const FilterArgs a) {
    // LDS-staged predicate operands + compaction scratch
    __shared__ uint64_t s_lo[PQPS_MAX_LEAVES];
    __shared__ uint64_t s_span[PQPS_MAX_LEAVES];
    __shared__ uint32_t s_wave_total[kWaves];
    __shared__ uint64_t s_tile_excl;

    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint32_t wave = tid >> 6;

    if (tid < PQPS_MAX_LEAVES) {
        s_lo[tid] = a.lo[tid];
        s_span[tid] = a.span[tid];
    }
    __syncthreads();

    uint64_t begin = 0, n_rows = a.n_rows;
    if (GATHER) {
        begin = a.range[0];
        const uint64_t end = a.range[1];
        n_rows = end > begin ? end - begin : 0;
        if (n_rows > a.n_rows) n_rows = a.n_rows;     // never past the caller's bound
    }
    const uint64_t num_tiles = (n_rows + kTileRows - 1) / kTileRows;
    const uint64_t out_base = (GATHER && MODE == MODE_IDS) ? *a.out_count : 0;

    uint64_t block_count = 0;      // MODE_COUNT / MODE_FLAGS accumulate locally

    for (uint64_t tile = blockIdx.x; tile < num_tiles; tile += gridDim.x) {
        const uint64_t tile_row0 = tile * kTileRows;
        const uint64_t wave_row0 = tile_row0 + (uint64_t)wave * (kChunksPerWave * kChunkRows);
        const bool full = !GATHER && (tile_row0 + kTileRows <= n_rows);

        // per row: bit k = result of leaf k
        uint32_t idx[kChunksPerWave][kRowsPerLane];
#pragma unroll
        for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
            for (int j = 0; j < kRowsPerLane; j++) idx[u][j] = 0;

        // gather mode / partial tiles: resolve row numbers once
        uint64_t rowno[kChunksPerWave][kRowsPerLane];
        if (!full) {
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                for (int j = 0; j < kRowsPerLane; j++) {
                    const uint64_t pos = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane + j;
                    uint64_t r = ~0ull;
                    if (pos < n_rows) r = GATHER ? (uint64_t)a.cand[begin + pos] : pos;
                    rowno[u][j] = r;
                }
        }

        for (uint32_t c = 0; c < a.n_cols; c++) {             // uniform
            const char *base = (const char *)a.col[c];
            const int wl = a.width_log2[c];
            const uint32_t kb = a.leaf_begin[c], ke = a.leaf_begin[c + 1];
            if (wl == 3) {
                uint64_t v[kChunksPerWave][kRowsPerLane];
                if (full) {
#pragma unroll
                    for (int u = 0; u < kChunksPerWave; u++) {
                        const uint64_t r0 = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane;
                        const uint4 q0 = *(const uint4 *)(base + r0 * 8);
                        const uint4 q1 = *(const uint4 *)(base + r0 * 8 + 16);
                        v[u][0] = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
                        v[u][1] = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
                        v[u][2] = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
                        v[u][3] = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                        for (int j = 0; j < kRowsPerLane; j++)
                            v[u][j] = rowno[u][j] != ~0ull ? load_one(base, 3, rowno[u][j]) : 0;
                }
                for (uint32_t k = kb; k < ke; k++) {            // uniform
                    const uint64_t lo = s_lo[k], span = s_span[k];
                    const uint32_t neg = (a.negmask >> k) & 1u, bit = 1u << k;
#pragma unroll
                    for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                        for (int j = 0; j < kRowsPerLane; j++) {
                            const uint32_t hit = ((v[u][j] - lo) <= span) ? 1u : 0u;
                            idx[u][j] |= (hit ^ neg) ? bit : 0u;
                        }
                }
            } else {
                uint32_t v[kChunksPerWave][kRowsPerLane];
                if (full) {
                    if (wl == 2) {
#pragma unroll
                        for (int u = 0; u < kChunksPerWave; u++) {
                            const uint64_t r0 = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane;
                            const uint4 q = *(const uint4 *)(base + r0 * 4);
                            v[u][0] = q.x; v[u][1] = q.y; v[u][2] = q.z; v[u][3] = q.w;
                        }
                    } else if (wl == 1) {
#pragma unroll
                        for (int u = 0; u < kChunksPerWave; u++) {
                            const uint64_t r0 = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane;
                            const uint2 q = *(const uint2 *)(base + r0 * 2);
                            v[u][0] = q.x & 0xFFFFu; v[u][1] = q.x >> 16;
                            v[u][2] = q.y & 0xFFFFu; v[u][3] = q.y >> 16;
                        }
                    } else {
#pragma unroll
                        for (int u = 0; u < kChunksPerWave; u++) {
                            const uint64_t r0 = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane;
                            const uint32_t q = *(const uint32_t *)(base + r0);
                            v[u][0] = q & 0xFFu; v[u][1] = (q >> 8) & 0xFFu;
                            v[u][2] = (q >> 16) & 0xFFu; v[u][3] = q >> 24;
                        }
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                        for (int j = 0; j < kRowsPerLane; j++)
                            v[u][j] = rowno[u][j] != ~0ull ? (uint32_t)load_one(base, wl, rowno[u][j]) : 0u;
                }
                for (uint32_t k = kb; k < ke; k++) {            // uniform
                    const uint32_t lo = (uint32_t)s_lo[k], span = (uint32_t)s_span[k];
                    const uint32_t neg = (a.negmask >> k) & 1u, bit = 1u << k;
#pragma unroll
                    for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                        for (int j = 0; j < kRowsPerLane; j++) {
                            const uint32_t hit = ((v[u][j] - lo) <= span) ? 1u : 0u;
                            idx[u][j] |= (hit ^ neg) ? bit : 0u;
                        }
                }
            }
        }

        // boolean tree -> one match bit per row; bit (4u+j) of mbits
        uint32_t mbits = 0;
        if (a.n_leaves <= PQPS_TT_LEAVES) {
            const uint64_t tt = a.truth;
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                for (int j = 0; j < kRowsPerLane; j++)
                    mbits |= ((uint32_t)(tt >> idx[u][j]) & 1u) << (u * kRowsPerLane + j);
        } else {
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                for (int j = 0; j < kRowsPerLane; j++) {
                    uint32_t state = 0;                         // step index, ACCEPT or REJECT
                    for (uint32_t s = 0; s < a.n_leaves; s++) {   // uniform bound
                        const uint32_t r = (idx[u][j] >> a.order[s]) & 1u;
                        const uint32_t nxt = r ? a.on_true[s] : a.on_false[s];
                        state = (state == s) ? nxt : state;
                    }
                    mbits |= (state == PQPS_ACCEPT ? 1u : 0u) << (u * kRowsPerLane + j);
                }
        }
        // rows past the end of a partial tile never match
        if (!full) {
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++)
#pragma unroll
                for (int j = 0; j < kRowsPerLane; j++)
                    if (rowno[u][j] == ~0ull) mbits &= ~(1u << (u * kRowsPerLane + j));
        }

        if (MODE == MODE_FLAGS) {
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++) {
                const uint64_t r0 = wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane;
                const uint32_t m4 = (mbits >> (u * kRowsPerLane)) & 0xFu;
                const uint32_t packed = (m4 & 1u) | ((m4 & 2u) << 7) | ((m4 & 4u) << 14) | ((m4 & 8u) << 21);
                if (r0 + kRowsPerLane <= n_rows) {
                    *(uint32_t *)(a.out_flags + r0) = packed;
                } else {
                    for (int j = 0; j < kRowsPerLane; j++)
                        if (r0 + j < n_rows) a.out_flags[r0 + j] = (uint8_t)((m4 >> j) & 1u);
                }
            }
        }
        if (MODE != MODE_IDS) {
            block_count += __popc(mbits);
            continue;
        }

        // ---- order-preserving compaction -----------------------------------
        // lane prefix inside each chunk via ballots of the 3 count bits
        uint32_t lane_off[kChunksPerWave];
        uint32_t wave_total = 0;
#pragma unroll
        for (int u = 0; u < kChunksPerWave; u++) {
            const uint32_t cnt = __popc((mbits >> (u * kRowsPerLane)) & 0xFu);    // 0..4
            const uint64_t b0 = __ballot(cnt & 1u), b1 = __ballot(cnt & 2u), b2 = __ballot(cnt & 4u);
            lane_off[u] = wave_total + mbcnt(b0) + 2u * mbcnt(b1) + 4u * mbcnt(b2);
            wave_total += (uint32_t)__popcll(b0) + 2u * (uint32_t)__popcll(b1) + 4u * (uint32_t)__popcll(b2);
        }
        if (lane == 0) s_wave_total[wave] = wave_total;
        __syncthreads();
        uint32_t wave_off = 0, tile_total = 0;
#pragma unroll
        for (int w = 0; w < kWaves; w++) {
            const uint32_t t = s_wave_total[w];
            wave_off += (w < (int)wave) ? t : 0u;
            tile_total += t;
        }
        if (wave == 0) {
            const uint64_t excl = lookback(a.scratch, tile, tile_total);
            if (lane == 0) {
                s_tile_excl = excl;
                if (tile + 1 == num_tiles) a.scratch[HDR_TOTAL] = excl + tile_total;
            }
        }
        __syncthreads();
        const uint64_t tile_excl = s_tile_excl;
        if (mbits) {
#pragma unroll
            for (int u = 0; u < kChunksPerWave; u++) {
                uint64_t pos = out_base + tile_excl + wave_off + lane_off[u];
#pragma unroll
                for (int j = 0; j < kRowsPerLane; j++) {
                    if (mbits & (1u << (u * kRowsPerLane + j))) {
                        uint32_t id;
                        if (full) id = (uint32_t)(wave_row0 + (uint64_t)u * kChunkRows + lane * kRowsPerLane + j);
                        else id = (uint32_t)rowno[u][j];
                        if (pos < a.out_cap) a.out_ids[pos] = id + a.id_base;
                        pos++;
                    }
                }
            }
        }
        // s_wave_total / s_tile_excl are rewritten only after the next tile's
        // first barrier, which every thread reaches after reading them here.
    }

    if (MODE != MODE_IDS) {
        const uint64_t w = wave_sum_u64(block_count);
        if (lane == 0 && w) atomicAdd((unsigned long long *)a.out_count, (unsigned long long)w);
    }
}

// Scan mode: the last tile's owner left the total in the header; publish it.
__global__ void finish_kernel(uint64_t *scratch, uint64_t *out_count, int accumulate) {
    const uint64_t total = scratch[HDR_TOTAL];
    if (accumulate) *out_count += total; else *out_count = total;
}

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

template <typename K>
__global__ void probe_kernel(const K *keys, uint64_t n, K lo, K hi, uint64_t *range) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    uint64_t l = 0, r = n;
    while (l < r) { const uint64_t m = l + (r - l) / 2; if (keys[m] < lo) l = m + 1; else r = m; }
    const uint64_t b = l;
    r = n;
    while (l < r) { const uint64_t m = l + (r - l) / 2; if (!(hi < keys[m])) l = m + 1; else r = m; }
    range[0] = b;
    range[1] = l < b ? b : l;
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

// ---- all-gatherv tail: padded per-rank segments -> one contiguous ID list -------
// segs[r * seg_cap .. + counts[r]) are rank r's ascending IDs (what an equal-size
// all-gather delivered); merged = their rank-order concatenation, the layout
// MPI_Allgatherv produces from recvCounts/displs (engine/mpi/executeEngine-mpi.c:758-765).
__global__ __launch_bounds__(256) void merge_segments_kernel(const uint32_t *segs, const uint64_t *counts,
                                                             uint32_t world, uint64_t seg_cap,
                                                             uint32_t *merged, uint64_t merged_cap, uint64_t *total_out) {
    const uint32_t r = blockIdx.y;
    uint64_t displ = 0, total = 0;
    for (uint32_t i = 0; i < world; i++) {
        const uint64_t c = counts[i] < seg_cap ? counts[i] : seg_cap;
        if (i < r) displ += c;
        total += c;
    }
    const uint64_t cnt = counts[r] < seg_cap ? counts[r] : seg_cap;
    const uint32_t *src = segs + (uint64_t)r * seg_cap;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * blockDim.x)
        if (displ + i < merged_cap) merged[displ + i] = src[i];
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0 && total_out) {
        uint64_t raw = 0;
        for (uint32_t i = 0; i < world; i++) raw += counts[i];
        total_out[0] = total;          // IDs actually merged
        total_out[1] = raw;            // IDs the ranks reported (> total means a segment overflowed)
    }
}

// ---- streaming read probe ----------------------------------------------------
__global__ __launch_bounds__(256) void read_probe_kernel(const uint4 *p, uint64_t n16, uint64_t *out) {
    uint64_t acc = 0;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        const uint4 a = p[i], b = p[i + stride], c = p[i + 2 * stride], d = p[i + 3 * stride];
        acc += (uint64_t)a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w
             + c.x + c.y + c.z + c.w + d.x + d.y + d.z + d.w;
    }
    for (; i < n16; i += stride) { const uint4 a = p[i]; acc += (uint64_t)a.x + a.y + a.z + a.w; }
    acc = wave_sum_u64(acc);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd((unsigned long long *)out, (unsigned long long)acc);
}

// ---------------------------------------------------------------------------
// host side of the shim
// ---------------------------------------------------------------------------
thread_local char g_err[512] = "";

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

struct pqps_ctx {
    int device;
    int compute_units;
    hipStream_t stream;
    uint64_t *scratch;          // HDR_WORDS + status words
    uint64_t scratch_words;
    void *sort_tmp;
    size_t sort_tmp_bytes;
    // optional per-launch timing of the filter kernel (bench.py roofline)
    bool timing;
    int timed;                  // launches recorded since the last reset
    hipEvent_t *ev_start, *ev_stop;
};

namespace {

hipStream_t pick_stream(pqps_ctx *ctx, void *stream) { return stream ? (hipStream_t)stream : ctx->stream; }

int ensure_scratch(pqps_ctx *ctx, uint64_t tiles) {
    const uint64_t need = HDR_WORDS + tiles + 64;
    if (ctx->scratch_words >= need) return PQPS_OK;
    if (ctx->scratch) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(ctx->scratch)); ctx->scratch = nullptr; ctx->scratch_words = 0; }
    const uint64_t words = need + need / 2;
    HIP_TRY(hipMalloc((void **)&ctx->scratch, words * sizeof(uint64_t)));
    ctx->scratch_words = words;
    return PQPS_OK;
}

int check_pred(const pqps_column *cols, uint32_t n_cols, const pqps_predicate *pred) {
    if (!pred) return fail(PQPS_EINVAL, "predicate is NULL");
    if (n_cols > PQPS_MAX_COLUMNS) return fail(PQPS_EINVAL, "too many columns: %u", n_cols);
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

void fill_args(FilterArgs &a, const pqps_column *cols, uint32_t n_cols, const pqps_predicate *pred) {
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
}

uint32_t grid_for(pqps_ctx *ctx, uint64_t tiles) {
    const uint64_t cap = (uint64_t)ctx->compute_units * kBlocksPerCU;
    const uint64_t g = tiles < cap ? tiles : cap;
    return (uint32_t)(g ? g : 1);
}

template <int MODE, bool GATHER>
int launch_filter(pqps_ctx *ctx, FilterArgs &a, uint64_t tiles, hipStream_t s) {
    int rc = ensure_scratch(ctx, tiles);
    if (rc) return rc;
    a.scratch = ctx->scratch;
    // zero header + status words this launch can touch (16-byte multiple from the allocation start)
    const size_t zero_bytes = ((HDR_WORDS + tiles) * sizeof(uint64_t) + 15) & ~(size_t)15;
    HIP_TRY(hipMemsetAsync(ctx->scratch, 0, zero_bytes, s));
    const bool timed = ctx->timing && ctx->timed < kMaxTimedLaunches;
    if (timed) HIP_TRY(hipEventRecord(ctx->ev_start[ctx->timed], s));
    hipLaunchKernelGGL((filter_kernel<MODE, GATHER>), dim3(grid_for(ctx, tiles)), dim3(kBlock), 0, s, a);
    HIP_TRY(hipGetLastError());
    if (timed) { HIP_TRY(hipEventRecord(ctx->ev_stop[ctx->timed], s)); ctx->timed++; }
    return PQPS_OK;
}

}  // namespace

extern "C" {

const char *pqps_last_error(void) { return g_err; }

int pqps_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pqps_ctx_create(int device, pqps_ctx **out) {
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
    ctx->scratch = nullptr;
    ctx->scratch_words = 0;
    ctx->sort_tmp = nullptr;
    ctx->sort_tmp_bytes = 0;
    ctx->timing = false;
    ctx->timed = 0;
    ctx->ev_start = ctx->ev_stop = nullptr;
    hipError_t se = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (se != hipSuccess) { delete ctx; return fail(PQPS_EHIP, "hipStreamCreate: %s", hipGetErrorString(se)); }
    *out = ctx;
    return PQPS_OK;
}

void pqps_ctx_destroy(pqps_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->scratch) (void)hipFree(ctx->scratch);
    if (ctx->sort_tmp) (void)hipFree(ctx->sort_tmp);
    if (ctx->ev_start) {
        for (int i = 0; i < kMaxTimedLaunches; i++) { (void)hipEventDestroy(ctx->ev_start[i]); (void)hipEventDestroy(ctx->ev_stop[i]); }
        delete[] ctx->ev_start;
        delete[] ctx->ev_stop;
    }
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pqps_ctx_set_timing(pqps_ctx *ctx, int enable) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    if (enable && !ctx->ev_start) {
        ctx->ev_start = new (std::nothrow) hipEvent_t[kMaxTimedLaunches];
        ctx->ev_stop = new (std::nothrow) hipEvent_t[kMaxTimedLaunches];
        if (!ctx->ev_start || !ctx->ev_stop) return fail(PQPS_ENOMEM, "out of host memory");
        for (int i = 0; i < kMaxTimedLaunches; i++) {
            HIP_TRY(hipEventCreate(&ctx->ev_start[i]));
            HIP_TRY(hipEventCreate(&ctx->ev_stop[i]));
        }
    }
    ctx->timing = enable != 0;
    ctx->timed = 0;
    return PQPS_OK;
}

int pqps_ctx_kernel_time(pqps_ctx *ctx, double *total_ms, int *launches) {
    if (!ctx || !total_ms || !launches) return fail(PQPS_EINVAL, "NULL argument");
    double sum = 0.0;
    for (int i = 0; i < ctx->timed; i++) {
        HIP_TRY(hipEventSynchronize(ctx->ev_stop[i]));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start[i], ctx->ev_stop[i]));
        sum += ms;
    }
    *total_ms = sum;
    *launches = ctx->timed;
    ctx->timed = 0;
    return PQPS_OK;
}

int pqps_ctx_sync(pqps_ctx *ctx, void *stream) {
    if (!ctx) return fail(PQPS_EINVAL, "ctx is NULL");
    HIP_TRY(hipStreamSynchronize(pick_stream(ctx, stream)));
    if (ctx->scratch) {
        uint64_t err = 0;
        HIP_TRY(hipMemcpy(&err, ctx->scratch + HDR_ERROR, sizeof err, hipMemcpyDeviceToHost));
        if (err) return fail(PQPS_EHIP, "look-back spin limit hit: results of the last filter are invalid");
    }
    return PQPS_OK;
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
    hipStream_t s = pick_stream(ctx, stream);
    FilterArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    a.id_base = id_base;
    a.out_ids = out_ids;
    a.out_cap = out_capacity;
    a.out_count = out_count;
    const uint64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    rc = launch_filter<MODE_IDS, false>(ctx, a, tiles, s);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1), 0, s, ctx->scratch, out_count, 0);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_filter_count(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count) return fail(PQPS_EINVAL, "ctx/out_count is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    FilterArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    a.out_count = out_count;
    HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint64_t), s));
    const uint64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    return launch_filter<MODE_COUNT, false>(ctx, a, tiles, s);
}

int pqps_filter_flags(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred,
                      uint8_t *out_flags, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count || !out_flags) return fail(PQPS_EINVAL, "ctx/out_flags/out_count is NULL");
    if (((uintptr_t)out_flags & 3u) != 0) return fail(PQPS_EINVAL, "out_flags must be 4-byte aligned");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    FilterArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = n_rows;
    a.out_flags = out_flags;
    a.out_count = out_count;
    HIP_TRY(hipMemsetAsync(out_count, 0, sizeof(uint64_t), s));
    const uint64_t tiles = (n_rows + kTileRows - 1) / kTileRows;
    return launch_filter<MODE_FLAGS, false>(ctx, a, tiles, s);
}

int pqps_filter_gather(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                       const uint32_t *cand, const uint64_t *range, uint64_t max_candidates,
                       uint32_t id_base, const pqps_predicate *pred,
                       uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream) {
    if (!ctx || !out_count || !cand || !range) return fail(PQPS_EINVAL, "ctx/cand/range/out_count is NULL");
    if (!out_ids && out_capacity) return fail(PQPS_EINVAL, "out_ids is NULL");
    int rc = check_pred(cols, n_cols, pred);
    if (rc) return rc;
    hipStream_t s = pick_stream(ctx, stream);
    FilterArgs a;
    fill_args(a, cols, n_cols, pred);
    a.n_rows = max_candidates;
    a.id_base = id_base;
    a.cand = cand;
    a.range = range;
    a.out_ids = out_ids;
    a.out_cap = out_capacity;
    a.out_count = out_count;
    const uint64_t tiles = (max_candidates + kTileRows - 1) / kTileRows;
    rc = launch_filter<MODE_IDS, true>(ctx, a, tiles, s);
    if (rc) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(1), 0, s, ctx->scratch, out_count, 1);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

}  // extern "C"

namespace {

template <typename K>
int index_build_t(pqps_ctx *ctx, const void *col, uint64_t n, uint32_t *perm, void *sorted_keys, hipStream_t s) {
    if (n == 0) return PQPS_OK;
    K *keys_rev = nullptr;
    uint32_t *rows_rev = nullptr;
    HIP_TRY(hipMalloc((void **)&keys_rev, n * sizeof(K)));
    hipError_t e = hipMalloc((void **)&rows_rev, n * sizeof(uint32_t));
    if (e != hipSuccess) { (void)hipFree(keys_rev); return fail(PQPS_ENOMEM, "hipMalloc: %s", hipGetErrorString(e)); }
    const uint32_t blocks = (uint32_t)((n + 255) / 256);
    hipLaunchKernelGGL((reverse_gather_kernel<K>), dim3(blocks), dim3(256), 0, s, (const K *)col, n, keys_rev, rows_rev);
    size_t tmp_bytes = 0;
    // stable LSD radix sort (rocPRIM): equal keys keep the descending-row input order
    e = rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_rev, (K *)sorted_keys, rows_rev, perm, n, 0, sizeof(K) * 8, s);
    if (e == hipSuccess) {
        if (tmp_bytes > ctx->sort_tmp_bytes) {
            if (ctx->sort_tmp) (void)hipFree(ctx->sort_tmp);
            ctx->sort_tmp = nullptr; ctx->sort_tmp_bytes = 0;
            e = hipMalloc(&ctx->sort_tmp, tmp_bytes);
            if (e == hipSuccess) ctx->sort_tmp_bytes = tmp_bytes;
        }
        if (e == hipSuccess)
            e = rocprim::radix_sort_pairs(ctx->sort_tmp, tmp_bytes, keys_rev, (K *)sorted_keys, rows_rev, perm, n, 0, sizeof(K) * 8, s);
    }
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(keys_rev);
    (void)hipFree(rows_rev);
    if (e != hipSuccess) return fail(PQPS_EHIP, "radix_sort_pairs: %s", hipGetErrorString(e));
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
        return index_build_t<int32_t>(ctx, col->data, n_rows, perm, sorted_keys, s);
    }
    switch (col->width) {
    case 1: return index_build_t<uint8_t>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 2: return index_build_t<uint16_t>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 4: return index_build_t<uint32_t>(ctx, col->data, n_rows, perm, sorted_keys, s);
    case 8: return index_build_t<uint64_t>(ctx, col->data, n_rows, perm, sorted_keys, s);
    default: return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", col->width);
    }
}

int pqps_index_probe(pqps_ctx *ctx, const void *sorted_keys, uint32_t width, int key_kind,
                     uint64_t n_rows, uint64_t key_lo, uint64_t key_hi, uint64_t *range, void *stream) {
    if (!ctx || !range || (!sorted_keys && n_rows)) return fail(PQPS_EINVAL, "NULL argument");
    hipStream_t s = pick_stream(ctx, stream);
    if (key_kind == 1) {
        if (width != 4) return fail(PQPS_EINVAL, "signed keys must be 4 bytes wide");
        hipLaunchKernelGGL((probe_kernel<int32_t>), dim3(1), dim3(64), 0, s, (const int32_t *)sorted_keys, n_rows,
                           (int32_t)(uint32_t)key_lo, (int32_t)(uint32_t)key_hi, range);
    } else if (width == 1) {
        hipLaunchKernelGGL((probe_kernel<uint8_t>), dim3(1), dim3(64), 0, s, (const uint8_t *)sorted_keys, n_rows,
                           (uint8_t)key_lo, (uint8_t)key_hi, range);
    } else if (width == 2) {
        hipLaunchKernelGGL((probe_kernel<uint16_t>), dim3(1), dim3(64), 0, s, (const uint16_t *)sorted_keys, n_rows,
                           (uint16_t)key_lo, (uint16_t)key_hi, range);
    } else if (width == 4) {
        hipLaunchKernelGGL((probe_kernel<uint32_t>), dim3(1), dim3(64), 0, s, (const uint32_t *)sorted_keys, n_rows,
                           (uint32_t)key_lo, (uint32_t)key_hi, range);
    } else if (width == 8) {
        hipLaunchKernelGGL((probe_kernel<uint64_t>), dim3(1), dim3(64), 0, s, (const uint64_t *)sorted_keys, n_rows,
                           key_lo, key_hi, range);
    } else {
        return fail(PQPS_EINVAL, "width %u not in {1,2,4,8}", width);
    }
    HIP_TRY(hipGetLastError());
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

int pqps_merge_segments(pqps_ctx *ctx, const uint32_t *segments, const uint64_t *counts, uint32_t world,
                        uint64_t segment_capacity, uint32_t *merged, uint64_t merged_capacity,
                        uint64_t *totals, void *stream) {
    if (!ctx || !segments || !counts || !merged) return fail(PQPS_EINVAL, "NULL argument");
    if (world == 0 || world > 1024) return fail(PQPS_EINVAL, "world %u out of range", world);
    hipStream_t s = pick_stream(ctx, stream);
    uint64_t bx = (segment_capacity + 255) / 256;
    if (bx > 1024) bx = 1024;
    if (bx == 0) bx = 1;
    hipLaunchKernelGGL(merge_segments_kernel, dim3((uint32_t)bx, world), dim3(256), 0, s,
                       segments, counts, world, segment_capacity, merged, merged_capacity, totals);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

int pqps_read_probe(pqps_ctx *ctx, const void *data, uint64_t bytes, uint64_t *out_sum, void *stream) {
    if (!ctx || !data || !out_sum) return fail(PQPS_EINVAL, "NULL argument");
    if (((uintptr_t)data & 15u) != 0) return fail(PQPS_EINVAL, "data not 16-byte aligned");
    hipStream_t s = pick_stream(ctx, stream);
    HIP_TRY(hipMemsetAsync(out_sum, 0, sizeof(uint64_t), s));
    const uint32_t blocks = (uint32_t)ctx->compute_units * 8u;
    hipLaunchKernelGGL(read_probe_kernel, dim3(blocks), dim3(256), 0, s, (const uint4 *)data, bytes / 16, out_sum);
    HIP_TRY(hipGetLastError());
    return PQPS_OK;
}

}  // extern "C"
