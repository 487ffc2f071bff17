// filter_kernels.hpp -- device code of the SELECT/WHERE filter (included once by pqps_hip.hip).
//
// Replaces linearSearchRecords + evaluateWhereClause + checkCondition + CMP_* of the
// reference (engine/serial/executeEngine-serial.c:854-878, :292-316, :251-289, :18-123)
// with three stream-ordered kernels, none of which ever waits on another workgroup:
//
//  K1 eval    one wave = one STEP of 1024 consecutive rows (16 rows per lane); grid-stride
//             over steps, no LDS, no barriers.  Lane l owns RPL = 16/Wmax consecutive rows
//             of each 64*RPL-row chunk, so the widest predicate column is read with ONE fully
//             coalesced global_load_dwordx4 per chunk and narrower ones with dwordx2 / dword /
//             ushort loads that are just as contiguous across the wave.  Every predicate
//             column is read exactly once.  A leaf is the unsigned window test
//             ((x - lo) <= span) ^ neg; the boolean tree is a 64-entry truth table (<= 6
//             leaves) or a jump table.  Output per step: 16 match bits per lane (one coalesced
//             128-byte store, skipped when the step has no match) + the step's match count.
//  K2 scan    per 64 steps a wave sums the counts; the last workgroup to finish scans those
//             sums (threadfence + ticket) -> exclusive offset of every 64-step group + total.
//  K3 expand  wave per group: wave-prefix of the 64 step counts, then for every non-empty step
//             the match bits become ascending row IDs (ballot + mbcnt lane prefix).
//
// K1 dominates (it is the only kernel that touches the table) and is bound by HBM reads.
// Width-specialised instantiations (1-3 predicate columns, widths non-increasing) keep all
// loads statically scheduled; everything else takes the generic kernel.
#pragma once

namespace {

constexpr int kBlock = 256;                 // threads per workgroup (4 waves)
constexpr int kWaves = kBlock / 64;
constexpr int kStepRows = 1024;             // rows per wave per step = 64 lanes x 16 rows
constexpr int kGroupSteps = 64;             // steps per scan group (64 K rows)
constexpr int kRplGeneric = 4;              // generic kernel: 4 consecutive rows per lane per chunk

enum Mode { MODE_IDS = 0, MODE_COUNT = 1, MODE_FLAGS = 2 };

struct EvalArgs {
    const void *col[PQPS_MAX_COLUMNS];
    uint64_t lo[PQPS_MAX_LEAVES];
    uint64_t span[PQPS_MAX_LEAVES];
    uint64_t truth;
    uint64_t n_rows;                 // scan: rows; gather: caller's upper bound (range is on the device)
    uint16_t *masks;                 // [steps][64] match bits of every lane
    uint32_t *counts;                // [steps]     matches | log2(RPL) << 28
    uint8_t *out_flags;              // MODE_FLAGS
    unsigned long long *out_count;   // MODE_COUNT / MODE_FLAGS: device total
    const uint32_t *cand;            // gather: candidate row numbers
    const uint64_t *range;           // gather: [begin, end) into cand, device resident
    uint32_t n_cols;
    uint32_t n_leaves;
    uint32_t negmask;
    uint32_t pad0;
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

__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// One value of a column at an arbitrary row (partial steps, gather mode).
__device__ __forceinline__ uint64_t load_one(const void *base, int wlog2, uint64_t row) {
    switch (wlog2) {
    case 0: return ((const uint8_t *)base)[row];
    case 1: return ((const uint16_t *)base)[row];
    case 2: return ((const uint32_t *)base)[row];
    default: return ((const uint64_t *)base)[row];
    }
}

// Window test of every leaf of one column on R values; sets bit k of idx[r].
template <typename T, int R>
__device__ __forceinline__ void apply_leaves(const EvalArgs &a, uint32_t kb, uint32_t ke,
                                             const T (&v)[R], uint32_t (&idx)[R]) {
    for (uint32_t k = kb; k < ke; k++) {                        // uniform; operands come in SGPRs
        const T lo = (T)a.lo[k], span = (T)a.span[k];
        const uint32_t neg = (a.negmask >> k) & 1u, bit = 1u << k;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t hit = ((T)(v[r] - lo) <= span) ? 1u : 0u;
            idx[r] |= (hit ^ neg) ? bit : 0u;
        }
    }
}

// Boolean tree on the leaf bits of R rows -> R match bits (bit r).
template <int R>
__device__ __forceinline__ uint32_t combine_leaves(const EvalArgs &a, const uint32_t (&idx)[R]) {
    uint32_t m = 0;
    if (a.n_leaves <= PQPS_TT_LEAVES) {
        const uint64_t tt = a.truth;
#pragma unroll
        for (int r = 0; r < R; r++) m |= ((uint32_t)(tt >> idx[r]) & 1u) << r;
    } else {
#pragma unroll
        for (int r = 0; r < R; r++) {
            uint32_t state = 0;                                 // step index, ACCEPT or REJECT
            for (uint32_t s = 0; s < a.n_leaves; s++) {         // uniform bound
                const uint32_t bitv = (idx[r] >> a.order[s]) & 1u;
                const uint32_t nxt = bitv ? a.on_true[s] : a.on_false[s];
                state = (state == s) ? nxt : state;
            }
            m |= (state == PQPS_ACCEPT ? 1u : 0u) << r;
        }
    }
    return m;
}

// ---- per-step output ---------------------------------------------------------------
// Bit p of a lane's 16 match bits <-> row  step_row0 + (p / RPL) * 64 * RPL + lane * RPL + p % RPL.
template <int MODE>
__device__ __forceinline__ void emit_step(const EvalArgs &a, uint64_t step, uint32_t mbits, uint32_t rpl_log2,
                                          uint64_t n_rows, uint32_t lane, uint64_t &wave_total) {
    const uint32_t cnt = wave_sum_u32(__popc(mbits));
    if (MODE == MODE_IDS) {
        if (cnt) a.masks[step * 64 + lane] = (uint16_t)mbits;    // 128 B per step, only if needed
        if (lane == 0) a.counts[step] = cnt | (rpl_log2 << 28);
    } else {
        wave_total += cnt;
        if (MODE == MODE_FLAGS) {
            const uint32_t rpl = 1u << rpl_log2;
            for (uint32_t p = 0; p < 16; p++) {
                const uint64_t row = step * kStepRows + (uint64_t)(p >> rpl_log2) * 64 * rpl + lane * rpl + (p & (rpl - 1));
                if (row < n_rows) a.out_flags[row] = (uint8_t)((mbits >> p) & 1u);
            }
        }
    }
}

// ---- generic evaluators (any number of columns / leaves), RPL = 4 -----------------------
// Fast path: all 1024 rows of the step exist and are contiguous.
__device__ __forceinline__ uint32_t eval_step_full(const EvalArgs &a, uint64_t step_row0, uint32_t lane) {
    constexpr int R = 16;
    uint32_t idx[R];
#pragma unroll
    for (int r = 0; r < R; r++) idx[r] = 0;
    const uint64_t lane_row0 = step_row0 + lane * kRplGeneric;

    for (uint32_t c = 0; c < a.n_cols; c++) {                   // uniform
        const char *base = (const char *)a.col[c];
        const int wl = a.width_log2[c];
        const uint32_t kb = a.leaf_begin[c], ke = a.leaf_begin[c + 1];
        if (wl == 3) {
#pragma unroll
            for (int h = 0; h < 4; h += 2) {                    // two halves keep live registers down
                uint64_t v[8];
                uint32_t sub[8];
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const char *p = base + (lane_row0 + (uint64_t)(h + u) * 256) * 8;
                    const uint4 q0 = *(const uint4 *)p;
                    const uint4 q1 = *(const uint4 *)(p + 16);
                    v[4 * u + 0] = (uint64_t)q0.x | ((uint64_t)q0.y << 32);
                    v[4 * u + 1] = (uint64_t)q0.z | ((uint64_t)q0.w << 32);
                    v[4 * u + 2] = (uint64_t)q1.x | ((uint64_t)q1.y << 32);
                    v[4 * u + 3] = (uint64_t)q1.z | ((uint64_t)q1.w << 32);
                }
#pragma unroll
                for (int r = 0; r < 8; r++) sub[r] = idx[4 * h + r];
                apply_leaves<uint64_t, 8>(a, kb, ke, v, sub);
#pragma unroll
                for (int r = 0; r < 8; r++) idx[4 * h + r] = sub[r];
            }
        } else {
            uint32_t v[R];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint64_t r0 = lane_row0 + (uint64_t)u * 256;
                if (wl == 2) {
                    const uint4 q = *(const uint4 *)(base + r0 * 4);
                    v[4 * u] = q.x; v[4 * u + 1] = q.y; v[4 * u + 2] = q.z; v[4 * u + 3] = q.w;
                } else if (wl == 1) {
                    const uint2 q = *(const uint2 *)(base + r0 * 2);
                    v[4 * u] = q.x & 0xFFFFu; v[4 * u + 1] = q.x >> 16;
                    v[4 * u + 2] = q.y & 0xFFFFu; v[4 * u + 3] = q.y >> 16;
                } else {
                    const uint32_t q = *(const uint32_t *)(base + r0);
                    v[4 * u] = q & 0xFFu; v[4 * u + 1] = (q >> 8) & 0xFFu;
                    v[4 * u + 2] = (q >> 16) & 0xFFu; v[4 * u + 3] = q >> 24;
                }
            }
            apply_leaves<uint32_t, R>(a, kb, ke, v, idx);
        }
    }
    return combine_leaves<R>(a, idx);
}

// Guarded path (last partial step, gather mode): one chunk (4 rows per lane) at a time
// with element loads.  `pos` counts rows of the scan / positions of the candidate list.
template <bool GATHER>
__device__ __forceinline__ uint32_t eval_step_guarded(const EvalArgs &a, uint64_t step_row0, uint64_t n_rows,
                                                      uint64_t begin, uint32_t lane) {
    uint32_t mbits = 0;
#pragma unroll 1
    for (int u = 0; u < 4; u++) {
        const uint64_t r0 = step_row0 + (uint64_t)u * 256 + lane * kRplGeneric;
        uint64_t row[4];
        uint32_t idx[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            idx[j] = 0;
            row[j] = ~0ull;
            if (r0 + j < n_rows) row[j] = GATHER ? (uint64_t)a.cand[begin + r0 + j] : r0 + j;
        }
        for (uint32_t c = 0; c < a.n_cols; c++) {               // uniform
            const char *base = (const char *)a.col[c];
            const int wl = a.width_log2[c];
            const uint32_t kb = a.leaf_begin[c], ke = a.leaf_begin[c + 1];
            if (wl == 3) {
                uint64_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = row[j] != ~0ull ? load_one(base, 3, row[j]) : 0;
                apply_leaves<uint64_t, 4>(a, kb, ke, v, idx);
            } else {
                uint32_t v[4];
#pragma unroll
                for (int j = 0; j < 4; j++) v[j] = row[j] != ~0ull ? (uint32_t)load_one(base, wl, row[j]) : 0u;
                apply_leaves<uint32_t, 4>(a, kb, ke, v, idx);
            }
        }
        uint32_t m4 = combine_leaves<4>(a, idx);
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (row[j] == ~0ull) m4 &= ~(1u << j);              // rows past the end never match
        mbits |= m4 << (u * 4);
    }
    return mbits;
}

// Generic K1: any predicate; scan (full steps vectorised) or gather (always guarded).
template <int MODE, bool GATHER>
__global__ __launch_bounds__(kBlock) void eval_generic_kernel(const EvalArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
    uint64_t begin = 0, n_rows = a.n_rows;
    if (GATHER) {
        begin = a.range[0];
        const uint64_t end = a.range[1];
        n_rows = end > begin ? end - begin : 0;
        if (n_rows > a.n_rows) n_rows = a.n_rows;               // never past the caller's bound
    }
    // gather: the host sized counts[] for a.n_rows; steps past the device-side range report 0
    const uint64_t steps_alloc = (a.n_rows + kStepRows - 1) / kStepRows;
    uint64_t wave_total = 0;
    for (uint64_t step = wave; step < steps_alloc; step += n_waves) {
        const uint64_t step_row0 = step * kStepRows;
        uint32_t mbits = 0;
        if (!GATHER && step_row0 + kStepRows <= n_rows) mbits = eval_step_full(a, step_row0, lane);
        else if (step_row0 < n_rows) mbits = eval_step_guarded<GATHER>(a, step_row0, n_rows, begin, lane);
        emit_step<MODE>(a, step, mbits, 2, n_rows, lane, wave_total);
    }
    if (MODE != MODE_IDS && lane == 0 && wave_total) atomicAdd(a.out_count, (unsigned long long)wave_total);
}

// ---- width-specialised K1 ---------------------------------------------------------------
// Raw bytes of RPL consecutive rows of a W-byte column, as dwords.
template <int W, int RPL>
struct RawChunk {
    static constexpr int kBytes = W * RPL;                      // 16, 8, 4 or 2
    static constexpr int kDwords = kBytes >= 4 ? kBytes / 4 : 1;
    uint32_t d[kDwords];
    __device__ __forceinline__ void load(const char *p) {
        if constexpr (kBytes == 16) { const uint4 q = *(const uint4 *)p; d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }
        else if constexpr (kBytes == 8) { const uint2 q = *(const uint2 *)p; d[0] = q.x; d[1] = q.y; }
        else if constexpr (kBytes == 4) { d[0] = *(const uint32_t *)p; }
        else { d[0] = *(const uint16_t *)p; }
    }
    template <int R>
    __device__ __forceinline__ uint32_t get32() const {          // row R of the chunk, W <= 4
        if constexpr (W == 4) return d[R];
        else if constexpr (W == 2) return (R & 1) ? (d[R / 2] >> 16) : (d[R / 2] & 0xFFFFu);
        else return (d[R / 4] >> (8 * (R & 3))) & 0xFFu;
    }
    template <int R>
    __device__ __forceinline__ uint64_t get64() const { return (uint64_t)d[2 * R] | ((uint64_t)d[2 * R + 1] << 32); }
};

template <int W, int RPL, int U>
struct RawCol {
    RawChunk<W, RPL> c[U];
    __device__ __forceinline__ void load(const void *base, uint64_t lane_row0) {
#pragma unroll
        for (int u = 0; u < U; u++) c[u].load((const char *)base + (lane_row0 + (uint64_t)u * 64 * RPL) * W);
    }
};

template <int W, int RPL, int U, int... Rs>
__device__ __forceinline__ void unpack32(const RawCol<W, RPL, U> &raw, uint32_t (&v)[16], std::integer_sequence<int, Rs...>) {
    ((v[Rs] = raw.c[Rs / RPL].template get32<Rs % RPL>()), ...);
}
template <int W, int RPL, int U, int... Rs>
__device__ __forceinline__ void unpack64(const RawCol<W, RPL, U> &raw, uint64_t (&v)[16], std::integer_sequence<int, Rs...>) {
    ((v[Rs] = raw.c[Rs / RPL].template get64<Rs % RPL>()), ...);
}

template <int W, int RPL, int U>
__device__ __forceinline__ void eval_col(const EvalArgs &a, int slot, const RawCol<W, RPL, U> &raw, uint32_t (&idx)[16]) {
    const uint32_t kb = a.leaf_begin[slot], ke = a.leaf_begin[slot + 1];
    if constexpr (W == 8) {
        uint64_t v[16];
        unpack64(raw, v, std::make_integer_sequence<int, 16>{});
        apply_leaves<uint64_t, 16>(a, kb, ke, v, idx);
    } else {
        uint32_t v[16];
        unpack32(raw, v, std::make_integer_sequence<int, 16>{});
        apply_leaves<uint32_t, 16>(a, kb, ke, v, idx);
    }
}

constexpr int log2i(int x) { return x <= 1 ? 0 : 1 + log2i(x / 2); }

// W0 >= W1 >= W2 are the byte widths of the predicate columns (0 = slot unused).
template <int MODE, int W0, int W1, int W2>
__global__ __launch_bounds__(kBlock) void eval_spec_kernel(const EvalArgs a) {
    constexpr int RPL = 16 / W0;                                // consecutive rows per lane per chunk
    constexpr int U = 16 / RPL;                                 // chunks per step
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t n_rows = a.n_rows;
    const uint64_t full_steps = n_rows / kStepRows;
    uint64_t wave_total = 0;

    for (uint64_t step = wave; step < full_steps; step += n_waves) {
        const uint64_t lane_row0 = step * kStepRows + lane * RPL;
        RawCol<W0, RPL, U> r0;
        r0.load(a.col[0], lane_row0);
        RawCol<(W1 ? W1 : 1), RPL, U> r1;
        if constexpr (W1 != 0) r1.load(a.col[1], lane_row0);
        RawCol<(W2 ? W2 : 1), RPL, U> r2;
        if constexpr (W2 != 0) r2.load(a.col[2], lane_row0);

        uint32_t idx[16];
#pragma unroll
        for (int r = 0; r < 16; r++) idx[r] = 0;
        eval_col<W0, RPL, U>(a, 0, r0, idx);
        if constexpr (W1 != 0) eval_col<W1, RPL, U>(a, 1, r1, idx);
        if constexpr (W2 != 0) eval_col<W2, RPL, U>(a, 2, r2, idx);
        const uint32_t mbits = combine_leaves<16>(a, idx);
        emit_step<MODE>(a, step, mbits, log2i(RPL), n_rows, lane, wave_total);
    }
    // the partial last step (if any) goes through the guarded evaluator, RPL = 4 layout
    if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {
        const uint32_t mbits = eval_step_guarded<false>(a, full_steps * kStepRows, n_rows, 0, lane);
        emit_step<MODE>(a, full_steps, mbits, 2, n_rows, lane, wave_total);
    }
    if (MODE != MODE_IDS && lane == 0 && wave_total) atomicAdd(a.out_count, (unsigned long long)wave_total);
}

// ---- K2: group sums + scan by the last workgroup ------------------------------------------
struct ScanArgs {
    const uint32_t *counts;          // [steps]
    uint64_t steps;
    uint64_t groups;                 // ceil(steps / 64)
    uint64_t *group_sum;             // [groups]  scratch
    uint64_t *group_excl;            // [groups]  out: exclusive offset (includes the base)
    uint32_t *ticket;                // zero before the launch; reset by the last workgroup
    uint64_t *out_count;             // device: *out_count = base + total
    int accumulate;                  // 1: base = *out_count (index probes append)
};

__global__ __launch_bounds__(kBlock) void scan_kernel(const ScanArgs a) {
    __shared__ uint32_t s_last;
    __shared__ uint64_t s_part[kBlock];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    const uint64_t group = (uint64_t)blockIdx.x * kWaves + (tid >> 6);
    if (group < a.groups) {
        const uint64_t step = group * kGroupSteps + lane;
        const uint32_t c = step < a.steps ? (a.counts[step] & 0x0FFFFFFFu) : 0u;
        const uint32_t sum = wave_sum_u32(c);
        if (lane == 0) a.group_sum[group] = sum;
    }
    // last workgroup to arrive scans the group sums (classic threadfence + ticket hand-off)
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = (atomicAdd(a.ticket, 1u) == gridDim.x - 1) ? 1u : 0u;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    const uint64_t per = (a.groups + kBlock - 1) / kBlock;
    const uint64_t g0 = (uint64_t)tid * per, g1 = (g0 + per < a.groups) ? g0 + per : a.groups;
    uint64_t local = 0;
    for (uint64_t g = g0; g < g1; g++) local += __hip_atomic_load(&a.group_sum[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_part[tid] = local;
    __syncthreads();
    if (tid == 0) {
        uint64_t run = a.accumulate ? *a.out_count : 0;
        for (int i = 0; i < kBlock; i++) { const uint64_t t = s_part[i]; s_part[i] = run; run += t; }
        *a.out_count = run;
        *a.ticket = 0;                                           // ready for the next launch
    }
    __syncthreads();
    uint64_t run = s_part[tid];
    for (uint64_t g = g0; g < g1; g++) {
        const uint64_t t = __hip_atomic_load(&a.group_sum[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.group_excl[g] = run;
        run += t;
    }
}

// ---- K3: match bits -> ascending row IDs ---------------------------------------------------
struct ExpandArgs {
    const uint16_t *masks;
    const uint32_t *counts;
    const uint64_t *group_excl;
    uint64_t steps;
    uint64_t groups;
    uint32_t *out_ids;
    uint64_t out_cap;
    const uint32_t *cand;            // gather: candidate list
    const uint64_t *range;
    uint32_t id_base;
    uint32_t gather;
};

__global__ __launch_bounds__(kBlock) void expand_kernel(const ExpandArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t begin = a.gather ? a.range[0] : 0;
    for (uint64_t group = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); group < a.groups;
         group += (uint64_t)gridDim.x * kWaves) {
        const uint64_t my_step = group * kGroupSteps + lane;
        const uint32_t cw = my_step < a.steps ? a.counts[my_step] : 0u;
        const uint32_t my_cnt = cw & 0x0FFFFFFFu;
        uint64_t todo = __ballot(my_cnt != 0);
        if (todo == 0) continue;                                   // nothing matched in these 64 K rows
        // exclusive prefix of the 64 step counts (Hillis-Steele over the wave)
        uint32_t incl = my_cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off, 64);
            if ((int)lane >= off) incl += up;
        }
        const uint64_t my_off = a.group_excl[group] + (incl - my_cnt);
        while (todo) {                                             // uniform loop over non-empty steps
            const int src = __builtin_ctzll(todo);
            todo &= todo - 1;
            const uint64_t step = group * kGroupSteps + (uint64_t)src;
            const uint64_t step_off = __shfl(my_off, src, 64);
            const uint32_t rpl_log2 = __shfl(cw, src, 64) >> 28;
            const uint32_t rpl = 1u << rpl_log2, chunks = 16u >> rpl_log2;
            const uint32_t m16 = a.masks[step * 64 + lane];
            uint64_t base = step_off;
            for (uint32_t u = 0; u < chunks; u++) {                 // rows ascend as (chunk, lane, j)
                const uint32_t m = (m16 >> (u << rpl_log2)) & ((1u << rpl) - 1u);
                const uint32_t cnt = __popc(m);                     // 0..rpl
                // exclusive lane prefix from ballots of the count bits
                uint32_t pre = 0, tot = 0;
                for (uint32_t b = 0; b <= rpl_log2; b++) {
                    const uint64_t bal = __ballot((cnt >> b) & 1u);
                    pre += mbcnt(bal) << b;
                    tot += (uint32_t)__popcll(bal) << b;
                }
                if (m) {
                    uint64_t pos = base + pre;
                    const uint64_t r0 = step * kStepRows + (uint64_t)u * 64 * rpl + lane * rpl;
                    for (uint32_t j = 0; j < rpl; j++) {
                        if (m & (1u << j)) {
                            const uint32_t id = a.gather ? a.cand[begin + r0 + j] : (uint32_t)(r0 + j);
                            if (pos < a.out_cap) a.out_ids[pos] = id + a.id_base;
                            pos++;
                        }
                    }
                }
                base += tot;
            }
        }
    }
}

}  // namespace
