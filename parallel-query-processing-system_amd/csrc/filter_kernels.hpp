// filter_kernels.hpp -- device code of the SELECT/WHERE filter (included once by pqps_hip.hip).
//
// Replaces linearSearchRecords + evaluateWhereClause + checkCondition + CMP_* of the
// reference (engine/serial/executeEngine-serial.c:854-878, :292-316, :251-289, :18-123)
// with three stream-ordered kernels, none of which ever waits on another workgroup:
//
//  K1 eval    one wave = one STEP of 1024 consecutive rows (16 rows per lane); grid-stride
//             over steps, no LDS, no barriers.  Lane l owns RPL = 16/Wmax (4 for 8-byte) consecutive rows
//             of each 64*RPL-row chunk, so the widest predicate column is read with ONE fully
//             coalesced global_load_dwordx4 per chunk and narrower ones with dwordx2 / dword /
//             ushort loads that are just as contiguous across the wave.  Every predicate
//             column is read exactly once.  A leaf is the unsigned window test
//             ((x - lo) <= span) ^ neg; the boolean tree is a 64-entry truth table (<= 6
//             leaves) or a jump table.  Output per step: 16 match bits per lane (one coalesced
//             128-byte store, skipped when the step has no match) + the step's match count.
//  K2 sums    per group of 64 steps a wave sums the counts; group sums are also added (one
//             atomic per non-empty group) into supergroup sums of 64 groups.
//  K3 expand  workgroup per group: its first output slot = supergroup sums before it + earlier
//             group sums of its supergroup (a few hundred values, no serial scan); wave-prefix of
//             the 64 step counts; for every non-empty step the match bits become ascending row
//             IDs (ballot + mbcnt lane prefix).
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
    uint64_t *partials;              // MODE_COUNT / MODE_FLAGS: [gridDim.x] workgroup totals
    unsigned long long *super_sum;   // MODE_IDS: [n_super] zeroed here for the K2 that follows
    uint32_t n_super;
    uint32_t pad1;
    const uint32_t *cand;            // gather: candidate row numbers
    const uint64_t *range;           // gather: [begin, end) into cand, device resident
    uint32_t n_cols;
    uint32_t n_leaves;
    uint32_t negmask;
    uint32_t streaming;              // host side only: the scan outgrows the Infinity Cache (grid + load policy)
    uint32_t steps_per_iter;         // host side only: S of the chosen kernel (grid sizing)
    uint32_t valu_chain;             // host side only: the one-leaf vector-unit kernel variant was chosen
    uint32_t chain;                  // 0: general tree; 1: AND of leaves; 2: NOT(AND) = OR form (spec kernels)
    uint32_t chain_want;             // bit k: raw window hit that leaf slot k must have inside the AND
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

// ---- wave-wide reductions on the DPP path (no LDS crossbar, ~6 VALU) ----------------------
template <int CTRL, int ROW_MASK = 0xf, int BANK_MASK = 0xf>
__device__ __forceinline__ uint32_t dpp_or_zero(uint32_t v) {
    // lanes whose DPP source is invalid or masked receive 0
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, BANK_MASK, false);
}

// sum over the 64 lanes, returned in every lane
__device__ __forceinline__ uint32_t wave_sum_u32(uint32_t v) {
    v += dpp_or_zero<0xb1>(v);                  // quad_perm [1,0,3,2]
    v += dpp_or_zero<0x4e>(v);                  // quad_perm [2,3,0,1]
    v += dpp_or_zero<0x124>(v);                 // row_ror:4
    v += dpp_or_zero<0x128>(v);                 // row_ror:8   -> every lane holds its row total
    v += dpp_or_zero<0x142, 0xa>(v);            // row_bcast:15 into rows 1 and 3
    v += dpp_or_zero<0x143, 0xc>(v);            // row_bcast:31 into rows 2 and 3 -> lane 63 = total
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int lane) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, lane);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), lane);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t x) {
    uint32_t t = x;
    t += dpp_or_zero<0x111>(x);                 // row_shr:1
    t += dpp_or_zero<0x112>(x);                 // row_shr:2
    t += dpp_or_zero<0x113>(x);                 // row_shr:3   -> windows of 4
    t += dpp_or_zero<0x114, 0xf, 0xe>(t);       // row_shr:4   -> windows of 8
    t += dpp_or_zero<0x118, 0xf, 0xc>(t);       // row_shr:8   -> prefix inside each row of 16
    t += dpp_or_zero<0x142, 0xa>(t);            // + total of the previous row (rows 1, 3)
    t += dpp_or_zero<0x143, 0xc>(t);            // + lanes 0..31 (rows 2, 3)
    return t;
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
        // > 6 leaves: the jump program, evaluated for all R rows at once and BACKWARDS -- val[s] = rows that end
        // in ACCEPT when evaluation stands at step s = (leaf & val[on_true]) | (~leaf & val[on_false]); jumps only
        // go forward, so both operands are known.  ~45 VALU per leaf instead of a per-row walk of the program.
        uint32_t val[PQPS_MAX_LEAVES];
        const uint32_t full = R >= 32 ? 0xFFFFFFFFu : ((1u << R) - 1u);
        for (int s = (int)a.n_leaves - 1; s >= 0; s--) {        // uniform
            const uint32_t k = a.order[s], t = a.on_true[s], f = a.on_false[s];
            uint32_t leaf = 0;
#pragma unroll
            for (int r = 0; r < R; r++) leaf |= ((idx[r] >> k) & 1u) << r;
            const uint32_t vt = t == PQPS_ACCEPT ? full : (t == PQPS_REJECT ? 0u : val[t]);
            const uint32_t vf = f == PQPS_ACCEPT ? full : (f == PQPS_REJECT ? 0u : val[f]);
            val[s] = (leaf & vt) | (~leaf & vf);
        }
        m = val[0] & full;
    }
    return m;
}

// ---- row-mask evaluation (<= 6 leaves) -----------------------------------------------------
// A leaf is evaluated for the R rows of a lane into an R-bit mask (bit r = row r) with the
// cheapest compare that decides it: equality (span == 0), one-sided (lo == 0) or the window.
// The boolean tree is then applied ONCE per step on the leaf masks -- a handful of AND / OR /
// NOT on 16-bit masks -- instead of a truth-table lookup per row.
template <typename T, int R>
__device__ __forceinline__ uint32_t leaf_mask(const T (&v)[R], T lo, T span) {
    uint32_t m = 0;
    if (span == 0) {                                            // uniform: x == lo
#pragma unroll
        for (int r = R - 1; r >= 0; r--) m = (m << 1) | (v[r] == lo ? 1u : 0u);
    } else if (lo == 0) {                                       // uniform: x <= span
#pragma unroll
        for (int r = R - 1; r >= 0; r--) m = (m << 1) | (v[r] <= span ? 1u : 0u);
    } else {
#pragma unroll
        for (int r = R - 1; r >= 0; r--) m = (m << 1) | ((T)(v[r] - lo) <= span ? 1u : 0u);
    }
    return m;
}

struct LeafMasks { uint32_t m[PQPS_TT_LEAVES]; };

template <typename T, int R>
__device__ __forceinline__ void apply_leaves_masks(const EvalArgs &a, uint32_t kb, uint32_t ke,
                                                   const T (&v)[R], LeafMasks &lm) {
    for (uint32_t k = kb; k < ke; k++) {                        // uniform; operands come in SGPRs
        uint32_t m = leaf_mask<T, R>(v, (T)a.lo[k], (T)a.span[k]);
        if ((a.negmask >> k) & 1u) m = ~m;
        switch (k) {                                            // uniform: keeps lm in registers
        case 0: lm.m[0] = m; break;
        case 1: lm.m[1] = m; break;
        case 2: lm.m[2] = m; break;
        case 3: lm.m[3] = m; break;
        case 4: lm.m[4] = m; break;
        default: lm.m[5] = m; break;
        }
    }
}

// OR over the true rows of the truth table of AND over leaves (leaf or its complement).
// When more than half of the table is true the complement is expanded instead.
__device__ __forceinline__ uint32_t combine_masks(const EvalArgs &a, const LeafMasks &lm, uint32_t full) {
    const uint32_t n = a.n_leaves;
    const uint64_t all = n >= 6 ? ~0ull : ((1ull << (1u << n)) - 1ull);
    uint64_t tt = a.truth & all;
    const bool invert = (uint32_t)__popcll(tt) > (1u << n) / 2;
    if (invert) tt = ~tt & all;
    uint32_t res = 0;
    while (tt) {                                                // uniform loop over true entries
        const uint32_t e = (uint32_t)__builtin_ctzll(tt);
        tt &= tt - 1;
        uint32_t term = full;
#pragma unroll
        for (uint32_t k = 0; k < PQPS_TT_LEAVES; k++)
            if (k < n) term &= ((e >> k) & 1u) ? lm.m[k] : ~lm.m[k];
        res |= term;
    }
    return (invert ? ~res : res) & full;
}

// ---- per-step output ---------------------------------------------------------------
// Bit p of a lane's 16 match bits <-> row  step_row0 + (p / RPL) * 64 * RPL + lane * RPL + p % RPL.
// 128 B of match bits of one step: the 16-bit words of 8 neighbouring lanes are gathered into
// one lane (DPP), 8 lanes store 16 B each.
__device__ __forceinline__ void store_mask(const EvalArgs &a, uint64_t step, uint32_t mbits, uint32_t lane) {
    const uint32_t w2 = (mbits & 0xFFFFu) | (dpp_or_zero<0xb1>(mbits) << 16);     // lane pairs (even lanes valid)
    const uint32_t w2b = dpp_or_zero<0x4e>(w2);                                    // lane+2's pair
    const uint32_t q0 = w2, q1 = w2b;                                              // lanes 0 mod 4: words 0..3
    const uint32_t q2 = dpp_or_zero<0x104>(q0), q3 = dpp_or_zero<0x104>(q1);       // row_shl:4 -> lane+4's words
    if ((lane & 7u) == 0) {
        uint4 v; v.x = q0; v.y = q1; v.z = q2; v.w = q3;
        *(uint4 *)(a.masks + step * 64 + lane) = v;
    }
}

template <int MODE>
__device__ __forceinline__ void emit_step(const EvalArgs &a, uint64_t step, uint32_t mbits, uint32_t rpl_log2,
                                          uint64_t n_rows, uint32_t lane, uint64_t &wave_total) {
    const uint32_t cnt = wave_sum_u32(__popc(mbits));
    if (MODE == MODE_IDS) {
        if (cnt) store_mask(a, step, mbits, lane);               // 128 B per step, only if needed
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

// ---- chain path: the predicate is an AND of (possibly complemented) leaves, or its negation ----
// Every leaf-row compare writes its 64-lane result straight into an SGPR pair (one VALU, SDWA
// picks the byte / halfword); AND-ing the leaves and counting the matches is scalar-unit work;
// per-lane match bits are only materialised for steps that contain a match.
// A step is evaluated in two halves of 8 row slots: 8 planes (16 SGPRs) live at a time instead of 16,
// which is what brings the kernel under 96 SGPRs, i.e. to 8 waves per SIMD instead of 7.
struct RowPlanes { uint64_t p[8]; };                            // half H: p[i] bit l = row slot 8H + i of lane l

template <typename T, int H>
__device__ __forceinline__ void chain_leaf(const T (&v)[16], T lo, T span, bool want, RowPlanes &acc) {
    // `want`: the raw window hit this leaf needs; the four branches are wave-uniform
    if (span == 0) {
        if (want) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot(v[8 * H + r] == lo);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot(v[8 * H + r] != lo);
        }
    } else if (lo == 0) {
        if (want) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot(v[8 * H + r] <= span);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot(v[8 * H + r] > span);
        }
    } else {
        if (want) {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot((T)(v[8 * H + r] - lo) <= span);
        } else {
#pragma unroll
            for (int r = 0; r < 8; r++) acc.p[r] &= __ballot((T)(v[8 * H + r] - lo) > span);
        }
    }
}

// Folds one evaluated half into the step's match count and (ID output, only if the half has a match)
// into the lanes' match-bit words.
template <int MODE, int H>
__device__ __forceinline__ void fold_half(const EvalArgs &a, RowPlanes &acc, uint32_t &cnt, uint32_t &mbits) {
    if (a.chain == 2) {                                         // OR form: NOT of the AND
#pragma unroll
        for (int r = 0; r < 8; r++) acc.p[r] = ~acc.p[r];
    }
    uint64_t any = 0;
#pragma unroll
    for (int r = 0; r < 8; r++) any |= acc.p[r];
    if (any) {                                                  // uniform
#pragma unroll
        for (int r = 0; r < 8; r++) cnt += (uint32_t)__popcll(acc.p[r]);
        if (MODE == MODE_IDS) {
#pragma unroll
            for (int r = 0; r < 8; r++) mbits |= __builtin_amdgcn_inverse_ballot_w64(acc.p[r]) ? (1u << (8 * H + r)) : 0u;
        }
    }
}

template <int MODE>
__device__ __forceinline__ void emit_chain_step(const EvalArgs &a, uint64_t step, uint32_t cnt, uint32_t mbits,
                                                uint32_t rpl_log2, uint32_t lane, uint64_t &wave_total) {
    static_assert(MODE != MODE_FLAGS, "flag output goes through the generic kernel");
    if (MODE == MODE_IDS) {
        if (cnt) store_mask(a, step, mbits, lane);
        if (lane == 0) a.counts[step] = cnt | (rpl_log2 << 28);
    } else {
        wave_total += cnt;
    }
}

// COUNT / FLAGS modes: every workgroup adds its total into one of kPartialSlots counters (the grid
// can be far larger than that; spread over 4096 addresses the atomics do not queue up), which
// reduce_totals_kernel sums and leaves zeroed for the next query.
constexpr uint32_t kPartialSlots = 4096;

template <int MODE>
__device__ __forceinline__ void finish_totals(const EvalArgs &a, uint64_t wave_total) {
    if (MODE == MODE_IDS) return;
    __shared__ uint64_t s_tot[kWaves];
    if ((threadIdx.x & 63) == 0) s_tot[threadIdx.x >> 6] = wave_total;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int i = 0; i < kWaves; i++) t += s_tot[i];
        if (t) atomicAdd((unsigned long long *)&a.partials[blockIdx.x & (kPartialSlots - 1)], (unsigned long long)t);
    }
}

// plain or streaming (`nt`) loads, see RawChunk::load
template <bool NT> __device__ __forceinline__ uint4 ld_x4(const void *p) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    if constexpr (NT) { const u32x4 q = __builtin_nontemporal_load((const u32x4 *)p); return make_uint4(q.x, q.y, q.z, q.w); }
    else return *(const uint4 *)p;
}
template <bool NT> __device__ __forceinline__ uint2 ld_x2(const void *p) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    if constexpr (NT) { const u32x2 q = __builtin_nontemporal_load((const u32x2 *)p); return make_uint2(q.x, q.y); }
    else return *(const uint2 *)p;
}
template <bool NT> __device__ __forceinline__ uint32_t ld_x1(const void *p) {
    if constexpr (NT) return __builtin_nontemporal_load((const uint32_t *)p);
    else return *(const uint32_t *)p;
}

// ---- generic evaluators (any number of columns / leaves), RPL = 4 -----------------------
// Fast path: all 1024 rows of the step exist and are contiguous.
template <bool NT>
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
                    const uint4 q0 = ld_x4<NT>(p);
                    const uint4 q1 = ld_x4<NT>(p + 16);
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
                    const uint4 q = ld_x4<NT>(base + r0 * 4);
                    v[4 * u] = q.x; v[4 * u + 1] = q.y; v[4 * u + 2] = q.z; v[4 * u + 3] = q.w;
                } else if (wl == 1) {
                    const uint2 q = ld_x2<NT>(base + r0 * 2);
                    v[4 * u] = q.x & 0xFFFFu; v[4 * u + 1] = q.x >> 16;
                    v[4 * u + 2] = q.y & 0xFFFFu; v[4 * u + 3] = q.y >> 16;
                } else {
                    const uint32_t q = ld_x1<NT>(base + r0);
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
// K2 accumulates supergroup sums with atomics; the first workgroup of K1 clears them -- at its END,
// so that no workgroup starts with a kernel-argument round trip for something only K2 needs.
template <int MODE>
__device__ __forceinline__ void clear_super_sums(const EvalArgs &a) {
    if (MODE == MODE_IDS && blockIdx.x == 0)
        for (uint32_t i = threadIdx.x; i < a.n_super; i += kBlock) a.super_sum[(uint64_t)i * 512] = 0ull;   // kSuperStride
}

template <int MODE, bool GATHER, bool NT = false>
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
    // gather: the host sized the launch and counts[] for a.n_rows candidates, but only the steps of the
    // device-side range are evaluated (K2 / K3 derive the same bound): a narrow probe costs three
    // near-empty launches, not a pass over the whole table's step array
    const uint64_t steps_used = (n_rows + kStepRows - 1) / kStepRows;
    uint64_t wave_total = 0;
    for (uint64_t step = wave; step < steps_used; step += n_waves) {
        const uint64_t step_row0 = step * kStepRows;
        uint32_t mbits = 0;
        if (!GATHER && step_row0 + kStepRows <= n_rows) mbits = eval_step_full<NT>(a, step_row0, lane);
        else if (step_row0 < n_rows) mbits = eval_step_guarded<GATHER>(a, step_row0, n_rows, begin, lane);
        emit_step<MODE>(a, step, mbits, 2, n_rows, lane, wave_total);
    }
    clear_super_sums<MODE>(a);
    finish_totals<MODE>(a, wave_total);
}

// ---- width-specialised K1 ---------------------------------------------------------------
// Raw bytes of RPL consecutive rows of a W-byte column, as dwords.
template <int W, int RPL>
struct RawChunk {
    static constexpr int kBytes = W * RPL;                      // 32 (8-byte column), 16, 8 or 4
    static constexpr int kDwords = kBytes >= 4 ? kBytes / 4 : 1;
    uint32_t d[kDwords];
    // NT: streaming hint (`nt` modifier) -- the lines are not kept in L2 / Infinity Cache.  Pays once
    // the scan's footprint no longer fits the 256 MB Infinity Cache (measured crossover ~320 MB:
    // +9..12 % on 0.4..3 GB scans, -2..5 % on <= 300 MB ones that a repeated query finds cached).
    template <bool NT>
    __device__ __forceinline__ void load(const char *p) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
        if constexpr (NT) {
            if constexpr (kBytes == 32) {
                const u32x4 q = __builtin_nontemporal_load((const u32x4 *)p), r = __builtin_nontemporal_load((const u32x4 *)(p + 16));
                d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; d[4] = r.x; d[5] = r.y; d[6] = r.z; d[7] = r.w;
            } else if constexpr (kBytes == 16) { const u32x4 q = __builtin_nontemporal_load((const u32x4 *)p); d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }
            else if constexpr (kBytes == 8) { const u32x2 q = __builtin_nontemporal_load((const u32x2 *)p); d[0] = q.x; d[1] = q.y; }
            else if constexpr (kBytes == 4) { d[0] = __builtin_nontemporal_load((const uint32_t *)p); }
            else { d[0] = __builtin_nontemporal_load((const uint16_t *)p); }
        } else {
            if constexpr (kBytes == 32) {
                const uint4 q = *(const uint4 *)p, r = *(const uint4 *)(p + 16);
                d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; d[4] = r.x; d[5] = r.y; d[6] = r.z; d[7] = r.w;
            } else if constexpr (kBytes == 16) { const uint4 q = *(const uint4 *)p; d[0] = q.x; d[1] = q.y; d[2] = q.z; d[3] = q.w; }
            else if constexpr (kBytes == 8) { const uint2 q = *(const uint2 *)p; d[0] = q.x; d[1] = q.y; }
            else if constexpr (kBytes == 4) { d[0] = *(const uint32_t *)p; }
            else { d[0] = *(const uint16_t *)p; }
        }
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
    template <bool NT>
    __device__ __forceinline__ void load(const void *base, uint64_t lane_row0) {
#pragma unroll
        for (int u = 0; u < U; u++) c[u].template load<NT>((const char *)base + (lane_row0 + (uint64_t)u * 64 * RPL) * W);
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
__device__ __forceinline__ void eval_col_masks(const EvalArgs &a, int slot, const RawCol<W, RPL, U> &raw, LeafMasks &lm) {
    const uint32_t kb = a.leaf_begin[slot], ke = a.leaf_begin[slot + 1];
    if constexpr (W == 8) {
        uint64_t v[16];
        unpack64(raw, v, std::make_integer_sequence<int, 16>{});
        apply_leaves_masks<uint64_t, 16>(a, kb, ke, v, lm);
    } else {
        uint32_t v[16];
        unpack32(raw, v, std::make_integer_sequence<int, 16>{});
        apply_leaves_masks<uint32_t, 16>(a, kb, ke, v, lm);
    }
}

template <int W, int RPL, int U, int H>
__device__ __forceinline__ void eval_col_chain(const EvalArgs &a, int slot, const RawCol<W, RPL, U> &raw, RowPlanes &acc) {
    const uint32_t kb = a.leaf_begin[slot], ke = a.leaf_begin[slot + 1];
    if constexpr (W == 8) {
        uint64_t v[16];
        unpack64(raw, v, std::make_integer_sequence<int, 16>{});
        for (uint32_t k = kb; k < ke; k++) chain_leaf<uint64_t, H>(v, a.lo[k], a.span[k], (a.chain_want >> k) & 1u, acc);
    } else {
        uint32_t v[16];
        unpack32(raw, v, std::make_integer_sequence<int, 16>{});
        for (uint32_t k = kb; k < ke; k++)
            chain_leaf<uint32_t, H>(v, (uint32_t)a.lo[k], (uint32_t)a.span[k], (a.chain_want >> k) & 1u, acc);
    }
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

// All raw registers of one step for the (W0, W1, W2) shape.
template <int W0, int W1, int W2, int RPL, int U>
struct RawStep {
    RawCol<W0, RPL, U> r0;
    RawCol<(W1 ? W1 : 1), RPL, U> r1;
    RawCol<(W2 ? W2 : 1), RPL, U> r2;
    template <bool NT>
    __device__ __forceinline__ void load(const EvalArgs &a, uint64_t lane_row0) {
        r0.template load<NT>(a.col[0], lane_row0);
        if constexpr (W1 != 0) r1.template load<NT>(a.col[1], lane_row0);
        if constexpr (W2 != 0) r2.template load<NT>(a.col[2], lane_row0);
    }
    template <int MODE, int H>
    __device__ __forceinline__ void eval_chain_half(const EvalArgs &a, uint32_t &cnt, uint32_t &mbits) const {
        RowPlanes acc;
#pragma unroll
        for (int r = 0; r < 8; r++) acc.p[r] = ~0ull;
        eval_col_chain<W0, RPL, U, H>(a, 0, r0, acc);
        if constexpr (W1 != 0) eval_col_chain<W1, RPL, U, H>(a, 1, r1, acc);
        if constexpr (W2 != 0) eval_col_chain<W2, RPL, U, H>(a, 2, r2, acc);
        fold_half<MODE, H>(a, acc, cnt, mbits);
    }
    // One comparison on one column (`sudo_used = TRUE`, `risk_level > 3`): all on the vector unit.  Each row
    // slot is one compare plus one add-with-carry -- the compare's per-lane result enters `m + m + hit` (match
    // bits, MSB first) or `total + hit` (COUNT) as the carry -- so a step costs ~35 VALU and next to no SALU.
    // The ballot path above costs ~65 SALU per step even for one leaf, and a CU has ONE scalar unit: a 1-byte
    // column needs a step per 79 cycles per CU to keep up with HBM, which the scalar unit cannot deliver.
    template <int MODE, typename T, typename Hit>
    __device__ __forceinline__ void one_leaf_rows(const T (&v)[16], Hit hit, uint32_t &m, uint32_t &lane_total) const {
        if (MODE == MODE_IDS) {
#pragma unroll
            for (int r = 15; r >= 0; r--) m = m + m + (hit(v[r]) ? 1u : 0u);
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) lane_total += hit(v[r]) ? 1u : 0u;
        }
    }
    template <int MODE, typename T>
    __device__ __forceinline__ void one_leaf(const EvalArgs &a, const T (&v)[16], uint32_t &m, uint32_t &lane_total) const {
        const T lo = (T)a.lo[0], span = (T)a.span[0];
        const bool want = ((a.chain_want & 1u) != 0) != (a.chain == 2);      // OR form of one leaf = its negation
        if (span == 0) {                                                     // the six branches are wave-uniform
            if (want) one_leaf_rows<MODE>(v, [lo](T x) { return x == lo; }, m, lane_total);
            else one_leaf_rows<MODE>(v, [lo](T x) { return x != lo; }, m, lane_total);
        } else if (lo == 0) {
            if (want) one_leaf_rows<MODE>(v, [span](T x) { return x <= span; }, m, lane_total);
            else one_leaf_rows<MODE>(v, [span](T x) { return x > span; }, m, lane_total);
        } else {
            if (want) one_leaf_rows<MODE>(v, [lo, span](T x) { return (T)(x - lo) <= span; }, m, lane_total);
            else one_leaf_rows<MODE>(v, [lo, span](T x) { return (T)(x - lo) > span; }, m, lane_total);
        }
    }
    // VC: the vector-unit variant for ONE comparison on ONE column -- a kernel of its own, so that its registers
    // do not weigh on the ballot path's occupancy.  (The same idea for chains of two or three leaves was
    // measured too: no gain -- with 3+ bytes per row the scalar unit is not what limits the scan.)
    template <int MODE, bool VC>
    __device__ __forceinline__ void eval_chain_emit(const EvalArgs &a, uint64_t step, uint32_t rpl_log2, uint32_t lane,
                                                    uint64_t &wave_total, uint32_t &lane_total) const {
        if constexpr (VC) {
            static_assert(W1 == 0 && W2 == 0, "one column");
            uint32_t m = 0;
            if constexpr (W0 == 8) {
                uint64_t v[16];
                unpack64(r0, v, std::make_integer_sequence<int, 16>{});
                one_leaf<MODE, uint64_t>(a, v, m, lane_total);
            } else {
                uint32_t v[16];
                unpack32(r0, v, std::make_integer_sequence<int, 16>{});
                one_leaf<MODE, uint32_t>(a, v, m, lane_total);
            }
            if (MODE == MODE_IDS) {
                const uint32_t cnt = wave_sum_u32(__popc(m));
                if (cnt) store_mask(a, step, m, lane);
                if (lane == 0) a.counts[step] = cnt | (rpl_log2 << 28);
            }
        } else {
            uint32_t cnt = 0, mbits = 0;
            eval_chain_half<MODE, 0>(a, cnt, mbits);
            eval_chain_half<MODE, 1>(a, cnt, mbits);
            emit_chain_step<MODE>(a, step, cnt, mbits, rpl_log2, lane, wave_total);
        }
    }
    __device__ __forceinline__ uint32_t eval(const EvalArgs &a) const {     // <= 6 leaves: row-mask path
        LeafMasks lm;
#pragma unroll
        for (int k = 0; k < PQPS_TT_LEAVES; k++) lm.m[k] = 0;
        eval_col_masks<W0, RPL, U>(a, 0, r0, lm);
        if constexpr (W1 != 0) eval_col_masks<W1, RPL, U>(a, 1, r1, lm);
        if constexpr (W2 != 0) eval_col_masks<W2, RPL, U>(a, 2, r2, lm);
        return combine_masks(a, lm, 0xFFFFu);
    }
};

// W0 >= W1 >= W2 are the byte widths of the predicate columns (0 = slot unused).
// The kernel arguments the first loads depend on, fetched together at the very top: left to itself the
// compiler fetches them where first used, three dependent scalar-load round trips (~0.6 us) before a
// wave has a byte of the table in flight -- which a one-shot workgroup pays on every launch.
#define PQPS_HOIST_KERNARGS(a)                                                                        \
    asm volatile("" :: "s"((a).n_rows), "s"((a).col[0]), "s"((a).col[1]), "s"((a).col[2]), "s"(gridDim.x),  \
                 "s"((a).masks), "s"((a).counts), "s"((a).chain), "s"((a).chain_want), "s"((a).negmask),     \
                 "s"((uint32_t)(a).leaf_begin[0]), "s"((uint32_t)(a).leaf_begin[1]), "s"((uint32_t)(a).leaf_begin[2]), \
                 "s"((uint32_t)(a).leaf_begin[3]))

// General tree of <= 6 leaves (row-mask path), one step per iteration.
template <int MODE, int W0, int W1, int W2, bool NT>
__global__ __launch_bounds__(kBlock) void eval_spec_kernel(const EvalArgs a) {
    // consecutive rows per lane per chunk: the widest column is one dwordx4 per chunk
    // (an 8-byte column: two, so that RPL stays in {4, 8, 16})
    constexpr int RPL = W0 == 8 ? 4 : 16 / W0;
    constexpr int U = 16 / RPL;                                 // chunks per step
    PQPS_HOIST_KERNARGS(a);
    const uint64_t n_rows = a.n_rows;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t full_steps = n_rows / kStepRows;
    const uint64_t lane_off = lane * RPL;
    uint64_t wave_total = 0;
    for (uint64_t step = wave; step < full_steps; step += n_waves) {
        RawStep<W0, W1, W2, RPL, U> A;
        A.template load<NT>(a, step * kStepRows + lane_off);
        emit_step<MODE>(a, step, A.eval(a), log2i(RPL), n_rows, lane, wave_total);
    }
    // the partial last step (if any) goes through the guarded evaluator, RPL = 4 layout
    if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {
        const uint32_t mbits = eval_step_guarded<false>(a, full_steps * kStepRows, n_rows, 0, lane);
        emit_step<MODE>(a, full_steps, mbits, 2, n_rows, lane, wave_total);
    }
    clear_super_sums<MODE>(a);
    finish_totals<MODE>(a, wave_total);
}

// Chain predicates (AND of possibly complemented leaves, or the negation of one): SGPR planes.
// A wave keeps the loads of S steps in flight (all S x columns loads are issued, then the steps
// are evaluated one after the other).
// Steps per loop iteration (adjacent steps, so the chip-wide access window stays one contiguous range).
// Measured, fraction of 8 TB/s at 100 M / 1 B rows: a lone 1-byte column wants 2 (COUNT 0.71 / 0.86, IDs
// 0.61 / 0.69; 1 step: 0.56 / 0.48; 4: 0.72 / 0.86 and 0.60 / 0.65; 8: worse) -- 1 KB per wave and step is
// too little in flight; from 2 bytes per row on, 1 is best (u16+u8: 0.83 / 0.82 against 0.80 / 0.82 with 2
// and 0.77 / 0.80 with 4).
constexpr int chain_steps(int w0, int w1, int w2) { return w0 + w1 + w2 == 1 ? 2 : 1; }

template <int MODE, int W0, int W1, int W2, int S, bool NT, bool VC>
__global__ __launch_bounds__(kBlock) void eval_chain_kernel(const EvalArgs a) {
    constexpr int RPL = W0 == 8 ? 4 : 16 / W0;
    constexpr int U = 16 / RPL;
    PQPS_HOIST_KERNARGS(a);
    const uint64_t n_rows = a.n_rows;
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6);
    const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
    const uint64_t full_steps = n_rows / kStepRows;
    const uint64_t lane_off = lane * RPL;
    uint64_t wave_total = 0;
    uint32_t lane_total = 0;                                    // COUNT, one-leaf path: per-lane matches, summed once at the end
    RawStep<W0, W1, W2, RPL, U> A[S];
    // a wave takes S ADJACENT steps per iteration: the chip-wide access window stays one contiguous range
    for (uint64_t step0 = wave * S; step0 < full_steps; step0 += n_waves * S) {
#pragma unroll
        for (int i = 0; i < S; i++) {
            const uint64_t step = step0 + (uint64_t)i;
            if (step < full_steps) A[i].template load<NT>(a, step * kStepRows + lane_off);      // uniform guard
        }
#pragma unroll
        for (int i = 0; i < S; i++) {
            const uint64_t step = step0 + (uint64_t)i;
            if (step >= full_steps) break;
            A[i].template eval_chain_emit<MODE, VC>(a, step, log2i(RPL), lane, wave_total, lane_total);
        }
    }
    if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {
        const uint32_t mbits = eval_step_guarded<false>(a, full_steps * kStepRows, n_rows, 0, lane);
        emit_step<MODE>(a, full_steps, mbits, 2, n_rows, lane, wave_total);
    }
    if (MODE != MODE_IDS) wave_total += wave_sum_u32(lane_total);
    clear_super_sums<MODE>(a);
    finish_totals<MODE>(a, wave_total);
}

// ---- K2: group sums ---------------------------------------------------------------------
// group = 64 steps (64 K rows); supergroup = 64 groups (4 M rows).
constexpr int kSuperGroups = 64;
// Device atomics that hit one cache line serialise (~5 ns each, measured), so every
// supergroup counter lives in its own 4 KiB slot.
constexpr int kSuperStride = 512;           // in u64

struct SumArgs {
    const uint32_t *counts;          // [steps]
    uint64_t steps;
    uint64_t groups;                 // ceil(steps / 64)
    uint32_t *group_sum;             // [groups]
    unsigned long long *super_sum;   // [ceil(groups / 64) * kSuperStride], zeroed by K1
    uint64_t *base_slot;             // scratch: first output slot of this query
    const uint64_t *out_count;       // device result counter (read when accumulate)
    int accumulate;                  // 1: IDs are appended after *out_count (index probes)
    const uint64_t *range;           // gather: device-side candidate range (else nullptr)
    uint64_t max_rows;               // gather: the caller's bound on the range length
};

// Steps K1 evaluated in gather mode (same clamp as eval_generic_kernel).
__device__ __forceinline__ uint64_t gather_steps(const uint64_t *range, uint64_t max_rows) {
    const uint64_t b = range[0], e = range[1];
    uint64_t n = e > b ? e - b : 0;
    if (n > max_rows) n = max_rows;
    return (n + kStepRows - 1) / kStepRows;
}

__global__ __launch_bounds__(kBlock) void group_sum_kernel(const SumArgs a) {
    const uint32_t lane = threadIdx.x & 63;
    if (blockIdx.x == 0 && threadIdx.x == 0) *a.base_slot = a.accumulate ? *a.out_count : 0;
    const uint64_t steps = a.range ? gather_steps(a.range, a.max_rows) : a.steps;
    const uint64_t groups = (steps + kGroupSteps - 1) / kGroupSteps;
    for (uint64_t group = (uint64_t)blockIdx.x * kWaves + (threadIdx.x >> 6); group < groups;
         group += (uint64_t)gridDim.x * kWaves) {
        const uint64_t step = group * kGroupSteps + lane;
        const uint32_t c = step < steps ? (a.counts[step] & 0x0FFFFFFFu) : 0u;
        const uint32_t sum = wave_sum_u32(c);
        if (lane == 0) {
            a.group_sum[group] = sum;
            if (sum) atomicAdd(&a.super_sum[(group / kSuperGroups) * kSuperStride], (unsigned long long)sum);
        }
    }
}

// COUNT / FLAGS modes: workgroup partial totals -> one number (no same-address atomics)
__global__ __launch_bounds__(kBlock) void reduce_totals_kernel(uint64_t *partials, uint64_t *out_count) {
    __shared__ uint64_t s_wave[kWaves];
    uint64_t local = 0;
    for (uint32_t i = threadIdx.x; i < kPartialSlots; i += kBlock) {
        const uint64_t v = partials[i];
        if (v) { local += v; partials[i] = 0; }                 // left zeroed for the next query
    }
    local = wave_sum_u64(local);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint64_t t = 0;
        for (int i = 0; i < kWaves; i++) t += s_wave[i];
        *out_count = t;
    }
}

// ---- K3: match bits -> ascending row IDs ---------------------------------------------------
struct ExpandArgs {
    const uint16_t *masks;
    const uint32_t *counts;
    const uint32_t *group_sum;
    const unsigned long long *super_sum;
    const uint64_t *base_slot;
    uint64_t *out_count;             // written by the workgroup that owns the last group
    uint64_t steps;
    uint64_t groups;
    uint32_t *out_ids;
    uint64_t out_cap;
    const uint32_t *cand;            // gather: candidate list
    const uint64_t *range;
    uint64_t max_rows;               // gather: the caller's bound on the range length
    uint32_t id_base;
    uint32_t gather;
    uint64_t wave_groups_min;        // from this many groups on: one group per WAVE (see expand_kernel)
};

constexpr int kStageIds = kStepRows;        // a step yields at most 1024 IDs

// One step: K1 left 16 match bits per lane in its load layout (bit p of lane l <-> row
// (p / RPL) * 64 * RPL + l * RPL + p % RPL).  First bring them into ROW order -- lane d gets the
// bits of rows 16d .. 16d+15, which sit in 16/RPL source lanes -- then one wave scan gives every
// lane its output rank; IDs are staged in LDS and written out with fully coalesced stores.
template <int RL>                                               // log2(RPL): 2, 3 or 4
__device__ __forceinline__ void expand_step(const ExpandArgs &a, uint64_t step, uint32_t m16, uint32_t count,
                                            uint64_t out_off, uint64_t begin, uint32_t lane, uint32_t *stage) {
    constexpr uint32_t RPL = 1u << RL, S = 16u / RPL;              // S source lanes per destination lane
    uint32_t word;
    if constexpr (S == 1) {
        word = m16;
    } else {
        constexpr uint32_t LPC = 64u / S;                           // destination lanes per chunk
        const uint32_t u = lane / LPC, first = S * (lane % LPC);
        word = 0;
#pragma unroll
        for (uint32_t q = 0; q < S; q++) {
            const uint32_t src = (uint32_t)__shfl((int)m16, (int)(first + q), 64);
            word |= ((src >> (RPL * u)) & ((1u << RPL) - 1u)) << (RPL * q);
        }
    }
    const uint32_t cnt = __popc(word);
    const uint32_t incl = wave_incl_scan_u32(cnt);
    uint32_t pos = incl - cnt;
    const uint32_t r0 = (uint32_t)(step * kStepRows) + lane * 16u;
    if (a.gather) {                                                 // uniform
        // candidate numbers of the set bits, all requested before any is used (one memory latency per
        // step, not one per ID); a set bit implies the row lies inside the probed range
        uint32_t c[16];
#pragma unroll
        for (uint32_t j = 0; j < 16; j++) c[j] = ((word >> j) & 1u) ? a.cand[begin + r0 + j] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 16; j++)
            if ((word >> j) & 1u) stage[pos++] = c[j] + a.id_base;
    } else {
        while (word) {                                              // set bits only, ascending rows
            const uint32_t j = (uint32_t)__builtin_ctz(word);
            word &= word - 1;
            stage[pos++] = r0 + j + a.id_base;
        }
    }
    // same wave wrote and reads: DS operations of one wave complete in order; the asm only
    // stops the compiler from moving the reads above the writes
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (uint32_t k = lane; k < count; k += 64) {
        const uint64_t o = out_off + k;
        if (o < a.out_cap) a.out_ids[o] = stage[k];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // reads done before the next step overwrites
}

__device__ __forceinline__ void expand_step_any(const ExpandArgs &a, uint64_t step, uint32_t m16, uint32_t rpl_log2,
                                                uint32_t count, uint64_t out_off, uint64_t begin, uint32_t lane,
                                                uint32_t *stage) {
    switch (rpl_log2) {                                             // uniform
    case 2: expand_step<2>(a, step, m16, count, out_off, begin, lane, stage); break;
    case 3: expand_step<3>(a, step, m16, count, out_off, begin, lane, stage); break;
    default: expand_step<4>(a, step, m16, count, out_off, begin, lane, stage); break;
    }
}

// Workgroup per group of 64 steps.  Its first output slot is computed on the fly from the
// supergroup sums (<= a few hundred values) and the <= 63 earlier group sums of its own
// supergroup, so no serial scan kernel is needed.  The 4 waves then share the non-empty steps
// round-robin and fetch the match bits of up to 4 steps at a time.
__global__ __launch_bounds__(kBlock) void expand_kernel(const ExpandArgs a) {
    __shared__ uint32_t s_part[kWaves];
    __shared__ uint32_t s_stage[kWaves][kStageIds];
    __shared__ uint16_t s_mask[kWaves][kGroupSteps / kWaves][64];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t begin = a.gather ? a.range[0] : 0;
    const uint64_t base0 = *a.base_slot;
    const uint64_t steps = a.gather ? gather_steps(a.range, a.max_rows) : a.steps;
    const uint64_t groups = (steps + kGroupSteps - 1) / kGroupSteps;
    if (groups >= a.wave_groups_min) {                                // uniform for the grid
        // very many groups (>= 0.5 G rows): a group per WAVE.  The per-group chain of dependent loads
        // (sums in front -> counts -> match words -> IDs) is latency, not work; with four times as
        // many groups in flight the same chip finishes in a quarter of the rounds.  No barriers.
        for (uint64_t group = (uint64_t)blockIdx.x * kWaves + wave; group < groups; group += (uint64_t)gridDim.x * kWaves) {
            const uint64_t my_step = group * kGroupSteps + lane;
            const uint32_t cw = my_step < steps ? a.counts[my_step] : 0u;
            const uint64_t sg = group / kSuperGroups, g_in = group % kSuperGroups;
            uint32_t psum = 0;
            for (uint64_t j = lane; j < sg; j += 256) {              // four independent loads per trip
                const uint64_t j1 = j + 64, j2 = j + 128, j3 = j + 192;
                const unsigned long long s0 = a.super_sum[j * kSuperStride];
                const unsigned long long s1 = j1 < sg ? a.super_sum[j1 * kSuperStride] : 0ull;
                const unsigned long long s2 = j2 < sg ? a.super_sum[j2 * kSuperStride] : 0ull;
                const unsigned long long s3 = j3 < sg ? a.super_sum[j3 * kSuperStride] : 0ull;
                psum += (uint32_t)s0 + (uint32_t)s1 + (uint32_t)s2 + (uint32_t)s3;
            }
            if (lane < g_in) psum += a.group_sum[sg * kSuperGroups + lane];
            const uint32_t my_cnt = cw & 0x0FFFFFFFu;
            const uint64_t nonempty = __ballot(my_cnt != 0);
            const bool last_group = group + 1 == groups;
            if (nonempty == 0 && !last_group) continue;             // uniform for the wave
            const uint64_t group_off = base0 + wave_sum_u32(psum);
            const uint32_t incl = wave_incl_scan_u32(my_cnt);
            const uint64_t my_off = group_off + (incl - my_cnt);
            if (last_group && lane == 63) *a.out_count = group_off + incl;
            const uint16_t *gmask = a.masks + group * kGroupSteps * 64 + lane;
            for (uint32_t c = 0; c < kGroupSteps; c += 16) {          // 16 consecutive steps at a time
                const uint32_t bits = (uint32_t)(nonempty >> c) & 0xFFFFu;
                if (!bits) continue;
                uint32_t mreg[16];
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    mreg[i] = 0;
                    if ((bits >> i) & 1u) mreg[i] = gmask[(size_t)(c + i) * 64];                          // uniform branch
                }
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) s_mask[wave][i][lane] = (uint16_t)mreg[i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                for (uint32_t i = 0; i < 16; i++) {
                    if (!((bits >> i) & 1u)) continue;
                    const int sidx = (int)(c + i);
                    const uint64_t step_off = readlane_u64(my_off, sidx);           // sidx is wave-uniform: v_readlane, no LDS crossbar
                    const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, sidx);
                    expand_step_any(a, group * kGroupSteps + (uint64_t)sidx, s_mask[wave][i][lane], cwi >> 28,
                                    cwi & 0x0FFFFFFFu, step_off, begin, lane, s_stage[wave]);
                }
            }
        }
        return;
    }
    // few groups (a narrow index probe, a small table): split each over `parts` workgroups, every one
    // expanding 64 / parts of the group's steps, so that the work still spreads over the chip
    const uint32_t parts = groups >= 512 ? 1u : (groups >= 128 ? 4u : 16u);
    const uint32_t slots = (kGroupSteps / kWaves) / parts;            // step slots per wave: 16, 4 or 1
    for (uint64_t v = blockIdx.x; v < groups * parts; v += gridDim.x) {
        const uint64_t group = v / parts;
        const uint32_t part = (uint32_t)(v % parts);
        // (1) counts of the 64 steps (every wave loads the same 256 bytes) + the sums in front
        const uint64_t my_step = group * kGroupSteps + lane;
        const uint32_t cw = my_step < steps ? a.counts[my_step] : 0u;
        const uint64_t sg = group / kSuperGroups, g_in = group % kSuperGroups;
        uint32_t psum = 0;                                          // < 2^32: row IDs are u32
        for (uint64_t j = tid; j < sg; j += kBlock) psum += (uint32_t)a.super_sum[j * kSuperStride];
        if (tid < g_in) psum += a.group_sum[sg * kSuperGroups + tid];
        const uint32_t my_cnt = cw & 0x0FFFFFFFu;
        const uint64_t nonempty = __ballot(my_cnt != 0);
        const bool last_group = group + 1 == groups;
        if (nonempty == 0 && !last_group) continue;                 // uniform for the workgroup
        // (2) matches before this group
        psum = wave_sum_u32(psum);
        __syncthreads();                                            // previous iteration done with s_part
        if (lane == 0) s_part[wave] = psum;
        __syncthreads();
        uint64_t group_off = base0;
#pragma unroll
        for (int i = 0; i < kWaves; i++) group_off += s_part[i];
        // (3) exclusive prefix of the step counts inside the group
        const uint32_t incl = wave_incl_scan_u32(my_cnt);
        const uint64_t my_off = group_off + (incl - my_cnt);
        if (last_group && part == 0 && wave == 0 && lane == 63) *a.out_count = group_off + incl;
        // (4) wave w owns steps w, w+4, ... of the group (slot i <-> step w + 4i); this workgroup takes
        // the slots [part * slots, (part + 1) * slots).  All their match-bit words are requested at
        // once (one memory latency for up to 16 steps), parked in LDS, then expanded.
        const uint16_t *gmask = a.masks + (group * kGroupSteps + wave) * 64 + lane;
        const uint32_t slot0 = part * slots, slot1 = slot0 + slots;
        uint32_t mreg[kGroupSteps / kWaves];
#pragma unroll
        for (uint32_t i = 0; i < kGroupSteps / kWaves; i++) {
            mreg[i] = 0;
            if (i >= slot0 && i < slot1 && ((nonempty >> (wave + kWaves * i)) & 1ull)) mreg[i] = gmask[(size_t)i * kWaves * 64];   // uniform branch
        }
#pragma unroll
        for (uint32_t i = 0; i < kGroupSteps / kWaves; i++) s_mask[wave][i][lane] = (uint16_t)mreg[i];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        for (uint32_t i = slot0; i < slot1; i++) {
            const int sidx = (int)(wave + kWaves * i);
            if (!((nonempty >> sidx) & 1ull)) continue;
            const uint64_t step_off = readlane_u64(my_off, sidx);           // sidx is wave-uniform: v_readlane, no LDS crossbar
            const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, sidx);
            expand_step_any(a, group * kGroupSteps + (uint64_t)sidx, s_mask[wave][i][lane], cwi >> 28,
                            cwi & 0x0FFFFFFFu, step_off, begin, lane, s_stage[wave]);
        }
    }
}

}  // namespace
