// filter_kernels.hpp -- device code of the SELECT/WHERE filter (included once by pqps_hip.hip).
//
// Replaces linearSearchRecords + evaluateWhereClause + checkCondition + CMP_* of the
// reference (engine/serial/executeEngine-serial.c:854-878, :292-316, :251-289, :18-123).
//
// ID output (the row-ID list of linearSearchRecords) is ONE launch whose workgroups play two roles:
//
//  scan      a workgroup = one TILE of 4 (8 for a lone 1-byte column) consecutive steps, a wave = one STEP
//            of 1024 consecutive rows (16 rows per lane), no loop: the dispatcher deals the tiles out in
//            address order, so the chip reads one contiguous, advancing window of every predicate column.
//            Lane l owns RPL = 16/Wmax (4 for 8-byte) consecutive rows of each 64*RPL-row chunk, so the
//            widest predicate column is ONE fully coalesced global_load_dwordx4 per chunk and narrower
//            ones dwordx2 / dword / ushort loads that are just as contiguous across the wave.  A leaf is
//            the unsigned window test ((x - lo) <= span) ^ neg; the boolean tree is a 64-entry truth
//            table (<= 6 leaves) or a jump table.  Output per step: 16 match bits per lane (128 bytes,
//            skipped when the step has no match) and the step's match count; per tile ONE agent-scope
//            atomic add of (steps, matches) to the word of its GROUP (64 steps = 64 K rows).
//  expand    a workgroup per group, placed in the grid `lag` groups behind the group's scan tiles: waits
//            (normally not at all) until its group and everything in front of it has arrived, derives
//            its first output slot from the supergroup words (64 groups each) and the earlier group
//            words of its own supergroup, and turns the match bits into ascending row IDs.  Its integer
//            work runs in the shadow of the bandwidth-bound scan tiles around it; only the last `lag`
//            groups are expanded after the last table byte has been read.
//
// Hand-off between the roles follows the write-through form of the CDNA4 guide: payload (match bits,
// step counts) stored sc1, every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's
// barrier, one lane signals with an agent-scope atomic; the consumer polls with sc1 loads and reads the
// payload with sc1 loads only.  No result depends on dispatch order: an expander's wait is bounded, a
// group whose wait ran out is left to the expander that is last to leave its wait (one workgroup, so it
// cannot starve the tiles it waits for), and the words of a query are zeroed by the next query on the
// other half of a ping-pong pair.
//
// COUNT(*) / DELETE flags keep the grid-stride form of the scan (no ID list, no hand-off).
// Width-specialised instantiations (1-3 predicate columns, widths non-increasing) keep all loads
// statically scheduled; everything else takes the generic kernel.
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
    uint32_t *counts;                // [steps]     matches | log2(RPL) << 28; all zero between queries
    uint8_t *out_flags;              // MODE_FLAGS
    uint64_t *partials;              // MODE_COUNT / MODE_FLAGS: [gridDim.x] workgroup totals
    const uint32_t *cand;            // gather: candidate row numbers
    const uint64_t *range;           // gather: [begin, end) into cand, device resident
    // ---- ID output: hand-off words of this query (one half of the ping-pong pair) and the result ----
    uint64_t *gword;                 // [groups]  steps arrived << 48 | matches
    uint64_t *sword;                 // [supers]  groups forwarded << 48 | matches
    uint32_t *ctl;                   // [kCtlWords] expanders past their wait, deferred groups
    uint32_t *deferred;              // [groups]  1 = left to the recovery pass, | 2 = already forwarded
    uint64_t *zgword, *zsword;       // the other half: zeroed here for the query after this one
    uint32_t *zctl, *zdeferred;
    uint64_t zero_groups;            // ... as far as its last query can have written
    uint64_t *base_slot;             // gather: first output slot (= *out_count when the launch began)
    uint32_t *status;                // sticky error word of the context (a wait that never ended)
    uint32_t *out_ids;
    uint64_t out_cap;
    uint64_t *out_count;
    uint64_t block_base;             // added to blockIdx.x (the trailing expanders can be a launch of their own)
    uint32_t id_base;
    uint32_t lag;                    // groups between a group's scan tiles and its expander in the grid
    uint32_t spin_limit;             // polls before an expander leaves its group to the recovery pass
    uint32_t accumulate;             // gather: append behind *out_count
    uint32_t n_cols;
    uint32_t n_leaves;
    uint32_t negmask;
    uint32_t streaming;              // host side only: the scan outgrows the Infinity Cache (grid + load policy)
    uint32_t steps_per_iter;         // host side only: S of the chosen kernel (grid sizing)
    uint32_t valu_chain;             // host side only: the one-leaf vector-unit kernel variant was chosen
    uint32_t chain;                  // 0: general tree; 1: AND of leaves; 2: NOT(AND) = OR form (spec kernels)
    uint32_t chain_want;             // bit k: raw window hit that leaf slot k must have inside the AND
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
// 128 B of match bits of one step: the 16-bit words of 4 neighbouring lanes are gathered into one lane
// (DPP), 16 lanes store 8 B each -- write-through (sc1): the expander that reads them runs on another CU,
// possibly on another XCD whose L2 never sees this one's dirty lines.
#define PQPS_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
__device__ __forceinline__ uint64_t ld_sc1(const uint64_t *p) { return __hip_atomic_load(p, PQPS_AGENT); }
__device__ __forceinline__ uint32_t ld_sc1(const uint32_t *p) { return __hip_atomic_load(p, PQPS_AGENT); }
__device__ __forceinline__ uint16_t ld_sc1(const uint16_t *p) { return __hip_atomic_load(p, PQPS_AGENT); }
__device__ __forceinline__ void st_sc1(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, PQPS_AGENT); }
__device__ __forceinline__ void st_sc1(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, PQPS_AGENT); }
// every store of the calling wave has reached the memory side (the asm is invisible to the passes that
// drop a builtin wait they can prove redundant)
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__device__ __forceinline__ void store_mask(const EvalArgs &a, uint64_t step, uint32_t mbits, uint32_t lane) {
    const uint32_t w2 = (mbits & 0xFFFFu) | (dpp_or_zero<0xb1>(mbits) << 16);     // even lanes: own word | next lane's
    const uint32_t w2b = dpp_or_zero<0x4e>(w2);                                    // lanes 0 mod 4: the pair of lane + 2
    if ((lane & 3u) == 0)
        st_sc1((uint64_t *)(a.masks + step * 64 + lane), (uint64_t)w2 | ((uint64_t)w2b << 32));
}

// COUNT / FLAGS modes (grid-stride scan, no hand-off)
template <int MODE>
__device__ __forceinline__ void emit_step(const EvalArgs &a, uint64_t step, uint32_t mbits, uint32_t rpl_log2,
                                          uint64_t n_rows, uint32_t lane, uint64_t &wave_total) {
    static_assert(MODE != MODE_IDS, "ID output is the tile / expander form");
    wave_total += wave_sum_u32(__popc(mbits));
    if (MODE == MODE_FLAGS) {
        const uint32_t rpl = 1u << rpl_log2;
        for (uint32_t p = 0; p < 16; p++) {
            const uint64_t row = step * kStepRows + (uint64_t)(p >> rpl_log2) * 64 * rpl + lane * rpl + (p & (rpl - 1));
            if (row < n_rows) a.out_flags[row] = (uint8_t)((mbits >> p) & 1u);
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

// COUNT / FLAGS modes: every workgroup adds its total into one of kPartialSlots counters (the grid
// can be far larger than that; spread over 4096 addresses the atomics do not queue up), which
// reduce_totals_kernel sums and leaves zeroed for the next query.
constexpr uint32_t kPartialSlots = 4096;

template <int MODE>
__device__ __forceinline__ void finish_totals(const EvalArgs &a, uint64_t wave_total) {
    static_assert(MODE != MODE_IDS, "ID output is the tile / expander form");
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

// ---- ID output: tiles, groups, and the hand-off between scan and expand workgroups ------------------
// group = 64 steps (64 K rows); supergroup = 64 groups (4 M rows).  A hand-off word = arrivals << 48 | matches.
constexpr int kSuperGroups = 64;
constexpr int kArrShift = 48;
constexpr uint64_t kSumMask = (1ull << kArrShift) - 1ull;
constexpr int kCtlWords = 16;               // ctl[0]: expanders past their wait; ctl[1]: groups left to the recovery pass
constexpr uint32_t kDirectIds = 192;        // a step with at most this many matches writes its IDs lane by lane
constexpr uint32_t kRecoverSpins = 1u << 26; // the recovery pass gives up (sticky status word) after this many polls

struct FusedShared {
    uint16_t mask[kWaves][kGroupSteps / kWaves][64];   // expander: match words of a wave's 16 steps
    uint32_t counts[kGroupSteps];                       // expander: step counts of the group
    uint32_t tile_cnt[16];                              // scan: step counts of the tile
    uint64_t group_off;                                 // expander: first output slot of the group
    uint32_t state;                                     // expander: 1 = expand now, 0 = deferred
    uint32_t recover;                                   // expander: this workgroup runs the recovery pass
};

// What this launch covers: a scan of n_rows rows, or (gather) the device-side candidate range clamped
// to the caller's bound.
struct Extent { uint64_t begin, n_rows, steps, groups; };

template <bool GATHER>
__device__ __forceinline__ Extent scan_extent(const EvalArgs &a) {
    Extent e;
    e.begin = 0;
    e.n_rows = a.n_rows;
    if (GATHER) {
        e.begin = a.range[0];
        const uint64_t end = a.range[1];
        uint64_t n = end > e.begin ? end - e.begin : 0;
        if (n > a.n_rows) n = a.n_rows;                         // never past the caller's bound
        e.n_rows = n;
    }
    e.steps = (e.n_rows + kStepRows - 1) / kStepRows;
    e.groups = (e.steps + kGroupSteps - 1) / kGroupSteps;
    return e;
}

// Grid layout (TPG tiles per group): the tiles of group q are followed by the expander of group q - lag;
// the expanders of the last `lag` groups come after the last tile.  Placement is for speed only -- an
// expander checks what it needs and waits (bounded) if it is early.
enum { ROLE_NONE = 0, ROLE_SCAN = 1, ROLE_EXPAND = 2 };
struct Role { uint32_t kind; uint32_t index; };

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uniform_u64(uint64_t v) {
    return (uint64_t)uniform_u32((uint32_t)v) | ((uint64_t)uniform_u32((uint32_t)(v >> 32)) << 32);
}

// All in 32 bits (the host keeps a launch under 2^31 workgroups) and pinned to SGPRs: the role and everything
// derived from it is wave-uniform, and a 64-bit division would be done -- and then kept -- in vector registers.
template <int TPG>
__device__ __forceinline__ Role fused_role(const EvalArgs &a, uint32_t groups) {
    constexpr uint32_t period = TPG + 1;
    const uint32_t b = blockIdx.x + (uint32_t)a.block_base;
    const uint32_t main_blocks = groups * period;
    const uint32_t lag = a.lag < groups ? a.lag : groups;
    Role r;
    r.kind = ROLE_NONE;
    r.index = 0;
    if (b < main_blocks) {
        const uint32_t q = b / period, rr = b % period;
        if (rr < TPG) { r.kind = ROLE_SCAN; r.index = q * TPG + rr; }
        else if (q >= lag) { r.kind = ROLE_EXPAND; r.index = q - lag; }
    } else if (b - main_blocks < lag) {
        r.kind = ROLE_EXPAND;
        r.index = groups - lag + (b - main_blocks);
    }
    r.kind = uniform_u32(r.kind);
    r.index = uniform_u32(r.index);
    return r;
}

// The hand-off words of a query are zeroed by the NEXT query of the context, which runs on the other half
// of the ping-pong pair (plain stores: the kernel boundary publishes them).
__device__ __forceinline__ void zero_other_half(const EvalArgs &a) {
    if (a.block_base != 0) return;
    const uint64_t n = a.zero_groups, ns = (n + kSuperGroups - 1) / kSuperGroups;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (uint64_t)gridDim.x * kBlock) {
        a.zgword[i] = 0ull;
        a.zdeferred[i] = 0u;
        if (i < ns) a.zsword[i] = 0ull;
    }
    if (blockIdx.x == 0 && threadIdx.x < kCtlWords) a.zctl[threadIdx.x] = 0u;
}

// End of a scan tile of TS steps.  Every wave has left the counts of its steps in sh.tile_cnt, drained its
// match-word stores and passed the workgroup's barrier; wave 0 publishes: the non-zero counts (the array is
// all zero between queries), then ONE atomic add of (steps, matches) to the tile's group word.
template <int TS>
__device__ __forceinline__ void publish_tile(const EvalArgs &a, const FusedShared &sh, const Extent &ex, uint64_t tile, uint32_t lane) {
    const uint64_t first = tile * TS;
    const uint32_t steps_in_tile = ex.steps - first < (uint64_t)TS ? (uint32_t)(ex.steps - first) : (uint32_t)TS;
    const uint32_t c = lane < steps_in_tile ? sh.tile_cnt[lane] : 0u;
    const uint32_t total = wave_sum_u32(c & 0x0FFFFFFFu);
    bool stored = false;
    if (tile == 0 && a.accumulate) {                            // gather: results are appended behind *out_count
        if (lane == 0) st_sc1(a.base_slot, *a.out_count);
        stored = true;
    }
    if (total) {
        if (c & 0x0FFFFFFFu) st_sc1(a.counts + first + lane, c);
        stored = true;
    }
    if (stored) drain_stores();
    if (lane == 0)
        __hip_atomic_fetch_add(a.gword + first / kGroupSteps, ((uint64_t)steps_in_tile << kArrShift) | (uint64_t)total, PQPS_AGENT);
}

// One wave's share of a scan tile: count word into LDS, match words (if any) to memory and drained.
__device__ __forceinline__ void tile_step_out(const EvalArgs &a, FusedShared &sh, uint32_t slot, uint64_t step, uint32_t cnt,
                                              uint32_t mbits, uint32_t rpl_log2, uint32_t lane) {
    if (cnt) store_mask(a, step, mbits, lane);
    if (lane == 0) sh.tile_cnt[slot] = cnt | (rpl_log2 << 28);
}

// ---- expand: match words -> ascending row IDs ------------------------------------------------------------
// One step: the scan left 16 match bits per lane in its load layout (bit p of lane l <-> row
// (p / RPL) * 64 * RPL + l * RPL + p % RPL).  First bring them into ROW order -- lane d gets the bits of rows
// 16d .. 16d+15, which sit in 16/RPL source lanes.  Then
//   few matches:  one wave scan of the per-lane popcounts gives every lane its first output slot; a lane writes
//                 the IDs of its set bits one after the other (neighbouring lanes write neighbouring slots);
//   many matches: 64 rows at a time -- their match bits are the words of 4 lanes, read into an SGPR pair with
//                 v_readlane; rank inside the 64 = mbcnt, so the 64 lanes store to consecutive slots.
// Neither form touches LDS.
template <int RL>                                               // log2(RPL): 2, 3 or 4
__device__ __forceinline__ uint32_t row_order_word(uint32_t m16, uint32_t lane) {
    constexpr uint32_t RPL = 1u << RL, S = 16u / RPL;              // S source lanes per destination lane
    if constexpr (S == 1) {
        return m16;
    } else {
        constexpr uint32_t LPC = 64u / S;                           // destination lanes per chunk
        const uint32_t u = lane / LPC, first = S * (lane % LPC);
        uint32_t word = 0;
#pragma unroll
        for (uint32_t q = 0; q < S; q++) {
            const uint32_t src = (uint32_t)__shfl((int)m16, (int)(first + q), 64);
            word |= ((src >> (RPL * u)) & ((1u << RPL) - 1u)) << (RPL * q);
        }
        return word;
    }
}

__device__ __forceinline__ void expand_step(const EvalArgs &a, uint64_t begin, uint64_t step, uint32_t m16, uint32_t rpl_log2,
                                            uint32_t count, uint64_t out_off, uint32_t lane) {
    uint32_t word;
    switch (rpl_log2) {                                             // uniform
    case 2: word = row_order_word<2>(m16, lane); break;
    case 3: word = row_order_word<3>(m16, lane); break;
    default: word = m16; break;
    }
    const bool gather = a.cand != nullptr;                          // uniform
    const uint32_t step_row0 = (uint32_t)(step * kStepRows);
    if (count <= kDirectIds) {
        const uint32_t cnt = __popc(word);
        const uint32_t incl = wave_incl_scan_u32(cnt);
        uint64_t pos = out_off + (incl - cnt);
        const uint32_t r0 = step_row0 + lane * 16u;
        while (word) {                                              // set bits only, ascending rows
            const uint32_t j = (uint32_t)__builtin_ctz(word);
            word &= word - 1;
            // gather: the candidate number of the row (a set bit implies the row lies inside the probed range)
            const uint32_t id = gather ? a.cand[begin + r0 + j] : r0 + j;
            if (pos < a.out_cap) a.out_ids[pos] = id + a.id_base;
            pos++;
        }
    } else {
        uint64_t base = out_off;
#pragma unroll 1
        for (uint32_t s = 0; s < 16; s++) {
            // the match bits of rows 64s .. 64s+63 are the words of lanes 4s .. 4s+3: wave-uniform lane numbers, so
            // v_readlane (SGPR result), no LDS crossbar
            const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s));
            const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 1));
            const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 2));
            const uint32_t w3 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 3));
            const uint64_t b = (uint64_t)(w0 | (w1 << 16)) | ((uint64_t)(w2 | (w3 << 16)) << 32);
            if (b) {                                                // uniform
                if ((b >> lane) & 1ull) {
                    const uint64_t o = base + mbcnt(b);
                    const uint32_t row = step_row0 + 64u * s + lane;
                    const uint32_t id = gather ? a.cand[begin + row] : row;
                    if (o < a.out_cap) a.out_ids[o] = id + a.id_base;
                }
                base += (uint64_t)__popcll(b);
            }
        }
    }
}

// Steps of group g that exist (the last group can be short).
__device__ __forceinline__ uint32_t group_steps(const Extent &ex, uint64_t g) {
    const uint64_t left = ex.steps - g * kGroupSteps;
    return left < (uint64_t)kGroupSteps ? (uint32_t)left : (uint32_t)kGroupSteps;
}

// Polls (one wave, all lanes the same word) until every step of group g has arrived.
__device__ __forceinline__ bool wait_group(const EvalArgs &a, uint64_t g, uint32_t need, uint32_t limit, uint64_t &sum) {
    for (uint32_t spins = 0;; spins++) {
        const uint64_t w = ld_sc1(a.gword + g);
        if ((uint32_t)(w >> kArrShift) == need) { sum = w & kSumMask; return true; }
        if (spins >= limit) return false;
        __builtin_amdgcn_s_sleep(8);
    }
}

// Polls until everything in front of group g has arrived -- the supergroups before its own (every group
// forwarded) and the groups before it in its own supergroup -- and sums their matches.
__device__ __forceinline__ bool wait_prefix(const EvalArgs &a, uint64_t g, uint32_t limit, uint32_t lane, uint64_t &psum) {
    const uint64_t sg = g / kSuperGroups, g_in = g % kSuperGroups;
    for (uint32_t spins = 0;; spins++) {
        bool ok = true;
        uint64_t acc = 0;
        for (uint64_t j = lane; j < sg; j += 64) {
            const uint64_t w = ld_sc1(a.sword + j);
            ok = ok && (uint32_t)(w >> kArrShift) == (uint32_t)kSuperGroups;
            acc += w & kSumMask;
        }
        if (lane < g_in) {
            const uint64_t w = ld_sc1(a.gword + sg * kSuperGroups + lane);
            ok = ok && (uint32_t)(w >> kArrShift) == (uint32_t)kGroupSteps;
            acc += w & kSumMask;
        }
        if (__all(ok)) { psum = wave_sum_u64(acc); return true; }
        if (spins >= limit) return false;
        __builtin_amdgcn_s_sleep(8);
    }
}

// Wave 0, once group g and everything in front of it has arrived: the group's step counts go to LDS for the
// four waves, and back to zero in memory for the next query.
__device__ __forceinline__ void fetch_group_counts(const EvalArgs &a, FusedShared &sh, const Extent &ex, uint64_t g, uint32_t lane) {
    const uint64_t step = g * kGroupSteps + lane;
    uint32_t c = 0;
    if (step < ex.steps) {
        c = ld_sc1(a.counts + step);
        if (c) st_sc1(a.counts + step, 0u);
    }
    sh.counts[lane] = c;
}

// All four waves: wave w turns the match words of steps 16w .. 16w+15 of group g into row IDs.
__device__ __forceinline__ void expand_group(const EvalArgs &a, FusedShared &sh, const Extent &ex, uint64_t g, uint32_t lane, uint32_t wave) {
    const uint32_t cw = sh.counts[lane];
    const uint64_t group_off = sh.group_off;
    const uint32_t my_cnt = cw & 0x0FFFFFFFu;
    const uint32_t incl = wave_incl_scan_u32(my_cnt);
    const uint64_t my_off = group_off + (incl - my_cnt);
    if (g + 1 == ex.groups && wave == 0 && lane == 63) *a.out_count = group_off + incl;
    const uint64_t nonempty = __ballot(my_cnt != 0);
    const uint32_t c0 = wave * (kGroupSteps / kWaves);
    const uint32_t bits = uniform_u32((uint32_t)(nonempty >> c0) & 0xFFFFu);
    if (!bits) return;                                              // uniform for the wave
    // All match words of the wave's non-empty steps are requested at once (one memory latency) and parked in LDS:
    // slot k takes the k-th non-empty step; slots past the last one re-read the first (no branch between the
    // loads, and a line this wave has just asked for).
    const uint16_t *gmask = a.masks + (g * kGroupSteps + c0) * 64 + lane;
    const uint32_t first = (uint32_t)__builtin_ctz(bits);
    uint32_t rest = bits;
    uint32_t mreg[kGroupSteps / kWaves];
#pragma unroll
    for (uint32_t k = 0; k < kGroupSteps / kWaves; k++) {
        const uint32_t i = rest ? (uint32_t)__builtin_ctz(rest) : first;                 // wave-uniform
        rest &= rest - 1;
        mreg[k] = ld_sc1(gmask + (size_t)i * 64);
    }
#pragma unroll
    for (uint32_t k = 0; k < kGroupSteps / kWaves; k++) sh.mask[wave][k][lane] = (uint16_t)mreg[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // same wave wrote and reads
    rest = bits;
    for (uint32_t k = 0; rest; k++) {
        const int sidx = (int)(c0 + (uint32_t)__builtin_ctz(rest));
        rest &= rest - 1;
        const uint64_t step_off = readlane_u64(my_off, sidx);       // sidx is wave-uniform: v_readlane, no LDS crossbar
        const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, sidx);
        expand_step(a, ex.begin, g * kGroupSteps + (uint64_t)sidx, sh.mask[wave][k][lane], cwi >> 28, cwi & 0x0FFFFFFFu,
                    step_off, lane);
    }
}

// The expander workgroup of group g.
//
// Recovery: the ONE expander workgroup that is last to leave its wait looks after the groups others gave up
// on (which in-order dispatch never produces).  By then every other expander is past its wait, so this
// workgroup is the only one on the chip that waits for anything, and what it waits for are scan tiles, which
// wait for nothing.  It first forwards what was never forwarded, then runs the same body over the deferred
// groups in ascending order.
__device__ __forceinline__ void expander(const EvalArgs &a, FusedShared &sh, const Extent &ex, uint64_t g) {
    const uint32_t lane = threadIdx.x & 63, wave = uniform_u32(threadIdx.x >> 6);
    uint32_t ticket = 0;
    bool recovery = false;                                          // uniform for the workgroup
    uint64_t g0 = 0, todo = 0;                                      // recovery: deferred groups of [g0, g0 + 64) still to do
    for (;;) {
        if (wave == 0) {
            uint64_t sum = 0, psum = 0;
            uint32_t flags = 1u;
            bool ok = true;
            if (!recovery) {
                ok = wait_group(a, g, group_steps(ex, g), a.spin_limit, sum);
                if (ok) {
                    // forward the group's matches to its supergroup word as soon as they are known: later supergroups wait for it
                    if (lane == 0) __hip_atomic_fetch_add(a.sword + g / kSuperGroups, (1ull << kArrShift) | sum, PQPS_AGENT);
                    flags |= 2u;
                }
            }
            if (ok) ok = wait_prefix(a, g, recovery ? kRecoverSpins : a.spin_limit, lane, psum);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: payload loads stay behind the polls
            if (ok) {
                const uint64_t base = a.accumulate ? ld_sc1(a.base_slot) : 0ull;
                fetch_group_counts(a, sh, ex, g, lane);
                if (lane == 0) sh.group_off = base + psum;
            } else if (recovery) {
                if (lane == 0) st_sc1(a.status, 1u);                // something never arrived: reported, never silent
            } else {
                if (lane == 0) {
                    st_sc1(a.deferred + g, flags);
                    __hip_atomic_fetch_add(a.ctl + 1, 1u, PQPS_AGENT);
                }
                drain_stores();                                     // ... before this workgroup counts as past its wait
            }
            if (lane == 0) {
                if (!recovery) ticket = __hip_atomic_fetch_add(a.ctl + 0, 1u, PQPS_AGENT);   // used after the expansion: latency hidden
                sh.state = ok ? 1u : 0u;
            }
        }
        __syncthreads();
        if (sh.state) expand_group(a, sh, ex, g, lane, wave);
        if (!recovery) {
            if (wave == 0 && lane == 0)
                sh.recover = ((uint64_t)ticket + 1 == ex.groups && ld_sc1(a.ctl + 1) != 0u) ? 1u : 0u;
            __syncthreads();
            if (!sh.recover) return;
            recovery = true;
            if (wave == 0) {                                        // pass 1: forward what was never forwarded
                bool alive = true;
                for (uint64_t f0 = 0; alive && f0 < ex.groups; f0 += 64) {
                    const uint32_t f = f0 + lane < ex.groups ? ld_sc1(a.deferred + f0 + lane) : 0u;
                    uint64_t fw = __ballot((f & 3u) == 1u);         // deferred and not forwarded
                    while (alive && fw) {
                        const uint64_t gg = f0 + (uint64_t)__builtin_ctzll(fw);
                        fw &= fw - 1;
                        uint64_t sum = 0;
                        alive = wait_group(a, gg, group_steps(ex, gg), kRecoverSpins, sum);
                        if (alive && lane == 0) __hip_atomic_fetch_add(a.sword + gg / kSuperGroups, (1ull << kArrShift) | sum, PQPS_AGENT);
                    }
                }
                if (!alive && lane == 0) st_sc1(a.status, 1u);
            }
            g0 = 0;
            const uint32_t f = lane < ex.groups ? ld_sc1(a.deferred + lane) : 0u;
            todo = __ballot((f & 1u) != 0u);
        }
        // next deferred group (every wave reads the same flags: nobody writes them any more)
        while (todo == 0) {
            g0 += 64;
            if (g0 >= ex.groups) return;
            const uint32_t f = g0 + lane < ex.groups ? ld_sc1(a.deferred + g0 + lane) : 0u;
            todo = __ballot((f & 1u) != 0u);
        }
        g = g0 + (uint64_t)__builtin_ctzll(todo);
        todo &= todo - 1;
        __syncthreads();                                            // everyone is done with the previous group's LDS
    }
}

// Generic scan: any predicate; scan (full steps vectorised) or gather (always guarded).
template <int MODE, bool GATHER, bool NT = false>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? 4 : 1) void eval_generic_kernel(const EvalArgs a) {
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if constexpr (MODE == MODE_IDS) {
        constexpr int TS = kWaves;                              // steps per tile
        __shared__ FusedShared sh;
        zero_other_half(a);
        const Extent ex = scan_extent<GATHER>(a);
        const Role role = fused_role<kGroupSteps / TS>(a, (uint32_t)ex.groups);
        if (role.kind == ROLE_EXPAND) { expander(a, sh, ex, role.index); return; }
        if (role.kind != ROLE_SCAN || (uint64_t)role.index * TS >= ex.steps) return;
        const uint64_t step = (uint64_t)role.index * TS + wv;
        uint32_t cnt = 0;
        if (step < ex.steps) {
            const uint64_t step_row0 = step * kStepRows;
            uint32_t mbits;
            if (!GATHER && step_row0 + kStepRows <= ex.n_rows) mbits = eval_step_full<NT>(a, step_row0, lane);
            else mbits = eval_step_guarded<GATHER>(a, step_row0, ex.n_rows, ex.begin, lane);
            cnt = wave_sum_u32(__popc(mbits));
            tile_step_out(a, sh, wv, step, cnt, mbits, 2, lane);
        } else if (lane == 0) {
            sh.tile_cnt[wv] = 0;
        }
        if (cnt) drain_stores();
        __syncthreads();
        if (wv == 0) publish_tile<TS>(a, sh, ex, role.index, lane);
    } else {
        static_assert(!GATHER, "gather mode produces ID lists");
        const uint64_t wave = (uint64_t)blockIdx.x * kWaves + wv;
        const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
        const uint64_t n_rows = a.n_rows;
        const uint64_t steps_used = (n_rows + kStepRows - 1) / kStepRows;
        uint64_t wave_total = 0;
        for (uint64_t step = wave; step < steps_used; step += n_waves) {
            const uint64_t step_row0 = step * kStepRows;
            uint32_t mbits = 0;
            if (step_row0 + kStepRows <= n_rows) mbits = eval_step_full<NT>(a, step_row0, lane);
            else mbits = eval_step_guarded<false>(a, step_row0, n_rows, 0, lane);
            emit_step<MODE>(a, step, mbits, 2, n_rows, lane, wave_total);
        }
        finish_totals<MODE>(a, wave_total);
    }
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
    // One step of a chain predicate: ID output -> the step's match count and the lanes' match bits;
    // COUNT -> cnt (ballot path) or lane_total (vector-unit path, summed once per wave at the end).
    template <int MODE, bool VC>
    __device__ __forceinline__ void eval_chain_step(const EvalArgs &a, uint32_t &cnt, uint32_t &mbits, uint32_t &lane_total) const {
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
                cnt = wave_sum_u32(__popc(m));
                mbits = m;
            }
        } else {
            eval_chain_half<MODE, 0>(a, cnt, mbits);
            eval_chain_half<MODE, 1>(a, cnt, mbits);
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
                 "s"((a).masks), "s"((a).chain), "s"((a).chain_want), "s"((a).negmask),     \
                 "s"((uint32_t)(a).leaf_begin[0]), "s"((uint32_t)(a).leaf_begin[1]), "s"((uint32_t)(a).leaf_begin[2]), \
                 "s"((uint32_t)(a).leaf_begin[3]))

// General tree of <= 6 leaves (row-mask path).
template <int MODE, int W0, int W1, int W2, bool NT>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? 8 : 1) void eval_spec_kernel(const EvalArgs a) {
    // consecutive rows per lane per chunk: the widest column is one dwordx4 per chunk
    // (an 8-byte column: two, so that RPL stays in {4, 8, 16})
    constexpr int RPL = W0 == 8 ? 4 : 16 / W0;
    constexpr int U = 16 / RPL;                                 // chunks per step
    PQPS_HOIST_KERNARGS(a);
    const uint64_t n_rows = a.n_rows;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t full_steps = n_rows / kStepRows;
    const uint64_t lane_off = lane * RPL;
    if constexpr (MODE == MODE_IDS) {
        constexpr int TS = kWaves;
        __shared__ FusedShared sh;
        zero_other_half(a);
        const Extent ex = scan_extent<false>(a);
        const Role role = fused_role<kGroupSteps / TS>(a, (uint32_t)ex.groups);
        if (role.kind == ROLE_EXPAND) { expander(a, sh, ex, role.index); return; }
        if (role.kind != ROLE_SCAN || (uint64_t)role.index * TS >= ex.steps) return;
        const uint64_t step = (uint64_t)role.index * TS + wv;
        uint32_t cnt = 0;
        if (step < full_steps) {
            RawStep<W0, W1, W2, RPL, U> A;
            A.template load<NT>(a, step * kStepRows + lane_off);
            const uint32_t mbits = A.eval(a);
            cnt = wave_sum_u32(__popc(mbits));
            tile_step_out(a, sh, wv, step, cnt, mbits, log2i(RPL), lane);
        } else if (step < ex.steps) {                           // the partial last step: guarded evaluator, RPL = 4 layout
            const uint32_t mbits = eval_step_guarded<false>(a, step * kStepRows, n_rows, 0, lane);
            cnt = wave_sum_u32(__popc(mbits));
            tile_step_out(a, sh, wv, step, cnt, mbits, 2, lane);
        } else if (lane == 0) {
            sh.tile_cnt[wv] = 0;
        }
        if (cnt) drain_stores();
        __syncthreads();
        if (wv == 0) publish_tile<TS>(a, sh, ex, role.index, lane);
    } else {
        const uint64_t wave = (uint64_t)blockIdx.x * kWaves + wv;
        const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
        uint64_t wave_total = 0;
        for (uint64_t step = wave; step < full_steps; step += n_waves) {
            RawStep<W0, W1, W2, RPL, U> A;
            A.template load<NT>(a, step * kStepRows + lane_off);
            emit_step<MODE>(a, step, A.eval(a), log2i(RPL), n_rows, lane, wave_total);
        }
        if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {
            const uint32_t mbits = eval_step_guarded<false>(a, full_steps * kStepRows, n_rows, 0, lane);
            emit_step<MODE>(a, full_steps, mbits, 2, n_rows, lane, wave_total);
        }
        finish_totals<MODE>(a, wave_total);
    }
}

// Chain predicates (AND of possibly complemented leaves, or the negation of one): SGPR planes.
// A wave keeps the loads of S steps in flight (all S x columns loads are issued, then the steps
// are evaluated one after the other).
// Steps per wave (adjacent steps, so the chip-wide access window stays one contiguous range).
// Measured, fraction of 8 TB/s at 100 M / 1 B rows: a lone 1-byte column wants 2 (COUNT 0.71 / 0.86;
// 1 step: 0.56 / 0.48; 4: 0.72 / 0.86; 8: worse) -- 1 KB per wave and step is too little in flight; from
// 2 bytes per row on, 1 is best (u16+u8: 0.83 / 0.82 against 0.80 / 0.82 with 2 and 0.77 / 0.80 with 4).
constexpr int chain_steps(int w0, int w1, int w2) { return w0 + w1 + w2 == 1 ? 2 : 1; }

template <int MODE, int W0, int W1, int W2, int S, bool NT, bool VC>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? 8 : 1) void eval_chain_kernel(const EvalArgs a) {
    constexpr int RPL = W0 == 8 ? 4 : 16 / W0;
    constexpr int U = 16 / RPL;
    PQPS_HOIST_KERNARGS(a);
    const uint64_t n_rows = a.n_rows;
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t full_steps = n_rows / kStepRows;
    const uint64_t lane_off = lane * RPL;
    RawStep<W0, W1, W2, RPL, U> A[S];
    if constexpr (MODE == MODE_IDS) {
        constexpr int TS = kWaves * S;                          // a tile = S adjacent steps per wave
        __shared__ FusedShared sh;
        zero_other_half(a);
        const Extent ex = scan_extent<false>(a);
        const Role role = fused_role<kGroupSteps / TS>(a, (uint32_t)ex.groups);
        if (role.kind == ROLE_EXPAND) { expander(a, sh, ex, role.index); return; }
        if (role.kind != ROLE_SCAN || (uint64_t)role.index * TS >= ex.steps) return;
        const uint64_t step0 = (uint64_t)role.index * TS + (uint64_t)wv * S;
#pragma unroll
        for (int i = 0; i < S; i++)
            if (step0 + i < full_steps) A[i].template load<NT>(a, (step0 + i) * kStepRows + lane_off);      // uniform guard
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < S; i++) {
            const uint64_t step = step0 + (uint64_t)i;
            uint32_t cnt = 0, mbits = 0, lane_total = 0;
            if (step < full_steps) {
                A[i].template eval_chain_step<MODE, VC>(a, cnt, mbits, lane_total);
                tile_step_out(a, sh, wv * S + i, step, cnt, mbits, log2i(RPL), lane);
            } else if (step < ex.steps) {
                mbits = eval_step_guarded<false>(a, step * kStepRows, n_rows, 0, lane);
                cnt = wave_sum_u32(__popc(mbits));
                tile_step_out(a, sh, wv * S + i, step, cnt, mbits, 2, lane);
            } else if (lane == 0) {
                sh.tile_cnt[wv * S + i] = 0;
            }
            any |= cnt;
        }
        if (any) drain_stores();
        __syncthreads();
        if (wv == 0) publish_tile<TS>(a, sh, ex, role.index, lane);
    } else {
        const uint64_t wave = (uint64_t)blockIdx.x * kWaves + wv;
        const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
        uint64_t wave_total = 0;
        uint32_t lane_total = 0;                                // one-leaf path: per-lane matches, summed once at the end
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
                uint32_t cnt = 0, mbits = 0;
                A[i].template eval_chain_step<MODE, VC>(a, cnt, mbits, lane_total);
                wave_total += cnt;
            }
        }
        if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {
            const uint32_t mbits = eval_step_guarded<false>(a, full_steps * kStepRows, n_rows, 0, lane);
            emit_step<MODE>(a, full_steps, mbits, 2, n_rows, lane, wave_total);
        }
        wave_total += wave_sum_u32(lane_total);
        finish_totals<MODE>(a, wave_total);
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

}  // namespace
