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
//            table (<= 6 leaves) or a jump table.  Output per step, skipped when the step has no match: up to 104
//            matches as 16-bit entries (row inside the step's group) in the step's slot (store_list), more as a list of
//            16-bit row numbers in the context's list area (store_list16) or -- no list area, 1-byte columns,
//            gathers -- as 16 match bits per lane (128 bytes); per tile ONE store of its steps' COUNT WORDS
//            (epoch << 16 | log2(RPL) << 11 | matches) by wave 0 after the tile's barrier.  The first two
//            tiles of a group also do SUM DUTY for a group / supergroup `sum_lag` groups back (a hint).
//  expand    turns what the steps of a GROUP (64 steps = 64 K rows) left into ascending row IDs (entries and lists: a
//            copy at the step's own output offset; bit masks: ranked).  It needs the
//            group's count words (all tagged with this query's epoch => its match words are in memory)
//            and the matches in front of it: supergroup sums (64 groups each) + the earlier group sums of
//            its own supergroup -- tagged words published by the tiles' sum duty or, failing that, by the
//            groups' own expanders.  Placed behind the last tile (a workgroup per group) or, for large
//            tables, `lag` groups behind the group's tiles (a wave per group), where its integer work
//            runs in the shadow of the bandwidth-bound scan.
//
// Hand-off between the roles follows the write-through form of the CDNA4 guide and uses NO atomics: payload
// (match bits) stored sc1, every storing wave drains (s_waitcnt vmcnt(0)) before the workgroup's barrier,
// one wave publishes the epoch-tagged count words; the consumer polls with sc1 loads and reads the payload
// with sc1 loads only.  Nothing has to be reset between queries (a stale word carries an older epoch).  No
// result depends on dispatch order: an expander's wait is bounded, a group whose wait ran out is left to
// the last group's leader once every leader is past its wait (sharded counters: the only atomics in the launch, no
// return values; zeroed by the query before, on the other half of a ping-pong pair), and a wait that never
// ends there sets the context's sticky status word.
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
constexpr uint32_t kGatherParts = 16;       // gather (index mode): expander workgroups per group (each 64 / 16 = 4 steps, one per wave)
constexpr int kSlotWords = 64;              // 16-bit entries of a step's slot in each of the two list areas of the sparse steps (128 bytes)

enum Mode { MODE_IDS = 0, MODE_COUNT = 1, MODE_FLAGS = 2 };

struct EvalArgs {
    const void *col[PQPS_MAX_COLUMNS];
    uint64_t lo[PQPS_MAX_LEAVES];
    uint64_t span[PQPS_MAX_LEAVES];
    uint64_t truth;
    uint64_t n_rows;                 // scan: rows; gather: caller's upper bound (range is on the device)
    uint16_t *masks;                 // [steps][64] match bits of every lane (steps that left a bit mask)
    uint16_t *slots;                 // [steps][64] 16-bit entries of the steps with few matches (store_list): entries 0 .. 63
    uint16_t *slots_hi;              // [steps][64] ... entries 64 .. 127 (two dense arrays of 128-byte lines: one array of 256-byte slots
                                     //            of which a few-percent answer fills the first half costs the scan 2 - 5 %)
    uint32_t *counts;                // [steps]     epoch << 16 | log2(RPL) << 11 | matches
    uint64_t *tiny;                  // [steps]     epoch << 48 | up to three 16-bit entries (store_tiny): the steps with 1 - 3 matches
    uint16_t *lists;                 // [steps][1024] row lists of the fuller steps (see store_list16), or nullptr: bit masks for those
    uint8_t *out_flags;              // MODE_FLAGS
    uint64_t *partials;              // MODE_COUNT / MODE_FLAGS: [gridDim.x] workgroup totals
    const uint32_t *cand;            // gather: candidate row numbers
    const uint64_t *range;           // gather: [begin, end) into cand, device resident
    const void *key_col;             // gather through an index (pqps_index_select): the column the index was built on, or nullptr
    const void *keys;                //   ... and its sorted keys: keys[i] = key_col[cand[i]], read in place of the scattered column
    // ---- ID output: hand-off words (epoch-tagged: valid when the tag is this query's) and the result ----
    uint64_t *gsum;                  // [groups]  epoch << 48 | matches of the group
    uint64_t *ssum;                  // [supers]  epoch << 48 | matches of the supergroup
    uint32_t *deferred;              // [groups]  epoch << 16 | 1 = given up, left to the recovery pass | 2 = its sum is published
    uint32_t *ctl;                   // [kCtlWords] expanders past their wait, deferred groups (zero at launch)
    uint32_t *zctl;                  // the other half of the ctl pair: zeroed here for the query after this one
    uint64_t *base_slot;             // gather: first output slot (= *out_count when the launch began)
    uint32_t *status;                // [kStatusWords] error words of the context (a wait that never ended): word epoch % kStatusWords = epoch
    uint32_t *out_ids;
    uint64_t out_cap;
    uint64_t *out_count;
    uint32_t id_base;
    uint32_t lag;                    // groups between a group's scan tiles and its expander in the grid
    uint32_t sum_lag;                // groups between a group's scan tiles and the tile that sums it up (< lag)
    uint32_t grid_groups;            // gather: groups the grid was sized for (a wider range: the workgroups loop)
    uint32_t spin_limit;             // polls before an expander leaves its group to the recovery pass
    // (32-bit fields: a 16-bit kernel argument picked by a run-time index is fetched with a VECTOR load, on the tile's path)
    uint32_t list16_min, list16_min_u8;   // a step with MORE matches than this leaves a 16-bit row list (_u8: widest predicate column 1 byte wide)
    uint32_t list_max, list_max_u8;       // a step with at most this many matches leaves them as 16-bit entries in its slot (<= kListIds; _u8: unused, 0)
    uint32_t tiny_max;                    // a step with at most this many matches (<= kTinyIds; 0: never) leaves them in its tiny word
    uint32_t tune;                   // A/B switches of tuning runs (PQPS_TUNE): bit 0 = no second look ahead of early expander waves
    uint32_t accumulate;             // gather: append behind *out_count
    uint32_t epoch;                  // 1 .. 65535, unique among the queries whose words can still be around
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
#ifdef PQPS_STAMPS   /* development builds: wall-clock stamps of tiles and expanders (scripts/stamps.py reads the dump) */
    uint64_t *stamps;                // [4 header][groups][8][tiles]
    uint64_t stamp_groups;
#endif
};

// Device code reads the arguments where the launch put them: the kernel-argument segment (constant address space, scalar
// loads).  The kernels take `EvalArgs` by value (that fixes the segment's layout) and bind this reference to its start; the
// expander paths go through a pointer the compiler cannot see through (args_for_expanders), so that the ~25 fields only
// they need are fetched after the role branch.  With the by-value parameter the compiler fetched everything at the top of
// the kernel, in five dependent rounds of scalar loads with spills to vector-register lanes in between -- 157 instructions
// and ~1 us before a scan tile had its first column load in flight (COUNT kernels: 61 instructions, one round), and a
// one-shot tile lives 2.5 - 3.5 us.
typedef const __attribute__((address_space(4))) EvalArgs CArgs;
__device__ __forceinline__ CArgs &kernel_args() { return *(CArgs *)__builtin_amdgcn_kernarg_segment_ptr(); }
__device__ __forceinline__ CArgs &args_for_expanders() {
    CArgs *p = (CArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return *p;
}

#ifdef PQPS_STAMPS
#define PQPS_STAMP_GROUP(a, g, k) do { if ((threadIdx.x & 63) == 0) (a).stamps[4 + (g) * 8 + (k)] = wall_clock64(); } while (0)
#define PQPS_STAMP_GROUP_MAX(a, g, k) do { if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long *)&(a).stamps[4 + (g) * 8 + (k)], (unsigned long long)wall_clock64()); } while (0)
#define PQPS_STAMP_VALUE(a, g, k, v) do { if ((threadIdx.x & 63) == 0) (a).stamps[4 + (g) * 8 + (k)] = (v); } while (0)
#define PQPS_STAMP_TILE(a, t) do { if (threadIdx.x == 0) (a).stamps[4 + (a).stamp_groups * 8 + (t)] = wall_clock64(); } while (0)
#define PQPS_STAMP_START(a) do { if (blockIdx.x == 0 && threadIdx.x == 0) (a).stamps[3] = wall_clock64(); } while (0)
#else
#define PQPS_STAMP_START(a) do { } while (0)
#define PQPS_STAMP_GROUP(a, g, k) do { } while (0)
#define PQPS_STAMP_GROUP_MAX(a, g, k) do { } while (0)
#define PQPS_STAMP_VALUE(a, g, k, v) do { } while (0)
#define PQPS_STAMP_TILE(a, t) do { } while (0)
#endif

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
__device__ __forceinline__ void apply_leaves(CArgs &a, uint32_t kb, uint32_t ke,
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
__device__ __forceinline__ uint32_t combine_leaves(CArgs &a, const uint32_t (&idx)[R]) {
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
__device__ __forceinline__ void apply_leaves_masks(CArgs &a, uint32_t kb, uint32_t ke,
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
__device__ __forceinline__ uint32_t combine_masks(CArgs &a, const LeafMasks &lm, uint32_t full) {
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

__device__ __forceinline__ void store_mask(CArgs &a, uint64_t step, uint32_t mbits, uint32_t lane) {
    const uint32_t w2 = (mbits & 0xFFFFu) | (dpp_or_zero<0xb1>(mbits) << 16);     // even lanes: own word | next lane's
    const uint32_t w2b = dpp_or_zero<0x4e>(w2);                                    // lanes 0 mod 4: the pair of lane + 2
    if ((lane & 3u) == 0)
        st_sc1((uint64_t *)(a.masks + step * 64 + lane), (uint64_t)w2 | ((uint64_t)w2b << 32));
}

// COUNT / FLAGS modes (grid-stride scan, no hand-off)
template <int MODE>
__device__ __forceinline__ void emit_step(CArgs &a, uint64_t step, uint32_t mbits, uint32_t rpl_log2,
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
__device__ __forceinline__ void fold_half(CArgs &a, RowPlanes &acc, uint32_t &cnt, uint32_t &mbits) {
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
__device__ __forceinline__ void finish_totals(CArgs &a, uint64_t wave_total) {
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

// Match bits of rows at or past n_rows never count.  Bit p of lane l <-> row step_row0 + (p / RPL) * 64 * RPL + l * RPL + p % RPL.
// (The partial last step of a scan is evaluated like a full one -- column buffers are readable up to the next
// multiple of 1024 rows, see pqps_filter_scan -- and trimmed with this.)
template <int RPL>
__device__ __forceinline__ uint32_t rows_below(uint64_t step_row0, uint64_t n_rows, uint32_t lane) {
    uint32_t m = 0;
#pragma unroll
    for (uint32_t p = 0; p < 16; p++)
        if (step_row0 + (uint64_t)(p / RPL) * 64 * RPL + (uint64_t)lane * RPL + p % RPL < n_rows) m |= 1u << p;
    return m;
}

// ---- generic evaluators (any number of columns / leaves), RPL = 4 -----------------------
// Fast path: all 1024 rows of the step exist and are contiguous.
template <bool NT>
__device__ __forceinline__ uint32_t eval_step_full(CArgs &a, uint64_t step_row0, uint32_t lane) {
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

// Gather (index mode), all 1024 positions of a step at once: the 16 candidate numbers of a lane in ONE round of loads, then
// one round per predicate column (16 element loads each) -- 1 + n_cols memory latencies per step.  The first form (a chunk of
// 4 positions per lane at a time) took 2 per chunk = 8 for the one-column predicate of an index query, most of the 22 us such a launch lasts (its
// 35 k candidates are 34 steps: nothing but latency).  The gather kernel may use 128 registers (two workgroups per CU).
__device__ __forceinline__ uint32_t eval_step_gather(CArgs &a, uint64_t step_row0, uint64_t n_rows, uint64_t begin, uint32_t lane) {
    constexpr int R = 16;
    uint32_t row[R], idx[R], live = 0;
#pragma unroll
    for (int p = 0; p < R; p++) {                                    // bit p <-> position (p / 4) * 256 + lane * 4 + p % 4 of the step
        const uint64_t r = step_row0 + (uint64_t)(p / 4) * 256 + lane * kRplGeneric + (p % 4);
        idx[p] = 0;
        if (r < n_rows) live |= 1u << p;
        row[p] = a.cand[begin + (r < n_rows ? r : 0)];              // (every lane loads: the range's first candidate is always there)
    }
    for (uint32_t c = 0; c < a.n_cols; c++) {                       // uniform
        const char *base = (const char *)a.col[c];
        const int wl = a.width_log2[c];
        const uint32_t kb = a.leaf_begin[c], ke = a.leaf_begin[c + 1];
        // The indexed column itself is not gathered: the value at candidate i is sorted key i, CONSECUTIVE in memory, whereas
        // the candidates' rows lie all over the table (one 64-byte sector per 4-byte value: `user_id` in [1001, 1100], 5.2 M
        // candidates of 100 M rows, moved 330 MB to look at 21 -- 135 us of a 140 us query).
        const bool sorted = a.keys != nullptr && (const void *)base == a.key_col;      // uniform
        if (sorted) base = (const char *)a.keys;
        uint32_t at[R];
#pragma unroll
        for (int p = 0; p < R; p++) {
            const uint64_t r = step_row0 + (uint64_t)(p / 4) * 256 + lane * kRplGeneric + (p % 4);
            at[p] = sorted ? (uint32_t)(begin + (r < n_rows ? r : 0)) : row[p];     // (row numbers and positions: below 2^32)
        }
        if (wl == 3) {
            uint64_t v[R];
#pragma unroll
            for (int p = 0; p < R; p++) v[p] = ((const uint64_t *)base)[at[p]];
            apply_leaves<uint64_t, R>(a, kb, ke, v, idx);
        } else {
            uint32_t v[R];
            if (wl == 2) {
#pragma unroll
                for (int p = 0; p < R; p++) v[p] = ((const uint32_t *)base)[at[p]];
            } else if (wl == 1) {
#pragma unroll
                for (int p = 0; p < R; p++) v[p] = ((const uint16_t *)base)[at[p]];
            } else {
#pragma unroll
                for (int p = 0; p < R; p++) v[p] = ((const uint8_t *)base)[at[p]];
            }
            apply_leaves<uint32_t, R>(a, kb, ke, v, idx);
        }
    }
    return combine_leaves<R>(a, idx) & live;                        // positions past the range never match
}

// ---- ID output: tiles, groups, and the hand-off between scan and expand workgroups ------------------
// group = 64 steps (64 K rows); supergroup = 64 groups (4 M rows).  Nothing in the hand-off is an atomic
// read-modify-write (agent-scope atomics that share a 128-byte line serialise at ~5 ns each -- 24 k of them cost
// more than the 100 M-row scan they signal for): every shared word is written per query by one lane with a
// write-through store and carries the query's EPOCH, so it is valid exactly when its tag matches -- no word
// needs zeroing between queries.
//   counts[step]  epoch << 16 | log2(RPL) << 11 | matches   written by the step's scan tile
//   gsum[group]   epoch << 48 | matches of the group        written by a scan tile `sum_lag` groups further on (sum duty)
//   ssum[super]   epoch << 48 | matches of the supergroup   written by a scan tile 2 * sum_lag groups past its end
// (the last groups of a table have no tile that far behind them: there the expanders write the words themselves).
// What lies in front of group g = the ssum of every supergroup before its own + the gsum of the earlier groups
// of its own: two dense arrays, read whole in one round of loads.  No word depends on another group's position
// in the output, so no chain of waits forms, and an expander placed 3 * sum_lag groups behind its group's tiles
// finds everything it needs at its first look.
constexpr int kSuperGroups = 64;
// ctl: the only words with atomic read-modify-writes, none of them on a path anything waits for, none with a return
// value.  A leader that is past its wait adds 1 to the shard of its group (64 shards, a 128-byte line each: neighbouring
// groups get there at the same time, and atomics that share a line serialise).  The leader of the LAST group is the one
// that looks whether any group was given up on -- once the shards add up to the number of groups, i.e. every leader is
// past its wait (expanders_past_their_wait).
constexpr int kCtlShards = 64, kCtlStride = 32;                     // u32 words per line
constexpr int kCtlWords = (kCtlShards + 2) * kCtlStride;            // (two spare lines)
constexpr uint32_t kDirectIds = 192;        // a step with at most this many matches stages its IDs in LDS (a fuller one stores 64 rows at a time)
constexpr uint32_t kBlockIds = 448;         // 1-byte columns: FOUR steps with at most this many matches between them are ranked as one block
constexpr uint32_t kStageRing = 512;        // >= max(kDirectIds, kBlockIds) + 63
constexpr uint32_t kSoloIds = 256;          // a trailing group with at most this many matches is expanded by its leader wave alone
constexpr uint32_t kRecoverSpins = 1u << 26; // "the long wait" (recovery pass, gather leaders): bounded by WALL CLOCK, see kRecoverTicks
// A wait that the grid's layout guarantees to end (what is waited for are scan tiles, which wait for nothing) still
// gets a deadline: 250 ms of the 100 MHz wall clock -- far beyond any launch, far below what a watchdog calls a hang.
// When it runs out the sticky status word is set; the host reports it once and resets the hand-off words.
constexpr uint64_t kRecoverTicks = 25000000ull;
constexpr uint32_t kCountMask = 0x7FFu;     // matches of a step: 0 .. 1024
// The status words name the LAUNCH that gave up (its epoch, in the word epoch % kStatusWords): two queries of a query
// stream can run on one lane context, and whoever awaits one of them must not be told about the other's failure.
constexpr uint32_t kStatusWords = 64;
constexpr int kRplShift = 11, kEpochShift = 16, kWordEpochShift = 48;
constexpr uint64_t kWordMask = (1ull << kWordEpochShift) - 1ull;

__device__ __forceinline__ void report_gave_up(CArgs &a) { st_sc1(a.status + (a.epoch & (kStatusWords - 1u)), a.epoch); }

struct alignas(16) FusedShared {
    uint16_t mask[kWaves][kGroupSteps / kWaves][64];   // expander waves: match words of 16 steps at a time
    uint32_t stage[kWaves][kStageRing];                // expander waves: row IDs on their way out, 64 to a store
    uint32_t counts[kGroupSteps];                       // trailing expander: step counts of the group, from its leader wave
    uint64_t group_off;                                 //   ... and the group's first output slot
    uint32_t state;                                     //   ... 1 = expand now, 0 = deferred
    uint32_t tile_cnt[16];                              // scan: step counts of the tile
    uint64_t tile_tiny[16];                             //   ... and the tiny words of its steps with at most kTinyIds matches
};

// What this launch covers: a scan of n_rows rows, or (gather) the device-side candidate range clamped
// to the caller's bound.
struct Extent { uint64_t begin, n_rows, steps, groups; };

template <bool GATHER>
__device__ __forceinline__ Extent scan_extent(CArgs &a) {
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

// Grid layout (TPG tiles per group).  An expander is a WAVE: what it mostly does is wait for loads, and every
// wave slot it holds meanwhile is one the scan cannot fill (measured: four waves per group among the tiles
// slow the scan by 19 %).  So among the scan tiles an expander workgroup takes a QUAD of four groups, one per
// wave, placed behind the tiles of the quad `lag` groups further on; its integer work hides under the scan.
// The last `lag` groups have no scan to hide under: their expander workgroups follow the last tile, one per
// group, each wave a quarter of the group's steps.  Placement is for speed only -- an expander checks what it
// needs and waits (bounded) if it is early.
enum { ROLE_NONE = 0, ROLE_SCAN = 1, ROLE_EXPAND_QUAD = 2, ROLE_EXPAND_GROUP = 3 };
struct Role { uint32_t kind; uint32_t index; };

__device__ __forceinline__ uint32_t uniform_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// Groups expanded by the trailing workgroups: at least the requested lag, and such that what is left in front is
// whole quads.
__host__ __device__ inline uint32_t trailing_groups(uint32_t groups, uint32_t lag_req) {
    const uint32_t quads = groups > lag_req ? (groups - lag_req) / 4u : 0u;
    return groups - 4u * quads;
}

// All in 32 bits (the host keeps a launch under 2^31 workgroups) and pinned to SGPRs: the role and everything
// derived from it is wave-uniform, and a 64-bit division would be done -- and then kept -- in vector registers.
template <int TPG>
__device__ __forceinline__ Role fused_role(CArgs &a, uint32_t groups) {
    constexpr uint32_t quad_tiles = 4u * TPG, period = quad_tiles + 1u;
    const uint32_t b = blockIdx.x;
    const uint32_t tile_quads = (groups + 3u) / 4u;
    const uint32_t main_blocks = tile_quads * period;
    const uint32_t lag = trailing_groups(groups, a.lag);               // groups with a trailing expander workgroup
    const uint32_t quads = (groups - lag) / 4u, lag_quads = tile_quads - quads;
    Role r;
    r.kind = ROLE_NONE;
    r.index = 0;
    if (b < main_blocks) {
        const uint32_t q = b / period, rr = b % period;
        if (rr < quad_tiles) { r.kind = ROLE_SCAN; r.index = q * quad_tiles + rr; }
        else if (q >= lag_quads) { r.kind = ROLE_EXPAND_QUAD; r.index = q - lag_quads; }   // < quads by construction
    } else if (b - main_blocks < lag) {
        r.kind = ROLE_EXPAND_GROUP;
        r.index = groups - lag + (b - main_blocks);
    }
    r.kind = uniform_u32(r.kind);
    r.index = uniform_u32(r.index);
    return r;
}

// The two counters of a query (ctl) are the only words that need zeroing: the query does it for the NEXT
// one, which uses the other half of a ping-pong pair (plain stores: the kernel boundary publishes them).
__device__ __forceinline__ void zero_other_ctl(CArgs &a) {
    if (blockIdx.x == 0 && threadIdx.x < kCtlShards + 2) a.zctl[threadIdx.x * kCtlStride] = 0u;
}

// A leader that has left its wait says so (and a group it gave up on is on record by then: settle_group drains).  No
// return value: the first form drew tickets -- a returning atomic per leader, a second one for the last of each shard, then
// a load, one after the other at the very end of the launch: 1 - 1.5 us on every ID query for a pass that never runs.
// (low half of a shard's word: leaders past their wait; high half: those of them that gave their group up -- one
// word, so that whoever sees all leaders through also sees what they left behind)
__device__ __forceinline__ void leader_past_its_wait(CArgs &a, uint64_t g, uint32_t lane, bool gave_up) {
    if (lane == 0) (void)__hip_atomic_fetch_add(a.ctl + (uint32_t)(g % kCtlShards) * kCtlStride, gave_up ? 0x10001u : 1u, PQPS_AGENT);
}

// The last group's leader, at the end of its work: waits (bounded: they wait for scan tiles, which wait for nothing) until
// every leader of the launch is past its wait, and tells whether any group was given up on.  One round of loads when
// the others are through already, which is the rule: they started before this one.
__device__ __forceinline__ bool expanders_past_their_wait(CArgs &a, uint64_t groups, uint32_t lane) {
    const uint64_t deadline = wall_clock64() + kRecoverTicks;
    for (;;) {
        const uint32_t c = lane < kCtlShards ? ld_sc1(a.ctl + lane * kCtlStride) : 0u;
        if ((uint64_t)wave_sum_u32(c & 0xFFFFu) >= groups) return wave_sum_u32(c >> 16) != 0u;     // (a shard has at most 1024 groups)
        if (wall_clock64() > deadline) { if (lane == 0) report_gave_up(a); return false; }
        __builtin_amdgcn_s_sleep(16);
    }
}

// End of a scan tile of TS steps.  Every wave has left the counts of its steps in sh.tile_cnt, drained its
// match-word stores and passed the workgroup's barrier; wave 0 publishes the count words with the query's
// epoch: an expander that finds all count words of its group tagged with it knows the group's match words are
// in memory.  One store instruction, no atomic, nothing to wait for.
template <int TS, bool TINY = true>
__device__ __forceinline__ void publish_tile(CArgs &a, const FusedShared &sh, const Extent &ex, uint64_t tile, uint32_t lane) {
    const uint64_t first = tile * TS;
    const uint32_t steps_in_tile = ex.steps - first < (uint64_t)TS ? (uint32_t)(ex.steps - first) : (uint32_t)TS;
    if (tile == 0 && a.accumulate) {                            // gather: results are appended behind *out_count
        if (lane == 0) st_sc1(a.base_slot, *a.out_count);
        drain_stores();                                         // in memory before anything an expander waits for
    }
    if (lane < steps_in_tile) {
        // (the tiny word carries the epoch itself: whoever finds a count word of 1 - 3 matches looks at the tag of the step's tiny word too)
        if constexpr (TINY) { if (sh.tile_tiny[lane]) st_sc1(a.tiny + first + lane, sh.tile_tiny[lane]); }
        st_sc1(a.counts + first + lane, sh.tile_cnt[lane] | (a.epoch << kEpochShift));
    }
}

// Sum duty of a scan tile.  The expander of a group needs the sums of the groups in front of it; a sum that only
// the group's own expander published would reach its neighbours a poll round too late (they start within
// nanoseconds of each other), and every round an expander waits is a wave slot the scan cannot use.  So the
// FIRST tile of group q also sums up group q - sum_lag, whose count words have long been written: one extra
// 256-byte load issued next to the tile's column loads, consumed after the tile's own work.  Pure hint: if
// those count words are not all there yet, the group's expander publishes the sum itself.
// The SECOND tile of a group does the same one level up, for the supergroup that ended sum_lag groups ago -- from
// its 4096 count words (16 KB read by one wave, once per 1024 tiles), not from its 64 group sums, so that the two
// duties do not wait for each other.
struct SumDuty { uint32_t c; uint32_t on; uint32_t g; };           // on: 1 = group sum, 2 = supergroup sum (on, g: wave-uniform)

template <int TPG>
__device__ __forceinline__ SumDuty sum_duty_load(CArgs &a, uint32_t tile, uint32_t wv, uint32_t lane) {
    SumDuty d;
    d.c = 0; d.on = 0; d.g = 0;
    if (wv != 0) return d;                                          // uniform
    const uint32_t q = tile / TPG, t_in = tile % TPG;
    if (t_in == 0 && q >= a.sum_lag) {
        d.g = q - a.sum_lag;
        d.on = 1;
        d.c = ld_sc1(a.counts + (uint64_t)d.g * kGroupSteps + lane);
    } else if (t_in == 1 && q >= a.sum_lag && (q - a.sum_lag) % kSuperGroups == kSuperGroups - 1) {
        d.g = (q - a.sum_lag) / kSuperGroups;                       // the supergroup whose last group is q - sum_lag
        d.on = 2;
    }
    d.g = uniform_u32(d.g);                                         // (kept in scalar registers through the tile's work)
    d.on = uniform_u32(d.on);
    return d;
}

__device__ __forceinline__ void sum_duty_finish(CArgs &a, const SumDuty &d, uint32_t lane) {
    const uint64_t tag = (uint64_t)a.epoch << kWordEpochShift;
    if (d.on == 1) {
        if (__all((d.c >> kEpochShift) == a.epoch)) {
            const uint32_t sum = wave_sum_u32(d.c & kCountMask);
            if (lane == 0) st_sc1(a.gsum + d.g, tag | (uint64_t)sum);
        }
    } else if (d.on == 2) {
        const uint64_t *pairs = (const uint64_t *)(a.counts + (uint64_t)d.g * kSuperGroups * kGroupSteps);     // 2048 pairs of count words
        const uint64_t both = ((uint64_t)a.epoch << 48) | ((uint64_t)a.epoch << kEpochShift);
        bool ok = true;
        uint32_t sum = 0;
#pragma unroll 1
        for (int r = 0; r < 4; r++) {                               // 8 loads in flight at a time
            uint64_t w[8];
#pragma unroll
            for (int i = 0; i < 8; i++) w[i] = ld_sc1(pairs + (r * 8 + i) * 64 + lane);
#pragma unroll
            for (int i = 0; i < 8; i++) {
                ok = ok && (w[i] & 0xFFFF0000FFFF0000ull) == both;
                sum += (uint32_t)(w[i] & kCountMask) + (uint32_t)((w[i] >> 32) & kCountMask);
            }
        }
        if (__all(ok)) {
            const uint64_t total = wave_sum_u64((uint64_t)sum);
            if (lane == 0) st_sc1(a.ssum + d.g, tag | total);
        }
    }
}

// A step with at most a.list_max matches (host: kListIds) leaves them as a LIST in its slot: 16-bit row numbers inside the
// step's GROUP (step-in-group << 10 | row-in-step), ascending, 2 bytes per match -- only the 8-byte words that hold
// entries are stored.  Turning bits into ranked row numbers is the expensive part of the expansion, and the expanders
// have nothing to hide it under, whereas a scan tile's vector units idle while it waits for memory.  What is left for
// the expander is a copy with NO per-step decoding and no LDS: lane i loads entry i of the step straight from the slot
// (one 2-byte load per lane), adds the group's first row and stores at the step's output offset (expand_direct).
// (The first form packed 10 bits per row into 128 bytes: one or two LDS atomics per match in the tile, and in the expander a
// slot fetch into LDS, two LDS reads, a 64-bit shift and ~100 instructions of per-step control -- the expanders were
// bound by their instructions, 2.2 - 2.6 TB/s of slot + ID traffic at 1 G rows.)  Denser steps keep a 16-bit list in the
// list area (store_list16) or the bit mask; which form a step took follows from its count word (step_form).
//
// Row order: bit p of lane l is row (p / RPL) * 64 * RPL + l * RPL + p % RPL, i.e. chunk by chunk (16 / RPL chunks),
// inside a chunk lane by lane.  The per-lane counts of the <= 4 chunks travel through ONE wave scan as 8-bit fields
// (no field exceeds 128), the chunks' bases come from the last lane's fields.
constexpr uint32_t kListIds = 2 * kSlotWords; // entries of a step's two 128-byte slots
// ... of which the host asks for 104 (a.list_max): an answer of 12 - 14 % (around 128 matches per step) would otherwise leave its
// steps in both forms, and a copied step in the middle of ranked ones interrupts their run of staged IDs (`risk_level > 2`,
// 138 per step: 947 us at 1 G rows with 128, 916 with 104 or with bit masks only; Q_B, 69 per step: 932 either way)
constexpr uint32_t kListDefault = 104;
enum { FORM_NONE = 0, FORM_DIRECT = 1, FORM_LIST16 = 2, FORM_MASK = 3, FORM_TINY = 4 };
// A step with at most kTinyIds matches -- what a SPARSE answer consists of: S1 has 0.07 matches per step -- leaves no slot at
// all: its entries ride in a 64-bit word of their own beside the count word (epoch << 48 | e2 << 32 | e1 << 16 | e0), stored by
// the same lane of wave 0 in the same breath.  The tile has nothing to drain (no write-through payload before the barrier), and the
// group's leader, which reads the count words anyway, has the IDs of such steps with the same round of loads: no slot fetch
// between "settled" and the last ID (stamps, S1 at 100 M rows: 2.4 us of the 8 us from the last tile to the end of the launch).
constexpr uint32_t kTinyIds = 3;

// ranks the step's matches into `stage` (LDS, the calling wave's): entry i = the i-th matching row of the step, as its 16-bit number
// inside the step's group
__device__ __forceinline__ void rank_to_stage(uint16_t *stage, uint64_t step, uint32_t mbits, uint32_t rl, uint32_t lane) {
    const uint32_t rpl = 1u << rl, chunks = 16u >> rl;              // rl = 2, 3, 4: 4, 2, 1 chunks
    uint32_t per = 0;
#pragma unroll
    for (uint32_t c = 0; c < 4; c++)
        if (c < chunks) per |= (uint32_t)__popc((mbits >> (c * rpl)) & ((1u << rpl) - 1u)) << (8 * c);
    const uint32_t incl = wave_incl_scan_u32(per);
    const uint32_t tot = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const uint32_t base = (tot << 8) + (tot << 16) + (tot << 24);   // field c: matches of the chunks before c
    uint32_t at = base + incl - per;                                // field c: list slot of this lane's first match in chunk c
    uint32_t m = mbits;
    const uint32_t lane_rows = (((uint32_t)step & (kGroupSteps - 1u)) << 10) + (lane << rl);
    while (m) {                                                     // set bits only: ascending rows inside a chunk
        const uint32_t p = (uint32_t)__builtin_ctz(m);
        m &= m - 1;
        const uint32_t c = p >> rl, sh8 = 8u * c;
        stage[(at >> sh8) & 0xFFu] = (uint16_t)((c << (6u + rl)) + lane_rows + (p & (rpl - 1u)));
        at += 1u << sh8;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the same wave wrote the stage)
}

__device__ __forceinline__ void store_list(CArgs &a, uint16_t *stage, uint64_t step, uint32_t mbits, uint32_t cnt, uint32_t rl, uint32_t lane) {
    rank_to_stage(stage, step, mbits, rl, lane);
    // whole 128-byte lines (entries past the count: whatever the stage held): a partly written line is a read-modify-write
    // on the memory side.  Lanes 0 .. 15: entries 0 .. 63, lanes 16 .. 31: the second line of a step with more than 64.
    if (lane < (cnt > 64u ? 32u : 16u))
        st_sc1((uint64_t *)((lane < 16u ? a.slots : a.slots_hi) + step * kSlotWords) + (lane & 15u), ((const uint64_t *)stage)[lane]);
}

// A step with more matches than its slot takes (a.list_max, and more than a.list16_min) leaves them as a list of another kind when
// the launch has a list area (a.lists): 16-bit row numbers inside the step, ascending, at lists[step * 1024 ...] -- 2 bytes per match
// instead of a 128-byte bit mask.  Turning a dense bit mask into IDs costs the expander ~400 vector instructions per step (64 rows at a time:
// read the bits of the 64 rows into a scalar pair, rank, store) and the expanders behind the last tile have nothing
// to hide them under: `risk_level > 1` (43 % of 100 M rows) took 225 us, 160 of them after the last tile.  The
// scan tile's vector units idle while it waits for memory; it ranks its own matches (two wave scans for the chunks'
// per-lane counts, one LDS store per match) and the expander is left with a copy: four list entries + first row of
// the step -> four IDs per lane and store.
// (`step_form` must agree between the tile that writes and the expander that reads: both see the count word.)
__device__ __forceinline__ uint32_t step_form(CArgs &a, uint32_t cnt, uint32_t rpl_log2) {
    if (cnt == 0u) return FORM_NONE;
    if (rpl_log2 < 4u) {                                            // (not for 1-byte predicates -- a step is 1 KB of input there: their kernels are spared the code)
        if (cnt <= a.tiny_max) return FORM_TINY;
        if (cnt <= a.list_max) return FORM_DIRECT;
    }
#ifndef PQPS_NO_LIST16   /* experiments: what the kernels cost without the list code in them */
    if (a.lists != nullptr && cnt > (rpl_log2 >= 4u ? a.list16_min_u8 : a.list16_min)) return FORM_LIST16;
#endif
    return FORM_MASK;
}
__device__ __forceinline__ uint32_t word_form(CArgs &a, uint32_t cw) { return step_form(a, cw & kCountMask, (cw >> kRplShift) & 7u); }
// the steps of a group that have a 128-byte bit mask to fetch (`cw` = a step's count word)
__device__ __forceinline__ bool step_has_slot(CArgs &a, uint32_t cw) { return word_form(a, cw) == FORM_MASK; }

template <int RL>
__device__ __forceinline__ void store_list16(CArgs &a, uint16_t *stage, uint64_t step, uint32_t mbits, uint32_t cnt, uint32_t lane) {
    constexpr uint32_t RPL = 1u << RL, CH = 16u >> RL;              // RL = 2, 3, 4: 4, 2, 1 chunks of 64 * RPL rows
    static_assert(RL >= 2 && RL <= 4, "widest predicate column: 8 / 4, 2 or 1 bytes");
    // rank of a lane's first match in each chunk: chunk by chunk, inside a chunk lane by lane (= ascending rows);
    // two chunks share a wave scan as 16-bit fields (a chunk has at most 512 matches)
    uint32_t at[CH];
    if constexpr (CH == 1) {
        const uint32_t per = (uint32_t)__popc(mbits);
        at[0] = wave_incl_scan_u32(per) - per;
    } else {
        const uint32_t nib = (1u << RPL) - 1u;
        const uint32_t per01 = (uint32_t)__popc(mbits & nib) | ((uint32_t)__popc((mbits >> RPL) & nib) << 16);
        const uint32_t incl01 = wave_incl_scan_u32(per01);
        const uint32_t tot01 = (uint32_t)__builtin_amdgcn_readlane((int)incl01, 63);
        const uint32_t ex01 = incl01 - per01;
        at[0] = ex01 & 0xFFFFu;
        at[1] = (tot01 & 0xFFFFu) + (ex01 >> 16);
        if constexpr (CH == 4) {
            const uint32_t per23 = (uint32_t)__popc((mbits >> (2 * RPL)) & nib) | ((uint32_t)__popc((mbits >> (3 * RPL)) & nib) << 16);
            const uint32_t incl23 = wave_incl_scan_u32(per23);
            const uint32_t tot23 = (uint32_t)__builtin_amdgcn_readlane((int)incl23, 63);
            const uint32_t ex23 = incl23 - per23, front = (tot01 & 0xFFFFu) + (tot01 >> 16);
            at[2] = front + (ex23 & 0xFFFFu);
            at[3] = front + (tot23 & 0xFFFFu) + (ex23 >> 16);
        }
    }
    const uint32_t lane_rows = lane << RL;
#pragma unroll
    for (uint32_t c = 0; c < CH; c++) {                             // trip count of each loop: the fullest lane's matches in the chunk (<= RPL)
        uint32_t m = (mbits >> (c * RPL)) & ((1u << RPL) - 1u);
        uint16_t *out = stage + at[c];
        const uint32_t row0 = (c << (6u + RL)) + lane_rows;
        while (m) {
            const uint32_t j = (uint32_t)__builtin_ctz(m);
            m &= m - 1;
            *out++ = (uint16_t)(row0 + j);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the same wave wrote the stage)
    uint64_t *dst = (uint64_t *)(a.lists + step * kStepRows);
#pragma unroll
    for (uint32_t r = 0; r < 4; r++)                                // 256 entries per store instruction
        if (r * 256u < cnt) {                                       // uniform
            if (r * 256u + lane * 4u < cnt) st_sc1(dst + r * 64u + lane, ((const uint64_t *)stage)[r * 64u + lane]);
        }
}

// Start of a scan tile: no step of it has a tiny word yet (only the steps with 1 - kTinyIds matches write theirs; a wave's
// LDS accesses keep their order, and a tile's waves meet at its barrier before wave 0 reads the words).
__device__ __forceinline__ void zero_tiny_words(FusedShared &sh, uint32_t first_slot, uint32_t n_slots) {
    if ((threadIdx.x & 63u) < n_slots) sh.tile_tiny[first_slot + (threadIdx.x & 63u)] = 0;
}

// One wave's share of a scan tile: count word into LDS, match words (if any) to memory.
__device__ __forceinline__ void tile_step_out(CArgs &a, FusedShared &sh, uint32_t slot, uint64_t step, uint32_t cnt,
                                              uint32_t mbits, uint32_t rpl_log2, uint32_t lane) {
    const uint32_t form = step_form(a, cnt, rpl_log2);              // uniform
    uint64_t tiny = 0;
    if (form == FORM_TINY) {
        uint16_t *stage = &sh.mask[threadIdx.x >> 6][0][0];
        rank_to_stage(stage, step, mbits, rpl_log2, lane);
        tiny = ((uint64_t)a.epoch << kWordEpochShift) | ((uint64_t)stage[0] | ((uint64_t)stage[1] << 16) | ((uint64_t)stage[2] << 32));   // (entries past the count: whatever the stage held)
        if (lane == 0) sh.tile_tiny[slot] = tiny;               // (the other steps' words stay zero: zero_tiny_words)
    }
    if (form == FORM_DIRECT) store_list(a, &sh.mask[threadIdx.x >> 6][0][0], step, mbits, cnt, rpl_log2, lane);
    else if (form == FORM_LIST16) {
        uint16_t *stage = (uint16_t *)sh.stage[threadIdx.x >> 6];                  // 2 KB per wave: 1024 entries
        if (rpl_log2 == 2u) store_list16<2>(a, stage, step, mbits, cnt, lane);      // uniform
        else if (rpl_log2 == 3u) store_list16<3>(a, stage, step, mbits, cnt, lane);
        else store_list16<4>(a, stage, step, mbits, cnt, lane);
    } else if (form == FORM_MASK) store_mask(a, step, mbits, lane);
    if (lane == 0) sh.tile_cnt[slot] = cnt | (rpl_log2 << kRplShift);
}

// ---- expand: match words -> ascending row IDs ------------------------------------------------------------
// One step: the scan left 16 match bits per lane in its load layout (bit p of lane l <-> row
// (p / RPL) * 64 * RPL + l * RPL + p % RPL).  First bring them into ROW order -- lane d gets the bits of rows
// 16d .. 16d+15, which sit in 16/RPL source lanes.  Then
//   few matches:  one wave scan of the per-lane popcounts gives every lane its rank; the IDs go to a ring in
//                 LDS and leave it 64 at a time, one full store instruction per 256 bytes of output -- among
//                 the scan tiles a CU's memory pipeline is full of their loads, and what an expander pays for
//                 is every instruction it puts into that queue, not the bytes;
//   many matches: 64 rows at a time -- their match bits are the words of 4 lanes, read into an SGPR pair with
//                 v_readlane; rank inside the 64 = mbcnt, so the 64 lanes store to consecutive slots.
// The IDs of consecutive steps are consecutive in the output, so the ring carries over from step to step.
// Row IDs that leave in runs of whole store instructions -- the copies of expand_direct, the 64 staged IDs of ring_flush --
// go out with streaming (`nt`) stores: nothing on the device reads them again, and as write-back lines an answer of a few
// percent displaces what the scan and the hand-off keep in L2 (same-process A/B, the copies only: Q_A at 1 G rows 724 -> 697
// us, Q_B 960 -> 921; at 100 M rows Q_B 98.0 -> 94.2; with ring_flush a lone u8 column 45.1 -> 42.9).  NOT the stores of
// the dense paths (64 rows at a time, lists of the list area): their instructions fill lines in parts that only L2 puts
// together -- streaming, `risk_level > 1` (43 %) at 1 G rows took 1482 us instead of 1267.
#define st_id(p, v) __builtin_nontemporal_store((v), (p))      /* (a macro: the pointee's `aligned(4)` must reach the builtin) */

struct OutRing {
    uint32_t head, pending;          // wave-uniform: ring position of the oldest staged ID, staged IDs
    uint64_t pos;                    // output slot of the oldest staged ID (= of the next ID when nothing is staged)
};

__device__ __forceinline__ void ring_flush(CArgs &a, uint32_t *ring, OutRing &r, uint32_t lane, uint32_t n) {
    // n <= 64 staged IDs leave in one store instruction
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the same wave wrote the ring
    const uint32_t v = ring[(r.head + lane) & (kStageRing - 1)];
    if (lane < n && r.pos + lane < a.out_cap) st_id(a.out_ids + r.pos + lane, v);
    r.head = (r.head + n) & (kStageRing - 1);
    r.pending -= n;
    r.pos += n;
}

// Room for `need` more IDs behind the staged ones WITHOUT wrapping round the ring's end (need + 63 <= kStageRing):
// if they would not fit, the < 64 staged IDs move to the front.  The rank loops can then walk a plain pointer --
// one add per ID instead of mask + shift + add, in loops whose trip count is the fullest lane's.
__device__ __forceinline__ void ring_reserve(uint32_t *ring, OutRing &r, uint32_t lane, uint32_t need) {
    if (r.head + r.pending + need <= kStageRing) return;            // uniform
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // the same wave wrote the ring
    const uint32_t v = ring[(r.head + lane) & (kStageRing - 1)];
    if (lane < r.pending) ring[lane] = v;                           // (one wave: its LDS reads and writes keep their order)
    r.head = 0;
}

// The next ID goes to output slot `want`: steps that were copied elsewhere (16-bit entries, lists) interrupt the run of
// the steps around them -- what is staged leaves first.
__device__ __forceinline__ void ring_seek(CArgs &a, uint32_t *ring, OutRing &r, uint32_t lane, uint64_t want) {
    if (r.pos + r.pending == want) return;                          // uniform
    if (r.pending) ring_flush(a, ring, r, lane, r.pending);         // (< 64 staged IDs between steps)
    r.pos = want;
}

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

// Row-ordered match word of a step (lane d: rows 16 d .. 16 d + 15) from the scan's load layout.
__device__ __forceinline__ uint32_t step_row_word(uint32_t m16, uint32_t rpl_log2, uint32_t lane) {
    switch (rpl_log2) {                                             // uniform
    case 2: return row_order_word<2>(m16, lane);
    case 3: return row_order_word<3>(m16, lane);
    default: return m16;
    }
}

// The match bits of rows 64 s .. 64 s + 63 of a step are the words of lanes 4 s .. 4 s + 3: wave-uniform lane
// numbers, so v_readlane (SGPR result), no LDS crossbar.
__device__ __forceinline__ uint64_t sub_block_bits(uint32_t word, uint32_t s) {
    const uint32_t w0 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s));
    const uint32_t w1 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 1));
    const uint32_t w2 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 2));
    const uint32_t w3 = (uint32_t)__builtin_amdgcn_readlane((int)word, (int)(4 * s + 3));
    return (uint64_t)(w0 | (w1 << 16)) | ((uint64_t)(w2 | (w3 << 16)) << 32);
}

// Gather (index mode): the candidate numbers of a dense step's matching rows, all 16 sub-blocks requested at once
// (lane L of sub-block s <-> row 64 s + L; a set bit implies the row lies inside the probed range): one memory
// latency per step instead of one per sub-block.
__device__ __forceinline__ void gather_candidates(CArgs &a, uint64_t begin, uint64_t step, uint32_t word, uint32_t lane, uint32_t (&cnd)[16]) {
    const uint32_t step_row0 = (uint32_t)(step * kStepRows);
#pragma unroll
    for (uint32_t s = 0; s < 16; s++) {
        // every lane loads (a lane without a match: the range's first candidate, always there) -- exactly 16 load
        // instructions per step, so the waits further down can be counted instead of being "for everything"
        const uint64_t b = sub_block_bits(word, s);
        const uint32_t row = ((b >> lane) & 1ull) ? step_row0 + 64u * s + lane : 0u;
        cnd[s] = a.cand[begin + row];
    }
}

// A step with many matches: 64 rows at a time, rank inside the 64 = mbcnt, so the lanes store to consecutive slots.
template <bool GATHER>
__device__ __forceinline__ void expand_step_dense(CArgs &a, uint64_t step, uint32_t word, uint32_t lane, uint32_t *ring, OutRing &r,
                                                  const uint32_t (&cnd)[16]) {
    const uint32_t step_row0 = (uint32_t)(step * kStepRows);
    if (r.pending) ring_flush(a, ring, r, lane, r.pending);         // < 64 staged IDs from the steps before
    auto sub_block = [&](uint32_t s, uint32_t cand_id) {
        const uint64_t b = sub_block_bits(word, s);
        if (b) {                                                    // uniform
            if ((b >> lane) & 1ull) {
                const uint64_t o = r.pos + mbcnt(b);
                const uint32_t id = GATHER ? cand_id : step_row0 + 64u * s + lane;
                if (o < a.out_cap) a.out_ids[o] = id + a.id_base;
            }
            r.pos += (uint64_t)__popcll(b);
        }
    };
    if constexpr (GATHER) {
#pragma unroll
        for (uint32_t s = 0; s < 16; s++) sub_block(s, cnd[s]);     // (the numbers sit in 16 registers)
    } else {
#pragma unroll 1
        for (uint32_t s = 0; s < 16; s++) sub_block(s, 0u);         // a loop: the scan tiles' code wants the instruction cache
    }
}

// A step with few matches: one wave scan of the per-lane popcounts gives every lane its rank; the IDs go to the ring.
template <bool GATHER>
__device__ __forceinline__ void expand_step_sparse(CArgs &a, uint64_t begin, uint64_t step, uint32_t word, uint32_t count, uint32_t lane,
                                                   uint32_t *ring, OutRing &r) {
    const uint32_t step_row0 = (uint32_t)(step * kStepRows);
    const uint32_t cnt = __popc(word);
    const uint32_t incl = wave_incl_scan_u32(cnt);
    ring_reserve(ring, r, lane, count);
    uint32_t *out = ring + (r.head + r.pending + (incl - cnt));
    const uint32_t r0 = step_row0 + lane * 16u, r0b = r0 + a.id_base;
    while (word) {                                                  // set bits only, ascending rows
        const uint32_t j = (uint32_t)__builtin_ctz(word);
        word &= word - 1;
        // gather: the candidate number of the row (a set bit implies the row lies inside the probed range)
        uint32_t id = r0b + j;
        if constexpr (GATHER) id = a.cand[begin + r0 + j] + a.id_base;
        *out++ = id;
    }
    r.pending += count;
    while (r.pending >= 64) ring_flush(a, ring, r, lane, 64);
}

// One-byte-wide predicates (RPL = 16: no row lists, and the match words are in row order as they are): FOUR
// consecutive steps at once, lane L takes rows [64 L, 64 L + 64) of the block -- 8 contiguous bytes of the parked
// words of step L / 16 -- one wave scan ranks all IDs of the block, and the rank loop's trip count is that of the
// fullest 64-row lane instead of four times that of the fullest 16-row lane.
template <bool GATHER>
__device__ __forceinline__ void expand_block16(CArgs &a, uint64_t begin, uint64_t step0, uint32_t total, uint32_t nb,
                                               const uint16_t (*park)[64], uint32_t lane, uint32_t *ring, OutRing &r) {
    const uint32_t st = lane >> 4, q = lane & 15u;
    uint64_t w = *(const uint64_t *)(park[st] + 4u * q);
    if (!((nb >> st) & 1u)) w = 0;                                  // an empty step: its slot was not filled
    const uint32_t cnt = (uint32_t)__popcll(w);
    const uint32_t incl = wave_incl_scan_u32(cnt);
    ring_reserve(ring, r, lane, total);
    uint32_t *out = ring + (r.head + r.pending + (incl - cnt));
    const uint32_t r0 = (uint32_t)(step0 * kStepRows) + lane * 64u, r0b = r0 + a.id_base;
    uint32_t half = (uint32_t)w;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        while (half) {                                              // set bits only, ascending rows
            const uint32_t j = (uint32_t)__builtin_ctz(half) + 32u * h;
            half &= half - 1;
            uint32_t id = r0b + j;
            if constexpr (GATHER) id = a.cand[begin + r0 + j] + a.id_base;
            *out++ = id;
        }
        half = (uint32_t)(w >> 32);
    }
    r.pending += total;
    while (r.pending >= 64) ring_flush(a, ring, r, lane, 64);
}

// `slot` = the step's 128 bytes as parked in LDS: a row list (count <= kListIds, see store_list) or 16 match bits per lane.
// A step whose matches were left as 16-bit row numbers (store_list16): a copy.  Lane l of round q takes entries
// 256 q + 4 l .. + 3 -- one 8-byte load, four adds, one 16-byte store (the output run of a step starts wherever the
// matches in front of it end, so the store is only 4-byte aligned: global memory accesses may be).
__device__ __forceinline__ void expand_step_list16(CArgs &a, uint64_t step, uint32_t count, uint32_t lane, uint32_t *ring, OutRing &r) {
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    if (r.pending) ring_flush(a, ring, r, lane, r.pending);         // < 64 staged IDs from the steps before
    const uint64_t *src = (const uint64_t *)(a.lists + step * kStepRows) + lane;
    const uint32_t base = (uint32_t)(step * kStepRows) + a.id_base;
    // entries that have a place in the caller's buffer (a result that does not fit is cut off, the count says so)
    const uint64_t room = r.pos < a.out_cap ? a.out_cap - r.pos : 0ull;
    const uint32_t lim = room < (uint64_t)count ? (uint32_t)room : count;
    uint64_t w[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++)                                // (every lane loads, one wait: see expand_lists16)
        w[q] = ld_sc1(q * 256u + lane * 4u < lim ? src + q * 64u : src - lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t e0 = q * 256u + lane * 4u;
        if (q * 256u < lim) {                                       // uniform
            const uint32_t lo = (uint32_t)w[q], hi = (uint32_t)(w[q] >> 32);
            const uint32_t i0 = base + (lo & 0xFFFFu), i1 = base + (lo >> 16), i2 = base + (hi & 0xFFFFu), i3 = base + (hi >> 16);
            uint32_t *o = a.out_ids + r.pos + e0;
            if (e0 + 4u <= lim) {
                u32x4_a4 v;
                v.x = i0; v.y = i1; v.z = i2; v.w = i3;
                *(u32x4_a4 *)o = v;
            } else if (e0 < lim) {                                  // the list's last lane: 1 - 3 entries
                o[0] = i0;
                if (e0 + 1u < lim) o[1] = i1;
                if (e0 + 2u < lim) o[2] = i2;
            }
        }
    }
    r.pos += count;
}

// The same for a BATCH of NS consecutive steps, NR rounds of 256 entries at a time (`which`: bit i = step0 + i left a
// 16-bit list; lane l of `cw` / `my_off` holds count word and first output slot of step l of the group, step0 is step
// `first` of it).  One step at a time the copy is a chain of memory latencies -- 138 entries are one 276-byte load per
// wave, and 16 steps per wave one after the other made the expansion of a 13 % answer take 23 us per group at 3.5 TB/s of
// copy traffic.  Here NS * NR = 8 loads are requested before any is awaited: eight steps of up to 256 matches each (what a
// few-percent answer looks like), or four steps with their first 512; fuller steps come round again.
template <int NS, int NR>
__device__ __forceinline__ void expand_lists16(CArgs &a, uint64_t step0, uint32_t which, uint32_t cw, uint64_t my_off, uint32_t first, uint32_t lane) {
    typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
    uint32_t lim[NS];
    uint64_t off[NS];
    uint32_t most = 0;
#pragma unroll
    for (uint32_t i = 0; i < NS; i++) {
        const uint32_t cnt = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(first + i)) & kCountMask;
        off[i] = readlane_u64(my_off, (int)(first + i));
        // entries that have a place in the caller's buffer (a result that does not fit is cut off, the count says so)
        const uint64_t room = off[i] < a.out_cap ? a.out_cap - off[i] : 0ull;
        lim[i] = !((which >> i) & 1u) ? 0u : (room < (uint64_t)cnt ? (uint32_t)room : cnt);
        most = lim[i] > most ? lim[i] : most;
    }
    auto put = [&](uint32_t i, uint32_t q, uint64_t w) {
        const uint32_t e0 = q * 256u + lane * 4u;
        const uint32_t base = (uint32_t)((step0 + i) * kStepRows) + a.id_base;
        const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
        const uint32_t i0 = base + (lo & 0xFFFFu), i1 = base + (lo >> 16), i2 = base + (hi & 0xFFFFu), i3 = base + (hi >> 16);
        uint32_t *o = a.out_ids + off[i] + e0;
        if (e0 + 4u <= lim[i]) {                                    // (the run of a step starts wherever the matches in front of it end:
            u32x4_a4 v;                                             //  4-byte aligned, which global memory accesses may be)
            v.x = i0; v.y = i1; v.z = i2; v.w = i3;
            *(u32x4_a4 *)o = v;
        } else if (e0 < lim[i]) {                                   // the list's last lane: 1 - 3 entries
            o[0] = i0;
            if (e0 + 1u < lim[i]) o[1] = i1;
            if (e0 + 2u < lim[i]) o[2] = i2;
        }
    };
#pragma unroll 1
    for (uint32_t q0 = 0; q0 * 256u < most; q0 += NR) {             // uniform
        uint64_t w[NS][NR];
#pragma unroll
        for (uint32_t i = 0; i < NS; i++)
#pragma unroll
            for (uint32_t q = 0; q < NR; q++) {
                // every lane loads (a lane past the list's end: the list's first entries) -- no branch around a load, and
                // ONE explicit wait for all of them below: gfx9's vmcnt counts loads and stores in one in-order queue, and
                // left to itself the compiler put a vmcnt(0) in front of every step's stores (it cannot count loads behind
                // branches), i.e. every step waited for the acknowledgement of the step's stores before it: eight memory
                // round trips in a row, 10 us for the 16 steps of a wave (Q_A at 100 M rows: 14 us from settled to done)
                const uint32_t e0 = (q0 + q) * 256u + lane * 4u;
                w[i][q] = ld_sc1((const uint64_t *)(a.lists + (step0 + i) * kStepRows) + (e0 < lim[i] ? (q0 + q) * 64u + lane : 0u));
            }
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        if (first == 0 && q0 == 0) PQPS_STAMP_GROUP(a, step0 / kGroupSteps, 7);
#pragma unroll
        for (uint32_t i = 0; i < NS; i++)
#pragma unroll
            for (uint32_t q = 0; q < NR; q++)
                if ((q0 + q) * 256u < lim[i]) put(i, q0 + q, w[i][q]);       // uniform
    }
}

// Steps whose matches were left as 16-bit entries in their slots (FORM_DIRECT): a copy without LDS and without
// per-step decoding.  `rest` = those steps of group g (bit = step in group) still to do; the first NS of them are taken:
// NS four-byte loads per lane go out before any is awaited (lane i: entries 2 i and 2 i + 1 of the step -- a slot's 128
// entries are one load instruction), then per step two adds and one 8-byte store per lane at the step's own output
// offset (4-byte aligned, which global memory accesses may be; lane l of `lim_v` / `off_v` holds, for step l of the
// group, the entries that have a place in the caller's buffer and the output slot of the first, counted from `obase`).
// `big`: the steps with more than 64 entries.  Returns the steps left.
typedef __attribute__((address_space(1))) char global_char;
template <bool GATHER, int NS>
__device__ __forceinline__ uint64_t expand_direct(CArgs &a, uint64_t begin, uint64_t g, uint64_t rest, uint64_t big, uint32_t lim_v, uint32_t off_v,
                                                  uint32_t *obase, uint32_t lane) {
    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    // lanes 0 .. 31: entries 0 .. 63 (first line); lanes 32 .. 63: entries 64 .. 127 from the second line if the step has
    // one (`big`), otherwise the first line once more (same 128 bytes: no second memory access)
    const uint32_t *lo = (const uint32_t *)(a.slots + g * kGroupSteps * kSlotWords) + (lane & 31u);
    const uint32_t *hi = lane < 32u ? lo : (const uint32_t *)(a.slots_hi + g * kGroupSteps * kSlotWords) + (lane & 31u);
    uint32_t st[NS];
    uint32_t v[NS];
#pragma unroll
    for (int i = 0; i < NS; i++) {
        // every lane loads (no step left: the group's first slot once more) -- no branch around a load, ONE wait for all
        st[i] = rest ? (uint32_t)__builtin_ctzll(rest) : 64u;
        rest &= rest - 1;                                           // (0 stays 0)
        const uint32_t sl = st[i] & 63u;
        v[i] = ld_sc1((((big >> sl) & 1ull) ? hi : lo) + sl * (uint32_t)(kSlotWords / 2));
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0)
    const uint32_t gbase = (uint32_t)(g * kGroupSteps * kStepRows), e0 = 2u * lane;
    const uint32_t add = GATHER ? a.id_base : gbase + a.id_base;
    global_char *ob = (global_char *)obase;
    uint32_t c0[GATHER ? NS : 1], c1[GATHER ? NS : 1];
    if constexpr (GATHER) {                                         // the candidate numbers of the listed rows, all requested before the stores
#pragma unroll
        for (int i = 0; i < NS; i++) {
            const uint32_t lim = st[i] < 64u ? (uint32_t)__builtin_amdgcn_readlane((int)lim_v, (int)(st[i] & 63u)) : 0u;
            c0[i] = a.cand[begin + (e0 < lim ? gbase + (v[i] & 0xFFFFu) : 0u)];
            c1[i] = a.cand[begin + (e0 + 1u < lim ? gbase + (v[i] >> 16) : 0u)];
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
    }
#pragma unroll
    for (int i = 0; i < NS; i++) {
        if (st[i] >= 64u) break;                                    // uniform
        const uint32_t lim = (uint32_t)__builtin_amdgcn_readlane((int)lim_v, (int)st[i]);
        const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)off_v, (int)st[i]);
        u32x2_a4 id;
        id.x = (GATHER ? c0[GATHER ? i : 0] : (v[i] & 0xFFFFu)) + add;
        id.y = (GATHER ? c1[GATHER ? i : 0] : (v[i] >> 16)) + add;
        // (a group's IDs are at most 256 KB: byte offsets in 32 bits, scalar base + vector offset)
        global_char *o = ob + ((off + e0) << 2);
        if (e0 + 1u < lim) st_id((__attribute__((address_space(1))) u32x2_a4 *)o, id);
        else if (e0 < lim) st_id((__attribute__((address_space(1))) uint32_t *)o, id.x);      // the last entry of an odd count
    }
    return rest;
}

template <bool GATHER>
__device__ __forceinline__ void expand_step(CArgs &a, uint64_t begin, uint64_t step, const uint16_t *slot, uint32_t rpl_log2,
                                            uint32_t count, uint32_t lane, uint32_t *ring, OutRing &r) {
    if constexpr (!GATHER)
        if (step_form(a, count, rpl_log2) == FORM_LIST16) { expand_step_list16(a, step, count, lane, ring, r); return; }   // uniform
    const uint32_t word = step_row_word(slot[lane], rpl_log2, lane);
    if (count <= kDirectIds) {
        expand_step_sparse<GATHER>(a, begin, step, word, count, lane, ring, r);
        return;
    }
    uint32_t cnd[16];
    if constexpr (GATHER) {
        gather_candidates(a, begin, step, word, lane, cnd);
        // All 16 numbers in before the first store goes out: gfx9's vmcnt counts loads and stores in ONE in-order
        // queue, so a wait for a load that is placed between stores also waits for the stores' acknowledgements
        // (measured: 2.8 us per step when the compiler places one wait per sub-block; requesting the NEXT step's
        // numbers early does not help either -- taking them over means a wait behind this step's stores).
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
    }
    expand_step_dense<GATHER>(a, step, word, lane, ring, r, cnd);
}

__device__ __forceinline__ bool word_valid(CArgs &a, uint64_t w) { return (uint32_t)(w >> kWordEpochShift) == a.epoch; }

// One look at what the expander of group g waits for, everything asked for in one round of loads.
//   its own group     every count word of the group carries the epoch (=> the group's match words are in
//                     memory); lane l keeps step l's word in `cw`;
//   what is in front  the ssum of the supergroups before its own -- for the NEAR ones just before it, whose words
//                     may not be out yet (always so at the end of the table, where no tile does sum duty), the 64
//                     gsums serve as well -- and the gsum of the earlier groups of its own supergroup.
// `left`: bit 0 = own group still incomplete, bit 1 = front still unknown; a part that is settled is not read
// again.  Returns the bits that are still open.
// `watch` (set when the front is still open): the LAST of the sum words found missing -- sums appear roughly in
// ascending order, so the caller can wait for that one word with one-load looks and come back for a full one
// (which costs ~250 vector instructions) when it has appeared.
template <int NEAR, bool NARROW = false>                        // NARROW: a kernel for 1-byte predicates -- no step of it has a tiny word or entries in its slot
__device__ __forceinline__ uint32_t poll_group(CArgs &a, const Extent &ex, uint64_t g, uint32_t lane, uint32_t left,
                                               uint32_t &cw, uint64_t &tw, uint64_t &psum, uint64_t &own_super, const uint64_t *&watch) {
    uint32_t now = 0;
    const uint64_t step = g * kGroupSteps + lane;
    uint32_t c = a.epoch << kEpochShift;
    uint64_t t = 0;                                                 // the step's tiny word (looked at if the count word says 1 - kTinyIds matches)
    if ((left & 1u) && step < ex.steps) {
        c = ld_sc1(a.counts + step);
        if constexpr (!NARROW) { if (a.tiny_max) t = ld_sc1(a.tiny + step); }      // (tiny_max: uniform)
    }
    if (left & 2u) {                                                // uniform
        const uint64_t sg = g / kSuperGroups, g_in = g % kSuperGroups;
        const uint64_t far = sg > (uint64_t)NEAR ? sg - NEAR : 0;   // supergroups [0, far): by their words only
        bool ok = true;
        uint64_t acc = 0, mine = 0;
        const uint64_t *missing = nullptr;                          // per lane: the last word this lane found missing
        for (uint64_t j = lane; j < far; j += 64) {
            const uint64_t w = ld_sc1(a.ssum + j);
            if (!word_valid(a, w)) { ok = false; missing = a.ssum + j; }
            acc += w & kWordMask;
        }
        uint64_t ps = 0, pg[NEAR > 0 ? NEAR : 1];                   // supergroups [far, sg): word (lane k) or groups
        if (far + lane < sg) ps = ld_sc1(a.ssum + far + lane);
#pragma unroll
        for (int k = 0; k < NEAR; k++) {
            pg[k] = 0;
            if (far + k < sg) pg[k] = ld_sc1(a.gsum + (far + k) * kSuperGroups + lane);
        }
        const uint64_t ps_ok = __ballot(far + lane < sg && word_valid(a, ps));
        if (far + lane < sg && word_valid(a, ps)) acc += ps & kWordMask;
        bool near_ok = true;
#pragma unroll
        for (int k = 0; k < NEAR; k++) {
            if (far + k < sg && !((ps_ok >> k) & 1ull)) {           // uniform
                if (__all(word_valid(a, pg[k]))) {
                    acc += pg[k] & kWordMask;
                } else {
                    near_ok = false;
                    if (!word_valid(a, pg[k])) missing = a.gsum + (far + k) * kSuperGroups + lane;
                }
            }
        }
        if (NEAR == 0 && __popcll(ps_ok) != (int)(sg - far)) {      // (the light look has no group sums to fall back on)
            near_ok = false;
            if (far + lane < sg && !word_valid(a, ps)) missing = a.ssum + far + lane;
        }
        if (lane < g_in) {                                          // earlier groups of the own supergroup
            const uint64_t w = ld_sc1(a.gsum + sg * kSuperGroups + lane);
            if (!word_valid(a, w)) { ok = false; missing = a.gsum + sg * kSuperGroups + lane; }
            mine = w & kWordMask;
        }
        if (__all(ok) && near_ok) {
            own_super = wave_sum_u64(mine);
            psum = wave_sum_u64(acc) + own_super;
        } else {
            now |= 2u;
            // one of the missing words, from the highest lane that has one (within an array that is the highest index)
            const uint64_t who = __ballot(missing != nullptr);
            if (who) watch = (const uint64_t *)(uintptr_t)readlane_u64((uint64_t)(uintptr_t)missing, 63 - (int)__builtin_clzll(who));
        }
    }
    if (left & 1u) {
        const bool there = (c >> kEpochShift) == a.epoch && (NARROW || word_form(a, c) != (uint32_t)FORM_TINY || word_valid(a, t));
        if (__all(there)) { cw = c; tw = t; } else now |= 1u;
    }
    return now;
}

// A wave of an expander workgroup that has nothing to do while its leader waits for the sums in front of the group
// asks for the match words of its OWN 16 steps [c0, c0 + 16) already -- if their count words are all there (one
// look, no waiting: a hint).  The words land in the wave's LDS slice, slot = step - c0, while the leader settles;
// expand_range then finds them there instead of paying a memory latency after the barrier (2.8 us of the 7 us a
// trailing group of Q_A took from settled to done).  Returns the steps requested (bits 0 .. 15) | 1 << 16, or 0.
__device__ __forceinline__ uint32_t prefetch_own_steps(CArgs &a, FusedShared &sh, const Extent &ex, uint64_t g, uint32_t lane,
                                                       uint32_t c0, uint32_t park) {
    typedef __attribute__((address_space(1))) const void global_cvoid;
    typedef __attribute__((address_space(3))) void lds_void;
    const uint64_t step = g * kGroupSteps + c0 + (lane & 15u);
    uint32_t c = a.epoch << kEpochShift;
    if (lane < 16 && step < ex.steps) c = ld_sc1(a.counts + step);
    if (!__all((c >> kEpochShift) == a.epoch)) return 0u;            // uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // no instruction: the payload loads stay behind the look
    const uint32_t bits = uniform_u32((uint32_t)__ballot(lane < 16 && step_has_slot(a, c)) & 0xFFFFu);
    const uint32_t *gmask = (const uint32_t *)(a.masks + (g * kGroupSteps + c0) * 64) + lane;
    for (uint32_t rest = bits; rest; rest &= rest - 1) {            // uniform
        const uint32_t k = (uint32_t)__builtin_ctz(rest);
        if (lane < 32) __builtin_amdgcn_global_load_lds((global_cvoid *)(gmask + (size_t)k * 32), (lds_void *)&sh.mask[park][k][0], 4, 0, 16 /* sc1 */);
    }
    return bits | (1u << 16);
}

// The LEADER's look ahead: it reads all 64 count words anyway.  A group with few matches (<= kSoloIds: the leader
// will expand it alone) and at most 16 non-empty steps gets those steps' slots requested in the packed order the
// sparse branch of expand_range uses (slot k = the k-th non-empty step) -> returns 2 << 16 and the steps in `mask`;
// otherwise the leader's own quarter (its 16 steps) as in prefetch_own_steps.
// (`c`: lane l holds the count word of step l of the group, all 64 known to carry this query's epoch)
__device__ __forceinline__ uint32_t leader_prefetch_with(CArgs &a, FusedShared &sh, uint64_t g, uint32_t lane, uint32_t park,
                                                         uint32_t c, uint64_t &mask) {
    typedef __attribute__((address_space(1))) const void global_cvoid;
    typedef __attribute__((address_space(3))) void lds_void;
    mask = 0;
    // steps with a 128-byte slot to fetch (a solo group may well hold steps with 16-bit lists: up to kSoloIds matches, a list
    // from list16_min + 1 on -- those have no slot).  The packed order is taken exactly when expand_range's sparse branch
    // will run: at most 16 NON-EMPTY steps, whichever form they left.
    const uint64_t nonempty = __ballot(step_has_slot(a, c));
    const uint64_t any_match = __ballot(word_form(a, c) >= (uint32_t)FORM_LIST16);   // (steps with 16-bit entries in their slots are copied without LDS)
    const uint32_t *gm = (const uint32_t *)(a.masks + g * kGroupSteps * 64) + lane;
    if (wave_sum_u32(c & kCountMask) <= kSoloIds && __popcll(any_match) <= (int)(kGroupSteps / kWaves)) {
        uint32_t k = 0;
        for (uint64_t rest = nonempty; rest; rest &= rest - 1, k++) {     // uniform
            const uint32_t st = (uint32_t)__builtin_ctzll(rest);
            if (lane < 32) __builtin_amdgcn_global_load_lds((global_cvoid *)(gm + (size_t)st * 32), (lds_void *)&sh.mask[park][k][0], 4, 0, 16 /* sc1 */);
        }
        mask = nonempty;
        return 2u << 16;
    }
    const uint32_t bits = uniform_u32((uint32_t)nonempty & 0xFFFFu);
    for (uint32_t rest = bits; rest; rest &= rest - 1) {            // uniform
        const uint32_t k = (uint32_t)__builtin_ctz(rest);
        if (lane < 32) __builtin_amdgcn_global_load_lds((global_cvoid *)(gm + (size_t)k * 32), (lds_void *)&sh.mask[park][k][0], 4, 0, 16 /* sc1 */);
    }
    return bits | (1u << 16);
}

__device__ __forceinline__ uint32_t prefetch_as_leader(CArgs &a, FusedShared &sh, const Extent &ex, uint64_t g, uint32_t lane,
                                                       uint32_t park, uint64_t &mask) {
    mask = 0;
    const uint64_t step = g * kGroupSteps + lane;
    uint32_t c = a.epoch << kEpochShift;
    if (step < ex.steps) c = ld_sc1(a.counts + step);
    if (!__all((c >> kEpochShift) == a.epoch)) return 0u;            // uniform
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return leader_prefetch_with(a, sh, g, lane, park, c, mask);
}

// What a leader that was early carries through its wait: the slots of its group are requested the moment its count words
// turn out complete (settle_group), not after the sums in front have arrived as well -- for the groups of the last
// wavefront of tiles, whose count words are never there at the first look, that is one memory latency (1 - 1.5 us) off
// the end of the launch.
struct LeaderPrefetch { FusedShared *sh; uint32_t park; uint32_t pre; uint64_t mask; };

// The calling wave turns the match words of steps [c0, c1) of group g (both multiples of 16) into row IDs
// (`park` = its LDS slice; `cw` = lane l holds the count word of step l; `group_off` = the group's first output slot).
// `pre` = what prefetch_own_steps returned for [c0, c0 + 16) (0: nothing is on its way).
template <bool GATHER, bool NARROW = false>
__device__ __forceinline__ void expand_range(CArgs &a, FusedShared &sh, const Extent &ex, uint64_t g, uint32_t lane, uint32_t c0,
                                             uint32_t c1, uint32_t park, uint32_t cw, uint64_t group_off, uint32_t pre = 0,
                                             uint64_t pre_mask = 0, uint64_t tiny_span = 0, uint64_t tw = 0) {
    const uint32_t my_cnt = cw & kCountMask;
    const uint32_t incl = wave_incl_scan_u32(my_cnt);
    const uint64_t my_off = group_off + (incl - my_cnt);
    if (g + 1 == ex.groups && c0 == 0 && lane == 63) *a.out_count = group_off + incl;
    const uint64_t span = (c1 >= 64 ? ~0ull : ((1ull << c1) - 1ull)) & ~((1ull << c0) - 1ull);
    const uint32_t form = word_form(a, cw);
    // Steps with 1 - kTinyIds matches: their entries came with the count words (`tw`, lane l = step l): lane l stores its
    // step's IDs itself -- three store instructions for the whole group, no load.  The steps of `tiny_span` are this wave's to do.
    const uint64_t tiny = NARROW ? 0ull : __ballot(form == (uint32_t)FORM_TINY);
    if (tiny & tiny_span) {                                         // uniform
        const uint32_t gbase = (uint32_t)(g * kGroupSteps * kStepRows);
        const bool mine = form == (uint32_t)FORM_TINY && ((tiny_span >> lane) & 1ull) != 0;
#pragma unroll
        for (uint32_t i = 0; i < kTinyIds; i++) {
            const uint32_t e = (uint32_t)(tw >> (16u * i)) & 0xFFFFu;
            if (mine && i < my_cnt && my_off + i < a.out_cap) {
                uint32_t id = gbase + e;
                if constexpr (GATHER) id = a.cand[ex.begin + id];   // (a listed position lies inside the probed range)
                st_id(a.out_ids + my_off + i, id + a.id_base);
            }
        }
    }
    const uint64_t all_nonempty = __ballot(my_cnt != 0) & span & ~tiny;     // non-empty steps of the range that left something to fetch (wave-uniform)
    if (!all_nonempty) return;
    // Steps that left 16-bit entries in their slots (what a few-percent answer consists of): copied first, up to
    // 16 steps to a round of loads, every step at its own output offset -- no LDS, no ring, no order among them.
    const uint64_t direct = NARROW ? 0ull : __ballot(form == FORM_DIRECT) & span;
    if (direct) {                                                   // uniform
        // entries that have a place in the caller's buffer (a result that does not fit is cut off, the count says so)
        const uint64_t room = my_off < a.out_cap ? a.out_cap - my_off : 0ull;
        const uint32_t lim_v = form != FORM_DIRECT ? 0u : (room < (uint64_t)my_cnt ? (uint32_t)room : my_cnt);
        const uint64_t big = __ballot(my_cnt > (uint32_t)kSlotWords) & direct;
        for (uint64_t rest = direct; rest;) rest = expand_direct<GATHER, GATHER ? 4 : 16>(a, ex.begin, g, rest, big, lim_v, incl - my_cnt, a.out_ids + group_off, lane);
    }
    const uint64_t nonempty = all_nonempty & ~direct;               // what is left: 16-bit lists in the list area, bit masks
    if (!nonempty) { if (c0 == 0) PQPS_STAMP_GROUP(a, g, 6); return; }
    const uint64_t slotted = __ballot(form == FORM_MASK) & span;    // ... those with a 128-byte bit mask to fetch
    const uint32_t rpl_log2 = ((uint32_t)__builtin_amdgcn_readlane((int)cw, (int)__builtin_ctzll(nonempty)) >> kRplShift) & 7u;   // one per query
    uint32_t *ring = sh.stage[park];
    OutRing r;
    r.head = 0;
    r.pending = 0;
    r.pos = readlane_u64(my_off, (int)__builtin_ctzll(nonempty));   // the range's IDs form one run of the output
    typedef __attribute__((address_space(1))) const void global_cvoid;
    typedef __attribute__((address_space(3))) void lds_void;
    // the prefetched words serve if they are exactly what this range needs (same 16 steps, same non-empty ones)
    const bool have = (pre >> 16) == 1u && c1 == c0 + 16u && (uint32_t)(slotted >> c0) == (pre & 0xFFFFu);    // uniform
    const bool have_packed = (pre >> 16) == 2u && slotted == pre_mask;      // the leader's look ahead for a group it expands alone
    // matches of the block of 4 steps a lane's step belongs to (DPP quad sums; used for 1-byte predicates)
    uint32_t quad = my_cnt + dpp_or_zero<0xb1>(my_cnt);
    quad += dpp_or_zero<0x4e>(quad);
    if (c1 - c0 > kGroupSteps / kWaves && __popcll(nonempty) <= (int)(kGroupSteps / kWaves)) {
        // At most 16 non-empty steps in the whole range (a sparse answer): ONE round of loads fetches all their
        // match words, slot k of the LDS slice = the k-th of them (prefetched: slot = step - c0), and they are
        // expanded step by step.
        const uint32_t *gm = (const uint32_t *)(a.masks + g * kGroupSteps * 64) + lane;
        uint32_t k = 0;
        if (!have && !have_packed)
            for (uint64_t rest = slotted; rest; rest &= rest - 1, k++) {      // uniform
                const uint32_t st = (uint32_t)__builtin_ctzll(rest);
                if (lane < 32) __builtin_amdgcn_global_load_lds((global_cvoid *)(gm + (size_t)st * 32), (lds_void *)&sh.mask[park][k][0], 4, 0, 16 /* sc1 */);
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // (the compiler does not count LDS-DMA as a write to LDS)
        if (c0 == 0) PQPS_STAMP_GROUP(a, g, 5);
        for (uint64_t rest = nonempty; rest; rest &= rest - 1) {
            const uint32_t st = (uint32_t)__builtin_ctzll(rest);
            const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)st);
            k = (uint32_t)__popcll(slotted & ((1ull << st) - 1ull));  // packed: slot k = the k-th step that has one
            ring_seek(a, ring, r, lane, readlane_u64(my_off, (int)st));
            expand_step<GATHER>(a, ex.begin, g * kGroupSteps + st, sh.mask[park][have ? st - c0 : k], rpl_log2, cwi & kCountMask, lane, ring, r);
        }
        if (r.pending) ring_flush(a, ring, r, lane, r.pending);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (c0 == 0) PQPS_STAMP_GROUP(a, g, 6);
        return;
    }
    for (uint32_t w0 = c0; w0 < c1; w0 += kGroupSteps / kWaves) {   // 16 steps at a time
        const uint32_t bits = uniform_u32((uint32_t)(nonempty >> w0) & 0xFFFFu);
        if (!bits) continue;
        const uint32_t fetch = uniform_u32((uint32_t)(slotted >> w0) & 0xFFFFu);
        // The match words of the window's non-empty steps go from memory straight into LDS (LDS-DMA: 128 bytes per
        // step by 32 lanes, no vector register in between), all requested before any is awaited: one memory
        // latency per window.  The slot of an empty step keeps whatever it held: nobody looks at it.
        {
            const uint32_t *gmask = (const uint32_t *)(a.masks + (g * kGroupSteps + w0) * 64) + lane;
            if (!(have && w0 == c0))                                // (prefetched: the words of this window are on their way already)
                for (uint32_t rest = fetch; rest; rest &= rest - 1) {   // uniform
                    const uint32_t k = (uint32_t)__builtin_ctz(rest);
                    if (lane < 32) __builtin_amdgcn_global_load_lds((global_cvoid *)(gmask + (size_t)k * 32), (lds_void *)&sh.mask[park][k][0], 4, 0, 16 /* sc1 */);
                }
            if (fetch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the compiler does not count LDS-DMA as a write to LDS)
            if (c0 == 0 && w0 == 0) PQPS_STAMP_GROUP(a, g, 5);
        }
        uint32_t lists16 = 0;                                       // steps of the window that left 16-bit lists: copied in batches
        if constexpr (!GATHER) lists16 = bits & ~fetch;
        if (lists16) {
            const uint32_t big = uniform_u32((uint32_t)(__ballot(my_cnt > 256u) >> w0) & 0xFFFFu) & lists16;
#pragma unroll 1
            for (uint32_t h = 0; h < 2; h++) {                      // halves of 8 steps
                const uint32_t lh = (lists16 >> (8 * h)) & 0xFFu;
                if (!lh) continue;
                if (!((big >> (8 * h)) & 0xFFu)) { expand_lists16<8, 1>(a, g * kGroupSteps + w0 + 8 * h, lh, cw, my_off, w0 + 8 * h, lane); continue; }
#pragma unroll 1
                for (uint32_t b = 0; b < 2; b++) {
                    const uint32_t lb = (lh >> (4 * b)) & 0xFu;
                    if (lb) expand_lists16<4, 2>(a, g * kGroupSteps + w0 + 8 * h + 4 * b, lb, cw, my_off, w0 + 8 * h + 4 * b, lane);
                }
            }
        }
#pragma unroll 1
        for (uint32_t bb = 0; bb < 4; bb++) {                       // blocks of 4 steps
            const uint32_t nb = (bits >> (4 * bb)) & 0xFu;
            if (!nb) continue;
            const uint32_t sidx0 = w0 + 4 * bb;
            const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)quad, (int)sidx0);
            const uint32_t lb = (lists16 >> (4 * bb)) & 0xFu;
            if (rpl_log2 == 4 && total <= kBlockIds && lb == 0u && ((uint32_t)(direct >> sidx0) & 0xFu) == 0u) {  // uniform
                ring_seek(a, ring, r, lane, readlane_u64(my_off, (int)sidx0 + (int)__builtin_ctz(nb)));
                expand_block16<GATHER>(a, ex.begin, g * kGroupSteps + sidx0, total, nb, &sh.mask[park][4 * bb], lane, ring, r);
                continue;
            }
#pragma unroll 1
            for (uint32_t i = 0; i < 4; i++) {                      // (one copy of expand_step: the instruction cache is the scan tiles')
                if (!((nb >> i) & 1u)) continue;
                const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)(sidx0 + i));
                if ((lb >> i) & 1u) continue;                       // (done above; its IDs interrupt the run of the steps around it)
                ring_seek(a, ring, r, lane, readlane_u64(my_off, (int)(sidx0 + i)));
                expand_step<GATHER>(a, ex.begin, g * kGroupSteps + sidx0 + i, sh.mask[park][4 * bb + i], rpl_log2, cwi & kCountMask, lane, ring, r);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // reads done before the slice is filled again
    }
    if (r.pending) ring_flush(a, ring, r, lane, r.pending);
    if (c0 == 0) PQPS_STAMP_GROUP(a, g, 6);
}

// The leader wave of group g settles what the group needs: waits (bounded) for its count words and for the sums
// in front of it, publishes the group's own sum (and the supergroup's, if it is its last group) unless a tile
// has done so.  Returns false if it gave the group up.
//
// Recovery: the LAST group's leader, once every leader is past its wait, looks after the groups others gave up on (which
// in-order dispatch never produces).  By then every other expander is past its wait, so this wave is the
// only one on the chip that waits for anything, and what it waits for are scan tiles, which wait for nothing.
// It first publishes the sums that are missing, in ascending order, then expands the deferred groups.
constexpr int kNearGroups = 6;     // supergroups in front whose group sums are read along with their words when a word is missing

template <int NEAR, bool NARROW = false>
__device__ __forceinline__ bool settle_group(CArgs &a, const Extent &ex, uint64_t g, uint32_t lane, uint32_t limit, bool recovery,
                                             uint32_t &cw, uint64_t &tw, uint64_t &psum, bool final_word = false, LeaderPrefetch *lp = nullptr) {
    const uint64_t tag = (uint64_t)a.epoch << kWordEpochShift;
    uint64_t own_super = 0;
    uint32_t left = 3u;
    bool sum_out = recovery;                                        // (the recovery pass has published every sum beforehand)
    // Far from the end of the table every word in front has been written by a tile's sum duty: the first look reads
    // just those; only if one is missing (or near the end, where no tile does sum duty) the group sums behind the
    // missing supergroup words are read as well.
    bool light = g + a.sum_lag + (uint64_t)(NEAR + 1) * kSuperGroups < ex.groups;
    const bool long_wait = limit >= kRecoverSpins;                  // uniform: bounded by the wall clock instead of a poll count
    const uint64_t deadline = long_wait ? wall_clock64() + kRecoverTicks : 0ull;
    for (uint32_t spins = 0;; spins++) {
        const uint64_t *watch = nullptr;
        left = light ? poll_group<0, NARROW>(a, ex, g, lane, left, cw, tw, psum, own_super, watch)
                     : poll_group<NEAR, NARROW>(a, ex, g, lane, left, cw, tw, psum, own_super, watch);
        light = false;
        PQPS_STAMP_VALUE(a, g, 4, (uint64_t)spins + 1);
        if (!(left & 1u) && !sum_out) {
            PQPS_STAMP_GROUP(a, g, 1);
            // the group's matches: normally a tile has summed them up long ago; at the end of the table nobody has
            const uint32_t sum = wave_sum_u32(cw & kCountMask);
            if (lane == 0) st_sc1(a.gsum + g, tag | (uint64_t)sum);
            sum_out = true;
            if (lp && lp->pre == 0u && !(a.tune & 1u)) {            // uniform: the look ahead at the start came too early
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                lp->pre = leader_prefetch_with(a, *lp->sh, g, lane, lp->park, g * kGroupSteps + lane < ex.steps ? cw : 0u, lp->mask);
            }
        }
        if (left == 0 || spins >= limit || (long_wait && wall_clock64() > deadline)) break;
        __builtin_amdgcn_s_sleep(16);
        // the front is what is missing: cheap looks at one of the missing words until it has appeared
        if (left == 2u && watch != nullptr)
            while (spins < limit && !word_valid(a, ld_sc1(watch))) {
                if (long_wait && wall_clock64() > deadline) break;
                __builtin_amdgcn_s_sleep(8);
                spins++;
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");          // no instruction: payload loads stay behind the polls
    const bool ok = left == 0;
    if (ok) {
        // likewise the supergroup's total, by the leader of its last group
        const uint32_t sum = wave_sum_u32(cw & kCountMask);
        if (lane == 0 && g % kSuperGroups == kSuperGroups - 1) st_sc1(a.ssum + g / kSuperGroups, tag | (own_super + (uint64_t)sum));
    } else if (recovery || final_word) {
        if (lane == 0) report_gave_up(a);                        // something never arrived: reported, never silent
    } else {
        if (lane == 0) st_sc1(a.deferred + g, (a.epoch << kEpochShift) | (sum_out ? 3u : 1u));
        drain_stores();                                             // ... in memory before this leader says it is past its wait
    }
    return ok;
}

// Recovery pass: see settle_group.  One wave, cold code -- and SMALL code: it used to call settle_group and expand_range
// like every expander, and that second copy of the two made the kernels 55 KB instead of 33 and cost 150 bytes per
// lane of spills, 1 - 1.5 us on every ID query.  Once the missing sums are out (first half) nothing has to be waited
// for any more: every deferred group's count words are complete and every group sum in front of it is published, so
// what is in front is a plain sum, and the group's steps are expanded one at a time.
template <bool GATHER, bool NARROW = false>
__device__ __forceinline__ void recover_deferred(CArgs &a, FusedShared &sh, const Extent &ex, uint32_t lane_in, uint32_t park) {
    uint32_t lane = lane_in;
    asm volatile("" : "+v"(lane));                                  // (keeps this cold code's address arithmetic out of the callers' registers)
    const uint64_t tag = (uint64_t)a.epoch << kWordEpochShift;
    bool alive = true;                                              // pass 1: the group sums nobody has published, in ascending order
#pragma unroll 1
    for (uint64_t p0 = 0; alive && p0 < ex.groups; p0 += 64) {
        const uint32_t f = p0 + lane < ex.groups ? ld_sc1(a.deferred + p0 + lane) : 0u;
        uint64_t fw = __ballot((f >> kEpochShift) == a.epoch && (f & 3u) == 1u);
        while (alive && fw) {
            const uint64_t gg = p0 + (uint64_t)__builtin_ctzll(fw);
            fw &= fw - 1;
            uint32_t cw = 0;
            alive = false;
            const uint64_t deadline = wall_clock64() + kRecoverTicks;
#pragma unroll 1
            for (uint32_t spins = 0; spins < kRecoverSpins; spins++) {
                cw = gg * kGroupSteps + lane < ex.steps ? ld_sc1(a.counts + gg * kGroupSteps + lane) : a.epoch << kEpochShift;
                if (__all((cw >> kEpochShift) == a.epoch)) { alive = true; break; }
                if (wall_clock64() > deadline) break;
                __builtin_amdgcn_s_sleep(16);
            }
            const uint32_t sum = wave_sum_u32(cw & kCountMask);
            if (alive && lane == 0) st_sc1(a.gsum + gg, tag | (uint64_t)sum);
        }
    }
    if (!alive && lane == 0) report_gave_up(a);                  // a scan tile never arrived: reported, never silent
    drain_stores();
    // ... and the supergroup words still missing (every group sum is out now)
#pragma unroll 1
    for (uint64_t j = lane; alive && j * kSuperGroups + kSuperGroups <= ex.groups; j += 64) {
        if (word_valid(a, ld_sc1(a.ssum + j))) continue;
        uint64_t sum = 0;
        bool all = true;
#pragma unroll 1
        for (uint32_t k = 0; k < kSuperGroups; k++) {
            const uint64_t w = ld_sc1(a.gsum + j * kSuperGroups + k);
            all = all && word_valid(a, w);
            sum += w & kWordMask;
        }
        if (all) st_sc1(a.ssum + j, tag | sum);
    }
    drain_stores();
    if (!alive) return;
    uint32_t *ring = sh.stage[park];
#pragma unroll 1
    for (uint64_t f0 = 0; f0 < ex.groups; f0 += 64) {               // pass 2: expand what was given up, in ascending order
        const uint32_t f = f0 + lane < ex.groups ? ld_sc1(a.deferred + f0 + lane) : 0u;
        uint64_t todo = __ballot((f >> kEpochShift) == a.epoch && (f & 1u) != 0u);
        while (todo) {
            const uint64_t g = f0 + (uint64_t)__builtin_ctzll(todo);
            todo &= todo - 1;
            const uint32_t cw = g * kGroupSteps + lane < ex.steps ? ld_sc1(a.counts + g * kGroupSteps + lane) : a.epoch << kEpochShift;
            uint64_t front = 0;
            bool known = (cw >> kEpochShift) == a.epoch;
#pragma unroll 1
            for (uint64_t j = lane; j < g; j += 64) {               // the matches in front: every group sum before g
                const uint64_t w = ld_sc1(a.gsum + j);
                known = known && word_valid(a, w);
                front += w & kWordMask;
            }
            if (!__all(known)) { if (lane == 0) report_gave_up(a); continue; }       // (cannot be: pass 1 saw to both)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");  // no instruction: the payload loads stay behind the looks
            const uint32_t my_cnt = g * kGroupSteps + lane < ex.steps ? (cw & kCountMask) : 0u;
            OutRing r;
            r.head = 0;
            r.pending = 0;
            r.pos = (a.accumulate ? ld_sc1(a.base_slot) : 0ull) + wave_sum_u64(front);
            if (g + 1 == ex.groups) {                               // (the last group's expander leaves the total)
                const uint64_t total = r.pos + wave_sum_u32(my_cnt);
                if (lane == 0) *a.out_count = total;
            }
#pragma unroll 1
            for (uint64_t rest = __ballot(my_cnt != 0); rest; rest &= rest - 1) {        // one step at a time
                const uint32_t st = (uint32_t)__builtin_ctzll(rest);
                const uint32_t cwi = (uint32_t)__builtin_amdgcn_readlane((int)cw, (int)st);
                const uint64_t step = g * kGroupSteps + st;
                if (!NARROW && word_form(a, cwi) == FORM_TINY) {    // 1 - kTinyIds matches: in the step's tiny word
                    if (r.pending) ring_flush(a, ring, r, lane, r.pending);
                    const uint32_t n = cwi & kCountMask;
                    uint64_t t = ld_sc1(a.tiny + step);
                    for (const uint64_t deadline = wall_clock64() + kRecoverTicks; !word_valid(a, t) && wall_clock64() < deadline;) {   // (stored beside the count word)
                        __builtin_amdgcn_s_sleep(8);
                        t = ld_sc1(a.tiny + step);
                    }
                    if (!word_valid(a, t)) { if (lane == 0) report_gave_up(a); }
                    else if (lane < n && r.pos + lane < a.out_cap) {
                        uint32_t id = (uint32_t)(g * kGroupSteps * kStepRows) + ((uint32_t)(t >> (16u * lane)) & 0xFFFFu);
                        if constexpr (GATHER) id = a.cand[ex.begin + id];
                        st_id(a.out_ids + r.pos + lane, id + a.id_base);
                    }
                    r.pos += n;
                    continue;
                }
                if (!NARROW && word_form(a, cwi) == FORM_DIRECT) {  // 16-bit entries in its slot: a copy (every lane: this step's limit and offset)
                    if (r.pending) ring_flush(a, ring, r, lane, r.pending);
                    const uint32_t n = cwi & kCountMask;
                    const uint64_t room = r.pos < a.out_cap ? a.out_cap - r.pos : 0ull;
                    const uint32_t lim = room < (uint64_t)n ? (uint32_t)room : n;
                    (void)expand_direct<GATHER, 1>(a, ex.begin, g, 1ull << st, n > (uint32_t)kSlotWords ? ~0ull : 0ull, lim, 0u, a.out_ids + r.pos, lane);
                    r.pos += n;
                    continue;
                }
                if (step_has_slot(a, cwi)) {                        // its 128 bytes into the wave's LDS slice, slot 0
                    const uint32_t v = lane < 32 ? ld_sc1((const uint32_t *)(a.masks + step * 64) + lane) : 0u;
                    if (lane < 32) ((uint32_t *)&sh.mask[park][0][0])[lane] = v;
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                expand_step<GATHER>(a, ex.begin, step, sh.mask[park][0], (cwi >> kRplShift) & 7u, cwi & kCountMask, lane, ring, r);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the slot is filled again
            }
            if (r.pending) ring_flush(a, ring, r, lane, r.pending);
        }
    }
}

// An expander workgroup.  Among the scan tiles: four independent waves, one group each.  Behind the last tile:
// one group, its leader wave settles it and hands count words and output slot to the other three through LDS.
template <bool GATHER, bool LOOPED = false, bool NARROW = false>   // LOOPED: called again for further groups -- every wave reaches the barrier
__device__ __forceinline__ void expander_workgroup(CArgs &a, FusedShared &sh, const Extent &ex, const Role &role) {
    const uint32_t lane = threadIdx.x & 63, wave = uniform_u32(threadIdx.x >> 6);
    const bool shared = role.kind == ROLE_EXPAND_GROUP;             // uniform for the workgroup
    const bool leader = !shared || wave == 0;
    // Gather (index mode): few groups, often dense, and every step of the expansion is a chain of memory latencies (bit mask,
    // candidate numbers, stores) -- a workgroup takes a SIXTEENTH of a group (kGatherParts: 4 steps, one per wave), so that a
    // probe is spread over sixteen times as many waves (with a quarter of a group per workgroup, 4 steps per wave: `sudo_used = TRUE AND
    // risk_level > 3` over 4.4 M candidates 140 us); role.index counts parts.  Every part's leader settles the group for itself
    // (and publishes its sum: the same value every time).
    const uint32_t part = GATHER ? role.index % kGatherParts : 0u;
    const uint64_t g = GATHER ? (uint64_t)(role.index / kGatherParts) : (shared ? (uint64_t)role.index : (uint64_t)role.index * 4u + wave);
    if (g >= ex.groups) return;                                     // (the last quad of a table can be short)
    uint32_t c0 = shared ? wave * (kGroupSteps / kWaves) : 0u, c1 = shared ? c0 + kGroupSteps / kWaves : (uint32_t)kGroupSteps;
    if constexpr (GATHER) { c0 = part * (kGroupSteps / kGatherParts) + wave * (kGroupSteps / kGatherParts / kWaves); c1 = c0 + kGroupSteps / kGatherParts / kWaves; }
    bool ok = false;
    uint32_t cnts = 0;
    uint64_t group_off = 0;
    uint64_t tiny_words = 0;                                        // leaders: lane l = the tiny word of step l (settle_group)
    if (shared && !leader && !LOOPED) {
        // A group with few matches is its leader's alone: the other three waves leave as soon as they know (from the
        // group's sum, which a tile has published long ago unless this is the end of the table) -- a wave that waits
        // at the barrier holds a slot the next query's scan could use.
        const uint64_t w = ld_sc1(a.gsum + g);
        if (word_valid(a, w) && (w & kWordMask) <= kSoloIds) return;        // uniform
    }
    uint32_t pre = 0;
    uint64_t pre_mask = 0;
    if constexpr (!GATHER) {                                        // (a gather workgroup's waves take 4 steps each)
        if (shared) {
            pre = leader ? prefetch_as_leader(a, sh, ex, g, lane, wave, pre_mask) : prefetch_own_steps(a, sh, ex, g, lane, c0, wave);
            // a wave of the last wavefront's groups is early: it has nothing else to do until its leader has settled the
            // group (which cannot be before these count words exist), so it looks again a few times
            if (!leader && !(a.tune & 1u))
                for (uint32_t tries = 0; pre == 0u && tries < 12u; tries++) { __builtin_amdgcn_s_sleep(24); pre = prefetch_own_steps(a, sh, ex, g, lane, c0, wave); }
        }
    }
    if (leader) {
        uint32_t cw = 0;
        uint64_t psum = 0;
        PQPS_STAMP_GROUP(a, g, 0);
        // A leader's polls come first on its SIMD: behind the last tile the neighbouring waves are busy expanding
        // (the vector units are the bottleneck there), and the sums this leader publishes are what other groups wait
        // for (a lone u8 column at 100 M rows: last group settled 9.1 -> 3.8 us after the last tile).
        __builtin_amdgcn_s_setprio(3);
        // (gather: the whole grid is resident at once, so a wait can only fail if a tile never ran -- no second
        // chance through the recovery pass there, the wait is long and its failure sets the status word)
        LeaderPrefetch lp{&sh, wave, pre, pre_mask};
        ok = settle_group<kNearGroups, NARROW>(a, ex, g, lane, GATHER ? kRecoverSpins : a.spin_limit, false, cw, tiny_words, psum, GATHER, (!GATHER && shared) ? &lp : nullptr);
        pre = lp.pre;
        pre_mask = lp.mask;
        __builtin_amdgcn_s_setprio(0);
        PQPS_STAMP_GROUP(a, g, 2);
        if constexpr (!GATHER) leader_past_its_wait(a, g, lane, !ok);   // (a group given up is on record by now)
        if (ok) {
            cnts = g * kGroupSteps + lane < ex.steps ? (cw & 0xFFFFu) : 0u;
            group_off = (a.accumulate ? ld_sc1(a.base_slot) : 0ull) + psum;
        }
    }
    if (shared) {
        const bool solo = ok && wave_sum_u32(cnts & kCountMask) <= kSoloIds;     // (leader's view; the others read sh.state)
        if (leader) {
            sh.counts[lane] = cnts;
            if (lane == 0) { sh.group_off = group_off; sh.state = !ok ? 0u : (solo ? 2u : 1u); }
        }
        __syncthreads();                                            // (waves that have left are not waited for)
        const uint32_t state = sh.state;
        if (state == 2u) {                                          // few matches: the leader does all 64 steps (gather: its workgroup's part)
            if (!leader) return;
            c0 = GATHER ? part * (kGroupSteps / kGatherParts) : 0u;
            c1 = GATHER ? c0 + kGroupSteps / kGatherParts : (uint32_t)kGroupSteps;
        }
        ok = state != 0u;
        cnts = sh.counts[lane];
        group_off = sh.group_off;
    }
    // The steps with 1 - kTinyIds matches are their group's LEADER's, whichever wave the other steps around them belong to: it has
    // their entries from the look that settled the group (gather: every part's leader settles for itself -- its part's steps).
    uint64_t tiny_span = 0;
    if (leader) tiny_span = !GATHER ? ~0ull : (((1ull << (kGroupSteps / kGatherParts)) - 1ull) << (part * (kGroupSteps / kGatherParts)));
    if (ok) expand_range<GATHER, NARROW>(a, sh, ex, g, lane, c0, c1, wave, cnts, group_off, pre, pre_mask, tiny_span, tiny_words);
    PQPS_STAMP_GROUP_MAX(a, g, 3);
    // the last group's leader looks after the groups others gave up on (if any), once all of them are past their waits
    if constexpr (!GATHER)
        if (leader && g + 1 == ex.groups && expanders_past_their_wait(a, ex.groups, lane)) recover_deferred<GATHER, NARROW>(a, sh, ex, lane, wave);
}

// Generic scan: any predicate; scan (full steps vectorised) or gather (always guarded).
template <int MODE, bool GATHER, bool NT = false>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? (GATHER ? 2 : 4) : 1) void eval_generic_kernel(const EvalArgs) {
    CArgs &a = kernel_args();
    const uint32_t lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if constexpr (MODE == MODE_IDS) {
        constexpr int TS = kWaves;                              // steps per tile
        __shared__ FusedShared sh;
        zero_other_ctl(a);
        const Extent ex = scan_extent<GATHER>(a);
        // Gather: the probed range is known on the device only, and a grid sized for the caller's upper bound (the
        // whole table) would spend tens of microseconds dispatching workgroups that find nothing to do.  The host
        // sizes the grid for at most a.grid_groups groups; a wider range is covered by every workgroup taking its
        // role again `layout` groups (tiles) further on.  (Scan tiles wait for nothing and the expanders come after
        // them in the grid, so the expanders' bounded waits always end.)
        const uint32_t layout = GATHER && ex.groups > a.grid_groups ? a.grid_groups : (uint32_t)ex.groups;
        Role role;
        uint32_t layout_tiles;
        if constexpr (GATHER) {
            // gather grid: the tiles of `layout` groups, then kGatherParts expander workgroups per group
            layout_tiles = layout * (uint32_t)(kGroupSteps / TS);
            role.kind = uniform_u32(blockIdx.x < layout_tiles ? (uint32_t)ROLE_SCAN : (blockIdx.x - layout_tiles < kGatherParts * layout ? (uint32_t)ROLE_EXPAND_GROUP : (uint32_t)ROLE_NONE));
            role.index = uniform_u32(blockIdx.x < layout_tiles ? blockIdx.x : blockIdx.x - layout_tiles);
            if (role.kind == ROLE_EXPAND_GROUP) {
                for (Role r = role;;) {
                    expander_workgroup<GATHER, true>(args_for_expanders(), sh, ex, r);
                    r.index += kGatherParts * layout;
                    if (r.index / kGatherParts >= ex.groups) break;
                    __syncthreads();                                // the workgroup's LDS hand-over is free again
                }
                return;
            }
        } else {
            role = fused_role<kGroupSteps / TS>(a, layout);
            layout_tiles = ((layout + 3u) / 4u) * 4u * (uint32_t)(kGroupSteps / TS);
            if (role.kind >= ROLE_EXPAND_QUAD) { expander_workgroup<GATHER>(args_for_expanders(), sh, ex, role); return; }
        }
        if (role.kind != ROLE_SCAN) return;
        for (uint64_t tile = role.index; tile * TS < ex.steps; tile += layout_tiles) {
            const uint64_t step = tile * TS + wv;
            const SumDuty duty = sum_duty_load<kGroupSteps / TS>(a, (uint32_t)tile, wv, lane);
            zero_tiny_words(sh, wv, 1);
            uint32_t cnt = 0;
            if (step < ex.steps) {
                const uint64_t step_row0 = step * kStepRows;
                uint32_t mbits;
                if (GATHER) mbits = eval_step_gather(a, step_row0, ex.n_rows, ex.begin, lane);
                else {
                    mbits = eval_step_full<NT>(a, step_row0, lane);
                    if (step_row0 + kStepRows > ex.n_rows) mbits &= rows_below<kRplGeneric>(step_row0, ex.n_rows, lane);   // the partial last step
                }
                cnt = wave_sum_u32(__popc(mbits));
                tile_step_out(a, sh, wv, step, cnt, mbits, 2, lane);
            } else if (lane == 0) {
                sh.tile_cnt[wv] = 0;
            }
            if (cnt) drain_stores();
            __syncthreads();
            if (wv == 0) { publish_tile<TS>(a, sh, ex, tile, lane); sum_duty_finish(a, duty, lane); PQPS_STAMP_TILE(a, tile); }
            if (!GATHER) break;                                     // a scan's grid covers every tile
            __syncthreads();                                        // sh.tile_cnt is free again
        }
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
            mbits = eval_step_full<NT>(a, step_row0, lane);
            if (step_row0 + kStepRows > n_rows) mbits &= rows_below<kRplGeneric>(step_row0, n_rows, lane);             // the partial last step
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
__device__ __forceinline__ void eval_col_masks(CArgs &a, int slot, const RawCol<W, RPL, U> &raw, LeafMasks &lm) {
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
__device__ __forceinline__ void eval_col_chain(CArgs &a, int slot, const RawCol<W, RPL, U> &raw, RowPlanes &acc) {
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
__device__ __forceinline__ void eval_col(CArgs &a, int slot, const RawCol<W, RPL, U> &raw, uint32_t (&idx)[16]) {
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

// ---- chain predicates on the VECTOR unit (kernel variant EV = 2) ---------------------------------------------
// t[r] holds bit r in the lanes whose row r still satisfies every leaf seen so far; a leaf clears it where the row
// fails (one compare + one select per row and leaf, no scalar instruction).  The ballot path costs ~50 scalar
// instructions per leaf and step, and a CU has ONE scalar unit for its four SIMDs (rocprofv3 on S1 at 100 M rows: 227
// SALU + 18 SMEM per 3 KB step, the scalar unit 80 % busy).  A kernel variant of its own: as a run-time branch next to
// the ballot path it cost the other path its registers (85 - 93 VGPRs instead of 55 - 63 in the COUNT kernels).
template <typename T>
__device__ __forceinline__ void valu_leaf(const T (&v)[16], T lo, T span, bool want, uint32_t (&t)[16]) {
    // `want`: the raw window hit this leaf needs; the six branches are wave-uniform
    if (span == 0) {
        if (want) {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = v[r] == lo ? t[r] : 0u;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = v[r] != lo ? t[r] : 0u;
        }
    } else if (lo == 0) {
        if (want) {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = v[r] <= span ? t[r] : 0u;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = v[r] > span ? t[r] : 0u;
        }
    } else {
        if (want) {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = (T)(v[r] - lo) <= span ? t[r] : 0u;
        } else {
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = (T)(v[r] - lo) > span ? t[r] : 0u;
        }
    }
}

template <int W, int RPL, int U>
__device__ __forceinline__ void eval_col_valu(CArgs &a, int slot, const RawCol<W, RPL, U> &raw, uint32_t (&t)[16]) {
    static_assert(W <= 4, "the vector-unit chain path is for columns of up to 4 bytes");
    const uint32_t kb = a.leaf_begin[slot], ke = a.leaf_begin[slot + 1];
    uint32_t v[16];
    unpack32(raw, v, std::make_integer_sequence<int, 16>{});
    for (uint32_t k = kb; k < ke; k++)                              // uniform
        valu_leaf<uint32_t>(v, (uint32_t)a.lo[k], (uint32_t)a.span[k], ((a.chain_want >> k) & 1u) != 0, t);
}

constexpr int log2i(int x) { return x <= 1 ? 0 : 1 + log2i(x / 2); }

// All raw registers of one step for the (W0, W1, W2) shape.
template <int W0, int W1, int W2, int RPL, int U>
struct RawStep {
    RawCol<W0, RPL, U> r0;
    RawCol<(W1 ? W1 : 1), RPL, U> r1;
    RawCol<(W2 ? W2 : 1), RPL, U> r2;
    template <bool NT>
    __device__ __forceinline__ void load(CArgs &a, uint64_t lane_row0) {
        r0.template load<NT>(a.col[0], lane_row0);
        if constexpr (W1 != 0) r1.template load<NT>(a.col[1], lane_row0);
        if constexpr (W2 != 0) r2.template load<NT>(a.col[2], lane_row0);
    }
    template <int MODE, int H>
    __device__ __forceinline__ void eval_chain_half(CArgs &a, uint32_t &cnt, uint32_t &mbits) const {
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
    __device__ __forceinline__ void one_leaf(CArgs &a, const T (&v)[16], uint32_t &m, uint32_t &lane_total) const {
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
    // EV 0: ballots into SGPR planes (any chain); EV 1: ONE comparison on ONE column on the vector unit; EV 2: a chain over
    // narrow columns on the vector unit (valu_leaf).
    template <int MODE, int EV>
    __device__ __forceinline__ void eval_chain_step(CArgs &a, uint32_t &cnt, uint32_t &mbits, uint32_t &lane_total) const {
        if constexpr (EV == 1) {
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
        } else if constexpr (EV == 2) {
            static_assert(W0 <= 4, "narrow columns");
            uint32_t t[16];
#pragma unroll
            for (int r = 0; r < 16; r++) t[r] = 1u << r;
            eval_col_valu<W0, RPL, U>(a, 0, r0, t);
            if constexpr (W1 != 0) eval_col_valu<W1, RPL, U>(a, 1, r1, t);
            if constexpr (W2 != 0) eval_col_valu<W2, RPL, U>(a, 2, r2, t);
            uint32_t m = 0;
#pragma unroll
            for (int r = 0; r < 16; r++) m |= t[r];
            if (a.chain == 2) m ^= 0xFFFFu;                         // OR form: NOT of the AND
            if (MODE == MODE_IDS) {
                // most steps of a sparse answer have no match: one compare + one scalar test instead of a wave reduction
                cnt = __ballot(m != 0) ? wave_sum_u32(__popc(m)) : 0u;
                mbits = m;
            } else {
                lane_total += __popc(m);
            }
        } else {
            eval_chain_half<MODE, 0>(a, cnt, mbits);
            eval_chain_half<MODE, 1>(a, cnt, mbits);
        }
    }
    __device__ __forceinline__ uint32_t eval(CArgs &a) const {     // <= 6 leaves: row-mask path
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
                 "s"((a).masks), "s"((a).chain), "s"((a).chain_want), "s"((a).negmask), "s"((a).lag), "s"((a).sum_lag), "s"((a).counts),     \
                 "s"((uint32_t)(a).leaf_begin[0]), "s"((uint32_t)(a).leaf_begin[1]), "s"((uint32_t)(a).leaf_begin[2]), \
                 "s"((uint32_t)(a).leaf_begin[3]), "s"((a).lists), "s"((a).list_max), "s"((a).list_max_u8), "s"((a).list16_min), "s"((a).list16_min_u8), "s"((a).tiny_max))

// General tree of <= 6 leaves (row-mask path).
template <int MODE, int W0, int W1, int W2, bool NT>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? (W0 + W1 + W2 >= 12 ? 7 : 8) : 1) void eval_spec_kernel(const EvalArgs) {
    CArgs &a = kernel_args();
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
        zero_other_ctl(a);
        const Extent ex = scan_extent<false>(a);
        const Role role = fused_role<kGroupSteps / TS>(a, (uint32_t)ex.groups);
        if (role.kind >= ROLE_EXPAND_QUAD) { expander_workgroup<false, false, (W0 == 1)>(args_for_expanders(), sh, ex, role); return; }
        if (role.kind != ROLE_SCAN || (uint64_t)role.index * TS >= ex.steps) return;
        const uint64_t step = (uint64_t)role.index * TS + wv;
        SumDuty duty;
        uint32_t cnt = 0;
        if constexpr (RPL < 16) zero_tiny_words(sh, wv, 1);
        if (step < ex.steps) {
            RawStep<W0, W1, W2, RPL, U> A;
            A.template load<NT>(a, step * kStepRows + lane_off);
            duty = sum_duty_load<kGroupSteps / TS>(a, role.index, wv, lane);       // behind the column loads, consumed after the tile's work
            uint32_t mbits = A.eval(a);
            if (step >= full_steps) mbits &= rows_below<RPL>(step * kStepRows, n_rows, lane);   // the partial last step
            cnt = wave_sum_u32(__popc(mbits));
            tile_step_out(a, sh, wv, step, cnt, mbits, log2i(RPL), lane);
        } else {
            duty = sum_duty_load<kGroupSteps / TS>(a, role.index, wv, lane);
            if (lane == 0) sh.tile_cnt[wv] = 0;
        }
        if (cnt) drain_stores();
        __syncthreads();
        if (wv == 0) { publish_tile<TS, (RPL < 16)>(a, sh, ex, role.index, lane); sum_duty_finish(a, duty, lane); PQPS_STAMP_TILE(a, role.index); }
    } else {
        const uint64_t wave = (uint64_t)blockIdx.x * kWaves + wv;
        const uint64_t n_waves = (uint64_t)gridDim.x * kWaves;
        uint64_t wave_total = 0;
        const uint64_t steps = (n_rows + kStepRows - 1) / kStepRows;
        for (uint64_t step = wave; step < steps; step += n_waves) {
            RawStep<W0, W1, W2, RPL, U> A;
            A.template load<NT>(a, step * kStepRows + lane_off);
            uint32_t mbits = A.eval(a);
            if (step >= full_steps) mbits &= rows_below<RPL>(step * kStepRows, n_rows, lane);   // the partial last step
            emit_step<MODE>(a, step, mbits, log2i(RPL), n_rows, lane, wave_total);
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
#ifndef PQPS_MULTI_BYTES
#define PQPS_MULTI_BYTES 1
#endif
constexpr int chain_steps(int w0, int w1, int w2) { return w0 + w1 + w2 <= PQPS_MULTI_BYTES ? 2 : 1; }

#ifndef PQPS_CHAIN_WGS
#define PQPS_CHAIN_WGS 8
#endif
template <int MODE, int W0, int W1, int W2, int S, bool NT, int EV>
__global__ __launch_bounds__(kBlock, MODE == MODE_IDS ? PQPS_CHAIN_WGS : 1) void eval_chain_kernel(const EvalArgs) {
    CArgs &a = kernel_args();
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
        PQPS_STAMP_START(a);
        zero_other_ctl(a);
        const Extent ex = scan_extent<false>(a);
        const Role role = fused_role<kGroupSteps / TS>(a, (uint32_t)ex.groups);
#ifdef PQPS_NO_EXPAND   /* experiments: what the scan tiles cost when the kernel holds nothing else (no results) */
        if (role.kind >= ROLE_EXPAND_QUAD) return;
#else
        if (role.kind >= ROLE_EXPAND_QUAD) { expander_workgroup<false, false, (W0 == 1)>(args_for_expanders(), sh, ex, role); return; }
#endif
        if (role.kind != ROLE_SCAN || (uint64_t)role.index * TS >= ex.steps) return;
        const uint64_t step0 = (uint64_t)role.index * TS + (uint64_t)wv * S;
#pragma unroll
        for (int i = 0; i < S; i++)
            if (step0 + i < ex.steps) A[i].template load<NT>(a, (step0 + i) * kStepRows + lane_off);        // uniform guard
        const SumDuty duty = sum_duty_load<kGroupSteps / TS>(a, role.index, wv, lane);   // behind the column loads, consumed after the tile's work
        if constexpr (RPL < 16) zero_tiny_words(sh, wv * S, S);
        uint32_t any = 0;
#pragma unroll
        for (int i = 0; i < S; i++) {
            const uint64_t step = step0 + (uint64_t)i;
            uint32_t cnt = 0, mbits = 0, lane_total = 0;
            if (step < ex.steps) {
                A[i].template eval_chain_step<MODE, EV>(a, cnt, mbits, lane_total);
                if (step >= full_steps) {                       // the partial last step
                    mbits &= rows_below<RPL>(step * kStepRows, n_rows, lane);
                    cnt = wave_sum_u32(__popc(mbits));
                }
                tile_step_out(a, sh, wv * S + i, step, cnt, mbits, log2i(RPL), lane);
            } else if (lane == 0) {
                sh.tile_cnt[wv * S + i] = 0;
            }
            any |= cnt;
        }
        if (any) drain_stores();
        __syncthreads();
        if (wv == 0) { publish_tile<TS, (RPL < 16)>(a, sh, ex, role.index, lane); sum_duty_finish(a, duty, lane); PQPS_STAMP_TILE(a, role.index); }
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
                A[i].template eval_chain_step<MODE, EV>(a, cnt, mbits, lane_total);
                wave_total += cnt;
            }
        }
        if ((n_rows % kStepRows) != 0 && wave == full_steps % n_waves) {      // the partial last step: match bits, trimmed
            uint32_t cnt = 0, mbits = 0, unused = 0;
            A[0].template load<NT>(a, full_steps * kStepRows + lane_off);
            A[0].template eval_chain_step<MODE_IDS, EV>(a, cnt, mbits, unused);
            wave_total += wave_sum_u32(__popc(mbits & rows_below<RPL>(full_steps * kStepRows, n_rows, lane)));
        }
        wave_total += wave_sum_u32(lane_total);
        finish_totals<MODE>(a, wave_total);
    }
}

// COUNT / FLAGS modes: workgroup partial totals -> one number (no same-address atomics)
// One workgroup, every thread's kPartialSlots / kBlock slots requested at once (one memory latency, not sixteen):
// the launch is the tail of every COUNT(*) query, 5.8 us in its first form, and a 100 M-row scan of one byte per row
// takes 16.
__global__ __launch_bounds__(kBlock) void reduce_totals_kernel(uint64_t *partials, uint64_t *out_count) {
    constexpr uint32_t kPer = kPartialSlots / kBlock;
    __shared__ uint64_t s_wave[kWaves];
    uint64_t v[kPer];
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) v[k] = partials[threadIdx.x + k * kBlock];
    uint64_t local = 0;
#pragma unroll
    for (uint32_t k = 0; k < kPer; k++) {
        local += v[k];
        if (v[k]) partials[threadIdx.x + k * kBlock] = 0;          // left zeroed for the next query
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
