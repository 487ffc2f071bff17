/* pqps_hip.h -- thin C-ABI shim over the hand-written gfx950 kernels.
 *
 * This is the lowest drop-in boundary of the HIP backend: plain pointers and
 * sizes, no C++ / torch types.  Everything behind it lives in ONE hipcc
 * translation unit (csrc/pqps_hip.hip).  The C11 engine
 * (engine/hip/executeEngine-hip.c) and the Python harness (ctypes) are its
 * only callers.
 *
 * What each entry point replaces in the reference (Jairik/Parallel-Query-
 * Processing-System, paths relative to its root):
 *
 *   pqps_filter_scan    linearSearchRecords over engine->all_records
 *                       (engine/serial/executeEngine-serial.c:854-878, called
 *                       from :466) with evaluateWhereClause :292-316 /
 *                       checkCondition :251-289 / CMP_* :18-123 fused in.
 *   pqps_filter_gather  linearSearchRecords over the index candidates
 *                       (executeEngine-serial.c:471) -- order preserving.
 *   pqps_filter_count   COUNT(*): resultSetS.numRecords without the ID list
 *                       (MPI shape: MPI_Allreduce at engine/mpi/executeEngine-mpi.c:745).
 *   pqps_filter_flags   the per-row flag array of DELETE
 *                       (engine/omp/executeEngine-omp.c:708-732, mpi :726-741).
 *   pqps_index_build    loadIntoBplusTree (engine/serial/buildEngine-serial.c:41-62)
 *                       -- as a permutation sorted (key asc, row desc), the
 *                       leaf order of engine/bplus.c:282-314,471-490.
 *   pqps_index_probe    findLeaf + the leaf walk of findRange (bplus.c:282-358).
 *   pqps_index_select   one probe of executeQuerySelectSerial, whole (executeEngine-serial.c:358-448).
 *   pqps_partition      the block partition of engine/mpi/executeEngine-mpi.c:703-715.
 *   pqps_exchange_*     the per-query exchange of the MPI engine (executeEngine-mpi.c:717-768:
 *                       local scan of the rank's rows, MPI_Allgather of the sizes + MPI_Allgatherv of
 *                       the IDs; :745 MPI_Allreduce for counts) over RCCL, same shape.
 *   pqps_merge_slots    the displacement arithmetic + placement of MPI_Allgatherv (:758-765) for
 *                       equal-size slots (index mode across processes).
 *   pqps_compact_rows   the survivor compaction of DELETE (executeEngine-serial.c:646-680).
 *   pqps_bump_codes     (no counterpart: keeps dictionary codes order-preserving on INSERT).
 *
 * All functions return 0 on success or a negative PQPS_E* code; the text of
 * the last error of the calling thread is at pqps_last_error().
 * Device pointers are plain `void *` (hipMalloc / torch tensor data_ptr).
 * `stream` is a hipStream_t passed as void*; NULL = the context's own stream.
 */
#ifndef PQPS_HIP_H
#define PQPS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PQPS_OK            0
#define PQPS_EINVAL       -1   /* bad argument (shape, width, alignment, capacity) */
#define PQPS_EHIP         -2   /* a HIP runtime call failed                        */
#define PQPS_ENOMEM       -3
#define PQPS_ENODEVICE    -4   /* no gfx950 device visible                         */
#define PQPS_EOVERFLOW    -5   /* out_ids capacity too small for the matches       */
#define PQPS_ETIMEOUT     -6   /* a bounded host wait of the exchange ran out: the communicator was aborted, the exchange is dead */

#define PQPS_MAX_COLUMNS  12   /* columns of `record` (include/logType.h)          */
#define PQPS_MAX_LEAVES   32   /* leaf comparisons in one WHERE tree               */
#define PQPS_TT_LEAVES     6   /* <= 6 leaves: 64-entry truth table path           */
#define PQPS_TILE_ROWS  4096   /* column buffers are padded to a multiple of this  */

/* One device-resident column.  `width` in {1,2,4,8} bytes per row; values are
 * compared as unsigned after the host has biased signed columns (see
 * pqps_leaf).  `data` must be 16-byte aligned. */
typedef struct pqps_column {
    const void *data;
    uint32_t width;
    uint32_t reserved;
} pqps_column;

/* One leaf comparison, normalised by the host to an unsigned window test
 *      hit = ((value - lo) <= span) XOR negate            (mod 2^w arithmetic)
 * which covers =, !=, <, <=, >, >= on u64 / i32 / bool / dictionary codes
 * (signed i32 windows work unchanged in two's complement). */
typedef struct pqps_leaf {
    uint32_t column;          /* index into the pqps_column array of the call */
    uint32_t negate;          /* 0 or 1                                       */
    uint64_t lo;
    uint64_t span;
} pqps_leaf;

/* Jump-table form of the short-circuit evaluation of the WHERE tree
 * (executeEngine-serial.c:292-316): after leaf k, go to on_true[k] /
 * on_false[k]; targets are a later leaf index, PQPS_ACCEPT or PQPS_REJECT. */
#define PQPS_ACCEPT 0xFE
#define PQPS_REJECT 0xFF

typedef struct pqps_predicate {
    uint32_t n_leaves;                       /* 0: constant predicate (truth bit 0) */
    uint32_t n_columns;                      /* columns referenced, <= 12            */
    uint64_t truth;                          /* truth table over leaf bits, n<=6     */
    pqps_leaf leaf[PQPS_MAX_LEAVES];         /* sorted by `column`                   */
    uint8_t on_true[PQPS_MAX_LEAVES];        /* in ORIGINAL evaluation order ...     */
    uint8_t on_false[PQPS_MAX_LEAVES];
    uint8_t order[PQPS_MAX_LEAVES];          /* ... order[i] = leaf slot of step i   */
} pqps_predicate;

typedef struct pqps_ctx pqps_ctx;

const char *pqps_last_error(void);
/* The kernel instantiation the calling thread's last filter call chose, as rocprofv3 would name it
 * (e.g. "eval_chain_kernel<MODE_IDS, W0=2, W1=1, W2=0, S=1, NT=true, VC=false>"): what bench.py reports. */
const char *pqps_last_kernel(void);

/* Context = device ordinal + stream + filter scratch (match bits, step / group / supergroup
 * counts; grown on demand).  One query at a time per context; use one context per host thread. */
int  pqps_ctx_create(int device, pqps_ctx **out);
void pqps_ctx_destroy(pqps_ctx *ctx);
/* Allocates the filter scratch for tables of up to n_rows now (otherwise the first query does it:
 * half a dozen device allocations, several ms). */
int  pqps_ctx_reserve(pqps_ctx *ctx, uint64_t n_rows);
int  pqps_ctx_sync(pqps_ctx *ctx, void *stream);
/* Launch parameters of this context's ID queries (tests, A/B runs inside one process): "list16" 0 / 1 (the list area),
 * "list16_min" / "list16_min_u8" (a step with more matches leaves a 16-bit list), "list_max" (a step with at
 * most this many matches leaves 16-bit entries in its slot; 0 .. 128), "tiny_max" (... in its tiny word; 0 .. 3),
 * "expand_lag" / "sum_lag" (groups),
 * "tune" (bits); value < 0 restores the default. */
int  pqps_ctx_set_option(pqps_ctx *ctx, const char *name, long value);
int  pqps_device_count(void);
int  pqps_ctx_device(pqps_ctx *ctx);
/* Per-launch HIP-event timing of the filter (up to 4096 launches per reset).
 * The scan kernel carries its own begin / end events on the dispatch packet, i.e. the timestamps rocprofv3
 * --kernel-trace reports.  pqps_ctx_kernel_time waits for the recorded launches and returns: *eval_ms = sum of
 * the scan-kernel durations, *total_ms = sum over the whole query, and their number; it then resets the recorder.
 * An ID query is ONE launch (scan tiles + expanders): eval_ms == total_ms; COUNT(*) / flags add the
 * one-workgroup reduction behind the scan.  While timing is on, pqps_qstream_scan / pqps_exchange_select run
 * each query whole on the caller's stream with the context's own scratch (so that the events mean the above):
 * issue them on ONE stream then. */
int  pqps_ctx_set_timing(pqps_ctx *ctx, int enable);
int  pqps_ctx_kernel_time(pqps_ctx *ctx, double *eval_ms, double *total_ms, int *launches);
/* Fills name (<=63 chars), CU count and total HBM bytes of the ctx device. */
int  pqps_device_info(pqps_ctx *ctx, char *name64, int *compute_units, uint64_t *hbm_bytes);

/* Plain device memory helpers so a C host needs no HIP headers. */
int  pqps_malloc(pqps_ctx *ctx, size_t bytes, void **dptr);
int  pqps_free(pqps_ctx *ctx, void *dptr);
/* Pinned host memory that the device addresses as well (zeroed): a kernel's few result words -- a match count --
 * written there are on the host when the query's completion event has fired, no download call (~20 us of host
 * time each) in between. */
int  pqps_malloc_mapped(pqps_ctx *ctx, size_t bytes, void **host_ptr, void **dev_ptr);
int  pqps_free_mapped(pqps_ctx *ctx, void *host_ptr);
int  pqps_memset(pqps_ctx *ctx, void *dptr, int value, size_t bytes, void *stream);
int  pqps_upload(pqps_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes, void *stream);
int  pqps_download(pqps_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes, void *stream);
/* Device-to-device copy between the devices of two contexts (or inside one), asynchronous on `stream` of the
 * DESTINATION context (NULL: its own stream): what the one-process engine gathers its shards' results with
 * (peer DMA over xGMI -- the one-process counterpart of the send / recv pairs of pqps_exchange, which in turn
 * stand for MPI_Allgatherv, engine/mpi/executeEngine-mpi.c:765). */
int  pqps_copy_peer(pqps_ctx *dst_ctx, void *dst, pqps_ctx *src_ctx, const void *src, size_t bytes, void *stream);

/* codes[i] += 1 for every i < n_rows with codes[i] >= threshold (`width` 1, 2 or 4 bytes).
 * Keeps the dictionary codes of a string column order-preserving when INSERT adds a value
 * at rank `threshold`. */
int pqps_bump_codes(pqps_ctx *ctx, void *codes, uint32_t width, uint64_t n_rows, uint32_t threshold, void *stream);

/* Scan mode.  Evaluates `pred` on rows [0, n_rows) of `cols` and writes the
 * matching row IDs (row + id_base, u32) in ASCENDING row order to out_ids and
 * their number to *out_count (device u64).  Asynchronous on `stream`.  One launch.
 * Columns must be readable up to n_rows rounded up to PQPS_STEP_ROWS (1024) rows: the last, partial step is
 * loaded whole and masked.  A launch whose bounded internal waits ran out (never seen with in-order
 * dispatch) sets the context's sticky status word: pqps_ctx_sync then fails with PQPS_EHIP. */
#define PQPS_STEP_ROWS 1024
int pqps_filter_scan(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                     uint64_t n_rows, uint32_t id_base, const pqps_predicate *pred,
                     uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count,
                     void *stream);

/* COUNT(*) only: no ID list, one u64. */
int pqps_filter_count(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred,
                      uint64_t *out_count, void *stream);

/* Per-row byte flags (1 = match), the DELETE / MPI_Allgatherv shape. */
int pqps_filter_flags(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                      uint64_t n_rows, const pqps_predicate *pred,
                      uint8_t *out_flags, uint64_t *out_count, void *stream);

/* Index mode.  Candidates are cand[range[0] .. range[1]) (device u32 row
 * numbers, `range` = 2 device u64 as written by pqps_index_probe); rows that
 * satisfy `pred` are APPENDED, candidate order preserved, at
 * out_ids[*out_count ...], and *out_count (device u64) is advanced -- several
 * probes concatenate without a host round trip (executeEngine-serial.c:444-448). */
int pqps_filter_gather(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols,
                       const uint32_t *cand, const uint64_t *range, uint64_t max_candidates,
                       uint32_t id_base, const pqps_predicate *pred,
                       uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count,
                       void *stream);

/* Builds perm[0..n) = row numbers sorted by (key ascending, row DESCENDING)
 * and sorted_keys[0..n) (same width as the column).  key_kind: 0 = unsigned,
 * 1 = signed i32. */
int pqps_index_build(pqps_ctx *ctx, const pqps_column *col, uint64_t n_rows, int key_kind,
                     uint32_t *perm, void *sorted_keys, void *stream);

/* range[0] = first position with key >= key_lo, range[1] = first position with
 * key > key_hi (inclusive window, findRange semantics); keys as raw 64-bit
 * patterns, compared signed when key_kind == 1. */
int pqps_index_probe(pqps_ctx *ctx, const void *sorted_keys, uint32_t width, int key_kind,
                     uint64_t n_rows, uint64_t key_lo, uint64_t key_hi,
                     uint64_t *range, void *stream);

/* One index probe of a query, whole: pqps_index_probe into `range`, then the probe's rows that satisfy `pred` appended
 * as pqps_filter_gather does (executeEngine-serial.c:358-448: findRange, then the complete WHERE on every row found).
 * `index_column` = the column the index (perm / sorted_keys) was built on.  When `pred` is nothing but the probed
 * comparison itself -- ONE leaf on the indexed column whose window is [key_lo, key_hi], accepted when it holds (the
 * reference's `risk_level > 3` with an index on risk_level) -- every row found passes, and the rows are copied in
 * index order instead of evaluated (PQPS_INDEX_COPY=0, tests: always evaluate; same result). */
int pqps_index_select(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols, const pqps_column *index_column,
                      const uint32_t *perm, const void *sorted_keys, int key_kind, uint64_t n_rows,
                      uint64_t key_lo, uint64_t key_hi, uint32_t id_base, const pqps_predicate *pred,
                      uint64_t *range, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *stream);

/* DELETE on the device (engine/serial/executeEngine-serial.c:646-680 removes the matching rows and
 * keeps the survivors in order): `delete_flags` is what pqps_filter_flags produced (1 = row goes).
 * Every column is compacted in place to the surviving rows, order preserved; *kept_out = survivors.
 * Synchronises the stream.  Dictionary codes stay valid: a code nobody carries any more is harmless. */
int pqps_compact_rows(pqps_ctx *ctx, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                      const uint8_t *delete_flags, uint64_t *kept_out, void *stream);

/* Projection on the device (the gather half of executeEngine-serial.c:504-515): out[i] = the column's
 * value (numeric, or the dictionary code of a string column) of result row ids[i], for the first
 * min(*count_dev, capacity) rows; `out` has the column's width.  Works for any ID order (scan or index
 * mode).  A caller that wants text materialises it from these values only for the rows it shows. */
int pqps_project_column(pqps_ctx *ctx, const pqps_column *col, const uint32_t *ids, const uint64_t *count_dev,
                        uint64_t capacity, uint32_t id_base, void *out, void *stream);

/* Index mode across shards (no counterpart in the reference: its MPI engine replicates the table).
 * A shard's index-mode result is ordered (key asc, row desc) within the shard; rows of a higher rank
 * are higher rows, so the table-wide leaf order of engine/bplus.c:282-358 is the sort of the union by
 * (key ascending, row descending).  pqps_gather_keys produces, for a result list, the order-preserving
 * u64 image of each row's key (count read on the device, e.g. a slot header; signed i32 keys biased);
 * pqps_merge_index_slots takes the all-gathered [count | ids] slots and the parallel key slots
 * (`world` x (slot_stride - PQPS_SLOT_HEADER_WORDS) u64) and writes the merged order.  One probed
 * condition per call; a query with several probed conditions (whose results the serial engine
 * concatenates, duplicates included) merges each condition's segment with its own call.
 * Synchronises the stream (the sort needs the total on the host). */
int pqps_gather_keys(pqps_ctx *ctx, const pqps_column *col, int key_kind, const uint32_t *ids, const uint64_t *count_dev,
                     uint64_t capacity, uint32_t id_base, uint64_t *keys_out, void *stream);
int pqps_merge_index_slots(pqps_ctx *ctx, const uint32_t *slots, const uint64_t *key_slots, uint32_t world,
                           uint64_t slot_stride, uint32_t *merged, uint64_t merged_capacity, uint64_t *totals, void *stream);

/* Tail of the all-gatherv merge.  A slot is what one rank's pqps_filter_scan produced when
 * given out_count = slot and out_ids = slot + PQPS_SLOT_HEADER_WORDS:
 *     [u64 match count][u64 reserved][u32 row IDs ...]
 * `slots` holds `world` such slots, `slot_stride` u32 apart -- what ONE equal-size RCCL
 * all-gather delivers (the count travels with the payload, so MPI_Allgather of the sizes +
 * MPI_Allgatherv of the data, engine/mpi/executeEngine-mpi.c:753-765, become a single
 * collective).  Writes the rank-order concatenation of the ID lists to `merged` and, if
 * `totals` != NULL, totals[0] = IDs merged, totals[1] = sum of the reported counts (larger
 * => a slot overflowed).  Everything stays on the device. */
#define PQPS_SLOT_HEADER_WORDS 4
int pqps_merge_slots(pqps_ctx *ctx, const uint32_t *slots, uint32_t world, uint64_t slot_stride,
                     uint32_t *merged, uint64_t merged_capacity, uint64_t *totals, void *stream);

/* ---- a stream of queries on one GPU ------------------------------------------------------------------
 * An ID query is one launch: scan tiles that read the table at HBM speed, and behind the last of them a tail
 * in which the last tiles drain and the expanders wait for sums and hand out the last IDs -- 7 us (sparse
 * answer) to 25 us (dense) of a 100 M-row query in which most of the chip idles.  pqps_qstream_scan is
 * pqps_filter_scan with TWO queries in flight: each runs whole on one of two HIP streams of the query stream's
 * own (with a scratch of its own), so the dispatcher fills the slots one query's tail leaves free with the
 * next query's scan tiles (the reference's OpenMP driver issues its queries concurrently too,
 * QPEOMP.c:234-291).  `scan_stream` only orders the beginning: the first query after create / sync waits for
 * what that stream holds at that moment.  Results are complete after pqps_qstream_sync() (or a device
 * synchronise); the caller keeps a ring of `depth` out_ids / out_count pairs, the call for query k uses pair
 * k % depth and blocks on the host only until query k - depth has finished.
 * The two streams must sit on different hardware queues of the runtime (two streams that share one run their
 * launches one after the other): they are created at the highest stream priority, both the same -- the runtime
 * keeps a pool of hardware queues per priority, so they get two queues of their own however many streams the
 * rest of the process has created. */
typedef struct pqps_qstream pqps_qstream;
int pqps_qstream_create(pqps_ctx *ctx, uint32_t depth, pqps_qstream **out);
int pqps_qstream_scan(pqps_qstream *q, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                      const pqps_predicate *pred, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count,
                      void *scan_stream);
/* COUNT(*) through the same two lanes: pqps_filter_count with two queries in flight; the caller keeps a ring of
 * `depth` out_count words. */
int pqps_qstream_count(pqps_qstream *q, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                       const pqps_predicate *pred, uint64_t *out_count, void *scan_stream);
int pqps_qstream_sync(pqps_qstream *q);
/* What the last answer looked like (the caller knows once it has a count: the engine at awaitQueryHIP).  While
 * answers hold a quarter of the rows or more, ID queries go to ONE lane: two dense expansions side by side get in each
 * other's way (`risk_level > 1`, 43 % of 100 M rows: 124 us one at a time, 140 - 147 per query with two in flight), a
 * sparse answer's end hides under the next query's scan.  Thread-safe against the issuing calls. */
void pqps_qstream_hint_answer(pqps_qstream *q, uint64_t matches, uint64_t n_rows);
/* Host time (ns) pqps_qstream_scan has spent waiting for an output pair to come free -- as opposed to time inside
 * runtime calls; `reset` != 0 clears the counter. */
uint64_t pqps_qstream_wait_ns(pqps_qstream *q, int reset);
int pqps_qstream_destroy(pqps_qstream *q);

/* The same with the SLOT named by the caller (0 .. depth-1) instead of call number % depth: what an engine with
 * several host threads needs -- a thread takes a free slot, issues, waits for THAT slot and reads its result, while
 * other threads do the same on other slots (the reference's OpenMP driver, QPEOMP.c:234-291).  Issuing calls on one
 * query stream must not overlap (the caller serialises them, e.g. under a mutex: they take microseconds);
 * pqps_qstream_wait may run concurrently with anything.  A slot is reused only after its query has been waited for.
 * Tables of 537 M rows and more use ONE lane (their launches keep expanders among the scan tiles, the tail is a few
 * percent of the launch, and two launches side by side lose more than the overlap gains). */
int pqps_qstream_scan_slot(pqps_qstream *q, uint32_t slot, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows, uint32_t id_base,
                           const pqps_predicate *pred, uint32_t *out_ids, uint64_t out_capacity, uint64_t *out_count, void *scan_stream);
int pqps_qstream_count_slot(pqps_qstream *q, uint32_t slot, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                            const pqps_predicate *pred, uint64_t *out_count, void *scan_stream);
/* Host wait for the query of ONE slot; reports (once) a launch OF THIS SLOT whose bounded waits ran out (the launches are
 * told apart by their epochs: two slots' queries may have run on the same lane). */
int pqps_qstream_wait(pqps_qstream *q, uint32_t slot);
/* A query that is more than one filter call (index probes + gather filters, flag passes in front of the last pass):
 * pqps_qstream_lane hands out the lane context + HIP stream the slot's query is to run on (call the pqps_filter_* /
 * pqps_index_probe functions with them), pqps_qstream_mark records its end on that stream. */
int pqps_qstream_lane(pqps_qstream *q, uint32_t slot, uint64_t n_rows, void *scan_stream, pqps_ctx **lane_ctx, void **lane_stream);
int pqps_qstream_mark(pqps_qstream *q, uint32_t slot);
/* Takes the lanes' scratch (hand-off words, slots, the 2-bytes-per-row list area of ID scans) for tables of up to n_rows rows
 * now rather than inside the first queries. */
int pqps_qstream_reserve(pqps_qstream *q, uint64_t n_rows);
/* Test hook: marks the slot's ID launch as one whose bounded waits ran out, the way the kernel does (its epoch in the lane's
 * status words).  pqps_qstream_wait(slot) must report it -- once -- and no other slot's wait may. */
int pqps_qstream_test_fail_slot(pqps_qstream *q, uint32_t slot);
/* Per-launch timing of the queries AS THEY RUN IN THE STREAM (two in flight): the lanes' own recorders, events on
 * the dispatch packets (recording does not change how the launches overlap).  pqps_qstream_kernel_time = the sums of
 * pqps_ctx_kernel_time over the lanes. */
int pqps_qstream_set_timing(pqps_qstream *q, int enable);
int pqps_qstream_kernel_time(pqps_qstream *q, double *eval_ms, double *total_ms, int *launches);

/* ---- multi-GPU SELECT: shard scan + all-gatherv of the matching row IDs over RCCL, one host call per query ----
 * Replaces the exchange step of engine/mpi/executeEngine-mpi.c:717-768 and keeps its shape: local scan of
 * the rank's row range, MPI_Allgather of the sizes (:753), displacements = exclusive prefix (:758-762),
 * MPI_Allgatherv of the payload (:765).  RCCL has no all-gatherv: the sizes travel in an 8-byte-per-rank
 * ncclAllGather, the payload as ONE group of ncclSend / ncclRecv of exactly count[r] IDs between every pair of
 * ranks, landing at its displacement -- nothing padded on the wire, no compaction pass, and no receive buffer
 * that could be too small (the gathered list is grown to the sizes before the payload moves).
 * WIRE FORM.  An answer of 32 768 matches and more (below that latency, not bytes, is the cost -- PQPS_WIRE_MIN_IDS moves the
 * floor) and of more than ~2 matches per 65 536 rows travels in compact form: the low 16 bits of every row number
 * (relative to the shard's first row) + one u32 per 65 536-row group saying where the group's entries begin -- 2 bytes per
 * match + 4 per group instead of 4 per match; the receiving GPU rebuilds the u32 IDs at the displacement (one copy kernel
 * for all the peers of a query).
 * The sender decides from its own count; the 32-byte-per-rank sizes all-gather carries (reported count, rows, first row,
 * form), so every receiver sizes its receives alike.  PQPS_EXCHANGE_COMPACT=0: always u32 (A/B runs, tests).
 * BOUNDED WAITS.  Every host wait of the exchange (a ring slot, the sizes, a result, pqps_exchange_sync) ends after
 * PQPS_EXCHANGE_TIMEOUT_S seconds (default 30; 0 = unbounded): the communicator is aborted (ncclCommAbort -- which also
 * ends the peers' matching calls), the call returns PQPS_ETIMEOUT and so does every later call on this exchange; the host
 * falls back to another exchange path or tears down.  Nothing of this re-starts a process that holds the GPU.
 * One process per GPU; every rank makes the same calls in the same order.  RCCL is loaded at run time from
 * `rccl_library` (e.g. the librccl.so of the process's torch build, or /opt/rocm/lib/librccl.so); the
 * 128-byte id is produced on rank 0 and handed to the other ranks by whatever bootstrap the host has
 * (torch.distributed broadcast, MPI_Bcast, a file).
 *
 * Bring-up in two steps so that a rank that fails locally cannot leave the others blocked:
 *   pqps_exchange_prepare   everything local (library, stream, buffers, scratch contexts); no communication
 *   (the host's bootstrap agrees that every rank prepared)
 *   pqps_exchange_connect   ncclCommInitRank -- returns once every rank of the world has called it
 * pqps_exchange_create = prepare + connect, for a host that has no such agreement step.
 *
 * pqps_exchange_select(x, ..., slot, scan_stream) enqueues, without blocking on the device:
 *   scan_stream     : the filter launch, writing [count | IDs] into ring slot `slot` (`slot_capacity` IDs: give it
 *                     the shard's row count and it can never be too small)
 *   exchange stream : (behind an event) the all-gather of the sizes and their copy to the host
 * -- after it has finished an EARLIER query: waited on the host for that query's sizes and enqueued its send /
 * recv group.  With a ring of 5 or more slots that is the query three calls back, whose scan ended while the two
 * scans in flight ran: the call does not block and both scan lanes stay supplied (a ring of 4 finishes the query
 * two calls back, a shorter one the previous query -- each step shorter makes the call wait for a scan that is
 * still running).  A slot may be reused after `ring` further calls; reuse waits on the host for the earlier
 * exchange.  pqps_exchange_result() finishes the slot if need be, waits for it and returns
 * the device pointer of the gathered ascending ID list (identical on every rank; valid until the slot is used
 * again), totals[0] = IDs gathered, totals[1] = IDs reported; PQPS_EOVERFLOW if a rank's own slot was too small. */
typedef struct { char internal[128]; } pqps_rccl_id;     /* = ncclUniqueId */
typedef struct pqps_exchange pqps_exchange;
int pqps_exchange_unique_id(const char *rccl_library, pqps_rccl_id *id);
int pqps_exchange_prepare(pqps_ctx *ctx, const char *rccl_library, uint32_t world, uint32_t rank,
                          uint64_t slot_capacity, uint32_t ring, pqps_exchange **out);
int pqps_exchange_connect(pqps_exchange *x, const pqps_rccl_id *id);
int pqps_exchange_create(pqps_ctx *ctx, const char *rccl_library, const pqps_rccl_id *id, uint32_t world,
                         uint32_t rank, uint64_t slot_capacity, uint32_t ring, pqps_exchange **out);
int pqps_exchange_select(pqps_exchange *x, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                         uint32_t id_base, const pqps_predicate *pred, uint32_t slot, void *scan_stream);
/* COUNT(*) across the shards: local count kernel + ncclAllReduce(sum, 1 x u64)
 * (engine/mpi/executeEngine-mpi.c:745).  pqps_exchange_result() then reports totals[0] = the global
 * count, *local_count = this rank's; the merged pointer is meaningless for such a slot. */
int pqps_exchange_count(pqps_exchange *x, const pqps_column *cols, uint32_t n_cols, uint64_t n_rows,
                        const pqps_predicate *pred, uint32_t slot, void *scan_stream);
int pqps_exchange_result(pqps_exchange *x, uint32_t slot, const uint32_t **merged_dev, uint64_t *local_count,
                         uint64_t totals[2]);
/* Finishes every query handed in so far (enqueues the payload groups still held back) and waits for the
 * exchange stream: call it before stopping a clock or tearing down. */
int pqps_exchange_sync(pqps_exchange *x);
/* Host time (ns) spent waiting -- for a ring slot to come free or for the sizes of a query -- as opposed to
 * time inside runtime / RCCL calls; `reset` != 0 clears the counter. */
uint64_t pqps_exchange_wait_ns(pqps_exchange *x, int reset);
/* The compact wire form by itself, for a host that moves the payload with its own collectives (merge.py over
 * torch.distributed): pqps_wire_pack reads a slot as the filter left it ([u64 count][u64][u32 IDs ...], `capacity` IDs), writes
 * this rank's four header words (reported count, rows, first row, form: 1 = compact) to header_dev and -- if `enabled` and
 * the compact form pays (pqps_wire_pays: >= 32 768 IDs and fewer bytes) -- the payload [u32 goff[groups + 1], padded to 16 bytes][u16 low[n]] to
 * `wire` (room for pqps_wire_bytes(n_rows, min(capacity, n_rows))); pqps_wire_expand turns a received payload into u32 IDs
 * at out_ids (the list's displacement in the gathered list). */
uint64_t pqps_wire_bytes(uint64_t n_rows, uint64_t n_ids);
int pqps_wire_pays(uint64_t n_rows, uint64_t n_ids);
int pqps_wire_pack(pqps_ctx *ctx, const uint32_t *slot, uint64_t capacity, uint64_t n_rows, uint32_t id_base, int enabled,
                   uint64_t *header_dev, void *wire, void *stream);
int pqps_wire_expand(pqps_ctx *ctx, const void *wire, uint64_t n_rows, uint32_t id_base, uint32_t *out_ids, void *stream);
/* Payload bytes this rank has RECEIVED from its peers so far: out[0] as they travelled (compact or u32), out[1] what the
 * same lists are as u32 IDs; `reset` != 0 clears both. */
void pqps_exchange_wire_bytes(pqps_exchange *x, uint64_t out[2], int reset);
/* SMALL ANSWERS IN ONE COLLECTIVE.  Behind each rank's 32-byte header the sizes all-gather of a SELECT has room for
 * PQPS_EXCHANGE_EAGER_IDS row numbers (default 16 384 = 64 KB per rank; 0: none; the smallest setting and the smallest slot of
 * the world win, agreed at connect; at most 4 MB gathered per query): a rank whose list fits puts it there, and when EVERY
 * rank's list fits the answer is complete after that one all-gather and one copy kernel -- no send / recv group, no size on
 * the host first.  When a list does not fit the query takes the two steps described above, and the next SELECT gathers bare
 * headers again until an answer that would have fitted has been seen (every rank reads the same sizes at the same point of its
 * call sequence, so all take the same turn).  out[0] = SELECTs that finished in the one collective, out[1] = SELECTs
 * finished, out[2] = the room agreed on (IDs per rank; 0 = off). */
void pqps_exchange_eager(pqps_exchange *x, uint64_t out[3], int reset);
int pqps_exchange_destroy(pqps_exchange *x);

/* Checksums of a device-resident ID list: out[0] = sum of ids[i], out[1] = sum of ids[i] * (2 i + 1), both mod 2^64 (the
 * second depends on the order).  Synchronous; what a bench or a test compares two lists with without downloading them. */
int pqps_ids_checksum(pqps_ctx *ctx, const uint32_t *ids, uint64_t count, uint64_t out[2], void *stream);

/* Row-range block partition of engine/mpi/executeEngine-mpi.c:703-715. */
void pqps_partition(uint64_t n_rows, int world, int rank, uint64_t *start, uint64_t *count);

/* Seeded on-device generator of the commands_* schema (SURVEY.md App. B /
 * generate_commands.py distributions); rows [row0, row0+n) of the global
 * table.  Any output pointer may be NULL.  `user_cdf` = 2000 u32 thresholds,
 * `user_shell` = 2000 u8 (both device), built by pqps_synth_user_tables. */
typedef struct pqps_synth_cols {
    uint64_t *command_id;
    int32_t  *exit_code;
    int32_t  *user_id;
    int32_t  *risk_level;
    uint8_t  *sudo_used;
    uint8_t  *shell_code;     /* rank in {"bash","fish","sh","zsh"}        */
    uint16_t *user_code;      /* rank of "student<id>" == user_id - 1000   */
    uint8_t  *host_code;      /* rank among the 16 host names              */
    uint8_t  *base_code;      /* rank among 111 base commands (uniform)    */
} pqps_synth_cols;

#define PQPS_SYNTH_USERS 2000
void pqps_synth_user_tables(uint64_t seed, uint32_t *cdf_host, uint8_t *shell_host);
int  pqps_synth_generate(pqps_ctx *ctx, uint64_t seed, uint64_t row0, uint64_t n,
                         const uint32_t *user_cdf_dev, const uint8_t *user_shell_dev,
                         const pqps_synth_cols *out, void *stream);
/* CPU twin of the generator (same bits), used by tests and the CPU baseline. */
void pqps_synth_generate_host(uint64_t seed, uint64_t row0, uint64_t n,
                              const uint32_t *user_cdf, const uint8_t *user_shell,
                              const pqps_synth_cols *out);

#ifdef __cplusplus
}
#endif
#endif /* PQPS_HIP_H */
