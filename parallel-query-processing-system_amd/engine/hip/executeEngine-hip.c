/* executeEngine-hip.c -- MI355X execute engine (see include/executeEngine-hip.h).
 *
 * Host orchestration in C11; every row predicate runs in the HIP filter
 * kernel behind include/pqps_hip.h.  There is no CPU evaluation path: if the
 * device or the shim is unavailable the engine prints the reason and exits.
 *
 * Reference being replaced: engine/serial/executeEngine-serial.c ("S"):
 *   executeQuerySelectSerial S:328-528, linearSearchRecords S:854-878,
 *   evaluateWhereClause S:292-316, executeQueryInsertSerial S:538-617,
 *   executeQueryDeleteSerial S:627-715, initialize/destroy S:727-814.
 */
#define _POSIX_C_SOURCE 200809L
#include "executeEngine-hip.h"
#include "buildEngine-hip.h"
#include "hipPredicate.h"

#include <limits.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <time.h>
#include <unistd.h>

static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* PQPS_TRACE=1: phase timings of every engine call on stderr. */
static int trace_on(void) {
    static int on = -1;
    if (on < 0) { const char *e = getenv("PQPS_TRACE"); on = e && atoi(e) != 0; }
    return on;
}
#define TRACE(...) do { if (trace_on()) fprintf(stderr, "[pqps] " __VA_ARGS__); } while (0)

/* A device call failed inside a query: the reason goes to stderr (like every diagnostic of the reference's
 * engines) and the query reports failure to its caller -- the process, and the engine, carry on. */
static int engine_error(const char *what) {
    fprintf(stderr, "HIP engine: %s: %s\n", what, pqps_last_error());
    return -1;
}

#define TRY(call, what) do { if ((call) != PQPS_OK) return engine_error(what); } while (0)

/* ---- predicate binding --------------------------------------------------------- */

/* The compiled WHERE of one query: usually one pass; see hipCompileWherePlan. */
static int bind_where(const struct hipTable *t, const struct whereClauseS *where, struct hipPlan *plan) {
    struct hipSchema schema;
    char err[160];
    hipSchemaOfTable(t, &schema);
    if (hipCompileWherePlan(&schema, where, plan, err, sizeof err) != 0) {
        fprintf(stderr, "HIP engine: cannot compile WHERE clause: %s\n", err);
        return -1;
    }
    return 0;
}

/* One shard's side of a plan: the flag buffers of the passes before the last, and the last pass's columns. */
struct shard_pred {
    const pqps_predicate *pred;
    pqps_column cols[PQPS_MAX_COLUMNS];
    uint32_t n_cols;
    uint8_t **flags;                 /* n_passes - 1 device buffers (NULL for a single pass) */
    int n_flags;
};

static void shard_pred_free(struct hipTable *sh, struct shard_pred *sp) {
    for (int k = 0; k < sp->n_flags; k++) if (sp->flags[k]) pqps_free(sh->ctx, sp->flags[k]);
    free(sp->flags);
    sp->flags = NULL;
    sp->n_flags = 0;
}

static void pass_columns(const struct hipTable *sh, const struct hipPass *pass, uint8_t *const *flags, pqps_column *cols) {
    for (uint32_t i = 0; i < pass->pred.n_columns; i++) {
        const int id = pass->column_ids[i];
        if (id >= PQPS_MAX_COLUMNS) { cols[i].data = flags[id - PQPS_MAX_COLUMNS]; cols[i].width = 1; }
        else cols[i] = sh->col[id];
    }
}

/* Inclusive key window of an indexed top-level condition, S:377-424.
 * v + 1 / v - 1 wrap the way the reference's machine arithmetic does. */
static void key_window_u64(const char *op, const char *value, uint64_t *lo, uint64_t *hi) {
    const uint64_t v = strtoull(value, NULL, 10);
    *lo = 0; *hi = UINT64_MAX;
    if (strcmp(op, "=") == 0) { *lo = v; *hi = v; }
    else if (strcmp(op, ">") == 0) *lo = v + 1;
    else if (strcmp(op, ">=") == 0) *lo = v;
    else if (strcmp(op, "<") == 0) *hi = v - 1;
    else if (strcmp(op, "<=") == 0) *hi = v;
}

static void key_window_i32(const char *op, const char *value, uint64_t *lo, uint64_t *hi) {
    const int v = atoi(value);
    int l = INT_MIN, h = INT_MAX;
    if (strcmp(op, "=") == 0) { l = v; h = v; }
    else if (strcmp(op, ">") == 0) l = (int)((unsigned)v + 1u);
    else if (strcmp(op, ">=") == 0) l = v;
    else if (strcmp(op, "<") == 0) h = (int)((unsigned)v - 1u);
    else if (strcmp(op, "<=") == 0) h = v;
    *lo = (uint64_t)(uint32_t)l;
    *hi = (uint64_t)(uint32_t)h;
}

/* The keys a condition on a BOOL index admits in the OpenMP / MPI engines (omp:424-459): = / != one key, the ordered
 * operators what they admit; `> true` and `< false` admit none (lo > hi: the probe finds nothing, but it counts as a
 * probe -- the query stays in index mode and comes back empty). */
static void key_window_bool(const char *op, const char *value, uint64_t *lo, uint64_t *hi) {
    const int val = (strcasecmp(value, "true") == 0 || strcmp(value, "1") == 0) ? 1 : 0;
    int l = 0, h = 1;
    if (strcmp(op, "=") == 0) { l = val; h = val; }
    else if (strcmp(op, "!=") == 0) { l = !val; h = !val; }
    else if (strcmp(op, ">") == 0) { l = 1; h = val ? 0 : 1; }
    else if (strcmp(op, ">=") == 0) { l = val; h = 1; }
    else if (strcmp(op, "<") == 0) { l = val ? 0 : 1; h = 0; }
    else if (strcmp(op, "<=") == 0) { l = 0; h = val; }
    *lo = (uint64_t)l;
    *hi = (uint64_t)h;
}

/* One index probe the serial engine would make for this WHERE (S:358-433), in its order (with probe_bool: the
 * OpenMP / MPI engines' one-thread order, omp:362-494 -- the same walk with BOOL indexes included). */
struct probe { int index; uint64_t lo, hi; };

static int list_probes(struct engineS *engine, const struct hipTable *t, struct whereClauseS *where, struct probe **out) {
    int n = 0, cap = 0;
    struct probe *pr = NULL;
    for (struct whereClauseS *wc = where; wc; wc = wc->next) {
        if (wc->attribute == NULL) continue;                           /* nested node, S:361-364 */
        for (int i = 0; i < engine->num_indexes; i++) {
            if (strcmp(wc->attribute, engine->indexed_attributes[i]) != 0) continue;
            const struct hipIndex *ix = &t->index[i];
            if (ix->column < 0 || wc->operator == NULL || wc->value == NULL) continue;
            uint64_t lo, hi;
            /* only u64 / int indexes are probed by the serial engine, S:377-433 */
            if (engine->attribute_types[i] == FIELD_UINT64 && t->col[ix->column].width == 8 && ix->key_kind == 0)
                key_window_u64(wc->operator, wc->value, &lo, &hi);
            else if (engine->attribute_types[i] == FIELD_INT && ix->key_kind == 1)
                key_window_i32(wc->operator, wc->value, &lo, &hi);
            else if (t->probe_bool && engine->attribute_types[i] == FIELD_BOOL && t->col[ix->column].width == 1 && ix->key_kind == 0)
                key_window_bool(wc->operator, wc->value, &lo, &hi);
            else
                continue;
            if (n == cap) {
                cap = cap ? 2 * cap : 8;
                struct probe *grown = realloc(pr, (size_t)cap * sizeof *grown);
                if (!grown) { free(pr); return -1; }
                pr = grown;
            }
            pr[n++] = (struct probe){ i, lo, hi };
        }
    }
    *out = pr;
    return n;
}

/* ---- one query on the device(s) -------------------------------------------------------------------------
 * Row selection of executeQuerySelectSerial, S:358-474.  A query is ISSUED -- everything it needs is enqueued on
 * every shard, nothing is waited for -- and later AWAITED.  Engine tables give it a lane: result buffers of its own
 * on every shard and a slot of the shards' query streams (so several queries are on the device at once); a table
 * without lanes (ad-hoc tables over caller-supplied rows) runs it on the table's own context and buffers.
 *
 * Result.  Scan mode: shard s holds count[s] ascending engine row numbers; the answer is their concatenation in
 * shard order.  Index mode: per probe, each shard appended its rows in (key asc, row desc) order; several shards
 * are merged by key on shard 0's device.  `ids_dev` = the whole answer on shard 0's device (one shard: its lane's
 * buffer as it stands; several: the gathered list). */
struct query {
    struct engineS *engine;
    struct hipTable *t;
    int lane;                            /* -1: the table's own buffers */
    int n_shards;
    bool count_only;
    struct hipPlan plan;
    bool have_plan;
    struct probe *probes;
    int n_probes;
    struct shard_pred sp[HIP_MAX_SHARDS];
    uint64_t *seg_dev[HIP_MAX_SHARDS];   /* index mode, several shards: running count after each probe */
    /* after await */
    uint64_t total;
    uint64_t count[HIP_MAX_SHARDS];
    const uint32_t *ids_dev;
    pqps_ctx *ids_ctx;                   /* a context of the device ids_dev lives on (downloads) */
    bool exchanged;                      /* issued through the ranks' exchange: the answer is the exchange slot's */
};

static struct hipLane *query_lane(const struct query *q, int s) {
    struct hipTable *sh = hipTableShard(q->t, s);
    return q->lane >= 0 ? &sh->lane[q->lane] : &sh->own;
}

/* context for the lane's own copies / gathers (never the stream the scans run on) */
static pqps_ctx *lane_copy_ctx(const struct query *q, int s) {
    struct hipLane *L = query_lane(q, s);
    return L->copy ? L->copy : hipTableShard(q->t, s)->ctx;
}

static void query_init(struct query *q, struct engineS *engine, struct hipTable *t, int lane, bool count_only) {
    memset(q, 0, sizeof *q);
    q->engine = engine;
    q->t = t;
    q->lane = lane;
    q->n_shards = hipTableShards(t);
    q->count_only = count_only;
}

static void query_free(struct query *q) {
    for (int s = 0; s < q->n_shards; s++) {
        struct hipTable *sh = hipTableShard(q->t, s);
        shard_pred_free(sh, &q->sp[s]);
        if (q->seg_dev[s]) { pqps_free(sh->ctx, q->seg_dev[s]); q->seg_dev[s] = NULL; }
    }
    free(q->probes);
    q->probes = NULL;
    if (q->have_plan) { hipPlanFree(&q->plan); q->have_plan = false; }
}

/* The filter calls of the query on shard s, on (ctx, stream): flag passes, then the count, the probes + gather
 * filters, or the scan. */
static int issue_calls(struct query *q, int s, pqps_ctx *ctx, void *stream) {
    struct hipTable *sh = hipTableShard(q->t, s);
    struct hipLane *L = query_lane(q, s);
    struct shard_pred *sp = &q->sp[s];
    const struct hipPlan *plan = &q->plan;
    const struct hipPass *last = &plan->pass[plan->n_passes - 1];
    /* the passes before the last: one byte of flags per row each */
    if (plan->n_passes > 1) {
        sp->flags = calloc((size_t)plan->n_passes - 1, sizeof *sp->flags);
        if (!sp->flags) { fprintf(stderr, "HIP engine: out of memory\n"); return -1; }
        sp->n_flags = plan->n_passes - 1;
        for (int k = 0; k < sp->n_flags; k++) {
            pqps_column cols[PQPS_MAX_COLUMNS];
            TRY(pqps_malloc(sh->ctx, sh->capacity_rows, (void **)&sp->flags[k]), "flag allocation");
            pass_columns(sh, &plan->pass[k], sp->flags, cols);
            TRY(pqps_filter_flags(ctx, cols, plan->pass[k].pred.n_columns, sh->n_rows, &plan->pass[k].pred,
                                  sp->flags[k], L->count_dev + 4, stream), "flag filter");
        }
    }
    sp->pred = &last->pred;
    sp->n_cols = last->pred.n_columns;
    pass_columns(sh, last, sp->flags, sp->cols);
    if (q->count_only) {
        TRY(pqps_filter_count(ctx, sp->cols, sp->n_cols, sh->n_rows, sp->pred, L->count_dev, stream), "count filter");
    } else if (q->n_probes > 0) {
        TRY(pqps_memset(ctx, L->count_dev, 0, sizeof(uint64_t), stream), "counter reset");
        if (q->n_shards > 1 && !q->seg_dev[s]) TRY(pqps_malloc(sh->ctx, (size_t)q->n_probes * sizeof(uint64_t), (void **)&q->seg_dev[s]), "segment counters");
        for (int k = 0; k < q->n_probes; k++) {
            const struct hipIndex *ix = &sh->index[q->probes[k].index];
            uint64_t *range_dev = L->count_dev + 2;
            /* the probe, then its rows that pass the complete WHERE appended, leaf order kept (S:441-448 + S:471) -- copied
             * when the WHERE is the probed comparison itself (the shim looks at the compiled predicate) */
            TRY(pqps_index_select(ctx, sp->cols, sp->n_cols, &sh->col[ix->column], ix->perm_dev, ix->keys_dev, ix->key_kind, sh->n_rows,
                                  q->probes[k].lo, q->probes[k].hi, (uint32_t)sh->row0, sp->pred, range_dev,
                                  L->ids_dev, L->capacity_ids, L->count_dev, stream), "index probe + filter");
            if (q->seg_dev[s])                                     /* where this probe's rows end on this shard */
                TRY(pqps_copy_peer(ctx, q->seg_dev[s] + k, ctx, L->count_dev, sizeof(uint64_t), stream), "segment counter");
        }
    } else {
        TRY(pqps_filter_scan(ctx, sp->cols, sp->n_cols, sh->n_rows, (uint32_t)sh->row0, sp->pred, L->ids_dev, L->capacity_ids,
                             L->count_dev, stream), "scan filter");
    }
    return 0;
}

/* Everything of the query that runs on shard s.  A single-pass scan or count on a lane is ONE call into the shard's
 * query stream, whose completion event rides on the launch itself. */
static int issue_on_shard(struct query *q, int s) {
    struct hipTable *sh = hipTableShard(q->t, s);
    struct hipLane *L = query_lane(q, s);
    struct shard_pred *sp = &q->sp[s];
    memset(sp, 0, sizeof *sp);
    const struct hipPlan *plan = &q->plan;
    const struct hipPass *last = &plan->pass[plan->n_passes - 1];
    const bool single = plan->n_passes == 1 && q->n_probes == 0;
    if (q->t->xch) {
        /* one process per GPU: the shard's scan + the exchange with the other ranks, one call (mpi:717-768).  Every rank has
         * to issue the same queries in the same order; index probes and WHERE lists of several passes stay local matters */
        if (!single || q->lane < 0) {
            fprintf(stderr, "HIP engine: across ranks only scan-mode queries of one pass are exchanged (no index probes, at most %d comparisons)\n", PQPS_MAX_LEAVES);
            return -1;
        }
        sp->pred = &last->pred;
        sp->n_cols = last->pred.n_columns;
        pass_columns(sh, last, NULL, sp->cols);
        if (q->count_only)
            TRY(pqps_exchange_count(q->t->xch, sp->cols, sp->n_cols, sh->n_rows, sp->pred, (uint32_t)q->lane, NULL), "count exchange");
        else
            TRY(pqps_exchange_select(q->t->xch, sp->cols, sp->n_cols, sh->n_rows, (uint32_t)sh->row0, sp->pred, (uint32_t)q->lane, NULL), "scan exchange");
        q->exchanged = true;
        return 0;
    }
    if (q->lane >= 0 && single) {
        sp->pred = &last->pred;
        sp->n_cols = last->pred.n_columns;
        pass_columns(sh, last, NULL, sp->cols);
        if (q->count_only)
            TRY(pqps_qstream_count_slot(sh->qs, (uint32_t)q->lane, sp->cols, sp->n_cols, sh->n_rows, sp->pred, L->count_dev, NULL), "count filter");
        else
            TRY(pqps_qstream_scan_slot(sh->qs, (uint32_t)q->lane, sp->cols, sp->n_cols, sh->n_rows, (uint32_t)sh->row0, sp->pred,
                                       L->ids_dev, L->capacity_ids, L->count_dev, NULL), "scan filter");
        return 0;
    }
    pqps_ctx *ctx = sh->ctx;
    void *stream = NULL;
    if (q->lane >= 0) TRY(pqps_qstream_lane(sh->qs, (uint32_t)q->lane, sh->n_rows, NULL, &ctx, &stream), "query lane");
    const int rc = issue_calls(q, s, ctx, stream);
    /* the end of whatever reached the lane's stream is marked even if a call failed half-way: the lane's buffers must
     * not be handed to another query while a launch of this one still runs */
    if (q->lane >= 0 && pqps_qstream_mark(sh->qs, (uint32_t)q->lane) != PQPS_OK && rc == 0) return engine_error("query lane");
    return rc;
}

static int query_issue_all(struct query *q) {
    int rc = 0;
    hipTableLockIssue(q->t);
    for (int s = 0; s < q->n_shards && rc == 0; s++) {
        shard_pred_free(hipTableShard(q->t, s), &q->sp[s]);            /* (a re-issue after a result buffer had to grow) */
        rc = issue_on_shard(q, s);
    }
    hipTableUnlockIssue(q->t);
    return rc;
}

/* Compiles the WHERE, lists the probes, enqueues the query on every shard. */
static int query_issue(struct query *q, struct whereClauseS *where) {
    if (bind_where(q->t, where, &q->plan) != 0) return -1;
    q->have_plan = true;
    if (!q->count_only) {
        q->n_probes = list_probes(q->engine, q->t, where, &q->probes);     /* COUNT(*) is the scan-mode count */
        if (q->n_probes < 0) { q->n_probes = 0; return -1; }
    }
    return query_issue_all(q);
}

static int wait_shard(struct query *q, int s) {
    struct hipTable *sh = hipTableShard(q->t, s);
    if (q->exchanged || (q->t->xch && q->lane >= 0)) return 0;      /* (the exchange's own slot waits cover it) */
    if (q->lane >= 0) TRY(pqps_qstream_wait(sh->qs, (uint32_t)q->lane), "filter execution");
    else TRY(pqps_ctx_sync(sh->ctx, NULL), "filter execution");
    return 0;
}

static int grow_lane_ids(struct hipTable *sh, struct hipLane *L, uint64_t need) {
    if (need <= L->capacity_ids) return 0;
    pqps_free(sh->ctx, L->ids_dev);
    L->ids_dev = NULL;
    L->capacity_ids = 0;
    const uint64_t cap = need + need / 8 + 1024;
    TRY(pqps_malloc(sh->ctx, cap * sizeof(uint32_t), (void **)&L->ids_dev), "result allocation");
    L->capacity_ids = cap;
    return 0;
}

/* Several shards: the answer in one piece on shard 0's device.  Scan mode = the concatenation of the shards' lists in
 * shard order, placed by peer copies at the displacements the counts give (MPI_Allgather of the sizes, exclusive prefix,
 * MPI_Allgatherv: mpi:753-765).  Index mode = per probe the union of the shards' segments sorted by (key asc, row
 * desc) -- every segment is in that order already and a later shard holds later rows, so this is the table-wide
 * leaf order of the reference's tree: keys gathered on each shard, segments and keys copied over, merged by the
 * device sort behind pqps_merge_index_slots. */
static int gather_shards(struct query *q) {
    struct hipTable *t = q->t;
    struct hipLane *L0 = query_lane(q, 0);
    pqps_ctx *c0 = lane_copy_ctx(q, 0);
    if (q->total > L0->merged_cap) {
        if (L0->merged_dev) pqps_free(t->ctx, L0->merged_dev);
        L0->merged_dev = NULL;
        L0->merged_cap = 0;
        const uint64_t cap = q->total + q->total / 8 + 1024;
        TRY(pqps_malloc(t->ctx, cap * sizeof(uint32_t), (void **)&L0->merged_dev), "gathered list allocation");
        L0->merged_cap = cap;
    }
    q->ids_dev = L0->merged_dev;
    q->ids_ctx = c0;
    if (q->total == 0) return 0;
    if (q->n_probes == 0) {
        uint64_t at = 0;
        for (int s = 0; s < q->n_shards; s++) {
            TRY(pqps_copy_peer(c0, L0->merged_dev + at, lane_copy_ctx(q, s), query_lane(q, s)->ids_dev, q->count[s] * sizeof(uint32_t), NULL), "peer copy");
            at += q->count[s];
        }
        TRY(pqps_ctx_sync(c0, NULL), "peer copy");
        return 0;
    }
    /* index mode */
    uint64_t *ends = malloc((size_t)q->n_probes * (size_t)q->n_shards * sizeof *ends);       /* [shard][probe] */
    if (!ends) return -1;
    int rc = 0;
#define RUN(call, what) do { if (rc == 0 && (call) != PQPS_OK) rc = engine_error(what); } while (0)
    for (int s = 0; s < q->n_shards; s++)
        RUN(pqps_download(lane_copy_ctx(q, s), ends + (size_t)s * (size_t)q->n_probes, q->seg_dev[s], (size_t)q->n_probes * sizeof *ends, NULL), "segment counters");
    uint64_t out_at = 0;
    for (int k = 0; k < q->n_probes && rc == 0; k++) {
        uint64_t seg_total = 0, seg_len[HIP_MAX_SHARDS], seg_begin[HIP_MAX_SHARDS];
        for (int s = 0; s < q->n_shards; s++) {
            const uint64_t *e = ends + (size_t)s * (size_t)q->n_probes;
            seg_begin[s] = k ? e[k - 1] : 0;
            seg_len[s] = e[k] - seg_begin[s];
            seg_total += seg_len[s];
        }
        if (seg_total == 0) continue;
        const uint64_t stride = ((seg_total + 1) & ~1ull) + PQPS_SLOT_HEADER_WORDS;
        uint32_t *slots = NULL;
        uint64_t *keys = NULL, *totals_dev = NULL;
        RUN(pqps_malloc(t->ctx, stride * sizeof(uint32_t), (void **)&slots), "merge buffer");
        RUN(pqps_malloc(t->ctx, (stride - PQPS_SLOT_HEADER_WORDS) * sizeof(uint64_t), (void **)&keys), "merge buffer");
        RUN(pqps_malloc(t->ctx, 2 * sizeof(uint64_t), (void **)&totals_dev), "merge buffer");
        const uint64_t header[2] = { seg_total, 0 };
        RUN(pqps_upload(c0, slots, header, sizeof header, NULL), "merge header");
        uint64_t at = 0;
        for (int s = 0; s < q->n_shards && rc == 0; s++) {
            if (seg_len[s] == 0) continue;
            struct hipTable *sh = hipTableShard(t, s);
            struct hipLane *L = query_lane(q, s);
            pqps_ctx *cs = lane_copy_ctx(q, s);
            const struct hipIndex *ix = &sh->index[q->probes[k].index];
            uint64_t *keys_s = NULL;
            RUN(pqps_malloc(sh->ctx, seg_len[s] * sizeof(uint64_t), (void **)&keys_s), "key buffer");
            /* (the count word on the device is the shard's total >= the segment's length: the capacity argument bounds the gather) */
            RUN(pqps_gather_keys(cs, &sh->col[ix->column], ix->key_kind, L->ids_dev + seg_begin[s], L->count_dev, seg_len[s], (uint32_t)sh->row0, keys_s, NULL), "key gather");
            RUN(pqps_ctx_sync(cs, NULL), "key gather");
            RUN(pqps_copy_peer(c0, slots + PQPS_SLOT_HEADER_WORDS + at, cs, L->ids_dev + seg_begin[s], seg_len[s] * sizeof(uint32_t), NULL), "peer copy");
            RUN(pqps_copy_peer(c0, keys + at, cs, keys_s, seg_len[s] * sizeof(uint64_t), NULL), "peer copy");
            RUN(pqps_ctx_sync(c0, NULL), "peer copy");
            if (keys_s) pqps_free(sh->ctx, keys_s);
            at += seg_len[s];
        }
        RUN(pqps_merge_index_slots(c0, slots, keys, 1, stride, L0->merged_dev + out_at, L0->merged_cap - out_at, totals_dev, NULL), "index merge");
        out_at += seg_total;
        if (slots) pqps_free(t->ctx, slots);
        if (keys) pqps_free(t->ctx, keys);
        if (totals_dev) pqps_free(t->ctx, totals_dev);
    }
#undef RUN
    free(ends);
    return rc;
}

/* Waits for the query; a result buffer that turned out too small (index-mode duplicates can exceed the table's rows)
 * is grown and the query issued again. */
static int query_await(struct query *q) {
    if (q->exchanged) {
        /* every rank holds the whole answer: the shards' lists gathered in rank order (= ascending row numbers) */
        const uint32_t *merged = NULL;
        uint64_t local = 0, totals[2] = { 0, 0 };
        TRY(pqps_exchange_result(q->t->xch, (uint32_t)q->lane, q->count_only ? NULL : &merged, &local, totals), "exchange result");
        q->total = totals[0];
        q->count[0] = local;
        q->ids_dev = merged;
        q->ids_ctx = lane_copy_ctx(q, 0);
        return 0;
    }
    for (;;) {
        bool again = false;
        int rc = 0;
        q->total = 0;
        for (int s = 0; s < q->n_shards; s++) {
            struct hipTable *sh = hipTableShard(q->t, s);
            struct hipLane *L = query_lane(q, s);
            if (rc == 0) rc = wait_shard(q, s);
            if (rc == 0 && L->count_host) q->count[s] = L->count_host[0];         /* written by the launch itself, on the host with its completion event */
            else if (rc == 0 && pqps_download(lane_copy_ctx(q, s), &q->count[s], L->count_dev, sizeof(uint64_t), NULL) != PQPS_OK) rc = engine_error("count download");
            if (rc == 0 && !q->count_only && q->count[s] > L->capacity_ids) {
                rc = grow_lane_ids(sh, L, q->count[s]);
                again = true;
            }
            q->total += q->count[s];
            if (rc == 0 && !q->count_only && q->n_probes == 0 && q->lane >= 0 && sh->qs) pqps_qstream_hint_answer(sh->qs, q->count[s], sh->n_rows);
        }
        if (rc != 0) return rc;
        if (!again) break;
        if (query_issue_all(q) != 0) return -1;
    }
    if (q->count_only) return 0;
    if (q->n_shards == 1) {
        q->ids_dev = query_lane(q, 0)->ids_dev;
        q->ids_ctx = lane_copy_ctx(q, 0);
        return 0;
    }
    return gather_shards(q);
}

/* ---- asynchronous tickets (include/executeEngine-hip.h) ------------------------------------------------ */

struct hipQueryTicket {
    struct query q;
    double t0;
    int state;                       /* 0 issued, 1 awaited, -1 failed */
};

static struct hipQueryTicket *ticket_begin(struct engineS *engine, struct whereClauseS *where, bool count_only) {
    if (!engine || !engine->record_block) return NULL;
    struct hipTable *t = engine->record_block;
    struct hipQueryTicket *tk = calloc(1, sizeof *tk);
    if (!tk) { fprintf(stderr, "HIP engine: out of memory\n"); return NULL; }
    tk->t0 = now_seconds();
    hipTableLockShared(t);                                            /* until releaseQueryHIP: no writer meanwhile */
    const int lane = hipTableAcquireLane(t);
    if (lane == HIP_LANE_REFUSED) {                                   /* reason on stderr: the caller holds every lane itself, or none came free in time */
        hipTableUnlockShared(t);
        free(tk);
        return NULL;
    }
    query_init(&tk->q, engine, t, lane, count_only);
    if (query_issue(&tk->q, where) != 0) tk->state = -1;            /* reason on stderr; awaitQueryHIP reports -1 */
    return tk;
}

struct hipQueryTicket *executeQuerySelectAsyncHIP(struct engineS *engine, struct whereClauseS *whereClause) {
    return ticket_begin(engine, whereClause, false);
}

struct hipQueryTicket *executeQueryCountAsyncHIP(struct engineS *engine, struct whereClauseS *whereClause) {
    return ticket_begin(engine, whereClause, true);
}

long long awaitQueryHIP(struct hipQueryTicket *tk, struct hipDeviceResult *result) {
    if (result) memset(result, 0, sizeof *result);
    if (!tk) return -1;
    if (tk->state == 0) tk->state = query_await(&tk->q) == 0 ? 1 : -1;
    if (tk->state < 0) { if (result) result->count = -1; return -1; }
    if (result) {
        result->count = (long long)tk->q.total;
        result->ids_dev = tk->q.count_only ? NULL : tk->q.ids_dev;
        result->device = pqps_ctx_device(tk->q.t->ctx);
        result->n_shards = tk->q.n_shards;
        for (int s = 0; s < tk->q.n_shards && s < 16; s++) result->shard_count[s] = tk->q.count[s];
    }
    return (long long)tk->q.total;
}

/* Checksums of the answer's ID list where it lies, on the device (pqps_ids_checksum): what a driver compares the list of
 * a timed query with -- against another path's list, or against numpy over the oracle's -- without moving it. */
int hipQueryChecksumHIP(struct hipQueryTicket *tk, unsigned long long out[2]) {
    if (!tk || !out) return -1;
    const long long count = awaitQueryHIP(tk, NULL);
    if (count < 0 || tk->q.count_only) return -1;
    uint64_t sums[2] = { 0, 0 };
    if (count > 0 && pqps_ids_checksum(tk->q.ids_ctx, tk->q.ids_dev, (uint64_t)count, sums, NULL) != PQPS_OK) return engine_error("ID checksum");
    out[0] = sums[0];
    out[1] = sums[1];
    return 0;
}

/* ---- one process per GPU ----------------------------------------------------------------------------------- */
static struct engineS *engine_shell(unsigned long long num_rows, const char *tableName);
static void probe_mode_from_env(struct engineS *engine);

struct engineS *initializeEngineSyntheticRankHIP(unsigned long long rows_total, unsigned long long seed, int world, int rank,
                                                 const char *tableName) {
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "HIP engine: rank %d of %d\n", rank, world); return NULL; }
    if (rows_total > 0xFFFFFFFFull) { fprintf(stderr, "HIP engine: row numbers are 32 bits wide: %llu rows\n", rows_total); return NULL; }
    uint64_t start = 0, count = 0;
    pqps_partition(rows_total, world, rank, &start, &count);       /* mpi:703-715 */
    struct engineS *engine = engine_shell(count, tableName);
    if (!engine) return NULL;
    /* six lanes: the exchange holds the payload of two queries back behind the scans in flight (pqps_exchange_select) */
    buildSyntheticShardDeviceTableHIP(engine, count, seed, start, 6);
    struct hipTable *t = engine->record_block;
    t->world = world;
    t->rank = rank;
    t->rows_total = rows_total;
    probe_mode_from_env(engine);
    return engine;
}

int hipEngineRcclIdHIP(const char *rccl_library, void *id128) {
    if (!id128) return -1;
    if (pqps_exchange_unique_id(rccl_library, (pqps_rccl_id *)id128) != PQPS_OK) return engine_error("RCCL id");
    return 0;
}

int hipEngineJoinPrepareHIP(struct engineS *engine, const char *rccl_library) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    if (t->world < 1 || hipTableShards(t) != 1 || t->xch) { fprintf(stderr, "HIP engine: not a rank's engine, or joined already\n"); return -1; }
    if (hipTableLockExclusive(t) != 0) return -1;
    const int rc = pqps_exchange_prepare(t->ctx, rccl_library, (uint32_t)t->world, (uint32_t)t->rank, t->capacity_rows, (uint32_t)t->n_lanes, &t->xch);
    hipTableUnlockExclusive(t);
    if (rc != PQPS_OK) { t->xch = NULL; return engine_error("exchange set-up"); }
    return 0;
}

int hipEngineJoinConnectHIP(struct engineS *engine, const void *id128) {
    if (!engine || !engine->record_block || !id128) return -1;
    struct hipTable *t = engine->record_block;
    if (!t->xch) return -1;
    if (pqps_exchange_connect(t->xch, (const pqps_rccl_id *)id128) != PQPS_OK) {
        engine_error("communicator");
        pqps_exchange_destroy(t->xch);
        t->xch = NULL;
        return -1;
    }
    return 0;
}

int hipEngineJoinRanksHIP(struct engineS *engine, const char *rccl_library, const void *id128) {
    if (hipEngineJoinPrepareHIP(engine, rccl_library) != 0) return -1;
    return hipEngineJoinConnectHIP(engine, id128);
}

int hipEngineLeaveRanksHIP(struct engineS *engine) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    if (hipTableLockExclusive(t) != 0) return -1;                   /* every ticket is released */
    if (t->xch) { pqps_exchange_destroy(t->xch); t->xch = NULL; }
    hipTableUnlockExclusive(t);
    return 0;
}

int hipEngineWireBytesHIP(struct engineS *engine, unsigned long long out[2], int reset) {
    if (!engine || !engine->record_block || !out) return -1;
    struct hipTable *t = engine->record_block;
    uint64_t b[2] = { 0, 0 };
    if (t->xch) pqps_exchange_wire_bytes(t->xch, b, reset);
    out[0] = b[0];
    out[1] = b[1];
    return 0;
}

int hipEngineEagerQueriesHIP(struct engineS *engine, unsigned long long out[3], int reset) {
    if (!engine || !engine->record_block || !out) return -1;
    struct hipTable *t = engine->record_block;
    uint64_t b[3] = { 0, 0, 0 };
    if (t->xch) pqps_exchange_eager(t->xch, b, reset);
    out[0] = b[0];
    out[1] = b[1];
    out[2] = b[2];
    return 0;
}

int hipEngineLanes(struct engineS *engine) {
    if (!engine || !engine->record_block) return -1;
    return hipTableLaneCount(engine->record_block);
}

/* The lane goes back (its device buffers are no longer this query's); the table stays locked shared. */
static void ticket_release_lane(struct hipQueryTicket *tk) {
    if (tk->state == 2) return;
    /* never leave a launch behind that writes into a freed lane */
    if (tk->state == 0) (void)query_await(&tk->q);
    else if (tk->state < 0) for (int sx = 0; sx < tk->q.n_shards; sx++) (void)wait_shard(&tk->q, sx);
    query_free(&tk->q);
    hipTableReleaseLane(tk->q.t, tk->q.lane);
    tk->state = 2;
}

void releaseQueryHIP(struct hipQueryTicket *tk) {
    if (!tk) return;
    struct hipTable *t = tk->q.t;
    ticket_release_lane(tk);
    hipTableUnlockShared(t);
    free(tk);
}

/* The answer's row numbers on the host (malloc'd); -1 on error. */
static long long ticket_download_ids(struct hipQueryTicket *tk, unsigned int **ids) {
    *ids = NULL;
    long long count = awaitQueryHIP(tk, NULL);
    if (count < 0) return -1;
    unsigned int *out = malloc((count ? (size_t)count : 1) * sizeof *out);
    if (!out) { fprintf(stderr, "HIP engine: out of memory for %lld result IDs\n", count); return -1; }
    if (count && pqps_download(tk->q.ids_ctx, out, tk->q.ids_dev, (size_t)count * sizeof *out, NULL) != PQPS_OK) {
        engine_error("ID download");
        free(out);
        return -1;
    }
    *ids = out;
    return count;
}

/* The filter without the projection: the row numbers on the host. */
long long executeQuerySelectIdsHIP(struct engineS *engine, struct whereClauseS *whereClause,
                                   unsigned int **ids, double *queryTime) {
    if (!engine || !engine->record_block || !ids) return -1;
    *ids = NULL;
    const double t0 = now_seconds();
    struct hipQueryTicket *tk = executeQuerySelectAsyncHIP(engine, whereClause);
    const long long count = tk ? ticket_download_ids(tk, ids) : -1;
    releaseQueryHIP(tk);
    if (queryTime) *queryTime = now_seconds() - t0;
    return count;
}

long long executeQueryCountHIP(struct engineS *engine, struct whereClauseS *whereClause) {
    if (!engine || !engine->record_block) return -1;
    struct hipQueryTicket *tk = executeQueryCountAsyncHIP(engine, whereClause);
    const long long count = awaitQueryHIP(tk, NULL);
    releaseQueryHIP(tk);
    return count;
}

/* ---- projection ------------------------------------------------------------------ */

/* get_attribute_string_value, S:216-248, with the column resolved once per query
 * instead of one strcmp chain per cell. */
static char *cell_text(const record *r, const FieldInfo *fi) {
    char buf[32];
    if (!fi) return strdup("NULL");                            /* unknown column, S:244 */
    const char *p = (const char *)r + fi->offset;
    switch (fi->type) {
    case FIELD_UINT64: snprintf(buf, sizeof buf, "%llu", *(const unsigned long long *)p); return strdup(buf);
    case FIELD_INT: snprintf(buf, sizeof buf, "%d", *(const int *)p); return strdup(buf);
    case FIELD_BOOL: return strdup(*(const bool *)p ? "true" : "false");
    default: return strdup(p);
    }
}

static const char *const k_all_columns[12] = {
    "command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
    "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"
};

/* projection of a row range (one task per thread; malloc is thread safe) */
struct project_job {
    struct engineS *engine; const unsigned int *ids; char ***data; const FieldInfo *const *cols; int n_cols;
    size_t begin, end;
};

static void *project_rows(void *arg) {
    struct project_job *j = arg;
    for (size_t i = j->begin; i < j->end; i++) {
        const record *r = j->engine->all_records[j->ids[i]];
        char **row = malloc((size_t)j->n_cols * sizeof(char *));
        if (!row) { perror("Failed to allocate result row"); exit(EXIT_FAILURE); }
        for (int c = 0; c < j->n_cols; c++) row[c] = cell_text(r, j->cols[c]);
        j->data[i] = row;
    }
    return NULL;
}

static struct resultSetS *select_device_only(struct engineS *engine, const char **selectItems, int numSelectItems, struct whereClauseS *whereClause);

struct resultSetS *executeQuerySelectHIP(struct engineS *engine, const char **selectItems, int numSelectItems,
                                         const char *tableName, struct whereClauseS *whereClause) {
    (void)tableName;                                   /* never checked by the reference either */
    if (engine && engine->record_block && ((struct hipTable *)engine->record_block)->device_only)
        return select_device_only(engine, selectItems, numSelectItems, whereClause);
    struct resultSetS *rs = malloc(sizeof *rs);
    if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    memset(rs, 0, sizeof *rs);

    unsigned int *ids = NULL;
    double qtime = 0.0;
    if (!engine || !engine->record_block) { rs->success = false; return rs; }
    const double t_sel = now_seconds();
    struct hipQueryTicket *tk = executeQuerySelectAsyncHIP(engine, whereClause);    /* holds the table shared: the projection below reads the host rows */
    const long long count = tk ? ticket_download_ids(tk, &ids) : -1;
    if (tk) ticket_release_lane(tk);                   /* the device side is done: the lane serves the next query while this one builds strings */
    qtime = now_seconds() - t_sel;
    if (count < 0) {                                   /* reason already on stderr */
        releaseQueryHIP(tk);
        rs->success = false;
        return rs;
    }

    rs->numRecords = (int)count;
    if (selectItems == NULL || numSelectItems == 0) {  /* SELECT *, S:490-492 */
        selectItems = (const char **)k_all_columns;
        rs->numColumns = 12;
    } else {
        rs->numColumns = numSelectItems;
    }
    rs->columnNames = malloc((size_t)rs->numColumns * sizeof(char *));
    const FieldInfo **cols = malloc((size_t)(rs->numColumns > 0 ? rs->numColumns : 1) * sizeof *cols);
    for (int j = 0; j < rs->numColumns; j++) {
        rs->columnNames[j] = strdup(selectItems[j]);
        cols[j] = get_field_info(selectItems[j]);
    }
    rs->data = malloc((count ? (size_t)count : 1) * sizeof(char **));

    /* S:504-515 -- rows x columns heap strings, the result-set contract of the reference */
    const double t_proj = now_seconds();
    int nt = 1;
    if (count >= 8192) {
        const char *env = getenv("PQPS_HOST_THREADS");
        long cpus = env ? atol(env) : sysconf(_SC_NPROCESSORS_ONLN);
        nt = cpus < 1 ? 1 : (cpus > 16 ? 16 : (int)cpus);
    }
    struct project_job job[16];
    pthread_t tid[16];
    for (int k = 0; k < nt; k++) {
        job[k] = (struct project_job){ engine, ids, rs->data, cols, rs->numColumns,
                                       (size_t)count * (size_t)k / (size_t)nt, (size_t)count * (size_t)(k + 1) / (size_t)nt };
        if (nt == 1 || pthread_create(&tid[k], NULL, project_rows, &job[k]) != 0) { project_rows(&job[k]); tid[k] = 0; }
    }
    for (int k = 0; k < nt; k++) if (nt > 1 && tid[k]) pthread_join(tid[k], NULL);
    releaseQueryHIP(tk);
    TRACE("SELECT: %lld rows x %d columns, selection %.3f ms, projection %.3f ms (%d threads)\n", count, rs->numColumns,
          qtime * 1e3, (now_seconds() - t_proj) * 1e3, nt);
    free(cols);
    free(ids);
    rs->columnTypes = calloc((size_t)rs->numColumns, sizeof(FieldType));   /* placeholder, S:524-525 */
    rs->queryTime = qtime;
    rs->success = true;
    return rs;
}

/* ---- columnar SELECT ------------------------------------------------------------------------------- */

/* Dictionary codes of a result column become self-contained: the codes are widened to u32 and re-numbered
 * 0..k-1 over the k distinct values the result holds (still ascending in strcmp order), and the result owns a
 * copy of those k strings -- it stays valid whatever INSERT / DELETE / destroy later do to the engine. */
static int own_dictionary(struct hipColumnarResult *res, int j, const struct hipDictionary *d, uint32_t w, uint64_t count) {
    void *raw = res->values[j];
    uint32_t *wide = w == 4 ? raw : malloc(count * sizeof *wide);
    uint32_t *local = calloc((size_t)(d->count > 0 ? d->count : 1), sizeof *local);
    if (!wide || !local) { if (wide != raw) free(wide); free(local); fprintf(stderr, "HIP engine: out of memory for the result set\n"); return -1; }
    if (w == 1) for (uint64_t i = 0; i < count; i++) wide[i] = ((const uint8_t *)raw)[i];
    else if (w == 2) for (uint64_t i = 0; i < count; i++) wide[i] = ((const uint16_t *)raw)[i];
    if (wide != raw) { free(raw); res->values[j] = wide; }
    for (uint64_t i = 0; i < count; i++) local[wide[i]] = 1;
    int k = 0;
    for (int v = 0; v < d->count; v++) if (local[v]) k++;
    char **values = malloc((size_t)(k ? k : 1) * sizeof *values);
    if (!values) { free(local); fprintf(stderr, "HIP engine: out of memory for the result set\n"); return -1; }
    k = 0;
    for (int v = 0; v < d->count; v++) {
        if (!local[v]) continue;
        values[k] = strdup(d->values[v]);
        if (!values[k]) {                                               /* hand over what exists: freeColumnarResultHIP frees it */
            free(local);
            res->dictionaries[j] = (const char *const *)values;
            res->dictionarySizes[j] = k;
            fprintf(stderr, "HIP engine: out of memory for the result set\n");
            return -1;
        }
        local[v] = (uint32_t)k++;
    }
    for (uint64_t i = 0; i < count; i++) wide[i] = local[wide[i]];
    free(local);
    res->dictionaries[j] = (const char *const *)values;
    res->dictionarySizes[j] = k;
    return 0;
}

/* Device gather of one column for the selected rows of every shard, into `raw` (result order).  `scattered`: the
 * shards' rows are interleaved in the result (index mode over several shards): shard s's k-th row goes to sub_pos[s][k]. */
static int project_column(struct query *q, int c, void *const *gathered, bool scattered, unsigned int *const *sub_pos, char *raw, char *tmp) {
    struct hipTable *t = q->t;
    const uint32_t w = t->col[c].width;
    uint64_t at = 0;
    for (int s = 0; s < q->n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        struct hipLane *L = query_lane(q, s);
        pqps_ctx *cs = lane_copy_ctx(q, s);
        const uint64_t k = q->count[s];
        if (k == 0) continue;
        TRY(pqps_project_column(cs, &sh->col[c], L->ids_dev, L->count_dev, k, (uint32_t)sh->row0, gathered[s], NULL), "device projection");
        if (!scattered) {
            TRY(pqps_download(cs, raw + at * w, gathered[s], k * w, NULL), "projection download");
        } else {
            TRY(pqps_download(cs, tmp, gathered[s], k * w, NULL), "projection download");
            for (uint64_t i = 0; i < k; i++) memcpy(raw + (size_t)sub_pos[s][i] * w, tmp + i * w, w);
        }
        at += k;
    }
    return 0;
}

/* Caller holds the table shared. */
static int select_columnar(struct engineS *engine, struct hipTable *t, const char **selectItems, int numSelectItems,
                           struct whereClauseS *whereClause, struct hipColumnarResult *res) {
    struct query q;
    const int lane = hipTableAcquireLane(t);
    if (lane == HIP_LANE_REFUSED) return -1;                        /* reason on stderr */
    query_init(&q, engine, t, lane, false);
    int rc = query_issue(&q, whereClause);
    if (rc == 0) rc = query_await(&q);
    else for (int s = 0; s < q.n_shards; s++) (void)wait_shard(&q, s);
    const int n_shards = q.n_shards;
    const uint64_t count = rc == 0 ? q.total : 0;
    struct hipSchema schema;
    hipSchemaOfTable(t, &schema);
    void *gathered[HIP_MAX_SHARDS];
    unsigned int *sub_pos[HIP_MAX_SHARDS];
    memset(gathered, 0, sizeof gathered);
    memset(sub_pos, 0, sizeof sub_pos);
    char *tmp = NULL;
    const bool scattered = rc == 0 && n_shards > 1 && q.n_probes > 0 && count > 0;
    if (scattered) {
        /* index mode over several shards: each shard gathers for its own rows in merged order (its sub-list
         * replaces the shard-order list in its lane), the values are scattered to their merged positions */
        uint64_t fill[HIP_MAX_SHARDS], biggest = 0;
        memset(fill, 0, sizeof fill);
        unsigned int *sub_ids[HIP_MAX_SHARDS];
        memset(sub_ids, 0, sizeof sub_ids);
        unsigned int *merged = malloc(count * sizeof *merged);
        if (!merged) rc = -1;
        if (rc == 0 && pqps_download(q.ids_ctx, merged, q.ids_dev, count * sizeof *merged, NULL) != PQPS_OK) rc = engine_error("ID download");
        for (int s = 0; s < n_shards; s++) {
            sub_ids[s] = malloc((q.count[s] ? q.count[s] : 1) * sizeof **sub_ids);
            sub_pos[s] = malloc((q.count[s] ? q.count[s] : 1) * sizeof **sub_pos);
            if (!sub_ids[s] || !sub_pos[s]) rc = -1;
            if (q.count[s] > biggest) biggest = q.count[s];
        }
        for (uint64_t i = 0; i < count && rc == 0; i++) {
            const unsigned int id = merged[i];
            int s = n_shards - 1;
            while (s > 0 && id < hipTableShard(t, s)->row0) s--;
            sub_ids[s][fill[s]] = id;
            sub_pos[s][fill[s]++] = (unsigned int)i;
        }
        for (int s = 0; s < n_shards && rc == 0; s++)
            if (q.count[s] && pqps_upload(lane_copy_ctx(&q, s), query_lane(&q, s)->ids_dev, sub_ids[s], q.count[s] * sizeof **sub_ids, NULL) != PQPS_OK)
                rc = engine_error("ID upload");
        for (int s = 0; s < n_shards; s++) free(sub_ids[s]);
        free(merged);
        tmp = malloc((biggest ? biggest : 1) * 8);
        if (!tmp) rc = -1;
    }
    for (int s = 0; s < n_shards && rc == 0; s++)
        if (q.count[s] && pqps_malloc(hipTableShard(t, s)->ctx, q.count[s] * 8, &gathered[s]) != PQPS_OK) rc = engine_error("projection buffer");
    for (int j = 0; j < numSelectItems && rc == 0; j++) {
        res->columnNames[j] = strdup(selectItems[j]);
        const int c = hipColumnId(selectItems[j]);
        res->columnKinds[j] = c < 0 ? -1 : schema.col[c].kind;
        if (c < 0 || count == 0) continue;
        const uint32_t w = t->col[c].width;
        if (w == 0) {                                                    /* a single-valued string column: every row carries code 0 */
            res->values[j] = calloc(count, sizeof(uint32_t));
            if (!res->values[j]) { fprintf(stderr, "HIP engine: out of memory for the result set\n"); rc = -1; break; }
            rc = own_dictionary(res, j, &t->dict[c], 4, count);
            continue;
        }
        char *raw = malloc(count * w);
        if (!raw) { fprintf(stderr, "HIP engine: out of memory for the result set\n"); rc = -1; break; }
        res->values[j] = raw;
        rc = project_column(&q, c, gathered, scattered, sub_pos, raw, tmp);
        if (rc == 0 && schema.col[c].kind == HIPKIND_DICT)
            rc = own_dictionary(res, j, &t->dict[c], w, count);
    }
    for (int s = 0; s < n_shards; s++) {
        if (gathered[s]) pqps_free(hipTableShard(t, s)->ctx, gathered[s]);
        free(sub_pos[s]);
    }
    free(tmp);
    res->numRecords = (int)count;
    query_free(&q);
    hipTableReleaseLane(t, q.lane);
    return rc;
}

struct hipColumnarResult *executeQuerySelectColumnarHIP(struct engineS *engine, const char **selectItems, int numSelectItems,
                                                        struct whereClauseS *whereClause) {
    struct hipColumnarResult *res = calloc(1, sizeof *res);
    if (!res) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    if (!engine || !engine->record_block) return res;
    struct hipTable *t = engine->record_block;
    if (selectItems == NULL || numSelectItems == 0) { selectItems = (const char **)k_all_columns; numSelectItems = 12; }
    res->numColumns = numSelectItems;
    res->columnNames = calloc((size_t)numSelectItems, sizeof(char *));
    res->columnKinds = calloc((size_t)numSelectItems, sizeof(int));
    res->values = calloc((size_t)numSelectItems, sizeof(void *));
    res->dictionaries = calloc((size_t)numSelectItems, sizeof(*res->dictionaries));
    res->dictionarySizes = calloc((size_t)numSelectItems, sizeof(int));
    if (!res->columnNames || !res->columnKinds || !res->values || !res->dictionaries || !res->dictionarySizes) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }

    const double t0 = now_seconds();
    hipTableLockShared(t);
    const int rc = select_columnar(engine, t, selectItems, numSelectItems, whereClause, res);
    hipTableUnlockShared(t);
    res->queryTime = now_seconds() - t0;
    res->success = rc == 0;
    if (rc != 0) res->numRecords = 0;
    TRACE("SELECT (columnar): %d rows x %d columns in %.3f ms\n", res->numRecords, res->numColumns, res->queryTime * 1e3);
    return res;
}

void freeColumnarResultHIP(struct hipColumnarResult *res) {
    if (!res) return;
    for (int j = 0; j < res->numColumns; j++) {
        if (res->columnNames) free(res->columnNames[j]);
        if (res->values) free(res->values[j]);
        if (res->dictionaries && res->dictionaries[j]) {
            for (int v = 0; v < res->dictionarySizes[j]; v++) free((void *)res->dictionaries[j][v]);
            free((void *)res->dictionaries[j]);
        }
    }
    free(res->columnNames); free(res->columnKinds); free(res->values); free((void *)res->dictionaries); free(res->dictionarySizes);
    free(res);
}

char *hipColumnarCellText(const struct hipColumnarResult *res, int row, int col) {
    char buf[32];
    if (!res || row < 0 || row >= res->numRecords || col < 0 || col >= res->numColumns) return NULL;
    const void *v = res->values[col];
    switch (res->columnKinds[col]) {                                    /* get_attribute_string_value, S:216-248 */
    case HIPKIND_U64: snprintf(buf, sizeof buf, "%llu", (unsigned long long)((const uint64_t *)v)[row]); return strdup(buf);
    case HIPKIND_I32: snprintf(buf, sizeof buf, "%d", ((const int32_t *)v)[row]); return strdup(buf);
    case HIPKIND_BOOL: return strdup(((const uint8_t *)v)[row] ? "true" : "false");
    case HIPKIND_DICT: return strdup(res->dictionaries[col][((const uint32_t *)v)[row]]);
    default: return strdup("NULL");                                     /* unknown column, S:244 */
    }
}

struct resultSetS *hipColumnarHead(const struct hipColumnarResult *res, int limit) {
    struct resultSetS *rs = calloc(1, sizeof *rs);
    if (!rs || !res) { free(rs); return NULL; }
    const int rows = limit <= 0 || limit > res->numRecords ? res->numRecords : limit;   /* printTable: limit <= 0 = every row */
    rs->numRecords = res->numRecords;                                   /* the footer counts every record */
    rs->numColumns = res->numColumns;
    rs->columnNames = malloc((size_t)(res->numColumns > 0 ? res->numColumns : 1) * sizeof(char *));
    for (int j = 0; j < res->numColumns; j++) rs->columnNames[j] = strdup(res->columnNames[j]);
    rs->columnTypes = calloc((size_t)(res->numColumns > 0 ? res->numColumns : 1), sizeof(FieldType));
    rs->data = malloc((size_t)(rows > 0 ? rows : 1) * sizeof(char **));
    for (int i = 0; i < rows; i++) {
        rs->data[i] = malloc((size_t)(res->numColumns > 0 ? res->numColumns : 1) * sizeof(char *));
        for (int j = 0; j < res->numColumns; j++) rs->data[i][j] = hipColumnarCellText(res, i, j);
    }
    rs->queryTime = res->queryTime;
    rs->success = res->success;
    return rs;
}

void freeResultSetHead(struct resultSetS *head, int rows) {
    if (!head) return;
    const int full = head->numRecords;
    head->numRecords = rows <= 0 || rows > full ? full : rows;          /* only these rows were materialised */
    freeResultSet(head);
}

/* executeQuerySelectHIP on an engine without host rows: the same strings, made from values gathered on the device. */
static struct resultSetS *select_device_only(struct engineS *engine, const char **selectItems, int numSelectItems, struct whereClauseS *whereClause) {
    struct hipColumnarResult *res = executeQuerySelectColumnarHIP(engine, selectItems, numSelectItems, whereClause);
    struct resultSetS *rs = res && res->success ? hipColumnarHead(res, 0) : NULL;      /* every row */
    if (!rs) {
        rs = calloc(1, sizeof *rs);
        if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
        rs->success = false;
    }
    if (res) freeColumnarResultHIP(res);
    return rs;
}

/* freeResultSet, S:881-908. */
void freeResultSet(struct resultSetS *result) {
    if (!result) return;
    const double t_free = now_seconds();
    const long long cells = (long long)result->numRecords * result->numColumns;
    if (result->columnNames) {
        for (int j = 0; j < result->numColumns; j++) free(result->columnNames[j]);
        free(result->columnNames);
    }
    free(result->columnTypes);
    if (result->data) {
        for (int i = 0; i < result->numRecords; i++) {
            if (!result->data[i]) continue;
            for (int j = 0; j < result->numColumns; j++) free(result->data[i][j]);
            free(result->data[i]);
        }
        free(result->data);
    }
    free(result);
    TRACE("freeResultSet: %lld cells, %.3f ms\n", cells, (now_seconds() - t_free) * 1e3);
}

int isAttributeIndexed(struct engineS *engine, const char *attributeName) {
    for (int i = 0; i < engine->num_indexes; i++)
        if (strcmp(engine->indexed_attributes[i], attributeName) == 0) return i;
    return -1;
}

/* ---- helpers over caller-supplied rows ---------------------------------------------- */

static pqps_ctx *g_adhoc_ctx;
static pthread_mutex_t g_adhoc_lock = PTHREAD_MUTEX_INITIALIZER;   /* one shared context: one ad-hoc filter at a time */

static pqps_ctx *adhoc_ctx(void) {
    if (!g_adhoc_ctx) {
        int device = 0;
        const char *env = getenv("PQPS_DEVICE");
        if (env) device = atoi(env);
        if (pqps_ctx_create(device, &g_adhoc_ctx) != PQPS_OK) { engine_error("cannot create a device context"); g_adhoc_ctx = NULL; }
    }
    return g_adhoc_ctx;
}

static int adhoc_search(struct hipTable *t, record **records, struct whereClauseS *whereClause, record ***out, int *matching) {
    struct engineS none;                                /* no indexes: always the scan path */
    memset(&none, 0, sizeof none);
    struct query q;
    query_init(&q, &none, t, -1, false);                /* a table without lanes: its own context and buffers */
    int rc = query_issue(&q, whereClause);
    if (rc == 0) rc = query_await(&q);
    const uint64_t count = rc == 0 ? q.total : 0;
    uint32_t *ids = malloc((count ? count : 1) * sizeof *ids);
    record **hit = malloc((count ? count : 1) * sizeof *hit);
    if (rc == 0 && (!ids || !hit)) { fprintf(stderr, "HIP engine: out of memory for results\n"); rc = -1; }
    if (rc == 0 && count && pqps_download(q.ids_ctx, ids, q.ids_dev, count * sizeof *ids, NULL) != PQPS_OK) rc = engine_error("ID download");
    query_free(&q);
    if (rc != 0) { free(ids); free(hit); return -1; }
    for (uint64_t i = 0; i < count; i++) hit[i] = records[ids[i]];
    free(ids);
    *out = hit;
    *matching = (int)count;
    return 0;
}

/* linearSearchRecords, S:854-878: the rows are columnarised, filtered on the
 * GPU (input order kept) and the surviving pointers returned.  NULL (and 0 matches) when the device or
 * the clause fails; the reason is on stderr. */
record **linearSearchRecords(record **records, int num_records, struct whereClauseS *whereClause,
                             int *matchingRecords) {
    *matchingRecords = 0;
    record **out = NULL;
    pthread_mutex_lock(&g_adhoc_lock);
    pqps_ctx *ctx = adhoc_ctx();
    if (ctx) {
        struct hipTable *t = hipTableFromRows(ctx, records, (size_t)(num_records > 0 ? num_records : 0));
        if (adhoc_search(t, records, whereClause, &out, matchingRecords) != 0) { out = NULL; *matchingRecords = 0; }
        hipTableFree(t, 0);
    }
    pthread_mutex_unlock(&g_adhoc_lock);
    return out;
}

/* evaluateWhereClause, S:292-316, for one row: a one-row table through the same kernel. */
bool evaluateWhereClause(record *r, struct whereClauseS *wc) {
    if (wc == NULL) return true;
    int n = 0;
    record *rows[1] = { r };
    record **hit = linearSearchRecords(rows, 1, wc, &n);
    free(hit);
    return n == 1;
}

/* ---- lifecycle ------------------------------------------------------------------------ */

/* PQPS_PROBE_BOOL=1: new engines follow the OpenMP / MPI engines' row selection (hipEngineProbeBoolIndexes) */
static void probe_mode_from_env(struct engineS *engine) {
    const char *env = getenv("PQPS_PROBE_BOOL");
    if (env && atoi(env) != 0 && engine->record_block) ((struct hipTable *)engine->record_block)->probe_bool = 1;
}

struct engineS *initializeEngineHIP(int num_indexes, const char *indexed_attributes[],
                                    const int attribute_types[], const char *datafile,
                                    const char *tableName) {
    struct engineS *engine = malloc(sizeof *engine);
    if (!engine) { perror("Failed to allocate memory for engine"); exit(EXIT_FAILURE); }
    memset(engine, 0, sizeof *engine);
    engine->tableName = strdup(tableName ? tableName : "");
    if (!datafile) datafile = "../data/commands_50k.csv";          /* S:757 */
    engine->datafile = strdup(datafile);
    const double t0 = now_seconds();
    struct hipContextFuture *device = hipBeginContextHIP();        /* HIP start-up runs beside the CSV parse */
    engine->all_records = getAllRecordsFromFileHIP(datafile, &engine->num_records, &engine->record_block);
    const double t1 = now_seconds();
    buildDeviceTableOnHIP(engine, device);                         /* exits loudly without a GPU */
    const double t2 = now_seconds();
    for (int i = 0; i < num_indexes; i++) {
        if (!makeIndexHIP(engine, indexed_attributes[i], attribute_types[i]))
            fprintf(stderr, "Failed to create index for attribute: %s\n", indexed_attributes[i]);
    }
    probe_mode_from_env(engine);
    TRACE("init: %d rows, CSV -> rows %.1f ms, rows -> device columns %.1f ms, %d indexes %.1f ms\n", engine->num_records,
          (t1 - t0) * 1e3, (t2 - t1) * 1e3, num_indexes, (now_seconds() - t2) * 1e3);
    return engine;
}

/* An engine without host rows: all_records NULL, datafile "" (no CSV is kept in step). */
static struct engineS *engine_shell(unsigned long long num_rows, const char *tableName) {
    if (num_rows > (unsigned long long)INT_MAX) {
        fprintf(stderr, "HIP engine: %llu rows: the engine API counts rows in int (include/executeEngine-serial.h:22)\n", num_rows);
        return NULL;
    }
    struct engineS *engine = malloc(sizeof *engine);
    if (!engine) { perror("Failed to allocate memory for engine"); exit(EXIT_FAILURE); }
    memset(engine, 0, sizeof *engine);
    engine->tableName = strdup(tableName ? tableName : "");
    engine->datafile = strdup("");
    engine->num_records = (int)num_rows;
    return engine;
}

static void engine_indexes(struct engineS *engine, int num_indexes, const char *indexed_attributes[], const int attribute_types[]) {
    for (int i = 0; i < num_indexes; i++)
        if (!makeIndexHIP(engine, indexed_attributes[i], attribute_types[i]))
            fprintf(stderr, "Failed to create index for attribute: %s\n", indexed_attributes[i]);
    probe_mode_from_env(engine);
}

struct engineS *initializeEngineColumnsHIP(unsigned long long num_rows, const struct hipColumnData columns[12],
                                           int num_indexes, const char *indexed_attributes[], const int attribute_types[],
                                           const char *tableName) {
    if (!columns) return NULL;
    struct engineS *engine = engine_shell(num_rows, tableName);
    if (!engine) return NULL;
    const double t0 = now_seconds();
    buildDeviceTableFromColumnsHIP(engine, num_rows, columns);      /* exits loudly without a GPU */
    const double t1 = now_seconds();
    engine_indexes(engine, num_indexes, indexed_attributes, attribute_types);
    TRACE("init (columns): %llu rows, device columns %.1f ms, %d indexes %.1f ms\n", num_rows, (t1 - t0) * 1e3, num_indexes, (now_seconds() - t1) * 1e3);
    return engine;
}

struct engineS *initializeEngineSyntheticHIP(unsigned long long num_rows, unsigned long long seed,
                                             int num_indexes, const char *indexed_attributes[], const int attribute_types[],
                                             const char *tableName) {
    struct engineS *engine = engine_shell(num_rows, tableName);
    if (!engine) return NULL;
    const double t0 = now_seconds();
    buildSyntheticDeviceTableHIP(engine, num_rows, seed);           /* exits loudly without a GPU */
    const double t1 = now_seconds();
    engine_indexes(engine, num_indexes, indexed_attributes, attribute_types);
    TRACE("init (synthetic): %llu rows, generation %.1f ms, %d indexes %.1f ms\n", num_rows, (t1 - t0) * 1e3, num_indexes, (now_seconds() - t1) * 1e3);
    return engine;
}

void destroyEngineHIP(struct engineS *engine) {
    if (!engine) { fprintf(stderr, "Attempted to destroy a NULL engine pointer\n"); return; }
    const double t_destroy = now_seconds();
    destroyDeviceTableHIP(engine);                                 /* frees the row block too */
    TRACE("destroy: device table + context %.3f ms\n", (now_seconds() - t_destroy) * 1e3);
    free(engine->bplus_tree_roots);
    if (engine->indexed_attributes) {
        for (int i = 0; i < engine->num_indexes; i++) free(engine->indexed_attributes[i]);
        free(engine->indexed_attributes);
    }
    free(engine->attribute_types);
    free(engine->all_records);
    free(engine->tableName);
    free(engine->datafile);
    free(engine);
}

bool addAttributeIndexHIP(struct engineS *engine, const char *tableName, const char *attributeName,
                          int attributeType) {
    (void)tableName;
    if (hipTableLockExclusive(engine->record_block) != 0) return false;
    const bool ok = makeIndexHIP(engine, attributeName, attributeType);
    hipTableUnlockExclusive(engine->record_block);
    return ok;
}

int hipEngineProbeBoolIndexes(struct engineS *engine, int enable) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    if (hipTableLockExclusive(t) != 0) return -1;                   /* no query in flight while the mode changes */
    const int before = t->probe_bool;
    t->probe_bool = enable != 0;
    hipTableUnlockExclusive(t);
    return before;
}

int hipEngineKernelTiming(struct engineS *engine, int enable) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    int rc = 0;
    if (hipTableLockExclusive(t) != 0) return -1;                   /* no query in flight while the recorders are switched */
    for (int s = 0; s < hipTableShards(t) && rc == 0; s++)
        if (hipTableShard(t, s)->qs && pqps_qstream_set_timing(hipTableShard(t, s)->qs, enable) != PQPS_OK) rc = engine_error("kernel timing");
    hipTableUnlockExclusive(t);
    return rc;
}

int hipEngineKernelTime(struct engineS *engine, double *scan_ms, double *query_ms, int *launches) {
    if (!engine || !engine->record_block || !scan_ms || !query_ms || !launches) return -1;
    struct hipTable *t = engine->record_block;
    *scan_ms = 0.0; *query_ms = 0.0; *launches = 0;
    int rc = 0;
    if (hipTableLockExclusive(t) != 0) return -1;
    for (int s = 0; s < hipTableShards(t) && rc == 0; s++) {
        double e = 0.0, q = 0.0;
        int k = 0;
        if (!hipTableShard(t, s)->qs) continue;
        if (pqps_qstream_kernel_time(hipTableShard(t, s)->qs, &e, &q, &k) != PQPS_OK) { rc = engine_error("kernel timing"); break; }
        *scan_ms += e; *query_ms += q; *launches += k;
    }
    hipTableUnlockExclusive(t);
    return rc;
}

int hipEngineShards(struct engineS *engine, unsigned long long *rows, int capacity) {
    if (!engine || !engine->record_block) return 0;
    struct hipTable *t = engine->record_block;
    hipTableLockShared(t);
    const int n = hipTableShards(t);
    for (int s = 0; rows && s < n && s < capacity; s++) rows[s] = hipTableShard(t, s)->n_rows;
    hipTableUnlockShared(t);
    return n;
}

/* ---- mutation (kept in step with the CSV like the reference) ------------------------------ */

static void write_csv_row(FILE *f, const record *r) {           /* S:562, S:687 */
    fprintf(f, "%llu,%s,%s,%s,%d,%s,%d,%s,%d,%s,%s,%d\n", r->command_id, r->raw_command, r->base_command,
            r->shell_type, r->exit_code, r->timestamp, r->sudo_used, r->working_directory, r->user_id,
            r->user_name, r->host_name, r->risk_level);
}

/* executeQueryInsertSerial, S:538-617. */
/* The whole table as CSV text, rows formatted exactly like write_csv_row (S:687-700), by host
 * threads into per-range buffers that are then written in order: the rewrite after a DELETE is
 * the reference's own file format and by far the longest phase of a DELETE on a large table. */
struct csv_job { const record *rows; size_t begin, end; char *buf; size_t len; };

static char *put_u64(char *p, unsigned long long v) {
    char tmp[24];
    int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) *p++ = tmp[--k];
    return p;
}

static char *put_i32(char *p, int v) {
    unsigned long long u = v < 0 ? (unsigned long long)(-(long long)v) : (unsigned long long)v;
    if (v < 0) *p++ = '-';
    return put_u64(p, u);
}

static char *put_str(char *p, const char *s) {
    const size_t k = strlen(s);
    memcpy(p, s, k);
    return p + k;
}

static void *csv_format_rows(void *arg) {
    struct csv_job *j = arg;
    /* exact room: the strings as "%s" would print them (a field filled to its last byte has no NUL
     * and runs on into the next one, in the reference's fprintf as well) + 5 numbers + separators */
    size_t room = 1;
    for (size_t i = j->begin; i < j->end; i++) {
        const record *r = &j->rows[i];
        room += strlen(r->raw_command) + strlen(r->base_command) + strlen(r->shell_type) + strlen(r->timestamp) +
                strlen(r->working_directory) + strlen(r->user_name) + strlen(r->host_name) + 5 * 21 + 12;
    }
    j->buf = malloc(room);
    if (!j->buf) { j->len = 0; return NULL; }
    char *p = j->buf;
    for (size_t i = j->begin; i < j->end; i++) {
        const record *r = &j->rows[i];
        p = put_u64(p, r->command_id); *p++ = ',';
        p = put_str(p, r->raw_command); *p++ = ',';
        p = put_str(p, r->base_command); *p++ = ',';
        p = put_str(p, r->shell_type); *p++ = ',';
        p = put_i32(p, r->exit_code); *p++ = ',';
        p = put_str(p, r->timestamp); *p++ = ',';
        p = put_i32(p, (int)r->sudo_used); *p++ = ',';
        p = put_str(p, r->working_directory); *p++ = ',';
        p = put_i32(p, r->user_id); *p++ = ',';
        p = put_str(p, r->user_name); *p++ = ',';
        p = put_str(p, r->host_name); *p++ = ',';
        p = put_i32(p, r->risk_level); *p++ = '\n';
    }
    j->len = (size_t)(p - j->buf);
    return NULL;
}

static void write_csv_table(FILE *f, const record *rows, size_t n) {
    enum { kMaxJobs = 16, kChunkRows = 65536 };
    for (size_t base = 0; base < n; ) {                         /* bounded memory: <= 16 chunks in flight */
        struct csv_job job[kMaxJobs];
        pthread_t tid[kMaxJobs];
        int nj = 0;
        while (nj < kMaxJobs && base < n) {
            const size_t end = base + kChunkRows < n ? base + kChunkRows : n;
            job[nj] = (struct csv_job){ rows, base, end, NULL, 0 };
            base = end;
            nj++;
        }
        for (int k = 0; k < nj; k++)
            if (nj == 1 || pthread_create(&tid[k], NULL, csv_format_rows, &job[k]) != 0) { csv_format_rows(&job[k]); tid[k] = 0; }
        for (int k = 0; k < nj; k++) {
            if (nj > 1 && tid[k]) pthread_join(tid[k], NULL);
            if (job[k].buf) fwrite(job[k].buf, 1, job[k].len, f);
            else for (size_t i = job[k].begin; i < job[k].end; i++) write_csv_row(f, &rows[i]);   /* out of memory: plain path */
            free(job[k].buf);
        }
    }
}

bool executeQueryInsertHIP(struct engineS *engine, const char *tableName, const record *r) {
    (void)tableName;
    if (r->command_id == 0 || !r->raw_command[0] || !r->base_command[0] || !r->shell_type[0] ||
        !r->timestamp[0] || !r->working_directory[0] || !r->user_name[0] || !r->host_name[0])
        return false;                                              /* S:544-551 */
    struct hipTable *t = engine->record_block;
    const double t0 = now_seconds();
    if (hipTableLockExclusive(t) != 0) return false;
    const size_t n = (size_t)engine->num_records;
    if (t->device_only) {
        /* no host rows, no CSV: the row goes to the device table alone */
        if (n + 1 > (size_t)INT_MAX) { hipTableUnlockExclusive(t); return false; }
        engine->num_records = (int)(n + 1);
        const bool ok = appendRowDeviceTableHIP(engine, r);
        if (!ok) {
            engine->num_records = (int)n;
            fprintf(stderr, "HIP engine: INSERT needs a table rebuild (head-room used up or a dictionary outgrowing its code width), which an engine without host rows cannot do\n");
        }
        TRACE("INSERT (device only): %.3f ms\n", (now_seconds() - t0) * 1e3);
        hipTableUnlockExclusive(t);
        return ok;
    }
    /* room for the row first: a failed allocation must not leave the CSV one row ahead of the engine.
     * The host row store grows geometrically; all_records[] is re-pointed only when the block moved */
    if (n + 1 > t->row_capacity) {
        const size_t cap = n + n / 8 + 64;
        record *block = realloc(t->row_block, cap * sizeof *block);
        if (block) {
            if (block != t->row_block) for (size_t i = 0; i < n; i++) engine->all_records[i] = &block[i];
            t->row_block = block;
        }
        record **rows = block ? realloc(engine->all_records, cap * sizeof *rows) : NULL;
        if (rows) engine->all_records = rows;
        if (!block || !rows) { hipTableUnlockExclusive(t); return false; }
        t->row_capacity = cap;
    }
    FILE *f = fopen(engine->datafile, "a");
    if (!f) { hipTableUnlockExclusive(t); return false; }
    write_csv_row(f, r);
    fclose(f);

    t->row_block[n] = *r;
    engine->all_records[n] = &t->row_block[n];
    engine->num_records = (int)(n + 1);
    const double t1 = now_seconds();
    appendRowDeviceTableHIP(engine, engine->all_records[n]);
    TRACE("INSERT: CSV append + host row %.3f ms, device append + indexes %.3f ms\n", (t1 - t0) * 1e3, (now_seconds() - t1) * 1e3);
    hipTableUnlockExclusive(t);
    return true;
}

/* executeQueryDeleteSerial, S:627-715: the per-row decision is the GPU flag
 * kernel (the flag-array shape of engine/omp/executeEngine-omp.c:708-732). */
/* Flags of the rows that go, per shard on the device; `flags` (may be NULL: an engine without host rows) receives them
 * for the whole table, *deleted their number. */
static int delete_flags(struct engineS *engine, struct hipTable *t, struct whereClauseS *whereClause, size_t n, uint8_t **flags_dev, uint8_t *flags,
                        uint64_t *deleted) {
    struct query q;
    query_init(&q, engine, t, -1, true);                            /* the writer is alone: the table's own context and buffers */
    *deleted = 0;
    if (bind_where(t, whereClause, &q.plan) != 0) return -1;
    q.have_plan = true;
    int rc = 0;
    for (int s = 0; s < q.n_shards && rc == 0; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        /* the passes in front of the last as for any query, the last pass as flags instead of a count */
        struct hipPlan head = q.plan;
        struct shard_pred *sp = &q.sp[s];
        memset(sp, 0, sizeof *sp);
        const struct hipPass *last = &head.pass[head.n_passes - 1];
        if (head.n_passes > 1) {
            sp->flags = calloc((size_t)head.n_passes - 1, sizeof *sp->flags);
            if (!sp->flags) { rc = -1; break; }
            sp->n_flags = head.n_passes - 1;
            for (int k = 0; k < sp->n_flags && rc == 0; k++) {
                pqps_column cols[PQPS_MAX_COLUMNS];
                if (pqps_malloc(sh->ctx, sh->capacity_rows, (void **)&sp->flags[k]) != PQPS_OK) { rc = engine_error("flag allocation"); break; }
                pass_columns(sh, &head.pass[k], sp->flags, cols);
                if (pqps_filter_flags(sh->ctx, cols, head.pass[k].pred.n_columns, sh->n_rows, &head.pass[k].pred, sp->flags[k], sh->own.count_dev + 4, NULL) != PQPS_OK)
                    rc = engine_error("flag filter");
            }
        }
        sp->pred = &last->pred;
        sp->n_cols = last->pred.n_columns;
        pass_columns(sh, last, sp->flags, sp->cols);
        if (rc == 0 && pqps_malloc(sh->ctx, sh->capacity_rows, (void **)&flags_dev[s]) != PQPS_OK) rc = engine_error("flag allocation");
        if (rc == 0 && pqps_filter_flags(sh->ctx, sp->cols, sp->n_cols, sh->n_rows, sp->pred, flags_dev[s], sh->own.count_dev, NULL) != PQPS_OK)
            rc = engine_error("flag filter");
    }
    for (int s = 0; s < q.n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        uint64_t k = 0;
        if (rc == 0 && pqps_ctx_sync(sh->ctx, NULL) != PQPS_OK) rc = engine_error("filter execution");
        if (rc == 0 && pqps_download(sh->ctx, &k, sh->own.count_dev, sizeof k, NULL) != PQPS_OK) rc = engine_error("count download");
        *deleted += k;
        if (rc == 0 && sh->row0 + sh->n_rows > n) { fprintf(stderr, "HIP engine: device shards hold more rows than the engine\n"); rc = -1; }
        if (rc == 0 && flags && sh->n_rows && pqps_download(sh->ctx, flags + sh->row0, flags_dev[s], sh->n_rows, NULL) != PQPS_OK) rc = engine_error("flag download");
    }
    query_free(&q);
    return rc;
}

struct resultSetS *executeQueryDeleteHIP(struct engineS *engine, const char *tableName,
                                         struct whereClauseS *whereClause) {
    (void)tableName;
    struct resultSetS *rs = calloc(1, sizeof *rs);
    if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    const double t0 = now_seconds();
    struct hipTable *t = engine->record_block;
    if (hipTableLockExclusive(t) != 0) { memset(rs, 0, sizeof *rs); rs->success = false; return rs; }
    const size_t n = (size_t)engine->num_records;
    const int n_shards = hipTableShards(t);
    uint8_t *flags_dev[HIP_MAX_SHARDS];
    memset(flags_dev, 0, sizeof flags_dev);
    uint8_t *flags = t->device_only ? NULL : malloc(n ? n : 1);
    uint64_t flagged = 0;
    if ((!flags && !t->device_only) || delete_flags(engine, t, whereClause, n, flags_dev, flags, &flagged) != 0) {
        if (!flags && !t->device_only) fprintf(stderr, "HIP engine: out of memory for %zu delete flags\n", n);
        for (int s = 0; s < n_shards; s++) if (flags_dev[s]) pqps_free(hipTableShard(t, s)->ctx, flags_dev[s]);
        free(flags);
        hipTableUnlockExclusive(t);
        rs->success = false;                                       /* nothing was deleted */
        return rs;
    }
    const double t1 = now_seconds();

    size_t keep = 0, deleted = 0;
    if (t->device_only) {                                           /* no host rows, no CSV */
        deleted = (size_t)flagged;
        keep = n - deleted;
        engine->num_records = (int)keep;
    } else {
        record *block = t->row_block;
        for (size_t i = 0; i < n; i++) {
            if (flags[i]) { deleted++; continue; }
            if (keep != i) block[keep] = block[i];
            keep++;
        }
        free(flags);
        for (size_t i = 0; i < keep; i++) engine->all_records[i] = &block[i];
        engine->num_records = (int)keep;
    }
    const double t2 = now_seconds();

    if (!t->device_only) {
        FILE *f = fopen(engine->datafile, "w");                   /* S:683-701: no header written */
        if (f) {
            write_csv_table(f, t->row_block, keep);
            fclose(f);
        }
    }
    const double t3 = now_seconds();
    /* device side: the same flags compact the 12 columns of every shard in place (order kept); dictionaries
     * stay as they are (a code without rows is harmless), indexes are re-sorted */
    if (deleted) compactDeviceTableHIP(engine, flags_dev, keep);
    for (int s = 0; s < n_shards; s++) pqps_free(hipTableShard(t, s)->ctx, flags_dev[s]);
    TRACE("DELETE: %zu of %zu rows, flags %.3f ms, host rows %.3f ms, CSV rewrite %.3f ms, device compaction + indexes %.3f ms\n",
          deleted, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now_seconds() - t3) * 1e3);
    hipTableUnlockExclusive(t);
    rs->numRecords = (int)deleted;
    rs->queryTime = now_seconds() - t0;
    rs->success = true;
    return rs;
}
