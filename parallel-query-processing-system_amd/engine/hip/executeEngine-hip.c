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

/* Enqueues the passes before the last on the shard's stream (each one byte of flags per row) and binds the last. */
static int shard_pred_prepare(struct hipTable *sh, const struct hipPlan *plan, struct shard_pred *sp) {
    memset(sp, 0, sizeof *sp);
    const struct hipPass *last = &plan->pass[plan->n_passes - 1];
    if (plan->n_passes > 1) {
        sp->flags = calloc((size_t)plan->n_passes - 1, sizeof *sp->flags);
        if (!sp->flags) { fprintf(stderr, "HIP engine: out of memory\n"); return -1; }
        sp->n_flags = plan->n_passes - 1;
        for (int k = 0; k < sp->n_flags; k++) {
            pqps_column cols[PQPS_MAX_COLUMNS];
            TRY(pqps_malloc(sh->ctx, sh->capacity_rows, (void **)&sp->flags[k]), "flag allocation");
            pass_columns(sh, &plan->pass[k], sp->flags, cols);
            TRY(pqps_filter_flags(sh->ctx, cols, plan->pass[k].pred.n_columns, sh->n_rows, &plan->pass[k].pred,
                                  sp->flags[k], sh->count_dev + 4, NULL), "flag filter");
        }
    }
    sp->pred = &last->pred;
    sp->n_cols = last->pred.n_columns;
    pass_columns(sh, last, sp->flags, sp->cols);
    return 0;
}

static int ensure_id_capacity(struct hipTable *t, uint64_t need) {
    if (need <= t->capacity_ids) return 0;
    pqps_free(t->ctx, t->ids_dev);
    t->ids_dev = NULL;
    t->capacity_ids = 0;
    const uint64_t cap = need + need / 8 + 1024;
    TRY(pqps_malloc(t->ctx, cap * sizeof(uint32_t), (void **)&t->ids_dev), "result allocation");
    t->capacity_ids = cap;
    return 0;
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

/* One index probe the serial engine would make for this WHERE (S:358-433), in its order. */
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

/* Result of one row selection.  Scan mode: shard s holds count[s] ascending engine row numbers in its ids_dev;
 * the answer is their concatenation in shard order.  Index mode: per probe, each shard appended its rows in
 * (key asc, row desc) order; on one shard that is the answer as it stands, on several `merged` holds it. */
struct selection {
    uint64_t total;
    uint64_t count[HIP_MAX_SHARDS];
    unsigned int *merged;            /* host, `total` entries; index mode over several shards only */
};

static uint64_t row_key(const record *r, const FieldInfo *fi) {
    const char *p = (const char *)r + fi->offset;
    return fi->type == FIELD_UINT64 ? *(const uint64_t *)p : (uint64_t)((int64_t)*(const int *)p + 0x80000000ll);
}

/* Row selection of executeQuerySelectSerial, S:358-474, on the device(s). */
static int run_selection(struct engineS *engine, struct hipTable *t, struct whereClauseS *where, struct selection *sel) {
    memset(sel, 0, sizeof *sel);
    const int n_shards = hipTableShards(t);
    struct hipPlan plan;
    if (bind_where(t, where, &plan) != 0) return -1;
    struct probe *probes = NULL;
    const int n_probes = list_probes(engine, t, where, &probes);
    struct shard_pred sp[HIP_MAX_SHARDS];
    memset(sp, 0, sizeof sp);
    uint64_t *seg_end = NULL;                                          /* [probe][shard] running counts */
    int rc = n_probes < 0 ? -1 : 0;
#define RUN(call, what) do { if (rc == 0 && (call) != PQPS_OK) rc = engine_error(what); } while (0)
    for (int s = 0; s < n_shards && rc == 0; s++) rc = shard_pred_prepare(hipTableShard(t, s), &plan, &sp[s]);
    if (rc == 0 && n_probes > 0 && n_shards > 1) {
        seg_end = malloc((size_t)n_probes * (size_t)n_shards * sizeof *seg_end);
        if (!seg_end) rc = -1;
    }
    for (bool again = true; rc == 0 && again; ) {
        again = false;
        if (n_probes > 0) {
            for (int s = 0; s < n_shards; s++) {
                struct hipTable *sh = hipTableShard(t, s);
                RUN(pqps_memset(sh->ctx, sh->count_dev, 0, sizeof(uint64_t), NULL), "counter reset");
            }
            for (int k = 0; k < n_probes && rc == 0; k++) {
                for (int s = 0; s < n_shards; s++) {
                    struct hipTable *sh = hipTableShard(t, s);
                    const struct hipIndex *ix = &sh->index[probes[k].index];
                    uint64_t *range_dev = sh->count_dev + 2;
                    RUN(pqps_index_probe(sh->ctx, ix->keys_dev, sh->col[ix->column].width, ix->key_kind,
                                         sh->n_rows, probes[k].lo, probes[k].hi, range_dev, NULL), "index probe");
                    /* append the probe's rows that pass the complete WHERE, leaf order kept (S:441-448 + S:471) */
                    RUN(pqps_filter_gather(sh->ctx, sp[s].cols, sp[s].n_cols, ix->perm_dev, range_dev, sh->n_rows,
                                           (uint32_t)sh->row0, sp[s].pred, sh->ids_dev, sh->capacity_ids, sh->count_dev, NULL), "index filter");
                }
                if (seg_end)                                           /* where this probe's rows end on every shard */
                    for (int s = 0; s < n_shards; s++) {
                        struct hipTable *sh = hipTableShard(t, s);
                        RUN(pqps_download(sh->ctx, &seg_end[(size_t)k * (size_t)n_shards + (size_t)s], sh->count_dev, sizeof(uint64_t), NULL), "count download");
                    }
            }
        } else {                                                       /* full scan, S:464-467: all shards at once */
            for (int s = 0; s < n_shards; s++) {
                struct hipTable *sh = hipTableShard(t, s);
                RUN(pqps_filter_scan(sh->ctx, sp[s].cols, sp[s].n_cols, sh->n_rows, (uint32_t)sh->row0, sp[s].pred,
                                     sh->ids_dev, sh->capacity_ids, sh->count_dev, NULL), "scan filter");
            }
        }
        sel->total = 0;
        for (int s = 0; s < n_shards; s++) {
            struct hipTable *sh = hipTableShard(t, s);
            RUN(pqps_ctx_sync(sh->ctx, NULL), "filter execution");
            RUN(pqps_download(sh->ctx, &sel->count[s], sh->count_dev, sizeof(uint64_t), NULL), "count download");
            if (rc == 0 && sel->count[s] > sh->capacity_ids) {         /* duplicates can exceed n rows: retry larger */
                rc = ensure_id_capacity(sh, sel->count[s]);
                again = true;
            }
            sel->total += sel->count[s];
        }
    }
    if (rc == 0 && seg_end && sel->total > 0) {
        /* several shards, index mode: every probe's segment is sorted by (key asc, row desc) on each shard, and
         * the shards hold ascending row ranges -- merge the segments probe by probe */
        unsigned int *part[HIP_MAX_SHARDS];
        memset(part, 0, sizeof part);
        sel->merged = malloc(sel->total * sizeof *sel->merged);
        if (!sel->merged) rc = -1;
        for (int s = 0; s < n_shards && rc == 0; s++) {
            struct hipTable *sh = hipTableShard(t, s);
            part[s] = malloc((sel->count[s] ? sel->count[s] : 1) * sizeof **part);
            if (!part[s]) { rc = -1; break; }
            if (sel->count[s]) RUN(pqps_download(sh->ctx, part[s], sh->ids_dev, sel->count[s] * sizeof **part, NULL), "ID download");
        }
        uint64_t w = 0, at[HIP_MAX_SHARDS];
        memset(at, 0, sizeof at);
        for (int k = 0; k < n_probes && rc == 0; k++) {
            const FieldInfo *fi = get_field_info(engine->indexed_attributes[probes[k].index]);
            const uint64_t *end = &seg_end[(size_t)k * (size_t)n_shards];
            for (;;) {
                int best = -1;
                uint64_t best_key = 0;
                for (int s = n_shards - 1; s >= 0; s--) {             /* ties: the higher rows (later shard) first */
                    if (at[s] >= end[s]) continue;
                    const uint64_t key = row_key(engine->all_records[part[s][at[s]]], fi);
                    if (best < 0 || key < best_key) { best = s; best_key = key; }
                }
                if (best < 0) break;
                sel->merged[w++] = part[best][at[best]++];
            }
        }
        for (int s = 0; s < n_shards; s++) free(part[s]);
    }
#undef RUN
    for (int s = 0; s < n_shards; s++) shard_pred_free(hipTableShard(t, s), &sp[s]);
    free(seg_end);
    free(probes);
    hipPlanFree(&plan);
    if (rc != 0) { free(sel->merged); sel->merged = NULL; }
    return rc;
}

/* Caller holds the rows lock (shared).  The device phase -- the contexts' scratch and the shards'
 * result buffers serve one query at a time -- runs under the device lock. */
static long long select_ids(struct engineS *engine, struct whereClauseS *whereClause,
                            unsigned int **ids, double *queryTime) {
    struct hipTable *t = engine->record_block;
    const double t0 = now_seconds();
    struct selection sel;
    *ids = NULL;
    hipTableLockDevice(t);
    int rc = run_selection(engine, t, whereClause, &sel);
    unsigned int *out = NULL;
    if (rc == 0 && sel.merged) {
        out = sel.merged;
    } else if (rc == 0) {
        out = malloc((sel.total ? sel.total : 1) * sizeof *out);
        if (!out) { fprintf(stderr, "HIP engine: out of memory for %llu result IDs\n", (unsigned long long)sel.total); rc = -1; }
        uint64_t at = 0;
        for (int s = 0; s < hipTableShards(t) && rc == 0; s++) {
            struct hipTable *sh = hipTableShard(t, s);
            if (sel.count[s] && pqps_download(sh->ctx, out + at, sh->ids_dev, sel.count[s] * sizeof *out, NULL) != PQPS_OK)
                rc = engine_error("ID download");
            at += sel.count[s];
        }
        if (rc != 0) { free(out); out = NULL; }
    }
    hipTableUnlockDevice(t);
    if (queryTime) *queryTime = now_seconds() - t0;
    *ids = out;
    return rc == 0 ? (long long)sel.total : -1;
}

long long executeQuerySelectIdsHIP(struct engineS *engine, struct whereClauseS *whereClause,
                                   unsigned int **ids, double *queryTime) {
    if (!engine || !engine->record_block || !ids) return -1;
    hipTableLockShared(engine->record_block);
    const long long count = select_ids(engine, whereClause, ids, queryTime);
    hipTableUnlock(engine->record_block);
    return count;
}

static int count_rows(struct hipTable *t, struct whereClauseS *whereClause, uint64_t *total) {
    struct hipPlan plan;
    if (bind_where(t, whereClause, &plan) != 0) return -1;
    const int n_shards = hipTableShards(t);
    struct shard_pred sp[HIP_MAX_SHARDS];
    memset(sp, 0, sizeof sp);
    int rc = 0;
    *total = 0;
    for (int s = 0; s < n_shards && rc == 0; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        rc = shard_pred_prepare(sh, &plan, &sp[s]);
        if (rc == 0 && pqps_filter_count(sh->ctx, sp[s].cols, sp[s].n_cols, sh->n_rows, sp[s].pred, sh->count_dev, NULL) != PQPS_OK)
            rc = engine_error("count filter");
    }
    for (int s = 0; s < n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        uint64_t count = 0;
        if (rc == 0 && pqps_ctx_sync(sh->ctx, NULL) != PQPS_OK) rc = engine_error("filter execution");
        if (rc == 0 && pqps_download(sh->ctx, &count, sh->count_dev, sizeof count, NULL) != PQPS_OK) rc = engine_error("count download");
        *total += count;
        shard_pred_free(sh, &sp[s]);
    }
    hipPlanFree(&plan);
    return rc;
}

long long executeQueryCountHIP(struct engineS *engine, struct whereClauseS *whereClause) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    uint64_t total = 0;
    hipTableLockShared(t);
    hipTableLockDevice(t);
    const int rc = count_rows(t, whereClause, &total);
    hipTableUnlockDevice(t);
    hipTableUnlock(t);
    return rc == 0 ? (long long)total : -1;
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

struct resultSetS *executeQuerySelectHIP(struct engineS *engine, const char **selectItems, int numSelectItems,
                                         const char *tableName, struct whereClauseS *whereClause) {
    (void)tableName;                                   /* never checked by the reference either */
    struct resultSetS *rs = malloc(sizeof *rs);
    if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    memset(rs, 0, sizeof *rs);

    unsigned int *ids = NULL;
    double qtime = 0.0;
    if (!engine || !engine->record_block) { rs->success = false; return rs; }
    hipTableLockShared(engine->record_block);          /* the projection below reads the host rows */
    const long long count = select_ids(engine, whereClause, &ids, &qtime);
    if (count < 0) {                                   /* reason already on stderr */
        hipTableUnlock(engine->record_block);
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
    hipTableUnlock(engine->record_block);
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
        local[v] = (uint32_t)k++;
    }
    for (uint64_t i = 0; i < count; i++) wide[i] = local[wide[i]];
    free(local);
    res->dictionaries[j] = (const char *const *)values;
    res->dictionarySizes[j] = k;
    return 0;
}

/* Device gather of one column for the selected rows of every shard, into `raw` (result order). */
static int project_column(struct hipTable *t, const struct selection *sel, int c, void *const *gathered,
                          unsigned int *const *sub_pos, char *raw, char *tmp) {
    const uint32_t w = t->col[c].width;
    uint64_t at = 0;
    for (int s = 0; s < hipTableShards(t); s++) {
        struct hipTable *sh = hipTableShard(t, s);
        const uint64_t k = sel->count[s];
        if (k == 0) continue;
        TRY(pqps_project_column(sh->ctx, &sh->col[c], sh->ids_dev, sh->count_dev, k, (uint32_t)sh->row0, gathered[s], NULL), "device projection");
        if (!sel->merged) {
            TRY(pqps_download(sh->ctx, raw + at * w, gathered[s], k * w, NULL), "projection download");
        } else {                                                        /* rows of this shard are scattered over the merged order */
            TRY(pqps_download(sh->ctx, tmp, gathered[s], k * w, NULL), "projection download");
            for (uint64_t i = 0; i < k; i++) memcpy(raw + (size_t)sub_pos[s][i] * w, tmp + i * w, w);
        }
        at += k;
    }
    return 0;
}

static int select_columnar(struct engineS *engine, struct hipTable *t, const char **selectItems, int numSelectItems,
                           struct whereClauseS *whereClause, struct hipColumnarResult *res) {
    struct selection sel;
    if (run_selection(engine, t, whereClause, &sel) != 0) return -1;    /* IDs stay in the shards' ids_dev, the counts in count_dev */
    const int n_shards = hipTableShards(t);
    const uint64_t count = sel.total;
    struct hipSchema schema;
    hipSchemaOfTable(t, &schema);
    void *gathered[HIP_MAX_SHARDS];
    unsigned int *sub_pos[HIP_MAX_SHARDS];
    memset(gathered, 0, sizeof gathered);
    memset(sub_pos, 0, sizeof sub_pos);
    char *tmp = NULL;
    int rc = 0;
    if (sel.merged) {
        /* index mode over several shards: each shard gathers for its own rows in merged order (its sub-list
         * replaces the shard-order list on the device), the values are scattered to their merged positions */
        uint64_t fill[HIP_MAX_SHARDS], biggest = 0;
        memset(fill, 0, sizeof fill);
        unsigned int *sub_ids[HIP_MAX_SHARDS];
        memset(sub_ids, 0, sizeof sub_ids);
        for (int s = 0; s < n_shards; s++) {
            sub_ids[s] = malloc((sel.count[s] ? sel.count[s] : 1) * sizeof **sub_ids);
            sub_pos[s] = malloc((sel.count[s] ? sel.count[s] : 1) * sizeof **sub_pos);
            if (!sub_ids[s] || !sub_pos[s]) rc = -1;
            if (sel.count[s] > biggest) biggest = sel.count[s];
        }
        for (uint64_t i = 0; i < count && rc == 0; i++) {
            const unsigned int id = sel.merged[i];
            int s = n_shards - 1;
            while (s > 0 && id < hipTableShard(t, s)->row0) s--;
            sub_ids[s][fill[s]] = id;
            sub_pos[s][fill[s]++] = (unsigned int)i;
        }
        for (int s = 0; s < n_shards && rc == 0; s++) {
            struct hipTable *sh = hipTableShard(t, s);
            if (sel.count[s] && pqps_upload(sh->ctx, sh->ids_dev, sub_ids[s], sel.count[s] * sizeof **sub_ids, NULL) != PQPS_OK)
                rc = engine_error("ID upload");
        }
        for (int s = 0; s < n_shards; s++) free(sub_ids[s]);
        tmp = malloc((biggest ? biggest : 1) * 8);
        if (!tmp) rc = -1;
    }
    for (int s = 0; s < n_shards && rc == 0; s++)
        if (sel.count[s] && pqps_malloc(hipTableShard(t, s)->ctx, sel.count[s] * 8, &gathered[s]) != PQPS_OK) rc = engine_error("projection buffer");
    for (int j = 0; j < numSelectItems && rc == 0; j++) {
        res->columnNames[j] = strdup(selectItems[j]);
        const int c = hipColumnId(selectItems[j]);
        res->columnKinds[j] = c < 0 ? -1 : schema.col[c].kind;
        if (c < 0 || count == 0) continue;
        const uint32_t w = t->col[c].width;
        char *raw = malloc(count * w);
        if (!raw) { fprintf(stderr, "HIP engine: out of memory for the result set\n"); rc = -1; break; }
        res->values[j] = raw;
        rc = project_column(t, &sel, c, gathered, sub_pos, raw, tmp);
        if (rc == 0 && schema.col[c].kind == HIPKIND_DICT)
            rc = own_dictionary(res, j, &t->dict[c], w, count);
    }
    for (int s = 0; s < n_shards; s++) {
        if (gathered[s]) pqps_free(hipTableShard(t, s)->ctx, gathered[s]);
        free(sub_pos[s]);
    }
    free(tmp);
    free(sel.merged);
    res->numRecords = (int)count;
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
    hipTableLockDevice(t);
    const int rc = select_columnar(engine, t, selectItems, numSelectItems, whereClause, res);
    hipTableUnlockDevice(t);
    hipTableUnlock(t);
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
    struct selection sel;
    struct engineS none;                                /* no indexes: always the scan path */
    memset(&none, 0, sizeof none);
    if (run_selection(&none, t, whereClause, &sel) != 0) return -1;
    const uint64_t count = sel.total;
    uint32_t *ids = malloc((count ? count : 1) * sizeof *ids);
    record **hit = malloc((count ? count : 1) * sizeof *hit);
    if (!ids || !hit) { free(ids); free(hit); fprintf(stderr, "HIP engine: out of memory for results\n"); return -1; }
    if (count && pqps_download(t->ctx, ids, t->ids_dev, count * sizeof *ids, NULL) != PQPS_OK) { free(ids); free(hit); return engine_error("ID download"); }
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
    TRACE("init: %d rows, CSV -> rows %.1f ms, rows -> device columns %.1f ms, %d indexes %.1f ms\n", engine->num_records,
          (t1 - t0) * 1e3, (t2 - t1) * 1e3, num_indexes, (now_seconds() - t2) * 1e3);
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
    hipTableLockExclusive(engine->record_block);
    const bool ok = makeIndexHIP(engine, attributeName, attributeType);
    hipTableUnlock(engine->record_block);
    return ok;
}

int hipEngineShards(struct engineS *engine, unsigned long long *rows, int capacity) {
    if (!engine || !engine->record_block) return 0;
    struct hipTable *t = engine->record_block;
    hipTableLockShared(t);
    const int n = hipTableShards(t);
    for (int s = 0; rows && s < n && s < capacity; s++) rows[s] = hipTableShard(t, s)->n_rows;
    hipTableUnlock(t);
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
    hipTableLockExclusive(t);
    const size_t n = (size_t)engine->num_records;
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
        if (!block || !rows) { hipTableUnlock(t); return false; }
        t->row_capacity = cap;
    }
    FILE *f = fopen(engine->datafile, "a");
    if (!f) { hipTableUnlock(t); return false; }
    write_csv_row(f, r);
    fclose(f);

    t->row_block[n] = *r;
    engine->all_records[n] = &t->row_block[n];
    engine->num_records = (int)(n + 1);
    const double t1 = now_seconds();
    appendRowDeviceTableHIP(engine);
    TRACE("INSERT: CSV append + host row %.3f ms, device append + indexes %.3f ms\n", (t1 - t0) * 1e3, (now_seconds() - t1) * 1e3);
    hipTableUnlock(t);
    return true;
}

/* executeQueryDeleteSerial, S:627-715: the per-row decision is the GPU flag
 * kernel (the flag-array shape of engine/omp/executeEngine-omp.c:708-732). */
/* Flags of the rows that go, per shard on the device and for the whole table on the host. */
static int delete_flags(struct hipTable *t, struct whereClauseS *whereClause, size_t n, uint8_t **flags_dev, uint8_t *flags) {
    struct hipPlan plan;
    if (bind_where(t, whereClause, &plan) != 0) return -1;
    const int n_shards = hipTableShards(t);
    struct shard_pred sp[HIP_MAX_SHARDS];
    memset(sp, 0, sizeof sp);
    int rc = 0;
    for (int s = 0; s < n_shards && rc == 0; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        rc = shard_pred_prepare(sh, &plan, &sp[s]);
        if (rc == 0 && pqps_malloc(sh->ctx, sh->capacity_rows, (void **)&flags_dev[s]) != PQPS_OK) rc = engine_error("flag allocation");
        if (rc == 0 && pqps_filter_flags(sh->ctx, sp[s].cols, sp[s].n_cols, sh->n_rows, sp[s].pred, flags_dev[s], sh->count_dev, NULL) != PQPS_OK)
            rc = engine_error("flag filter");
    }
    for (int s = 0; s < n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        if (rc == 0 && pqps_ctx_sync(sh->ctx, NULL) != PQPS_OK) rc = engine_error("filter execution");
        if (rc == 0 && sh->row0 + sh->n_rows > n) { fprintf(stderr, "HIP engine: device shards hold more rows than the host\n"); rc = -1; }
        if (rc == 0 && sh->n_rows && pqps_download(sh->ctx, flags + sh->row0, flags_dev[s], sh->n_rows, NULL) != PQPS_OK) rc = engine_error("flag download");
        shard_pred_free(sh, &sp[s]);
    }
    hipPlanFree(&plan);
    return rc;
}

struct resultSetS *executeQueryDeleteHIP(struct engineS *engine, const char *tableName,
                                         struct whereClauseS *whereClause) {
    (void)tableName;
    struct resultSetS *rs = calloc(1, sizeof *rs);
    if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    const double t0 = now_seconds();
    struct hipTable *t = engine->record_block;
    hipTableLockExclusive(t);
    const size_t n = (size_t)engine->num_records;
    const int n_shards = hipTableShards(t);
    uint8_t *flags_dev[HIP_MAX_SHARDS];
    memset(flags_dev, 0, sizeof flags_dev);
    uint8_t *flags = malloc(n ? n : 1);
    if (!flags || delete_flags(t, whereClause, n, flags_dev, flags) != 0) {
        if (!flags) fprintf(stderr, "HIP engine: out of memory for %zu delete flags\n", n);
        for (int s = 0; s < n_shards; s++) if (flags_dev[s]) pqps_free(hipTableShard(t, s)->ctx, flags_dev[s]);
        free(flags);
        hipTableUnlock(t);
        rs->success = false;                                       /* nothing was deleted */
        return rs;
    }
    const double t1 = now_seconds();

    size_t keep = 0, deleted = 0;
    record *block = t->row_block;
    for (size_t i = 0; i < n; i++) {
        if (flags[i]) { deleted++; continue; }
        if (keep != i) block[keep] = block[i];
        keep++;
    }
    free(flags);
    for (size_t i = 0; i < keep; i++) engine->all_records[i] = &block[i];
    engine->num_records = (int)keep;
    const double t2 = now_seconds();

    FILE *f = fopen(engine->datafile, "w");                       /* S:683-701: no header written */
    if (f) {
        write_csv_table(f, block, keep);
        fclose(f);
    }
    const double t3 = now_seconds();
    /* device side: the same flags compact the 12 columns of every shard in place (order kept); dictionaries
     * stay as they are (a code without rows is harmless), indexes are re-sorted */
    if (deleted) compactDeviceTableHIP(engine, flags_dev, keep);
    for (int s = 0; s < n_shards; s++) pqps_free(hipTableShard(t, s)->ctx, flags_dev[s]);
    TRACE("DELETE: %zu of %zu rows, flags %.3f ms, host rows %.3f ms, CSV rewrite %.3f ms, device compaction + indexes %.3f ms\n",
          deleted, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now_seconds() - t3) * 1e3);
    hipTableUnlock(t);
    rs->numRecords = (int)deleted;
    rs->queryTime = now_seconds() - t0;
    rs->success = true;
    return rs;
}
