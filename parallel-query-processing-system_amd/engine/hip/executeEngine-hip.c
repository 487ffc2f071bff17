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

static void engine_die(const char *what) {
    fprintf(stderr, "HIP engine: %s: %s\n", what, pqps_last_error());
    exit(EXIT_FAILURE);
}

#define SHIM(call, what) do { if ((call) != PQPS_OK) engine_die(what); } while (0)

/* ---- predicate binding --------------------------------------------------------- */

struct bound_pred {
    pqps_predicate pred;
    pqps_column cols[PQPS_MAX_COLUMNS];
    uint32_t n_cols;
};

static void bind_where(const struct hipTable *t, const struct whereClauseS *where, struct bound_pred *bp) {
    struct hipSchema schema;
    int col_ids[PQPS_MAX_COLUMNS];
    char err[160];
    hipSchemaOfTable(t, &schema);
    if (hipCompileWhere(&schema, where, &bp->pred, col_ids, err, sizeof err) != 0) {
        fprintf(stderr, "HIP engine: cannot compile WHERE clause: %s\n", err);
        exit(EXIT_FAILURE);
    }
    bp->n_cols = bp->pred.n_columns;
    for (uint32_t i = 0; i < bp->n_cols; i++) bp->cols[i] = t->col[col_ids[i]];
}

static void ensure_id_capacity(struct hipTable *t, uint64_t need) {
    if (need <= t->capacity_ids) return;
    pqps_free(t->ctx, t->ids_dev);
    t->capacity_ids = need + need / 8 + 1024;
    SHIM(pqps_malloc(t->ctx, t->capacity_ids * sizeof(uint32_t), (void **)&t->ids_dev), "result allocation");
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

/* Row selection of executeQuerySelectSerial, S:358-474, on the device.
 * Returns the number of result rows; the IDs are left in t->ids_dev. */
static uint64_t run_selection(struct engineS *engine, struct hipTable *t, struct whereClauseS *where) {
    struct bound_pred bp;
    bind_where(t, where, &bp);
    uint64_t *count_dev = t->count_dev, *range_dev = t->count_dev + 2;

    for (;;) {
        bool any_index = false;
        for (struct whereClauseS *wc = where; wc; wc = wc->next) {
            if (wc->attribute == NULL) continue;                       /* nested node, S:361-364 */
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
                if (!any_index) SHIM(pqps_memset(t->ctx, count_dev, 0, sizeof(uint64_t), NULL), "counter reset");
                any_index = true;
                SHIM(pqps_index_probe(t->ctx, ix->keys_dev, t->col[ix->column].width, ix->key_kind,
                                      t->n_rows, lo, hi, range_dev, NULL), "index probe");
                /* append the probe's rows that pass the complete WHERE, leaf order kept (S:441-448 + S:471) */
                SHIM(pqps_filter_gather(t->ctx, bp.cols, bp.n_cols, ix->perm_dev, range_dev, t->n_rows, 0,
                                        &bp.pred, t->ids_dev, t->capacity_ids, count_dev, NULL), "index filter");
            }
        }
        if (!any_index)                                                /* full scan, S:464-467 */
            SHIM(pqps_filter_scan(t->ctx, bp.cols, bp.n_cols, t->n_rows, 0, &bp.pred,
                                  t->ids_dev, t->capacity_ids, count_dev, NULL), "scan filter");
        SHIM(pqps_ctx_sync(t->ctx, NULL), "filter execution");
        uint64_t count = 0;
        SHIM(pqps_download(t->ctx, &count, count_dev, sizeof count, NULL), "count download");
        if (count <= t->capacity_ids) return count;
        ensure_id_capacity(t, count);                                  /* duplicates can exceed n rows: retry larger */
    }
}

/* Caller holds the rows lock (shared).  The device phase -- the context's scratch and the table's
 * result buffer serve one query at a time -- runs under the device lock. */
static long long select_ids(struct engineS *engine, struct whereClauseS *whereClause,
                            unsigned int **ids, double *queryTime) {
    struct hipTable *t = engine->record_block;
    const double t0 = now_seconds();
    hipTableLockDevice(t);
    const uint64_t count = run_selection(engine, t, whereClause);
    unsigned int *out = malloc((count ? count : 1) * sizeof *out);
    if (!out) { perror("Failed to allocate memory for result IDs"); exit(EXIT_FAILURE); }
    if (count) SHIM(pqps_download(t->ctx, out, t->ids_dev, count * sizeof *out, NULL), "ID download");
    hipTableUnlockDevice(t);
    if (queryTime) *queryTime = now_seconds() - t0;
    *ids = out;
    return (long long)count;
}

long long executeQuerySelectIdsHIP(struct engineS *engine, struct whereClauseS *whereClause,
                                   unsigned int **ids, double *queryTime) {
    if (!engine || !engine->record_block || !ids) return -1;
    hipTableLockShared(engine->record_block);
    const long long count = select_ids(engine, whereClause, ids, queryTime);
    hipTableUnlock(engine->record_block);
    return count;
}

long long executeQueryCountHIP(struct engineS *engine, struct whereClauseS *whereClause) {
    if (!engine || !engine->record_block) return -1;
    struct hipTable *t = engine->record_block;
    hipTableLockShared(t);
    struct bound_pred bp;
    bind_where(t, whereClause, &bp);
    hipTableLockDevice(t);
    SHIM(pqps_filter_count(t->ctx, bp.cols, bp.n_cols, t->n_rows, &bp.pred, t->count_dev, NULL), "count filter");
    SHIM(pqps_ctx_sync(t->ctx, NULL), "filter execution");
    uint64_t count = 0;
    SHIM(pqps_download(t->ctx, &count, t->count_dev, sizeof count, NULL), "count download");
    hipTableUnlockDevice(t);
    hipTableUnlock(t);
    return (long long)count;
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
    if (!res->columnNames || !res->columnKinds || !res->values || !res->dictionaries) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }

    const double t0 = now_seconds();
    hipTableLockShared(t);
    hipTableLockDevice(t);
    const uint64_t count = run_selection(engine, t, whereClause);      /* IDs stay in t->ids_dev, the count in t->count_dev */
    struct hipSchema schema;
    hipSchemaOfTable(t, &schema);
    void *gathered = NULL;
    if (count) SHIM(pqps_malloc(t->ctx, count * 8, &gathered), "projection buffer");
    for (int j = 0; j < numSelectItems; j++) {
        res->columnNames[j] = strdup(selectItems[j]);
        const int c = hipColumnId(selectItems[j]);
        res->columnKinds[j] = c < 0 ? -1 : schema.col[c].kind;
        if (c < 0 || count == 0) continue;
        const uint32_t w = t->col[c].width;
        SHIM(pqps_project_column(t->ctx, &t->col[c], t->ids_dev, t->count_dev, count, 0, gathered, NULL), "device projection");
        void *raw = malloc(count * w);
        if (!raw) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
        SHIM(pqps_download(t->ctx, raw, gathered, count * w, NULL), "projection download");
        if (schema.col[c].kind == HIPKIND_DICT) {                       /* codes are widened to u32 whatever the column stores */
            res->dictionaries[j] = schema.col[c].dict;
            if (w != 4) {
                uint32_t *wide = malloc(count * sizeof *wide);
                if (!wide) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
                if (w == 1) for (uint64_t i = 0; i < count; i++) wide[i] = ((const uint8_t *)raw)[i];
                else for (uint64_t i = 0; i < count; i++) wide[i] = ((const uint16_t *)raw)[i];
                free(raw);
                raw = wide;
            }
        }
        res->values[j] = raw;
    }
    if (gathered) pqps_free(t->ctx, gathered);
    hipTableUnlockDevice(t);
    hipTableUnlock(t);
    res->numRecords = (int)count;
    res->queryTime = now_seconds() - t0;
    res->success = true;
    TRACE("SELECT (columnar): %d rows x %d columns in %.3f ms\n", res->numRecords, res->numColumns, res->queryTime * 1e3);
    return res;
}

void freeColumnarResultHIP(struct hipColumnarResult *res) {
    if (!res) return;
    for (int j = 0; j < res->numColumns; j++) {
        if (res->columnNames) free(res->columnNames[j]);
        if (res->values) free(res->values[j]);
    }
    free(res->columnNames); free(res->columnKinds); free(res->values); free((void *)res->dictionaries);
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
        SHIM(pqps_ctx_create(device, &g_adhoc_ctx), "cannot create a device context");
    }
    return g_adhoc_ctx;
}

/* linearSearchRecords, S:854-878: the rows are columnarised, filtered on the
 * GPU (input order kept) and the surviving pointers returned. */
record **linearSearchRecords(record **records, int num_records, struct whereClauseS *whereClause,
                             int *matchingRecords) {
    *matchingRecords = 0;
    pthread_mutex_lock(&g_adhoc_lock);
    struct hipTable *t = hipTableFromRows(adhoc_ctx(), records, (size_t)(num_records > 0 ? num_records : 0));
    struct bound_pred bp;
    bind_where(t, whereClause, &bp);
    SHIM(pqps_filter_scan(t->ctx, bp.cols, bp.n_cols, t->n_rows, 0, &bp.pred,
                          t->ids_dev, t->capacity_ids, t->count_dev, NULL), "scan filter");
    SHIM(pqps_ctx_sync(t->ctx, NULL), "filter execution");
    uint64_t count = 0;
    SHIM(pqps_download(t->ctx, &count, t->count_dev, sizeof count, NULL), "count download");
    uint32_t *ids = malloc((count ? count : 1) * sizeof *ids);
    record **out = malloc((count ? count : 1) * sizeof *out);
    if (!ids || !out) { perror("Failed to allocate memory for results"); exit(EXIT_FAILURE); }
    if (count) SHIM(pqps_download(t->ctx, ids, t->ids_dev, count * sizeof *ids, NULL), "ID download");
    for (uint64_t i = 0; i < count; i++) out[i] = records[ids[i]];
    free(ids);
    hipTableFree(t, 0);
    pthread_mutex_unlock(&g_adhoc_lock);
    *matchingRecords = (int)count;
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
    FILE *f = fopen(engine->datafile, "a");
    if (!f) { hipTableUnlock(t); return false; }
    write_csv_row(f, r);
    fclose(f);

    const size_t n = (size_t)engine->num_records;
    /* host row store grows geometrically; all_records[] is re-pointed only when the block moved */
    if (n + 1 > t->row_capacity) {
        const size_t cap = n + n / 8 + 64;
        record *block = realloc(t->row_block, cap * sizeof *block);
        record **rows = realloc(engine->all_records, cap * sizeof *rows);
        if (!block || !rows) { hipTableUnlock(t); return false; }
        if (block != t->row_block) for (size_t i = 0; i < n; i++) rows[i] = &block[i];
        t->row_block = block;
        t->row_capacity = cap;
        engine->all_records = rows;
    }
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
struct resultSetS *executeQueryDeleteHIP(struct engineS *engine, const char *tableName,
                                         struct whereClauseS *whereClause) {
    (void)tableName;
    struct resultSetS *rs = calloc(1, sizeof *rs);
    if (!rs) { perror("Failed to allocate memory for result set"); exit(EXIT_FAILURE); }
    const double t0 = now_seconds();
    struct hipTable *t = engine->record_block;
    hipTableLockExclusive(t);
    const size_t n = (size_t)engine->num_records;
    struct bound_pred bp;
    bind_where(t, whereClause, &bp);
    uint8_t *flags_dev = NULL, *flags = malloc(t->capacity_rows);
    SHIM(pqps_malloc(t->ctx, t->capacity_rows, (void **)&flags_dev), "flag allocation");
    SHIM(pqps_filter_flags(t->ctx, bp.cols, bp.n_cols, t->n_rows, &bp.pred, flags_dev, t->count_dev, NULL), "flag filter");
    SHIM(pqps_ctx_sync(t->ctx, NULL), "filter execution");
    if (n) SHIM(pqps_download(t->ctx, flags, flags_dev, n, NULL), "flag download");
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
    /* device side: the same flags compact the 12 columns in place (order kept); dictionaries
     * stay as they are (a code without rows is harmless), indexes are re-sorted */
    if (deleted) compactDeviceTableHIP(engine, flags_dev, keep);
    pqps_free(t->ctx, flags_dev);
    TRACE("DELETE: %zu of %zu rows, flags %.3f ms, host rows %.3f ms, CSV rewrite %.3f ms, device compaction + indexes %.3f ms\n",
          deleted, n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (now_seconds() - t3) * 1e3);
    hipTableUnlock(t);
    rs->numRecords = (int)deleted;
    rs->queryTime = now_seconds() - t0;
    rs->success = true;
    return rs;
}
