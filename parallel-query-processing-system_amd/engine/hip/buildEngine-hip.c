/* buildEngine-hip.c -- load step of the HIP engine.
 *
 * CSV -> contiguous `record` block (host row store, used for projection) ->
 * per-column arrays -> device buffers.  String columns become
 * order-preserving dictionary codes (rank in strcmp order) so that the filter
 * kernel only ever compares integers.
 *
 * Replaces, in the reference: getAllRecordsFromFile / getRecordFromLine /
 * parseCSVField (engine/serial/buildEngine-serial.c:70-221, same CSV rules),
 * loadIntoBplusTree / makeIndexSerial (:13-62, as a device sort) and the
 * whole-file replication of buildEngine-mpi.c:71-127 (each GPU gets columns,
 * not records).
 */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include "buildEngine-hip.h"
#include "hipPredicate.h"

#include <errno.h>
#include <malloc.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

/* ---- CSV ------------------------------------------------------------------ */

/* One field, rules of parseCSVField (buildEngine-serial.c:111-151): absent
 * when the cursor is at end of line; ends at an unquoted comma (consumed) or
 * NUL/LF/CR; "" inside quotes is a quote; text after the closing quote is kept. */
static int csv_field(const char **cursor, char *out) {
    const char *p = *cursor;
    if (*p == '\0' || *p == '\n' || *p == '\r') return 0;
    size_t n = 0;
    int quoted = 0;
    if (*p == '"') { quoted = 1; p++; }
    for (;;) {
        const char ch = *p;
        if (ch == '\0' || ch == '\n' || ch == '\r') break;
        if (quoted) {
            if (ch != '"') { out[n++] = ch; p++; }
            else if (p[1] == '"') { out[n++] = '"'; p += 2; }
            else { quoted = 0; p++; }
        } else if (ch == ',') {
            p++;
            break;
        } else {
            out[n++] = ch;
            p++;
        }
    }
    out[n] = '\0';
    *cursor = p;
    return 1;
}

static bool csv_bool(const char *t) { return strcasecmp(t, "true") == 0 || strcmp(t, "1") == 0; }

/* getRecordFromLine, buildEngine-serial.c:159-221. */
void fillRecordFromLineHIP(record *dst, const char *line) {
    char tok[1100];
    const char *cur = line;
    memset(dst, 0, sizeof *dst);
#define STR_FIELD(f) if (csv_field(&cur, tok)) strncpy(dst->f, tok, sizeof dst->f)
    if (csv_field(&cur, tok)) dst->command_id = strtoull(tok, NULL, 10);
    STR_FIELD(raw_command);
    STR_FIELD(base_command);
    STR_FIELD(shell_type);
    if (csv_field(&cur, tok)) dst->exit_code = atoi(tok);
    STR_FIELD(timestamp);
    if (csv_field(&cur, tok)) dst->sudo_used = csv_bool(tok);
    STR_FIELD(working_directory);
    if (csv_field(&cur, tok)) dst->user_id = atoi(tok);
    STR_FIELD(user_name);
    STR_FIELD(host_name);
    if (csv_field(&cur, tok)) dst->risk_level = atoi(tok);
#undef STR_FIELD
}

record *getRecordFromLineHIP(char *line) {
    record *r = malloc(sizeof *r);
    if (!r) { fprintf(stderr, "Memory allocation failed\n"); return NULL; }
    fillRecordFromLineHIP(r, line);
    return r;
}

/* ---- small fork-join helper (pthreads) -------------------------------------------- */
struct par_task { void (*fn)(void *, size_t, size_t); void *arg; size_t begin, end; };

static void *par_entry(void *p) {
    struct par_task *t = p;
    t->fn(t->arg, t->begin, t->end);
    return NULL;
}

static int host_threads(void) {
    const char *env = getenv("PQPS_HOST_THREADS");
    long n = env ? atol(env) : sysconf(_SC_NPROCESSORS_ONLN);
    if (n < 1) n = 1;
    if (n > 32) n = 32;
    return (int)n;
}

/* fn(arg, begin, end) over [0, n) split into contiguous ranges, one per thread */
static void parallel_for(size_t n, void (*fn)(void *, size_t, size_t), void *arg) {
    int nt = host_threads();
    if ((size_t)nt > n / 1024 + 1) nt = (int)(n / 1024 + 1);
    if (nt <= 1) { fn(arg, 0, n); return; }
    pthread_t tid[32];
    struct par_task task[32];
    for (int i = 0; i < nt; i++) {
        task[i].fn = fn; task[i].arg = arg;
        task[i].begin = n * (size_t)i / (size_t)nt;
        task[i].end = n * (size_t)(i + 1) / (size_t)nt;
        if (pthread_create(&tid[i], NULL, par_entry, &task[i]) != 0) { fn(arg, task[i].begin, task[i].end); tid[i] = 0; }
    }
    for (int i = 0; i < nt; i++) if (tid[i]) pthread_join(tid[i], NULL);
}

/* getAllRecordsFromFile, buildEngine-serial.c:70-108, with the block
 * allocation of the OMP variant (buildEngine-omp.c:84): rows are the fgets()
 * chunks of the file (<= 1023 bytes each, a chunk also ends after '\n') after
 * the first one.  The chunk boundaries are found in one sequential pass, the
 * rows are parsed in parallel (reference analogue: buildEngine-omp.c:157). */
struct parse_job { const char *text; const size_t *start; const uint32_t *len; record *block; };

static void parse_range(void *arg, size_t begin, size_t end) {
    struct parse_job *j = arg;
    char line[1024];
    for (size_t i = begin; i < end; i++) {
        memcpy(line, j->text + j->start[i], j->len[i]);
        line[j->len[i]] = '\0';                          /* fgets() semantics: C string, at most 1023 bytes */
        fillRecordFromLineHIP(&j->block[i], line);
    }
}

record **getAllRecordsFromFileHIP(const char *filepath, int *num_records, void **record_block_out) {
    *num_records = 0;
    if (record_block_out) *record_block_out = NULL;
    FILE *f = fopen(filepath, "r");
    if (!f) {
        fprintf(stderr, "Error opening file: %s\n", filepath);
        return NULL;
    }
    fseek(f, 0, SEEK_END);
    const long fsize = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *text = malloc((size_t)(fsize > 0 ? fsize : 0) + 1);
    size_t got = text ? fread(text, 1, (size_t)(fsize > 0 ? fsize : 0), f) : 0;
    fclose(f);
    if (!text) { fprintf(stderr, "Memory allocation failed\n"); return NULL; }
    text[got] = '\0';

    /* pass 1: chunk table (what successive fgets(line, 1024) calls would return) */
    size_t cap = got / 64 + 16, chunks = 0;
    size_t *start = malloc(cap * sizeof *start);
    uint32_t *len = malloc(cap * sizeof *len);
    for (size_t pos = 0; start && len && pos < got;) {
        size_t l = 0;
        while (l < 1023 && pos + l < got) { const char ch = text[pos + l]; l++; if (ch == '\n') break; }
        if (chunks == cap) {
            cap *= 2;
            start = realloc(start, cap * sizeof *start);
            len = realloc(len, cap * sizeof *len);
            if (!start || !len) break;
        }
        start[chunks] = pos; len[chunks] = (uint32_t)l; chunks++;
        pos += l;
    }
    if (!start || !len) { fprintf(stderr, "Memory allocation failed\n"); free(text); free(start); free(len); return NULL; }
    const size_t n = chunks > 0 ? chunks - 1 : 0;        /* first chunk = header, dropped unconditionally */

    /* head-room for INSERTs (untouched pages cost nothing); buildDeviceTableHIP learns the capacity
     * from malloc_usable_size, so the block is still an ordinary malloc'd block to its owner */
    const size_t cap_rows = n + n / 16 + 1024;
    record *block = malloc(cap_rows * sizeof *block);
    record **rows = malloc(cap_rows * sizeof *rows);
    if (block && n * sizeof *block >= ((size_t)64 << 20)) {
        /* a large row block is its own mapping: ask for huge pages on its 2 MiB-aligned interior before
         * the first touch -- 1 M rows are 1 GB, and faulting in / releasing 260 k small pages costs
         * more than parsing them (release alone: 126 ms, against 2 ms with huge pages) */
        const uintptr_t lo = ((uintptr_t)block + (((uintptr_t)2 << 20) - 1)) & ~(((uintptr_t)2 << 20) - 1);
        const uintptr_t hi = ((uintptr_t)block + n * sizeof *block) & ~(((uintptr_t)2 << 20) - 1);
        if (hi > lo) (void)madvise((void *)lo, hi - lo, MADV_HUGEPAGE);
    }
    if (!block || !rows) { fprintf(stderr, "Memory allocation failed\n"); free(text); free(start); free(len); free(block); free(rows); return NULL; }
    struct parse_job job = { text, start + 1, len + 1, block };
    parallel_for(n, parse_range, &job);
    for (size_t i = 0; i < n; i++) rows[i] = &block[i];
    free(text); free(start); free(len);
    *num_records = (int)n;
    if (record_block_out) *record_block_out = block; else if (n == 0) free(block);
    return rows;
}

FieldType mapAttributeTypeHIP(int attributeType) {
    switch (attributeType) {
    case 0: return FIELD_UINT64;
    case 1: return FIELD_INT;
    case 2: return FIELD_STRING;
    case 3: return FIELD_BOOL;
    default: return (FieldType)-1;
    }
}

/* ---- rows -> columns -> device ------------------------------------------------ */

static const size_t k_offset[HIPCOL_COUNT] = {
    offsetof(record, command_id), offsetof(record, raw_command), offsetof(record, base_command),
    offsetof(record, shell_type), offsetof(record, exit_code), offsetof(record, timestamp),
    offsetof(record, sudo_used), offsetof(record, working_directory), offsetof(record, user_id),
    offsetof(record, user_name), offsetof(record, host_name), offsetof(record, risk_level)
};
static const int k_kind[HIPCOL_COUNT] = {
    HIPKIND_U64, HIPKIND_DICT, HIPKIND_DICT, HIPKIND_DICT, HIPKIND_I32, HIPKIND_DICT,
    HIPKIND_BOOL, HIPKIND_DICT, HIPKIND_I32, HIPKIND_DICT, HIPKIND_DICT, HIPKIND_I32
};

static int cmp_cstr_ptr_ptr(const void *a, const void *b) {
    return strcmp(**(const char *const *const *)a, **(const char *const *const *)b);
}

static uint64_t hash_cstr(const char *s) {                      /* FNV-1a */
    uint64_t h = 1469598103934665603ull;
    for (; *s; s++) { h ^= (unsigned char)*s; h *= 1099511628211ull; }
    return h;
}

/* Sorted distinct values of one string column + the code (rank) of every row.
 * The "value" of a row is the C string that starts at the field -- exactly what
 * strcmp() in the reference's CMP_STR sees, including the run-on into the next
 * field when strncpy left no terminator (buildEngine-serial.c:171).
 * Distinct values are collected with an open-addressing hash set (one pass over
 * the rows), only the distinct ones are sorted. */
static int build_dictionary(record *const *rows, size_t n, size_t off, struct hipDictionary *d,
                            uint32_t *codes) {
    size_t cap = 1024;
    while (cap < 2 * n + 16 && cap < ((size_t)1 << 31)) cap <<= 1;
    uint32_t *slot = malloc(cap * sizeof *slot);                /* distinct id + 1, 0 = empty */
    const char **first = malloc((n ? n : 1) * sizeof *first);   /* representative of each distinct id */
    if (!slot || !first) { free(slot); free(first); return -1; }
    memset(slot, 0, cap * sizeof *slot);
    size_t distinct = 0, bytes = 0;
    for (size_t i = 0; i < n; i++) {
        const char *s = (const char *)rows[i] + off;
        size_t h = (size_t)hash_cstr(s) & (cap - 1);
        for (;;) {
            const uint32_t v = slot[h];
            if (v == 0) {
                first[distinct] = s;
                bytes += strlen(s) + 1;
                slot[h] = (uint32_t)++distinct;
                codes[i] = (uint32_t)distinct - 1;
                break;
            }
            if (strcmp(first[v - 1], s) == 0) { codes[i] = v - 1; break; }
            h = (h + 1) & (cap - 1);
        }
    }
    free(slot);
    /* rank of every distinct id in strcmp order */
    const char ***order = malloc((distinct ? distinct : 1) * sizeof *order);
    uint32_t *rank = malloc((distinct ? distinct : 1) * sizeof *rank);
    d->count = (int)distinct;
    d->storage = malloc(bytes ? bytes : 1);
    d->storage_bytes = bytes;
    d->values = malloc((distinct ? distinct : 1) * sizeof *d->values);
    if (!order || !rank || !d->storage || !d->values) { free(first); free(order); free(rank); return -1; }
    for (size_t k = 0; k < distinct; k++) order[k] = &first[k];
    qsort(order, distinct, sizeof *order, cmp_cstr_ptr_ptr);
    char *w = d->storage;
    for (size_t r = 0; r < distinct; r++) {
        const size_t id = (size_t)(order[r] - first);
        rank[id] = (uint32_t)r;
        const size_t l = strlen(first[id]) + 1;
        memcpy(w, first[id], l);
        d->values[r] = w;
        w += l;
    }
    for (size_t i = 0; i < n; i++) codes[i] = rank[codes[i]];
    free(order); free(rank); free(first);
    return 0;
}

static void hip_die(const char *what) {
    fprintf(stderr, "HIP engine: %s: %s\n", what, pqps_last_error());
    exit(EXIT_FAILURE);
}

void hipSchemaOfTable(const struct hipTable *t, struct hipSchema *s) {
    memset(s, 0, sizeof *s);
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        s->col[c].present = t->col[c].data != NULL || t->n_rows == 0 || (k_kind[c] == HIPKIND_DICT && t->dict[c].count == 1);
        s->col[c].kind = k_kind[c];
        s->col[c].width = t->col[c].width;
        s->col[c].dict_count = t->dict[c].count;
        s->col[c].dict = t->dict[c].values;
    }
}

/* One column: gather the field of every row into its staging buffer (dictionary
 * columns: build the dictionary first).  Columns are independent -> one thread each. */
struct col_job { record *const *rows; size_t n, cap_rows; struct hipTable *t; void *stage[HIPCOL_COUNT]; int failed; };

static void stage_columns(void *arg, size_t begin, size_t end) {
    struct col_job *j = arg;
    const size_t n = j->n;
    for (size_t c = begin; c < end; c++) {
        void *stage = j->stage[c];
        switch (k_kind[c]) {
        case HIPKIND_U64:
            j->t->col[c].width = 8;
            for (size_t i = 0; i < n; i++) ((uint64_t *)stage)[i] = *(const uint64_t *)((const char *)j->rows[i] + k_offset[c]);
            break;
        case HIPKIND_I32:
            j->t->col[c].width = 4;
            for (size_t i = 0; i < n; i++) ((int32_t *)stage)[i] = *(const int *)((const char *)j->rows[i] + k_offset[c]);
            break;
        case HIPKIND_BOOL:
            j->t->col[c].width = 1;
            for (size_t i = 0; i < n; i++) ((uint8_t *)stage)[i] = *(const bool *)((const char *)j->rows[i] + k_offset[c]) ? 1 : 0;
            break;
        default: {
            uint32_t *codes = malloc((n ? n : 1) * sizeof *codes);
            if (!codes || build_dictionary(j->rows, n, k_offset[c], &j->t->dict[c], codes) != 0) { j->failed = 1; free(codes); break; }
            const uint32_t width = j->t->dict[c].count <= 256 ? 1 : j->t->dict[c].count <= 65536 ? 2 : 4;
            j->t->col[c].width = width;
            for (size_t i = 0; i < n; i++) {
                if (width == 1) ((uint8_t *)stage)[i] = (uint8_t)codes[i];
                else if (width == 2) ((uint16_t *)stage)[i] = (uint16_t)codes[i];
                else ((uint32_t *)stage)[i] = codes[i];
            }
            free(codes);
            break;
        }
        }
    }
}

static void *col_thread(void *p) {
    struct par_task *t = p;
    t->fn(t->arg, t->begin, t->end);
    return NULL;
}

/* The HIP runtime takes ~0.2 s to come up; an engine that is about to parse a CSV starts that in the
 * background (hipBeginContextHIP) and collects the context when the first device call is due. */
struct hipContextFuture {
    pthread_t tid; bool threaded;
    int n; int device[HIP_MAX_SHARDS]; pqps_ctx *ctx[HIP_MAX_SHARDS];
    int rc; char err[256];
};

static void *context_thread(void *p) {
    struct hipContextFuture *f = p;
    for (int i = 0; i < f->n; i++) {
        f->rc = pqps_ctx_create(f->device[i], &f->ctx[i]);
        if (f->rc != PQPS_OK) {
            snprintf(f->err, sizeof f->err, "device %d: %s", f->device[i], pqps_last_error());   /* last_error is per thread */
            for (int k = 0; k < i; k++) { pqps_ctx_destroy(f->ctx[k]); f->ctx[k] = NULL; }
            break;
        }
    }
    return NULL;
}

/* PQPS_DEVICES=0,1,... : one shard of the table per listed device (a device may be listed more than once:
 * its shards then share the card).  Otherwise PQPS_DEVICE (default 0) holds the whole table. */
static int device_list(int *device) {
    const char *list = getenv("PQPS_DEVICES");
    int n = 0;
    if (list && *list) {
        const char *p = list;
        while (*p && n < HIP_MAX_SHARDS) {
            char *end = NULL;
            const long d = strtol(p, &end, 10);
            if (end == p) break;
            device[n++] = (int)d;
            p = end;
            while (*p == ',' || *p == ' ') p++;
        }
    }
    if (n == 0) {
        const char *env = getenv("PQPS_DEVICE");
        device[n++] = env ? atoi(env) : 0;
    }
    return n;
}

struct hipContextFuture *hipBeginContextHIP(void) {
    struct hipContextFuture *f = calloc(1, sizeof *f);
    if (!f) { perror("Failed to allocate memory for device start-up"); exit(EXIT_FAILURE); }
    f->n = device_list(f->device);
    f->threaded = pthread_create(&f->tid, NULL, context_thread, f) == 0;
    if (!f->threaded) context_thread(f);
    return f;
}

static void await_contexts(struct hipContextFuture *f) {
    if (f->threaded) { pthread_join(f->tid, NULL); f->threaded = false; }
    if (f->rc != PQPS_OK) {
        fprintf(stderr, "HIP engine: cannot create a device context: %s\n", f->err);
        exit(EXIT_FAILURE);
    }
}

/* Columns of `n` host rows staged on the host (dictionaries built), then uploaded as `n_shards` contiguous
 * row ranges, one per context. */
static void table_fill(struct hipTable *t, pqps_ctx *const *ctxs, int n_shards, struct hipContextFuture *future,
                       record *const *rows, size_t n);

struct hipTable *hipTableFromRows(pqps_ctx *ctx, record *const *rows, size_t n) {
    struct hipTable *t = calloc(1, sizeof *t);
    if (!t) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
    table_fill(t, &ctx, 1, NULL, rows, n);
    return t;
}

/* Block partition of executeEngine-mpi.c:703-715: the first n % parts shards hold one row more. */
static void shard_range(size_t n, int parts, int s, size_t *start, size_t *count) {
    const size_t base = n / (size_t)parts, rem = n % (size_t)parts;
    *start = (size_t)s * base + ((size_t)s < rem ? (size_t)s : rem);
    *count = base + ((size_t)s < rem ? 1 : 0);
}

/* Result buffers of one lane of one shard (`copy`: a context of its own for the lane's downloads; NULL for the
 * table's own buffers). */
static void lane_alloc(struct hipTable *sh, struct hipLane *L, bool own_context) {
    memset(L, 0, sizeof *L);
    L->capacity_ids = sh->capacity_rows;
    if (pqps_malloc(sh->ctx, L->capacity_ids * sizeof(uint32_t), (void **)&L->ids_dev) != PQPS_OK) hip_die("result allocation");
    if (own_context) {
        /* a query lane: its count words live in pinned host memory the device writes to directly */
        void *host = NULL, *dev = NULL;
        if (pqps_malloc_mapped(sh->ctx, 8 * sizeof(uint64_t), &host, &dev) != PQPS_OK) hip_die("counter allocation");
        L->count_host = host;
        L->count_dev = dev;
        if (pqps_ctx_create(pqps_ctx_device(sh->ctx), &L->copy) != PQPS_OK) hip_die("lane context");
    } else if (pqps_malloc(sh->ctx, 8 * sizeof(uint64_t), (void **)&L->count_dev) != PQPS_OK) hip_die("counter allocation");
}

static void lane_free(struct hipTable *sh, struct hipLane *L) {
    if (L->ids_dev) pqps_free(sh->ctx, L->ids_dev);
    if (L->count_host) pqps_free_mapped(sh->ctx, (void *)L->count_host);
    else if (L->count_dev) pqps_free(sh->ctx, L->count_dev);
    if (L->merged_dev) pqps_free(sh->ctx, L->merged_dev);
    if (L->copy) pqps_ctx_destroy(L->copy);
    memset(L, 0, sizeof *L);
}

static int engine_lanes(void) {
    const char *env = getenv("PQPS_ENGINE_LANES");
    const int n = env ? atoi(env) : 4;
    return n < 1 ? 1 : (n > HIP_MAX_LANES ? HIP_MAX_LANES : n);
}

/* Engine tables: `n_lanes` queries in flight per shard -- result buffers each, and a query stream with as many slots. */
static void shard_lanes_create(struct hipTable *sh, int n_lanes) {
    sh->n_lanes = n_lanes;
    for (int k = 0; k < n_lanes; k++) lane_alloc(sh, &sh->lane[k], true);
    if (pqps_qstream_create(sh->ctx, (uint32_t)n_lanes, &sh->qs) != PQPS_OK) hip_die("query stream");
    /* the scan lanes' scratch and list area now, not inside the first ID queries (2.5 bytes per row and lane) */
    if (pqps_qstream_reserve(sh->qs, sh->capacity_rows) != PQPS_OK) hip_die("query stream scratch");
}

/* Buffers of one shard for rows [row0, row0 + count): capacity leaves head-room so that INSERT appends in place; the
 * rows past the last one are zero (the filter reads whole 1024-row steps). */
static void shard_alloc(struct hipTable *sh, pqps_ctx *ctx, const struct hipTable *widths, size_t row0, size_t count) {
    sh->ctx = ctx;
    sh->n_rows = count;
    sh->row0 = row0;
    sh->capacity_rows = (count + count / 16 + PQPS_TILE_ROWS) / PQPS_TILE_ROWS * PQPS_TILE_ROWS;
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        const uint32_t width = widths->col[c].width;
        sh->col[c].data = NULL;
        sh->col[c].width = width;
        if (width == 0) continue;                                 /* single-valued string column */
        void *dev = NULL;
        if (pqps_malloc(ctx, sh->capacity_rows * width, &dev) != PQPS_OK) hip_die("column allocation");
        /* only the tail needs clearing when the caller fills rows [0, count) */
        if (pqps_memset(ctx, (char *)dev + count * width, 0, (sh->capacity_rows - count) * width, NULL) != PQPS_OK) hip_die("column clear");
        sh->col[c].data = dev;
    }
    lane_alloc(sh, &sh->own, false);
    if (pqps_ctx_reserve(ctx, sh->capacity_rows) != PQPS_OK) hip_die("filter scratch allocation");   /* not inside the first query */
}

/* Device side of one shard: rows [row0, row0 + count) of the staged columns. */
static void shard_upload(struct hipTable *sh, pqps_ctx *ctx, const struct hipTable *widths, void *const *stage,
                         size_t row0, size_t count) {
    shard_alloc(sh, ctx, widths, row0, count);
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        const uint32_t width = widths->col[c].width;
        if (count && width && pqps_upload(ctx, (void *)sh->col[c].data, (const char *)stage[c] + row0 * width, count * width, NULL) != PQPS_OK)
            hip_die("column upload");
    }
}

/* The shard structs of a table over `n_shards` contexts (shard 0 = the table itself). */
static void table_make_shards(struct hipTable *t, int n_shards) {
    t->n_shards = 0;
    t->shard = NULL;
    if (n_shards > 1) {
        t->shard = calloc((size_t)n_shards, sizeof *t->shard);
        if (!t->shard) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
        t->n_shards = n_shards;
        t->shard[0] = t;
        for (int s = 1; s < n_shards; s++) {
            t->shard[s] = calloc(1, sizeof **t->shard);
            if (!t->shard[s]) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
        }
    }
}

static void table_fill(struct hipTable *t, pqps_ctx *const *ctxs, int n_shards, struct hipContextFuture *future,
                       record *const *rows, size_t n) {
    struct col_job job;
    memset(&job, 0, sizeof job);
    job.rows = rows; job.n = n; job.cap_rows = n; job.t = t;
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        job.stage[c] = malloc((n ? n : 1) * 8);
        if (!job.stage[c]) { perror("Failed to allocate staging memory"); exit(EXIT_FAILURE); }
    }
    /* one thread per column (12 independent tasks); tiny tables stay on the caller's thread */
    if (n < 4096 || host_threads() == 1) {
        stage_columns(&job, 0, HIPCOL_COUNT);
    } else {
        pthread_t tid[HIPCOL_COUNT];
        struct par_task task[HIPCOL_COUNT];
        for (int c = 0; c < HIPCOL_COUNT; c++) {
            task[c].fn = stage_columns; task[c].arg = &job; task[c].begin = (size_t)c; task[c].end = (size_t)c + 1;
            if (pthread_create(&tid[c], NULL, col_thread, &task[c]) != 0) { stage_columns(&job, (size_t)c, (size_t)c + 1); tid[c] = 0; }
        }
        for (int c = 0; c < HIPCOL_COUNT; c++) if (tid[c]) pthread_join(tid[c], NULL);
    }
    if (job.failed) { perror("Failed to build dictionary"); exit(EXIT_FAILURE); }
    if (future) {                                           /* host staging above ran beside the device start-up */
        await_contexts(future);
        ctxs = future->ctx;
        n_shards = future->n;
    }
    table_make_shards(t, n_shards);
    for (int s = 0; s < n_shards; s++) {
        size_t row0, count;
        shard_range(n, n_shards, s, &row0, &count);
        shard_upload(hipTableShard(t, s), ctxs[s], t, job.stage, row0, count);
    }
    for (int c = 0; c < HIPCOL_COUNT; c++) free(job.stage[c]);
}

static void dictionary_free(struct hipDictionary *d) {
    if (d->values) {
        for (int i = 0; i < d->count; i++) {                    /* values appended by INSERT live outside `storage` */
            const char *v = d->values[i];
            if (!(v >= d->storage && v < d->storage + d->storage_bytes)) free((void *)v);
        }
    }
    free(d->values);
    free(d->storage);
    memset(d, 0, sizeof *d);
}

/* Readers (SELECT / COUNT, each on a lane of its own) and writers (INSERT / DELETE / index creation, alone).  A
 * reader may end on another thread than it began on (asynchronous tickets), which a pthread rwlock does not allow.
 *
 * Rules that keep a caller from waiting for itself (include/executeEngine-hip.h states them for the API's users):
 *   - waiting writers go first, EXCEPT that a thread which already holds a lane (a ticket of its own that is not released
 *     yet) is let in beside them: the writer waits for that thread's tickets, so holding the thread back would hold the
 *     writer back for ever;
 *   - a writer call from a thread that holds a lane is refused (it would wait for its own ticket);
 *   - a thread that asks for a lane while it holds every lane itself is refused at once; any other wait for a lane ends
 *     after PQPS_LANE_WAIT_MS (default 10 000) with a refusal instead of hanging a process that has touched the GPU. */
struct hipLocks {
    pthread_mutex_t m;
    pthread_cond_t cv;
    int readers, writer, writers_waiting;
    unsigned busy_lanes;                 /* bit k: lane k is taken */
    int n_lanes;
    pthread_t lane_owner[HIP_MAX_LANES]; /* the thread that took lane k (valid while its busy bit is set) */
    long lane_wait_ms;
    pthread_mutex_t issue;               /* issuing calls on the shards' query streams */
};

static struct hipLocks *locks_create(int n_lanes) {
    struct hipLocks *l = calloc(1, sizeof *l);
    pthread_condattr_t ca;
    if (!l || pthread_mutex_init(&l->m, NULL) != 0 || pthread_condattr_init(&ca) != 0 ||
        pthread_condattr_setclock(&ca, CLOCK_MONOTONIC) != 0 || pthread_cond_init(&l->cv, &ca) != 0 ||
        pthread_mutex_init(&l->issue, NULL) != 0) {
        perror("Failed to create engine locks");
        exit(EXIT_FAILURE);
    }
    pthread_condattr_destroy(&ca);
    l->n_lanes = n_lanes;
    const char *env = getenv("PQPS_LANE_WAIT_MS");
    l->lane_wait_ms = env && atol(env) >= 0 ? atol(env) : 10000;
    return l;
}

static void locks_destroy(struct hipLocks *l) {
    if (!l) return;
    pthread_mutex_destroy(&l->m);
    pthread_cond_destroy(&l->cv);
    pthread_mutex_destroy(&l->issue);
    free(l);
}

/* Test hooks (tests/c/locks_test.c exercises the gate on a table that has nothing but locks -- no device). */
void hipTableLocksCreate(struct hipTable *t, int n_lanes) { t->locks = locks_create(n_lanes > HIP_MAX_LANES ? HIP_MAX_LANES : n_lanes); }
void hipTableLocksDestroy(struct hipTable *t) { locks_destroy(t->locks); t->locks = NULL; }

/* (caller holds l->m) lanes the calling thread has taken and not given back */
static int lanes_of_self(const struct hipLocks *l) {
    int k, n = 0;
    const pthread_t self = pthread_self();
    for (k = 0; k < l->n_lanes; k++)
        if ((l->busy_lanes & (1u << k)) && pthread_equal(l->lane_owner[k], self)) n++;
    return n;
}

void hipTableLockShared(struct hipTable *t) {
    if (!t || !t->locks) return;
    struct hipLocks *l = t->locks;
    pthread_mutex_lock(&l->m);
    /* writers first (a stream of readers cannot starve them) -- but not in front of a thread they are waiting for */
    while (l->writer || (l->writers_waiting && lanes_of_self(l) == 0)) pthread_cond_wait(&l->cv, &l->m);
    l->readers++;
    pthread_mutex_unlock(&l->m);
}

void hipTableUnlockShared(struct hipTable *t) {
    if (!t || !t->locks) return;
    struct hipLocks *l = t->locks;
    pthread_mutex_lock(&l->m);
    if (--l->readers == 0) pthread_cond_broadcast(&l->cv);
    pthread_mutex_unlock(&l->m);
}

int hipTableLockExclusive(struct hipTable *t) {
    if (!t || !t->locks) return 0;
    struct hipLocks *l = t->locks;
    pthread_mutex_lock(&l->m);
    if (lanes_of_self(l) > 0) {                                   /* would wait for a ticket only this thread can release */
        pthread_mutex_unlock(&l->m);
        fprintf(stderr, "HIP engine: INSERT / DELETE / index calls are refused while the calling thread holds a query ticket (release it first)\n");
        return -1;
    }
    l->writers_waiting++;
    while (l->writer || l->readers) pthread_cond_wait(&l->cv, &l->m);
    l->writers_waiting--;
    l->writer = 1;
    pthread_mutex_unlock(&l->m);
    return 0;
}

void hipTableUnlockExclusive(struct hipTable *t) {
    if (!t || !t->locks) return;
    struct hipLocks *l = t->locks;
    pthread_mutex_lock(&l->m);
    l->writer = 0;
    pthread_cond_broadcast(&l->cv);
    pthread_mutex_unlock(&l->m);
}

int hipTableAcquireLane(struct hipTable *t) {
    if (!t || !t->locks || t->locks->n_lanes == 0) return -1;
    struct hipLocks *l = t->locks;
    struct timespec deadline;
    clock_gettime(CLOCK_MONOTONIC, &deadline);
    deadline.tv_sec += l->lane_wait_ms / 1000;
    deadline.tv_nsec += (l->lane_wait_ms % 1000) * 1000000L;
    if (deadline.tv_nsec >= 1000000000L) { deadline.tv_sec++; deadline.tv_nsec -= 1000000000L; }
    pthread_mutex_lock(&l->m);
    int k;
    for (;;) {
        for (k = 0; k < l->n_lanes; k++) if (!(l->busy_lanes & (1u << k))) break;
        if (k < l->n_lanes) break;
        if (lanes_of_self(l) == l->n_lanes) {                     /* every lane is this thread's own: nobody else can free one */
            pthread_mutex_unlock(&l->m);
            fprintf(stderr, "HIP engine: the calling thread already holds all %d query lanes (PQPS_ENGINE_LANES): release a ticket first\n", l->n_lanes);
            return HIP_LANE_REFUSED;
        }
        if (pthread_cond_timedwait(&l->cv, &l->m, &deadline) == ETIMEDOUT) {
            for (k = 0; k < l->n_lanes; k++) if (!(l->busy_lanes & (1u << k))) break;
            if (k < l->n_lanes) break;
            pthread_mutex_unlock(&l->m);
            fprintf(stderr, "HIP engine: no query lane came free within %ld ms (%d lanes, all held by unreleased tickets)\n", l->lane_wait_ms, l->n_lanes);
            return HIP_LANE_REFUSED;
        }
    }
    l->busy_lanes |= 1u << k;
    l->lane_owner[k] = pthread_self();
    pthread_mutex_unlock(&l->m);
    return k;
}

void hipTableReleaseLane(struct hipTable *t, int lane) {
    if (!t || !t->locks || lane < 0) return;
    struct hipLocks *l = t->locks;
    pthread_mutex_lock(&l->m);
    l->busy_lanes &= ~(1u << lane);
    pthread_cond_broadcast(&l->cv);
    pthread_mutex_unlock(&l->m);
}

int hipTableLaneCount(const struct hipTable *t) { return t && t->locks ? t->locks->n_lanes : 0; }

void hipTableLockIssue(struct hipTable *t) { if (t && t->locks) pthread_mutex_lock(&t->locks->issue); }
void hipTableUnlockIssue(struct hipTable *t) { if (t && t->locks) pthread_mutex_unlock(&t->locks->issue); }

/* Device buffers of one shard. */
static void shard_release(struct hipTable *sh, int n_indexes) {
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        if (sh->col[c].data) pqps_free(sh->ctx, (void *)sh->col[c].data);
        sh->col[c].data = NULL;
    }
    if (sh->index) {
        for (int i = 0; i < n_indexes; i++) {
            if (sh->index[i].perm_dev) pqps_free(sh->ctx, sh->index[i].perm_dev);
            if (sh->index[i].keys_dev) pqps_free(sh->ctx, sh->index[i].keys_dev);
        }
        free(sh->index);
        sh->index = NULL;
    }
    if (sh->qs) { pqps_qstream_destroy(sh->qs); sh->qs = NULL; }
    for (int k = 0; k < sh->n_lanes; k++) lane_free(sh, &sh->lane[k]);
    sh->n_lanes = 0;
    lane_free(sh, &sh->own);
}

/* Frees what the table owns on its devices, its dictionaries and its peer shards -- not the struct
 * itself, not the contexts. */
static void table_release(struct hipTable *t, int n_indexes) {
    for (int s = hipTableShards(t) - 1; s >= 0; s--) {
        struct hipTable *sh = hipTableShard(t, s);
        shard_release(sh, n_indexes);
        if (sh != t) free(sh);
    }
    free(t->shard);
    t->shard = NULL;
    t->n_shards = 0;
    for (int c = 0; c < HIPCOL_COUNT; c++) dictionary_free(&t->dict[c]);
}

void hipTableFree(struct hipTable *t, int n_indexes) {
    if (!t) return;
    if (t->xch) { pqps_exchange_destroy(t->xch); t->xch = NULL; }   /* before the contexts its lanes were made on */
    table_release(t, n_indexes);
    locks_destroy(t->locks);
    free(t);
}

/* Device index of one shard for engine->indexed_attributes[slot]: row numbers are shard-local. */
static bool build_index(struct engineS *engine, struct hipTable *t, int slot) {
    struct hipIndex *ix = &t->index[slot];
    memset(ix, 0, sizeof *ix);
    ix->column = hipColumnId(engine->indexed_attributes[slot]);
    if (ix->column < 0) return false;
    if (!t->col[ix->column].data) { ix->column = -1; return false; }      /* a single-valued column has no buffer to sort */
    ix->key_kind = k_kind[ix->column] == HIPKIND_I32 ? 1 : 0;
    const size_t n = t->n_rows ? t->n_rows : 1;
    if (pqps_malloc(t->ctx, n * sizeof(uint32_t), (void **)&ix->perm_dev) != PQPS_OK) hip_die("index allocation");
    if (pqps_malloc(t->ctx, n * t->col[ix->column].width, &ix->keys_dev) != PQPS_OK) hip_die("index allocation");
    if (pqps_index_build(t->ctx, &t->col[ix->column], t->n_rows, ix->key_kind, ix->perm_dev, ix->keys_dev, NULL) != PQPS_OK)
        hip_die("index build");
    return true;
}

/* makeIndexSerial, buildEngine-serial.c:13-31: registers the attribute on the
 * engine and builds its index; success = index usable. */
bool makeIndexHIP(struct engineS *engine, const char *indexName, int attributeType) {
    struct hipTable *t = engine->record_block;
    const int slot = engine->num_indexes;
    engine->bplus_tree_roots = realloc(engine->bplus_tree_roots, (size_t)(slot + 1) * sizeof(node *));
    engine->indexed_attributes = realloc(engine->indexed_attributes, (size_t)(slot + 1) * sizeof(char *));
    engine->attribute_types = realloc(engine->attribute_types, (size_t)(slot + 1) * sizeof(FieldType));
    if (!engine->bplus_tree_roots || !engine->indexed_attributes || !engine->attribute_types) {
        perror("Failed to allocate memory for engine components");
        exit(EXIT_FAILURE);
    }
    engine->bplus_tree_roots[slot] = NULL;             /* no pointer tree in this engine */
    engine->indexed_attributes[slot] = strdup(indexName);
    engine->attribute_types[slot] = mapAttributeTypeHIP(attributeType);
    engine->num_indexes = slot + 1;
    bool ok = true;
    for (int s = 0; s < hipTableShards(t); s++) {
        struct hipTable *sh = hipTableShard(t, s);
        sh->index = realloc(sh->index, (size_t)(slot + 1) * sizeof *sh->index);
        if (!sh->index) { perror("Failed to allocate memory for engine components"); exit(EXIT_FAILURE); }
        ok = build_index(engine, sh, slot) && ok;
    }
    return ok;
}

/* What makes a table an ENGINE's table: query lanes on every shard and the locks. */
static void table_make_engine_lanes(struct hipTable *t, int min_lanes) {
    int n_lanes = engine_lanes();
    if (n_lanes < min_lanes) n_lanes = min_lanes > HIP_MAX_LANES ? HIP_MAX_LANES : min_lanes;
    for (int s = 0; s < hipTableShards(t); s++) shard_lanes_create(hipTableShard(t, s), n_lanes);
    if (!t->locks) t->locks = locks_create(n_lanes);
}

static void table_make_engine(struct hipTable *t) { table_make_engine_lanes(t, 1); }

bool buildDeviceTableHIP(struct engineS *engine) {
    struct hipContextFuture *f = hipBeginContextHIP();
    const bool ok = buildDeviceTableOnHIP(engine, f);
    return ok;
}

bool buildDeviceTableOnHIP(struct engineS *engine, struct hipContextFuture *future) {
    struct hipTable *t = calloc(1, sizeof *t);
    if (!t) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
    table_fill(t, NULL, 0, future, engine->all_records, (size_t)engine->num_records);
    free(future);
    t->row_block = engine->record_block;               /* block handed over by getAllRecordsFromFileHIP */
    t->row_capacity = (size_t)(engine->num_records > 0 ? engine->num_records : 0);
    if (t->row_block && engine->all_records) {                     /* what the two allocations really hold */
        const size_t cap_block = malloc_usable_size(t->row_block) / sizeof(record);
        const size_t cap_rows = malloc_usable_size(engine->all_records) / sizeof(record *);
        const size_t cap = cap_block < cap_rows ? cap_block : cap_rows;
        if (cap > t->row_capacity) t->row_capacity = cap;
    } else {
        t->row_capacity = 0;                                       /* no block yet: the first INSERT allocates one */
    }
    table_make_engine(t);
    engine->record_block = t;
    return true;
}

/* Re-creates columns, dictionaries and indexes from the host rows (after INSERT / DELETE), on the same
 * contexts, re-balanced over the shards. */
void rebuildDeviceTableHIP(struct engineS *engine) {
    struct hipTable *t = engine->record_block;          /* stays at this address: callers hold its locks */
    pqps_ctx *ctxs[HIP_MAX_SHARDS];
    const int n_shards = hipTableShards(t);
    for (int s = 0; s < n_shards; s++) ctxs[s] = hipTableShard(t, s)->ctx;
    record *block = t->row_block;
    const size_t row_capacity = t->row_capacity;
    struct hipLocks *locks = t->locks;
    table_release(t, engine->num_indexes);
    memset(t, 0, sizeof *t);
    table_fill(t, ctxs, n_shards, NULL, engine->all_records, (size_t)engine->num_records);
    t->row_block = block;
    t->row_capacity = row_capacity;
    t->locks = locks;
    table_make_engine(t);
    for (int s = 0; s < n_shards && engine->num_indexes > 0; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        sh->index = calloc((size_t)engine->num_indexes, sizeof *sh->index);
        if (!sh->index) { perror("Failed to allocate memory for engine components"); exit(EXIT_FAILURE); }
        for (int i = 0; i < engine->num_indexes; i++) build_index(engine, sh, i);
    }
}

/* Rebuilds every device index of one shard (a device radix sort each; cheap next to a table rebuild). */
static void rebuild_indexes(struct engineS *engine, struct hipTable *t) {
    for (int i = 0; i < engine->num_indexes; i++) {
        if (t->index[i].perm_dev) pqps_free(t->ctx, t->index[i].perm_dev);
        if (t->index[i].keys_dev) pqps_free(t->ctx, t->index[i].keys_dev);
        t->index[i].perm_dev = NULL;
        t->index[i].keys_dev = NULL;
        build_index(engine, t, i);
    }
}

/* INSERT: appends the last host row (engine->all_records[n-1]) to the device columns of the last shard
 * in place.  A string value that is new to its dictionary is inserted at its rank and the codes at or
 * above that rank are bumped on every shard (order-preserving codes stay order-preserving).
 * Falls back to a full rebuild only when a column has to change its code width or the
 * head-room is used up. */
bool appendRowDeviceTableHIP(struct engineS *engine, const record *r) {
    struct hipTable *t = engine->record_block;
    const int n_shards = hipTableShards(t);
    struct hipTable *last = hipTableShard(t, n_shards - 1);
    const size_t n = (size_t)engine->num_records;               /* rows after the insert */
    /* what only a rebuild from the host rows can do: a device-only engine cannot */
#define REBUILD_OR_FAIL() do { if (t->device_only) return false; rebuildDeviceTableHIP(engine); return true; } while (0)
    if (n == 0 || n - 1 != last->row0 + last->n_rows || last->n_rows + 1 > last->capacity_rows) REBUILD_OR_FAIL();
    /* first pass: would any dictionary outgrow its code width? */
    int pos[HIPCOL_COUNT], present[HIPCOL_COUNT];
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        if (k_kind[c] != HIPKIND_DICT) continue;
        struct hipDictionary *d = &t->dict[c];
        const char *s = (const char *)r + k_offset[c];
        int l = 0, h = d->count;
        while (l < h) { const int m = l + (h - l) / 2; if (strcmp(d->values[m], s) < 0) l = m + 1; else h = m; }
        pos[c] = l;
        present[c] = l < d->count && strcmp(d->values[l], s) == 0;
        const uint64_t limit = t->col[c].width == 0 ? 1 : t->col[c].width == 1 ? 256 : t->col[c].width == 2 ? 65536 : 0xFFFFFFFFull;
        if (!present[c] && (uint64_t)d->count + 1 > limit) REBUILD_OR_FAIL();
    }
#undef REBUILD_OR_FAIL
    bool bumped = false;
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        const uint32_t w = t->col[c].width;
        uint64_t value = 0;
        const char *p = (const char *)r + k_offset[c];
        switch (k_kind[c]) {
        case HIPKIND_U64: value = *(const uint64_t *)p; break;
        case HIPKIND_I32: value = (uint32_t)*(const int *)p; break;
        case HIPKIND_BOOL: value = *(const bool *)p ? 1 : 0; break;
        default: {
            struct hipDictionary *d = &t->dict[c];
            if (!present[c]) {
                const char **grown = realloc(d->values, ((size_t)d->count + 1) * sizeof *grown);
                char *copy = strdup(p);
                if (!grown || !copy) { perror("Failed to grow dictionary"); exit(EXIT_FAILURE); }
                d->values = grown;
                memmove(&d->values[pos[c] + 1], &d->values[pos[c]], ((size_t)d->count - (size_t)pos[c]) * sizeof *grown);
                d->values[pos[c]] = copy;
                d->count++;
                for (int s = 0; s < n_shards; s++) {
                    struct hipTable *sh = hipTableShard(t, s);
                    if (pqps_bump_codes(sh->ctx, (void *)sh->col[c].data, w, sh->n_rows, (uint32_t)pos[c], NULL) != PQPS_OK)
                        hip_die("dictionary code shift");
                }
                bumped = true;
            }
            value = (uint64_t)pos[c];
            break;
        }
        }
        if (w && pqps_upload(last->ctx, (char *)last->col[c].data + last->n_rows * w, &value, w, NULL) != PQPS_OK) hip_die("row upload");
    }
    last->n_rows += 1;
    for (int s = 0; s < n_shards; s++)                          /* shifted codes are keys of the other shards' indexes too */
        if (bumped || s == n_shards - 1) rebuild_indexes(engine, hipTableShard(t, s));
    return true;
}

/* DELETE: drops the flagged rows from the device columns of every shard in place and re-sorts the
 * indexes; the shards keep their (now shorter) contiguous row ranges. */
void compactDeviceTableHIP(struct engineS *engine, uint8_t *const *delete_flags_dev, size_t expected_rows) {
    struct hipTable *t = engine->record_block;
    uint64_t total = 0;
    for (int s = 0; s < hipTableShards(t); s++) {
        struct hipTable *sh = hipTableShard(t, s);
        uint64_t kept = 0;
        pqps_column cols[HIPCOL_COUNT];
        uint32_t n_cols = 0;
        for (int c = 0; c < HIPCOL_COUNT; c++) if (sh->col[c].data) cols[n_cols++] = sh->col[c];      /* (single-valued columns have no buffer) */
        if (pqps_compact_rows(sh->ctx, cols, n_cols, sh->n_rows, delete_flags_dev[s], &kept, NULL) != PQPS_OK)
            hip_die("row compaction");
        const bool changed = kept != sh->n_rows || sh->row0 != total;
        sh->n_rows = kept;
        sh->row0 = total;
        total += kept;
        if (changed) rebuild_indexes(engine, sh);
    }
    if (total != (uint64_t)expected_rows) {
        fprintf(stderr, "HIP engine: device kept %llu rows, host kept %zu\n", (unsigned long long)total, expected_rows);
        exit(EXIT_FAILURE);
    }
}

void destroyDeviceTableHIP(struct engineS *engine) {
    struct hipTable *t = engine->record_block;
    if (!t) return;
    pqps_ctx *ctxs[HIP_MAX_SHARDS];
    const int n_shards = hipTableShards(t);
    for (int s = 0; s < n_shards; s++) ctxs[s] = hipTableShard(t, s)->ctx;
    const char *trace = getenv("PQPS_TRACE");
    struct timespec t0, t1, t2, t3;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    free(t->row_block);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    hipTableFree(t, engine->num_indexes);
    clock_gettime(CLOCK_MONOTONIC, &t2);
    for (int s = 0; s < n_shards; s++) pqps_ctx_destroy(ctxs[s]);
    clock_gettime(CLOCK_MONOTONIC, &t3);
    if (trace && atoi(trace))
        fprintf(stderr, "[pqps] destroy: host rows %.3f ms, device table %.3f ms, %d context(s) %.3f ms\n",
                (double)(t1.tv_sec - t0.tv_sec) * 1e3 + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-6,
                (double)(t2.tv_sec - t1.tv_sec) * 1e3 + (double)(t2.tv_nsec - t1.tv_nsec) * 1e-6, n_shards,
                (double)(t3.tv_sec - t2.tv_sec) * 1e3 + (double)(t3.tv_nsec - t2.tv_nsec) * 1e-6);
    engine->record_block = NULL;
}

/* ---- engines over device-resident columns (no host rows) ------------------------------------------------ */

static int dictionary_from_strings(struct hipDictionary *d, const char *const *values, int count) {
    size_t bytes = 0;
    for (int i = 0; i < count; i++) {
        if (!values[i]) return -1;
        if (i > 0 && strcmp(values[i - 1], values[i]) >= 0) return -1;         /* ascending strcmp order, no duplicates */
        bytes += strlen(values[i]) + 1;
    }
    d->count = count;
    d->storage = malloc(bytes ? bytes : 1);
    d->storage_bytes = bytes;
    d->values = malloc((size_t)(count > 0 ? count : 1) * sizeof *d->values);
    if (!d->storage || !d->values) return -1;
    char *w = d->storage;
    for (int i = 0; i < count; i++) {
        const size_t l = strlen(values[i]) + 1;
        memcpy(w, values[i], l);
        d->values[i] = w;
        w += l;
    }
    return 0;
}

static const uint32_t k_numeric_width[HIPCOL_COUNT] = { 8, 0, 0, 0, 4, 0, 1, 0, 4, 0, 0, 4 };

/* The table struct of a device-only engine with its dictionaries and column widths; the shards' buffers exist
 * (allocated, tails cleared) and wait to be filled. */
static struct hipTable *columns_table(unsigned long long num_rows, const struct hipColumnData *columns, pqps_ctx ***ctxs_out, int *n_shards_out) {
    struct hipContextFuture *f = hipBeginContextHIP();
    struct hipTable *t = calloc(1, sizeof *t);
    if (!t) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        const struct hipColumnData *cd = &columns[c];
        if (k_kind[c] != HIPKIND_DICT) {
            if (cd->width != k_numeric_width[c] || (!cd->values && num_rows)) {
                fprintf(stderr, "HIP engine: column %d: a numeric column of %u bytes per row is expected\n", c, k_numeric_width[c]);
                exit(EXIT_FAILURE);
            }
            t->col[c].width = cd->width;
            continue;
        }
        if (cd->dictionary_count < 0 || (cd->dictionary_count > 0 && !cd->dictionary) ||
            dictionary_from_strings(&t->dict[c], cd->dictionary, cd->dictionary_count) != 0) {
            fprintf(stderr, "HIP engine: column %d: the dictionary must be %d distinct strings in ascending strcmp order\n", c, cd->dictionary_count);
            exit(EXIT_FAILURE);
        }
        if (!cd->values && cd->dictionary_count == 1) { t->col[c].width = 0; continue; }      /* every row carries the one value */
        const uint64_t room = cd->width == 1 ? 256 : cd->width == 2 ? 65536 : cd->width == 4 ? 0x100000000ull : 0;
        if (room == 0 || (uint64_t)cd->dictionary_count > room || (!cd->values && num_rows)) {
            fprintf(stderr, "HIP engine: column %d: codes of %u bytes cannot number %d dictionary values\n", c, cd->width, cd->dictionary_count);
            exit(EXIT_FAILURE);
        }
        t->col[c].width = cd->width;
    }
    await_contexts(f);
    pqps_ctx **ctxs = malloc((size_t)f->n * sizeof *ctxs);
    if (!ctxs) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
    for (int s = 0; s < f->n; s++) ctxs[s] = f->ctx[s];
    const int n_shards = f->n;
    free(f);
    table_make_shards(t, n_shards);
    for (int s = 0; s < n_shards; s++) {
        size_t row0, count;
        shard_range((size_t)num_rows, n_shards, s, &row0, &count);
        shard_alloc(hipTableShard(t, s), ctxs[s], t, row0, count);
    }
    t->device_only = 1;
    *ctxs_out = ctxs;
    *n_shards_out = n_shards;
    return t;
}

bool buildDeviceTableFromColumnsHIP(struct engineS *engine, unsigned long long num_rows, const struct hipColumnData *columns) {
    pqps_ctx **ctxs = NULL;
    int n_shards = 0;
    struct hipTable *t = columns_table(num_rows, columns, &ctxs, &n_shards);
    for (int s = 0; s < n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        for (int c = 0; c < HIPCOL_COUNT; c++) {
            const uint32_t w = sh->col[c].width;
            if (!w || !sh->n_rows) continue;
            const char *src = (const char *)columns[c].values + sh->row0 * w;
            const int rc = columns[c].on_device ? pqps_copy_peer(sh->ctx, (void *)sh->col[c].data, ctxs[0], src, sh->n_rows * w, NULL)
                                                : pqps_upload(sh->ctx, (void *)sh->col[c].data, src, sh->n_rows * w, NULL);
            if (rc != PQPS_OK) hip_die("column copy");
        }
        if (pqps_ctx_sync(sh->ctx, NULL) != PQPS_OK) hip_die("column copy");
    }
    free(ctxs);
    table_make_engine(t);
    engine->record_block = t;
    return true;
}

/* Dictionaries of the synthetic table: the strings behind the codes of pqps_synth_generate (include/pqps_hip.h:
 * shell_code = rank in {bash, fish, sh, zsh}, user_code = user_id - 1000, host_code = rank among 16 host names,
 * base_code = rank among 111 base commands); raw_command, timestamp and working_directory carry one value. */
static const char *const k_synth_shells[4] = { "bash", "fish", "sh", "zsh" };
static const char *const k_synth_hosts[16] = {
    "cs-lab-01", "cs-lab-02", "labpc-01", "labpc-02", "labpc-03", "labpc-04", "labpc-05", "labpc-06", "labpc-07", "labpc-08",
    "labpc-09", "labpc-10", "personal-laptop", "remote-ssh-01", "vm-ubuntu-01", "vm-ubuntu-02"
};
static const char *const k_synth_raw[1] = { "cmd" };
static const char *const k_synth_time[1] = { "2025-01-01T00:00:00.000Z" };
static const char *const k_synth_dir[1] = { "/home/u" };
static char k_synth_user_text[PQPS_SYNTH_USERS][12], k_synth_base_text[111][8];
static const char *k_synth_users[PQPS_SYNTH_USERS], *k_synth_bases[111];
static pthread_once_t k_synth_once = PTHREAD_ONCE_INIT;

static void synth_names_init(void) {
    for (int i = 0; i < PQPS_SYNTH_USERS; i++) { snprintf(k_synth_user_text[i], sizeof k_synth_user_text[i], "student%d", 1000 + i); k_synth_users[i] = k_synth_user_text[i]; }
    for (int i = 0; i < 111; i++) { snprintf(k_synth_base_text[i], sizeof k_synth_base_text[i], "cmd%03d", i); k_synth_bases[i] = k_synth_base_text[i]; }
}

const char *const *hipSyntheticDictionary(int column, int *count) {
    pthread_once(&k_synth_once, synth_names_init);
    int n = 0;
    const char *const *d = NULL;
    switch (column) {
    case HIPCOL_RAW_COMMAND: d = k_synth_raw; n = 1; break;
    case HIPCOL_BASE_COMMAND: d = k_synth_bases; n = 111; break;
    case HIPCOL_SHELL_TYPE: d = k_synth_shells; n = 4; break;
    case HIPCOL_TIMESTAMP: d = k_synth_time; n = 1; break;
    case HIPCOL_WORKING_DIRECTORY: d = k_synth_dir; n = 1; break;
    case HIPCOL_USER_NAME: d = k_synth_users; n = PQPS_SYNTH_USERS; break;
    case HIPCOL_HOST_NAME: d = k_synth_hosts; n = 16; break;
    default: break;
    }
    if (count) *count = n;
    return d;
}

static bool synthetic_table(struct engineS *engine, unsigned long long num_rows, unsigned long long seed, unsigned long long first_row,
                            bool one_shard, int min_lanes);

bool buildSyntheticDeviceTableHIP(struct engineS *engine, unsigned long long num_rows, unsigned long long seed) {
    return synthetic_table(engine, num_rows, seed, 0, false, 1);
}

bool buildSyntheticShardDeviceTableHIP(struct engineS *engine, unsigned long long num_rows, unsigned long long seed,
                                       unsigned long long first_row, int min_lanes) {
    return synthetic_table(engine, num_rows, seed, first_row, true, min_lanes);
}

static bool synthetic_table(struct engineS *engine, unsigned long long num_rows, unsigned long long seed, unsigned long long first_row,
                            bool one_shard, int min_lanes) {
    struct hipColumnData columns[HIPCOL_COUNT];
    memset(columns, 0, sizeof columns);
    static const char dummy = 0;                                     /* "values will be generated": not NULL */
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        if (k_kind[c] != HIPKIND_DICT) { columns[c].width = k_numeric_width[c]; columns[c].values = &dummy; continue; }
        columns[c].dictionary = hipSyntheticDictionary(c, &columns[c].dictionary_count);
        if (columns[c].dictionary_count > 1) { columns[c].width = columns[c].dictionary_count > 256 ? 2 : 1; columns[c].values = &dummy; }
    }
    pqps_ctx **ctxs = NULL;
    int n_shards = 0;
    struct hipTable *t = columns_table(num_rows, columns, &ctxs, &n_shards);
    if (one_shard) {
        if (n_shards != 1) { fprintf(stderr, "HIP engine: a rank's part of a table lives on ONE device (PQPS_DEVICES names several)\n"); exit(EXIT_FAILURE); }
        t->row0 = first_row;                                         /* the shard's rows carry the table-wide row numbers */
    }
    uint32_t *cdf = malloc(PQPS_SYNTH_USERS * sizeof *cdf);
    uint8_t *shell = malloc(PQPS_SYNTH_USERS);
    if (!cdf || !shell) { perror("Failed to allocate memory for the synthetic table"); exit(EXIT_FAILURE); }
    pqps_synth_user_tables(seed, cdf, shell);
    for (int s = 0; s < n_shards; s++) {
        struct hipTable *sh = hipTableShard(t, s);
        void *cdf_dev = NULL, *shell_dev = NULL;
        if (pqps_malloc(sh->ctx, PQPS_SYNTH_USERS * sizeof *cdf, &cdf_dev) != PQPS_OK || pqps_malloc(sh->ctx, PQPS_SYNTH_USERS, &shell_dev) != PQPS_OK ||
            pqps_upload(sh->ctx, cdf_dev, cdf, PQPS_SYNTH_USERS * sizeof *cdf, NULL) != PQPS_OK || pqps_upload(sh->ctx, shell_dev, shell, PQPS_SYNTH_USERS, NULL) != PQPS_OK)
            hip_die("synthetic table set-up");
        pqps_synth_cols out;
        memset(&out, 0, sizeof out);
        out.command_id = (uint64_t *)sh->col[HIPCOL_COMMAND_ID].data;
        out.exit_code = (int32_t *)sh->col[HIPCOL_EXIT_CODE].data;
        out.user_id = (int32_t *)sh->col[HIPCOL_USER_ID].data;
        out.risk_level = (int32_t *)sh->col[HIPCOL_RISK_LEVEL].data;
        out.sudo_used = (uint8_t *)sh->col[HIPCOL_SUDO_USED].data;
        out.shell_code = (uint8_t *)sh->col[HIPCOL_SHELL_TYPE].data;
        out.user_code = (uint16_t *)sh->col[HIPCOL_USER_NAME].data;
        out.host_code = (uint8_t *)sh->col[HIPCOL_HOST_NAME].data;
        out.base_code = (uint8_t *)sh->col[HIPCOL_BASE_COMMAND].data;
        if (sh->n_rows && pqps_synth_generate(sh->ctx, seed, sh->row0, sh->n_rows, cdf_dev, shell_dev, &out, NULL) != PQPS_OK) hip_die("synthetic table generation");
        if (pqps_ctx_sync(sh->ctx, NULL) != PQPS_OK) hip_die("synthetic table generation");
        pqps_free(sh->ctx, cdf_dev);
        pqps_free(sh->ctx, shell_dev);
    }
    free(cdf); free(shell); free(ctxs);
    table_make_engine_lanes(t, min_lanes);
    engine->record_block = t;
    return true;
}
