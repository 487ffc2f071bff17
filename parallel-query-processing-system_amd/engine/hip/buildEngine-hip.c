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
        s->col[c].present = t->col[c].data != NULL || t->n_rows == 0;
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

/* Device side of one shard: rows [row0, row0 + count) of the staged columns.  Capacity leaves head-room so
 * that INSERT appends in place; the rows past the last one are zero (the filter reads whole 1024-row steps). */
static void shard_upload(struct hipTable *sh, pqps_ctx *ctx, const struct hipTable *widths, void *const *stage,
                         size_t row0, size_t count) {
    sh->ctx = ctx;
    sh->n_rows = count;
    sh->row0 = row0;
    sh->capacity_rows = (count + count / 16 + PQPS_TILE_ROWS) / PQPS_TILE_ROWS * PQPS_TILE_ROWS;
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        const uint32_t width = widths->col[c].width;
        void *dev = NULL;
        if (pqps_malloc(ctx, sh->capacity_rows * width, &dev) != PQPS_OK) hip_die("column allocation");
        if (pqps_memset(ctx, dev, 0, sh->capacity_rows * width, NULL) != PQPS_OK) hip_die("column clear");
        if (count && pqps_upload(ctx, dev, (const char *)stage[c] + row0 * width, count * width, NULL) != PQPS_OK) hip_die("column upload");
        sh->col[c].data = dev;
        sh->col[c].width = width;
    }
    sh->capacity_ids = sh->capacity_rows;
    if (pqps_malloc(ctx, sh->capacity_ids * sizeof(uint32_t), (void **)&sh->ids_dev) != PQPS_OK) hip_die("result allocation");
    if (pqps_malloc(ctx, 8 * sizeof(uint64_t), (void **)&sh->count_dev) != PQPS_OK) hip_die("counter allocation");
    if (pqps_ctx_reserve(ctx, sh->capacity_rows) != PQPS_OK) hip_die("filter scratch allocation");   /* not inside the first query */
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
    t->n_shards = 0;
    t->shard = NULL;
    if (n_shards > 1) {
        t->shard = calloc((size_t)n_shards, sizeof *t->shard);
        if (!t->shard) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
        t->n_shards = n_shards;
        t->shard[0] = t;
    }
    for (int s = 0; s < n_shards; s++) {
        struct hipTable *sh = t;
        if (s > 0) {
            sh = calloc(1, sizeof *sh);
            if (!sh) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
            t->shard[s] = sh;
        }
        size_t row0, count;
        shard_range(n, n_shards, s, &row0, &count);
        shard_upload(sh, ctxs[s], t, job.stage, row0, count);
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

struct hipLocks {
    pthread_rwlock_t rows;
    pthread_mutex_t device;
};

void hipTableLockShared(struct hipTable *t) { if (t && t->locks) pthread_rwlock_rdlock(&t->locks->rows); }
void hipTableLockExclusive(struct hipTable *t) { if (t && t->locks) pthread_rwlock_wrlock(&t->locks->rows); }
void hipTableUnlock(struct hipTable *t) { if (t && t->locks) pthread_rwlock_unlock(&t->locks->rows); }
void hipTableLockDevice(struct hipTable *t) { if (t && t->locks) pthread_mutex_lock(&t->locks->device); }
void hipTableUnlockDevice(struct hipTable *t) { if (t && t->locks) pthread_mutex_unlock(&t->locks->device); }

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
    if (sh->ids_dev) pqps_free(sh->ctx, sh->ids_dev);
    if (sh->count_dev) pqps_free(sh->ctx, sh->count_dev);
    sh->ids_dev = NULL;
    sh->count_dev = NULL;
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
    table_release(t, n_indexes);
    if (t->locks) {
        pthread_rwlock_destroy(&t->locks->rows);
        pthread_mutex_destroy(&t->locks->device);
        free(t->locks);
    }
    free(t);
}

/* Device index of one shard for engine->indexed_attributes[slot]: row numbers are shard-local. */
static bool build_index(struct engineS *engine, struct hipTable *t, int slot) {
    struct hipIndex *ix = &t->index[slot];
    memset(ix, 0, sizeof *ix);
    ix->column = hipColumnId(engine->indexed_attributes[slot]);
    if (ix->column < 0) return false;
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
    t->locks = calloc(1, sizeof *t->locks);
    if (!t->locks || pthread_rwlock_init(&t->locks->rows, NULL) != 0 || pthread_mutex_init(&t->locks->device, NULL) != 0) {
        perror("Failed to create engine locks");
        exit(EXIT_FAILURE);
    }
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
        build_index(engine, t, i);
    }
}

/* INSERT: appends the last host row (engine->all_records[n-1]) to the device columns of the last shard
 * in place.  A string value that is new to its dictionary is inserted at its rank and the codes at or
 * above that rank are bumped on every shard (order-preserving codes stay order-preserving).
 * Falls back to a full rebuild only when a column has to change its code width or the
 * head-room is used up. */
void appendRowDeviceTableHIP(struct engineS *engine) {
    struct hipTable *t = engine->record_block;
    const int n_shards = hipTableShards(t);
    struct hipTable *last = hipTableShard(t, n_shards - 1);
    const size_t n = (size_t)engine->num_records;               /* rows after the insert */
    if (n == 0 || n - 1 != last->row0 + last->n_rows || last->n_rows + 1 > last->capacity_rows) { rebuildDeviceTableHIP(engine); return; }
    const record *r = engine->all_records[n - 1];
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
        const uint64_t limit = t->col[c].width == 1 ? 256 : t->col[c].width == 2 ? 65536 : 0xFFFFFFFFull;
        if (!present[c] && (uint64_t)d->count + 1 > limit) { rebuildDeviceTableHIP(engine); return; }
    }
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
        if (pqps_upload(last->ctx, (char *)last->col[c].data + last->n_rows * w, &value, w, NULL) != PQPS_OK) hip_die("row upload");
    }
    last->n_rows += 1;
    for (int s = 0; s < n_shards; s++)                          /* shifted codes are keys of the other shards' indexes too */
        if (bumped || s == n_shards - 1) rebuild_indexes(engine, hipTableShard(t, s));
}

/* DELETE: drops the flagged rows from the device columns of every shard in place and re-sorts the
 * indexes; the shards keep their (now shorter) contiguous row ranges. */
void compactDeviceTableHIP(struct engineS *engine, uint8_t *const *delete_flags_dev, size_t expected_rows) {
    struct hipTable *t = engine->record_block;
    uint64_t total = 0;
    for (int s = 0; s < hipTableShards(t); s++) {
        struct hipTable *sh = hipTableShard(t, s);
        uint64_t kept = 0;
        if (pqps_compact_rows(sh->ctx, sh->col, HIPCOL_COUNT, sh->n_rows, delete_flags_dev[s], &kept, NULL) != PQPS_OK)
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
