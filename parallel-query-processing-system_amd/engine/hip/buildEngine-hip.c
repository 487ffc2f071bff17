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
#define _POSIX_C_SOURCE 200809L
#include "buildEngine-hip.h"
#include "hipPredicate.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

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

/* getAllRecordsFromFile, buildEngine-serial.c:70-108, with the block
 * allocation of the OMP variant (buildEngine-omp.c:84): rows are the fgets()
 * chunks of the file (<= 1023 bytes each) after the first one. */
record **getAllRecordsFromFileHIP(const char *filepath, int *num_records, void **record_block_out) {
    *num_records = 0;
    if (record_block_out) *record_block_out = NULL;
    FILE *f = fopen(filepath, "r");
    if (!f) {
        fprintf(stderr, "Error opening file: %s\n", filepath);
        return NULL;
    }
    size_t cap = 4096, n = 0;
    record *block = malloc(cap * sizeof *block);
    char line[1024];
    int first = 1;
    while (block && fgets(line, sizeof line, f)) {
        if (first) { first = 0; continue; }
        if (n == cap) {
            cap *= 2;
            record *grown = realloc(block, cap * sizeof *block);
            if (!grown) { free(block); block = NULL; break; }
            block = grown;
        }
        fillRecordFromLineHIP(&block[n++], line);
    }
    fclose(f);
    if (!block) { fprintf(stderr, "Memory allocation failed\n"); return NULL; }
    record **rows = malloc((n ? n : 1) * sizeof *rows);
    if (!rows) { free(block); fprintf(stderr, "Memory allocation failed\n"); return NULL; }
    for (size_t i = 0; i < n; i++) rows[i] = &block[i];
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

static int cmp_cstr_ptr(const void *a, const void *b) {
    return strcmp(*(const char *const *)a, *(const char *const *)b);
}

/* Sorted distinct values of one string column.  The "value" of a row is the
 * C string that starts at the field -- exactly what strcmp() in the
 * reference's CMP_STR sees, including the run-on into the next field when
 * strncpy left no terminator (buildEngine-serial.c:171). */
static int build_dictionary(record *const *rows, size_t n, size_t off, struct hipDictionary *d,
                            uint32_t *codes) {
    const char **ptrs = malloc((n ? n : 1) * sizeof *ptrs);
    if (!ptrs) return -1;
    for (size_t i = 0; i < n; i++) ptrs[i] = (const char *)rows[i] + off;
    qsort(ptrs, n, sizeof *ptrs, cmp_cstr_ptr);
    size_t distinct = 0, bytes = 0;
    for (size_t i = 0; i < n; i++) {
        if (i == 0 || strcmp(ptrs[i], ptrs[distinct - 1]) != 0) {
            ptrs[distinct++] = ptrs[i];
            bytes += strlen(ptrs[i]) + 1;
        }
    }
    d->count = (int)distinct;
    d->storage = malloc(bytes ? bytes : 1);
    d->values = malloc((distinct ? distinct : 1) * sizeof *d->values);
    if (!d->storage || !d->values) { free(ptrs); return -1; }
    char *w = d->storage;
    for (size_t i = 0; i < distinct; i++) {
        const size_t len = strlen(ptrs[i]) + 1;
        memcpy(w, ptrs[i], len);
        d->values[i] = w;
        w += len;
    }
    free(ptrs);
    for (size_t i = 0; i < n; i++) {
        const char *s = (const char *)rows[i] + off;
        size_t l = 0, r = distinct;
        while (l < r) { size_t m = l + (r - l) / 2; if (strcmp(d->values[m], s) < 0) l = m + 1; else r = m; }
        codes[i] = (uint32_t)l;
    }
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

/* Builds a device table from `n` host rows.  ctx may be shared (owned by the caller). */
struct hipTable *hipTableFromRows(pqps_ctx *ctx, record *const *rows, size_t n) {
    struct hipTable *t = calloc(1, sizeof *t);
    if (!t) { perror("Failed to allocate memory for device table"); exit(EXIT_FAILURE); }
    t->ctx = ctx;
    t->n_rows = n;
    t->capacity_rows = (n + PQPS_TILE_ROWS - 1) / PQPS_TILE_ROWS * PQPS_TILE_ROWS;
    if (t->capacity_rows == 0) t->capacity_rows = PQPS_TILE_ROWS;
    uint32_t *codes = malloc((n ? n : 1) * sizeof *codes);
    void *stage = malloc(t->capacity_rows * 8);
    if (!codes || !stage) { perror("Failed to allocate staging memory"); exit(EXIT_FAILURE); }

    for (int c = 0; c < HIPCOL_COUNT; c++) {
        uint32_t width;
        memset(stage, 0, t->capacity_rows * 8);
        switch (k_kind[c]) {
        case HIPKIND_U64:
            width = 8;
            for (size_t i = 0; i < n; i++) ((uint64_t *)stage)[i] = *(const uint64_t *)((const char *)rows[i] + k_offset[c]);
            break;
        case HIPKIND_I32:
            width = 4;
            for (size_t i = 0; i < n; i++) ((int32_t *)stage)[i] = *(const int *)((const char *)rows[i] + k_offset[c]);
            break;
        case HIPKIND_BOOL:
            width = 1;
            for (size_t i = 0; i < n; i++) ((uint8_t *)stage)[i] = *(const bool *)((const char *)rows[i] + k_offset[c]) ? 1 : 0;
            break;
        default:
            if (build_dictionary(rows, n, k_offset[c], &t->dict[c], codes) != 0) {
                perror("Failed to build dictionary");
                exit(EXIT_FAILURE);
            }
            width = t->dict[c].count <= 256 ? 1 : t->dict[c].count <= 65536 ? 2 : 4;
            for (size_t i = 0; i < n; i++) {
                if (width == 1) ((uint8_t *)stage)[i] = (uint8_t)codes[i];
                else if (width == 2) ((uint16_t *)stage)[i] = (uint16_t)codes[i];
                else ((uint32_t *)stage)[i] = codes[i];
            }
            break;
        }
        void *dev = NULL;
        if (pqps_malloc(ctx, t->capacity_rows * width, &dev) != PQPS_OK) hip_die("column allocation");
        if (pqps_upload(ctx, dev, stage, t->capacity_rows * width, NULL) != PQPS_OK) hip_die("column upload");
        t->col[c].data = dev;
        t->col[c].width = width;
    }
    free(stage);
    free(codes);

    t->capacity_ids = t->capacity_rows;
    if (pqps_malloc(ctx, t->capacity_ids * sizeof(uint32_t), (void **)&t->ids_dev) != PQPS_OK) hip_die("result allocation");
    if (pqps_malloc(ctx, 8 * sizeof(uint64_t), (void **)&t->count_dev) != PQPS_OK) hip_die("counter allocation");
    return t;
}

void hipTableFree(struct hipTable *t, int n_indexes) {
    if (!t) return;
    for (int c = 0; c < HIPCOL_COUNT; c++) {
        if (t->col[c].data) pqps_free(t->ctx, (void *)t->col[c].data);
        free(t->dict[c].values);
        free(t->dict[c].storage);
    }
    if (t->index) {
        for (int i = 0; i < n_indexes; i++) {
            if (t->index[i].perm_dev) pqps_free(t->ctx, t->index[i].perm_dev);
            if (t->index[i].keys_dev) pqps_free(t->ctx, t->index[i].keys_dev);
        }
        free(t->index);
    }
    if (t->ids_dev) pqps_free(t->ctx, t->ids_dev);
    if (t->count_dev) pqps_free(t->ctx, t->count_dev);
    free(t);
}

/* Device index for engine->indexed_attributes[slot]. */
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
    t->index = realloc(t->index, (size_t)(slot + 1) * sizeof *t->index);
    if (!engine->bplus_tree_roots || !engine->indexed_attributes || !engine->attribute_types || !t->index) {
        perror("Failed to allocate memory for engine components");
        exit(EXIT_FAILURE);
    }
    engine->bplus_tree_roots[slot] = NULL;             /* no pointer tree in this engine */
    engine->indexed_attributes[slot] = strdup(indexName);
    engine->attribute_types[slot] = mapAttributeTypeHIP(attributeType);
    engine->num_indexes = slot + 1;
    return build_index(engine, t, slot);
}

bool buildDeviceTableHIP(struct engineS *engine) {
    pqps_ctx *ctx = NULL;
    int device = 0;
    const char *env = getenv("PQPS_DEVICE");
    if (env) device = atoi(env);
    if (pqps_ctx_create(device, &ctx) != PQPS_OK) hip_die("cannot create a device context");
    struct hipTable *t = hipTableFromRows(ctx, engine->all_records, (size_t)engine->num_records);
    t->row_block = engine->record_block;               /* block handed over by getAllRecordsFromFileHIP */
    engine->record_block = t;
    return true;
}

/* Re-creates columns, dictionaries and indexes from the host rows (after INSERT / DELETE). */
void rebuildDeviceTableHIP(struct engineS *engine) {
    struct hipTable *old = engine->record_block;
    pqps_ctx *ctx = old->ctx;
    record *block = old->row_block;
    hipTableFree(old, engine->num_indexes);
    struct hipTable *t = hipTableFromRows(ctx, engine->all_records, (size_t)engine->num_records);
    t->row_block = block;
    engine->record_block = t;
    if (engine->num_indexes > 0) {
        t->index = calloc((size_t)engine->num_indexes, sizeof *t->index);
        for (int i = 0; i < engine->num_indexes; i++) build_index(engine, t, i);
    }
}

void destroyDeviceTableHIP(struct engineS *engine) {
    struct hipTable *t = engine->record_block;
    if (!t) return;
    pqps_ctx *ctx = t->ctx;
    free(t->row_block);
    hipTableFree(t, engine->num_indexes);
    pqps_ctx_destroy(ctx);
    engine->record_block = NULL;
}
