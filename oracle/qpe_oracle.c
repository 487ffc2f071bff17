/* qpe_oracle.c -- CPU restatement of QPESeq's SELECT/WHERE path (plain C11).
 *
 * TEST INFRASTRUCTURE ONLY (see qpe_oracle.h).  Parity status: PINNED against
 * the compiled reference (oracle/_ref) and tests/golden/.
 *
 * Each function names the reference code it restates.  Paths are relative to
 * the reference root (Jairik/Parallel-Query-Processing-System):
 *   S  = engine/serial/executeEngine-serial.c
 *   B  = engine/serial/buildEngine-serial.c
 *   BP = engine/bplus.c
 *   RS = engine/recordSchema.c
 */
#define _GNU_SOURCE
#include "qpe_oracle.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------
 * Row view: the five numeric fields by value, the seven strings by pointer.
 * Lets one evaluator serve AoS `record`s and dictionary-coded columns.
 * --------------------------------------------------------------------- */
enum { S_RAW, S_BASE, S_SHELL, S_TS, S_WD, S_UNAME, S_HOST, S_COUNT };
static const char *const k_str_names[S_COUNT] = {
    "raw_command", "base_command", "shell_type", "timestamp",
    "working_directory", "user_name", "host_name"
};

struct row_view {
    unsigned long long command_id;
    int exit_code, user_id, risk_level;
    bool sudo_used;
    const char *s[S_COUNT];
};

static void view_of_record(const record *r, struct row_view *v) {
    v->command_id = r->command_id;
    v->exit_code = r->exit_code;
    v->user_id = r->user_id;
    v->risk_level = r->risk_level;
    v->sudo_used = r->sudo_used;
    v->s[S_RAW] = r->raw_command;
    v->s[S_BASE] = r->base_command;
    v->s[S_SHELL] = r->shell_type;
    v->s[S_TS] = r->timestamp;
    v->s[S_WD] = r->working_directory;
    v->s[S_UNAME] = r->user_name;
    v->s[S_HOST] = r->host_name;
}

/* Operator text -> 0..5 (= != > < >= <=), -1 when it is none of the six.
 * S:131-136 (same six spellings for every attribute). */
static int op_code(const char *op) {
    if (!op) return -1;
    if (strcmp(op, "=") == 0) return 0;
    if (strcmp(op, "!=") == 0) return 1;
    if (strcmp(op, ">") == 0) return 2;
    if (strcmp(op, "<") == 0) return 3;
    if (strcmp(op, ">=") == 0) return 4;
    if (strcmp(op, "<=") == 0) return 5;
    return -1;
}

/* Sign of a three-way compare applied to one of the six operators. */
static bool apply_op(int op, int cmp) {
    switch (op) {
    case 0: return cmp == 0;
    case 1: return cmp != 0;
    case 2: return cmp > 0;
    case 3: return cmp < 0;
    case 4: return cmp >= 0;
    case 5: return cmp <= 0;
    default: return false;
    }
}

/* checkCondition, S:251-289, with create_where_condition S:129-213 and the
 * CMP_NUM / CMP_STR bodies S:18-26 folded in.  The literal is typed by the
 * COLUMN: strtoull for command_id (S:258), atoi for the three int columns
 * (S:265), "true"/"1" for sudo_used (S:270), raw text for the strings
 * (S:275).  Unknown attribute or operator => false (S:212, S:279). */
static bool check_view(const struct row_view *v, const struct whereClauseS *c) {
    const char *a = c->attribute;
    int op = op_code(c->operator);
    if (strcmp(a, "command_id") == 0) {
        unsigned long long lit = strtoull(c->value, NULL, 10);
        if (op < 0) return false;
        return apply_op(op, v->command_id < lit ? -1 : (v->command_id > lit ? 1 : 0));
    }
    if (strcmp(a, "risk_level") == 0 || strcmp(a, "exit_code") == 0 || strcmp(a, "user_id") == 0) {
        int lit = atoi(c->value);
        int x = (a[0] == 'r') ? v->risk_level : (a[0] == 'e') ? v->exit_code : v->user_id;
        if (op < 0) return false;
        return apply_op(op, x < lit ? -1 : (x > lit ? 1 : 0));
    }
    if (strcmp(a, "sudo_used") == 0) {
        bool lit = (strcasecmp(c->value, "true") == 0 || strcmp(c->value, "1") == 0);
        if (op == 0) return v->sudo_used == lit;    /* S:122 */
        if (op == 1) return v->sudo_used != lit;    /* S:123 */
        return false;                               /* no ordering comparators exist, S:207-210 */
    }
    for (int k = 0; k < S_COUNT; k++) {
        if (strcmp(a, k_str_names[k]) == 0) {
            if (op < 0) return false;
            return apply_op(op, strcmp(v->s[k], c->value));   /* CMP_STR, S:23-26 */
        }
    }
    return false;
}

/* evaluateWhereClause, S:292-316: right-recursive, no precedence; a node
 * with `sub` evaluates the nested chain; a logical_op that is neither "OR"
 * nor "AND" (or NULL with a successor) behaves as AND (S:315). */
static bool eval_view(const struct row_view *v, const struct whereClauseS *wc) {
    if (wc == NULL) return true;
    bool cur = wc->sub ? eval_view(v, wc->sub) : check_view(v, wc);
    if (wc->next == NULL) return cur;
    if (wc->logical_op && strcmp(wc->logical_op, "OR") == 0)
        return cur || eval_view(v, wc->next);
    return cur && eval_view(v, wc->next);
}

bool orc_check_condition(const record *r, const struct whereClauseS *c) {
    struct row_view v;
    view_of_record(r, &v);
    return check_view(&v, c);
}

bool orc_eval_where(const record *r, const struct whereClauseS *wc) {
    struct row_view v;
    view_of_record(r, &v);
    return eval_view(&v, wc);
}

/* linearSearchRecords, S:854-878: input order kept, NULL clause keeps all. */
int orc_linear_search(const record *const *rows, int n,
                      const struct whereClauseS *wc, int *out_pos) {
    int m = 0;
    for (int i = 0; i < n; i++) {
        if (wc == NULL || orc_eval_where(rows[i], wc)) out_pos[m++] = i;
    }
    return m;
}

/* ------------------------------------------------------------------------
 * CSV ingest
 * --------------------------------------------------------------------- */

/* parseCSVField, B:111-151.  Returns false when the cursor already sits on
 * end of line (field absent; the caller leaves the zero from calloc).
 * A field ends at an unquoted ',' (consumed) or at NUL / LF / CR.  Inside
 * quotes "" is a literal quote; text after a closing quote is appended. */
static bool next_field(const char **cursor, char *out) {
    const char *p = *cursor;
    if (*p == '\0' || *p == '\n' || *p == '\r') return false;
    size_t len = 0;
    bool quoted = false;
    if (*p == '"') { quoted = true; p++; }
    while (*p != '\0' && *p != '\n' && *p != '\r') {
        if (quoted) {
            if (*p == '"') {
                if (p[1] == '"') { out[len++] = '"'; p += 2; }
                else { quoted = false; p++; }
            } else {
                out[len++] = *p++;
            }
        } else {
            if (*p == ',') { p++; break; }
            out[len++] = *p++;
        }
    }
    out[len] = '\0';
    *cursor = p;
    return true;
}

static bool parse_bool_text(const char *t) {      /* B:188-190, S:270 */
    return strcasecmp(t, "true") == 0 || strcmp(t, "1") == 0;
}

/* getRecordFromLine, B:159-221: twelve fields in schema order into a zeroed
 * record; strings copied with strncpy(dst, tok, sizeof dst) -- no forced NUL. */
void orc_fill_record(record *dst, const char *line) {
    char tok[1100];
    const char *cur = line;
    memset(dst, 0, sizeof *dst);
    if (next_field(&cur, tok)) dst->command_id = strtoull(tok, NULL, 10);
    if (next_field(&cur, tok)) strncpy(dst->raw_command, tok, sizeof dst->raw_command);
    if (next_field(&cur, tok)) strncpy(dst->base_command, tok, sizeof dst->base_command);
    if (next_field(&cur, tok)) strncpy(dst->shell_type, tok, sizeof dst->shell_type);
    if (next_field(&cur, tok)) dst->exit_code = atoi(tok);
    if (next_field(&cur, tok)) strncpy(dst->timestamp, tok, sizeof dst->timestamp);
    if (next_field(&cur, tok)) dst->sudo_used = parse_bool_text(tok);
    if (next_field(&cur, tok)) strncpy(dst->working_directory, tok, sizeof dst->working_directory);
    if (next_field(&cur, tok)) dst->user_id = atoi(tok);
    if (next_field(&cur, tok)) strncpy(dst->user_name, tok, sizeof dst->user_name);
    if (next_field(&cur, tok)) strncpy(dst->host_name, tok, sizeof dst->host_name);
    if (next_field(&cur, tok)) dst->risk_level = atoi(tok);
}

/* getAllRecordsFromFile, B:70-108: the first fgets() chunk is dropped as the
 * header; every further chunk of at most 1023 bytes is one row (a longer
 * physical line therefore yields several rows).  Row index = chunk order. */
int orc_load_csv(const char *path, record **rows_out) {
    FILE *f = fopen(path, "r");
    *rows_out = NULL;
    if (!f) return -1;
    char line[1024];
    size_t cap = 1024, n = 0;
    record *rows = malloc(cap * sizeof *rows);
    bool header = true;
    while (fgets(line, sizeof line, f)) {
        if (header) { header = false; continue; }
        if (n == cap) {
            cap *= 2;
            rows = realloc(rows, cap * sizeof *rows);
        }
        if (!rows) { fclose(f); return -1; }
        orc_fill_record(&rows[n++], line);
    }
    fclose(f);
    *rows_out = rows;
    return (int)n;
}

/* ------------------------------------------------------------------------
 * Index emulation
 * --------------------------------------------------------------------- */

struct sort_ctx { const record *rows; size_t off; FieldType type; };

/* compare_key, RS:88-127, on the field at ctx->off. */
static int key_cmp_rows(const struct sort_ctx *c, int ra, int rb) {
    const char *pa = (const char *)&c->rows[ra] + c->off;
    const char *pb = (const char *)&c->rows[rb] + c->off;
    switch (c->type) {
    case FIELD_UINT64: {
        uint64_t a = *(const uint64_t *)pa, b = *(const uint64_t *)pb;
        return a < b ? -1 : a > b;
    }
    case FIELD_INT: {
        int a = *(const int *)pa, b = *(const int *)pb;
        return a < b ? -1 : a > b;
    }
    case FIELD_BOOL: {
        bool a = *(const bool *)pa, b = *(const bool *)pb;
        return a == b ? 0 : (a ? 1 : -1);
    }
    default:
        return strcmp(pa, pb);
    }
}

static int perm_cmp(const void *x, const void *y, void *arg) {
    int ra = *(const int *)x, rb = *(const int *)y;
    int k = key_cmp_rows((const struct sort_ctx *)arg, ra, rb);
    if (k) return k;
    return ra > rb ? -1 : (ra < rb);          /* equal keys: later insertion first */
}

/* Schema lookup, RS:12-38. */
static bool schema_lookup(const char *attr, size_t *off, FieldType *type) {
    static const struct { const char *n; size_t o; FieldType t; } tab[] = {
        { "command_id", offsetof(record, command_id), FIELD_UINT64 },
        { "raw_command", offsetof(record, raw_command), FIELD_STRING },
        { "base_command", offsetof(record, base_command), FIELD_STRING },
        { "shell_type", offsetof(record, shell_type), FIELD_STRING },
        { "exit_code", offsetof(record, exit_code), FIELD_INT },
        { "timestamp", offsetof(record, timestamp), FIELD_STRING },
        { "sudo_used", offsetof(record, sudo_used), FIELD_BOOL },
        { "working_directory", offsetof(record, working_directory), FIELD_STRING },
        { "user_id", offsetof(record, user_id), FIELD_INT },
        { "user_name", offsetof(record, user_name), FIELD_STRING },
        { "host_name", offsetof(record, host_name), FIELD_STRING },
        { "risk_level", offsetof(record, risk_level), FIELD_INT },
    };
    for (size_t i = 0; i < sizeof tab / sizeof tab[0]; i++)
        if (strcmp(tab[i].n, attr) == 0) { *off = tab[i].o; *type = tab[i].t; return true; }
    return false;
}

/* Leaf order of the tree built by loadIntoBplusTree (B:41-62) with insert
 * (BP:723-740): keys ascending; a new duplicate always lands in front of the
 * equal keys already there (findLeaf goes left on equality BP:339-342,
 * insertIntoLeaf stops at the first key >= new BP:475-477, the split keeps
 * that position BP:511-517) => equal keys in DESCENDING row order. */
int orc_index_build(const record *rows, int n, const char *attr, int *perm) {
    struct sort_ctx c;
    c.rows = rows;
    if (!schema_lookup(attr, &c.off, &c.type)) return -1;
    for (int i = 0; i < n; i++) perm[i] = i;
    qsort_r(perm, (size_t)n, sizeof perm[0], perm_cmp, &c);
    return 0;
}

/* findRange, BP:282-314, over the sorted permutation: every entry with
 * key_start <= key <= key_end, leaf order.  Returns [*b, *e). */
static void range_u64(const record *rows, size_t off, const int *perm, int n,
                      uint64_t lo, uint64_t hi, int *b, int *e) {
    int l = 0, r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if (*(const uint64_t *)((const char *)&rows[perm[m]] + off) < lo) l = m + 1; else r = m; }
    *b = l;
    r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if (*(const uint64_t *)((const char *)&rows[perm[m]] + off) <= hi) l = m + 1; else r = m; }
    *e = l;
}

static void range_i32(const record *rows, size_t off, const int *perm, int n,
                      int lo, int hi, int *b, int *e) {
    int l = 0, r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if (*(const int *)((const char *)&rows[perm[m]] + off) < lo) l = m + 1; else r = m; }
    *b = l;
    r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if (*(const int *)((const char *)&rows[perm[m]] + off) <= hi) l = m + 1; else r = m; }
    *e = l;
}

static void range_bool(const record *rows, size_t off, const int *perm, int n,
                       int lo, int hi, int *b, int *e) {
    int l = 0, r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if ((int)*(const bool *)((const char *)&rows[perm[m]] + off) < lo) l = m + 1; else r = m; }
    *b = l;
    r = n;
    while (l < r) { int m = l + (r - l) / 2;
        if ((int)*(const bool *)((const char *)&rows[perm[m]] + off) <= hi) l = m + 1; else r = m; }
    *e = l;
}

/* The key window the OpenMP / MPI engines derive for a condition on a BOOL index
 * (engine/omp/executeEngine-omp.c:424-459; engine/mpi has the same block): = / != pick one key, the
 * ordered operators the keys they admit -- `> true` and `< false` admit none (start > end). */
static void bool_window(const char *op, const char *value, int *lo, int *hi) {
    const int val = (strcasecmp(value, "true") == 0 || strcmp(value, "1") == 0) ? 1 : 0;
    if (strcmp(op, "=") == 0) { *lo = val; *hi = val; }
    else if (strcmp(op, "!=") == 0) { *lo = !val; *hi = !val; }
    else {
        *lo = 0; *hi = 1;
        if (strcmp(op, ">") == 0) { if (!val) { *lo = 1; *hi = 1; } else { *lo = 1; *hi = 0; } }
        else if (strcmp(op, ">=") == 0) { if (!val) { *lo = 0; *hi = 1; } else { *lo = 1; *hi = 1; } }
        else if (strcmp(op, "<") == 0) { if (val) { *lo = 0; *hi = 0; } else { *lo = 1; *hi = 0; } }
        else if (strcmp(op, "<=") == 0) { if (val) { *lo = 0; *hi = 1; } else { *lo = 0; *hi = 0; } }
    }
}

/* executeQuerySelectSerial, S:358-474 (row selection only).
 * For every TOP-LEVEL condition in chain order (nested nodes have
 * attribute == NULL and are skipped, S:361-364) and every index whose name
 * matches (S:366-367): u64 / int indexes derive an inclusive key window from
 * the operator (S:377-424; v+1 / v-1 wrap like the machine does) and append
 * the whole findRange output (S:441-448); bool / string indexes are ignored
 * (S:425-433).  No index fired => full scan (S:464-467); otherwise the
 * candidate list is re-filtered with the complete WHERE (S:469-474).
 * Unlike the reference the candidate buffer cannot overflow (S:342 sizes it
 * num_records): results past `cap` are counted but not stored. */
long long orc_select_ids(const record *rows, int n,
                         int num_idx, const char *const *idx_attr, const int *idx_type,
                         const int *const *idx_perm,
                         const struct whereClauseS *wc,
                         uint32_t *out_ids, long long cap, long long *candidates) {
    return orc_select_ids_v(rows, n, num_idx, idx_attr, idx_type, idx_perm, wc, out_ids, cap, candidates, 0);
}

/* `probe_bool` != 0: the row selection of executeQuerySelectOMP / ...MPI (omp:362-494) run by ONE thread -- the same
 * walk, and BOOL indexes are probed too (omp:424-459).  (With several threads the reference appends the candidates
 * of different indexes in whatever order the threads get there, omp:366,481: not a target.) */
long long orc_select_ids_v(const record *rows, int n,
                           int num_idx, const char *const *idx_attr, const int *idx_type,
                           const int *const *idx_perm,
                           const struct whereClauseS *wc,
                           uint32_t *out_ids, long long cap, long long *candidates, int probe_bool) {
    bool any_index = false;
    long long out = 0, cand = 0;
    for (const struct whereClauseS *c = wc; c; c = c->next) {
        if (c->attribute == NULL) continue;
        for (int i = 0; i < num_idx; i++) {
            if (strcmp(c->attribute, idx_attr[i]) != 0) continue;
            size_t off; FieldType real;
            if (!schema_lookup(idx_attr[i], &off, &real)) continue;
            int b = 0, e = 0;
            int op = op_code(c->operator);
            if (idx_type[i] == FIELD_UINT64 && real == FIELD_UINT64) {
                uint64_t v = strtoull(c->value, NULL, 10), lo = 0, hi = UINT64_MAX;
                if (op == 0) { lo = v; hi = v; }
                else if (op == 2) { lo = v + 1; }
                else if (op == 4) { lo = v; }
                else if (op == 3) { hi = v - 1; }
                else if (op == 5) { hi = v; }
                range_u64(rows, off, idx_perm[i], n, lo, hi, &b, &e);
            } else if (idx_type[i] == FIELD_INT && real == FIELD_INT) {
                int v = atoi(c->value), lo = INT_MIN, hi = INT_MAX;
                if (op == 0) { lo = v; hi = v; }
                else if (op == 2) { lo = (int)((unsigned)v + 1u); }
                else if (op == 4) { lo = v; }
                else if (op == 3) { hi = (int)((unsigned)v - 1u); }
                else if (op == 5) { hi = v; }
                range_i32(rows, off, idx_perm[i], n, lo, hi, &b, &e);
            } else if (probe_bool && idx_type[i] == FIELD_BOOL && real == FIELD_BOOL) {
                int lo, hi;
                bool_window(c->operator, c->value, &lo, &hi);
                range_bool(rows, off, idx_perm[i], n, lo, hi, &b, &e);
            } else {
                continue;                      /* S:425-433 */
            }
            any_index = true;
            if (e < b) e = b;
            cand += e - b;
            for (int k = b; k < e; k++) {      /* append, then re-filter (fused) */
                int row = idx_perm[i][k];
                if (orc_eval_where(&rows[row], wc)) {
                    if (out < cap) out_ids[out] = (uint32_t)row;
                    out++;
                }
            }
        }
    }
    if (candidates) *candidates = any_index ? cand : -1;
    if (any_index) return out;
    for (int i = 0; i < n; i++) {
        if (wc == NULL || orc_eval_where(&rows[i], wc)) {
            if (out < cap) out_ids[out] = (uint32_t)i;
            out++;
        }
    }
    return out;
}

/* get_attribute_string_value, S:216-248. */
void orc_attr_string(const record *r, const char *attr, char *buf, size_t buflen) {
    size_t off; FieldType t;
    if (!schema_lookup(attr, &off, &t)) { snprintf(buf, buflen, "NULL"); return; }
    const char *p = (const char *)r + off;
    switch (t) {
    case FIELD_UINT64: snprintf(buf, buflen, "%llu", *(const unsigned long long *)p); break;
    case FIELD_INT: snprintf(buf, buflen, "%d", *(const int *)p); break;
    case FIELD_BOOL: snprintf(buf, buflen, "%s", *(const bool *)p ? "true" : "false"); break;
    default: snprintf(buf, buflen, "%s", p); break;
    }
}

/* ------------------------------------------------------------------------
 * Columnar twin
 * --------------------------------------------------------------------- */
static inline uint32_t code_at(const void *p, int w, uint64_t i) {
    switch (w) {
    case 1: return ((const uint8_t *)p)[i];
    case 2: return ((const uint16_t *)p)[i];
    default: return ((const uint32_t *)p)[i];
    }
}

static void view_of_columns(const struct orc_columns *t, uint64_t i, struct row_view *v) {
    v->command_id = t->command_id ? t->command_id[i] : 0;
    v->exit_code = t->exit_code ? t->exit_code[i] : 0;
    v->user_id = t->user_id ? t->user_id[i] : 0;
    v->risk_level = t->risk_level ? t->risk_level[i] : 0;
    v->sudo_used = t->sudo_used ? t->sudo_used[i] != 0 : false;
    for (int k = 0; k < S_COUNT; k++)
        v->s[k] = t->str_code[k] ? t->str_dict[k][code_at(t->str_code[k], t->str_code_width[k], i)] : "";
}

static long long scan_range(const struct orc_columns *t, const struct whereClauseS *wc,
                            uint64_t r0, uint64_t r1, uint32_t id_base,
                            uint32_t *out, long long cap) {
    long long m = 0;
    struct row_view v;
    for (uint64_t i = r0; i < r1; i++) {
        view_of_columns(t, i, &v);
        if (wc == NULL || eval_view(&v, wc)) {
            if (m < cap) out[m] = (uint32_t)(i + id_base);
            m++;
        }
    }
    return m;
}

long long orc_scan_columns(const struct orc_columns *t, const struct whereClauseS *wc,
                           uint32_t id_base, uint32_t *out_ids, long long cap, int nthreads) {
    if (nthreads <= 1) return scan_range(t, wc, 0, t->n_rows, id_base, out_ids, cap);
#ifdef _OPENMP
    long long *cnt = calloc((size_t)nthreads, sizeof *cnt);
    uint32_t **part = calloc((size_t)nthreads, sizeof *part);
    #pragma omp parallel num_threads(nthreads)
    {
        int r = omp_get_thread_num();
        uint64_t s, c;
        orc_partition(t->n_rows, nthreads, r, &s, &c);
        part[r] = malloc((c ? c : 1) * sizeof(uint32_t));
        cnt[r] = scan_range(t, wc, s, s + c, id_base, part[r], (long long)c);
    }
    long long total = 0;
    for (int r = 0; r < nthreads; r++) {
        for (long long k = 0; k < cnt[r]; k++, total++)
            if (total < cap) out_ids[total] = part[r][k];
        free(part[r]);
    }
    free(part); free(cnt);
    return total;
#else
    return scan_range(t, wc, 0, t->n_rows, id_base, out_ids, cap);
#endif
}

/* engine/mpi/executeEngine-mpi.c:703-715. */
void orc_partition(uint64_t n, int world, int rank, uint64_t *start, uint64_t *count) {
    uint64_t base = n / (uint64_t)world, rem = n % (uint64_t)world;
    if ((uint64_t)rank < rem) { *count = base + 1; *start = (uint64_t)rank * (base + 1); }
    else { *count = base; *start = rem * (base + 1) + ((uint64_t)rank - rem) * base; }
}
