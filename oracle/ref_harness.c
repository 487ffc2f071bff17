/* ref_harness.c -- OUR glue around the REAL reference, for oracle pinning.
 *
 * Compiled only by oracle/Makefile, only when /root/reference exists, together
 * with the reference's own sources (taken where they lie, never copied) into
 * oracle/_ref/libqpeseq_ref.so.  It drives the reference's public API
 * (tokenize -> parse_tokens -> convert_conditions -> executeQuerySelectSerial
 * -> printTable, exactly the chain of connectEngine.c:125-233) and serialises
 * what comes back so tests/golden/make_golden.py can turn it into fixtures
 * (tests/test_oracle_golden.py then checks the oracle against them; bench.py's
 * cpu_baseline leg times linearSearchRecords of this library).
 *
 * TEST INFRASTRUCTURE ONLY.  Not part of the product, not shipped.
 */
#define _POSIX_C_SOURCE 200809L
#include <limits.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* reference headers (resolved through -I/root/reference/include) */
#include "bplus.h"
#include "connectEngine.h"
#include "executeEngine-serial.h"
#include "printHelper.h"
#include "sql.h"

#define US "\x1f"   /* cell separator   */
#define RS "\x1e"   /* record separator */

struct sbuf { char *p; size_t len, cap; };

static void sb_put(struct sbuf *b, const char *s) {
    size_t n = strlen(s);
    if (b->len + n + 1 > b->cap) { b->len += n; return; }   /* count, do not write */
    memcpy(b->p + b->len, s, n);
    b->len += n;
    b->p[b->len] = '\0';
}

static void sb_int(struct sbuf *b, long long v) {
    char t[32];
    snprintf(t, sizeof t, "%lld", v);
    sb_put(b, t);
}

void *refh_open(const char *csv, int num_idx, const char **names, const int *types) {
    return initializeEngineSerial(num_idx, names, types, csv, "commands");
}

void refh_close(void *e) { destroyEngineSerial((struct engineS *)e); }

int refh_num_records(void *e) { return ((struct engineS *)e)->num_records; }

/* Copies row i of the engine (1040 bytes) to out. */
void refh_get_record(void *e, int i, void *out) {
    memcpy(out, ((struct engineS *)e)->all_records[i], sizeof(record));
}

/* Runs one SELECT statement through the reference chain and serialises the
 * result set:  numRecords US numColumns US queryTime RS  name US ... RS
 * cell US ... RS ...   Returns the number of bytes needed (excluding NUL);
 * output is truncated (never overflowed) when that exceeds cap.
 * Returns -1 when the statement is not a SELECT. */
long long refh_select(void *e, const char *sql, char *out, long long cap) {
    Token tokens[MAX_TOKENS];
    if (tokenize(sql, tokens, MAX_TOKENS) <= 0) return -1;
    ParsedSQL parsed = parse_tokens(tokens);
    if (parsed.command != CMD_SELECT) { free_parsed_sql(&parsed); return -1; }
    const char *items[10];
    int n_items = 0;
    if (!parsed.select_all) {
        n_items = parsed.num_columns;
        for (int i = 0; i < n_items; i++) items[i] = parsed.columns[i];
    }
    struct whereClauseS *wc = convert_conditions(&parsed);
    struct resultSetS *rs = executeQuerySelectSerial((struct engineS *)e, items, n_items, parsed.table, wc);
    struct sbuf b = { out, 0, (size_t)cap };
    if (cap > 0) out[0] = '\0';
    sb_int(&b, rs->numRecords); sb_put(&b, US);
    sb_int(&b, rs->numColumns); sb_put(&b, US);
    sb_put(&b, rs->success ? "1" : "0"); sb_put(&b, RS);
    for (int j = 0; j < rs->numColumns; j++) { sb_put(&b, rs->columnNames[j]); sb_put(&b, US); }
    sb_put(&b, RS);
    for (int i = 0; i < rs->numRecords; i++) {
        for (int j = 0; j < rs->numColumns; j++) { sb_put(&b, rs->data[i][j]); sb_put(&b, US); }
        sb_put(&b, RS);
    }
    freeResultSet(rs);
    free_where_clause_list(wc);
    free_parsed_sql(&parsed);
    return (long long)b.len;
}

/* The engine API below the parser: executeQuerySelectSerial on a caller-built whereClauseS list (the parser
 * takes at most 5 conditions per level; the backend takes any list).  Same serialisation as refh_select. */
long long refh_select_where(void *e, const char **items, int n_items, struct whereClauseS *wc, char *out, long long cap) {
    struct resultSetS *rs = executeQuerySelectSerial((struct engineS *)e, items, n_items, "commands", wc);
    struct sbuf b = { out, 0, (size_t)cap };
    if (cap > 0) out[0] = '\0';
    sb_int(&b, rs->numRecords); sb_put(&b, US);
    sb_int(&b, rs->numColumns); sb_put(&b, US);
    sb_put(&b, rs->success ? "1" : "0"); sb_put(&b, RS);
    for (int j = 0; j < rs->numColumns; j++) { sb_put(&b, rs->columnNames[j]); sb_put(&b, US); }
    sb_put(&b, RS);
    for (int i = 0; i < rs->numRecords; i++) {
        for (int j = 0; j < rs->numColumns; j++) { sb_put(&b, rs->data[i][j]); sb_put(&b, US); }
        sb_put(&b, RS);
    }
    freeResultSet(rs);
    return (long long)b.len;
}

/* Same chain, result printed by the reference's printTable into `path`. */
int refh_print(void *e, const char *sql, int limit, const char *path) {
    Token tokens[MAX_TOKENS];
    if (tokenize(sql, tokens, MAX_TOKENS) <= 0) return -1;
    ParsedSQL parsed = parse_tokens(tokens);
    if (parsed.command != CMD_SELECT) { free_parsed_sql(&parsed); return -1; }
    const char *items[10];
    int n_items = 0;
    if (!parsed.select_all) {
        n_items = parsed.num_columns;
        for (int i = 0; i < n_items; i++) items[i] = parsed.columns[i];
    }
    struct whereClauseS *wc = convert_conditions(&parsed);
    struct resultSetS *rs = executeQuerySelectSerial((struct engineS *)e, items, n_items, parsed.table, wc);
    rs->queryTime = 0.0;                 /* make the footer reproducible */
    FILE *f = fopen(path, "w");
    if (!f) return -2;
    printTable(f, rs, limit);
    fclose(f);
    freeResultSet(rs);
    free_where_clause_list(wc);
    free_parsed_sql(&parsed);
    return 0;
}

/* Token stream: type US value RS ... */
long long refh_tokens(const char *sql, char *out, long long cap) {
    Token tokens[MAX_TOKENS];
    int n = tokenize(sql, tokens, MAX_TOKENS);
    struct sbuf b = { out, 0, (size_t)cap };
    if (cap > 0) out[0] = '\0';
    for (int i = 0; i <= n && i < MAX_TOKENS; i++) {
        sb_int(&b, tokens[i].type); sb_put(&b, US); sb_put(&b, tokens[i].value); sb_put(&b, RS);
    }
    return (long long)b.len;
}

static void dump_where(struct sbuf *b, const struct whereClauseS *wc) {
    sb_put(b, "[");
    for (; wc; wc = wc->next) {
        if (wc->sub) {
            sb_put(b, "(");
            dump_where(b, wc->sub);
            sb_put(b, ")");
        } else {
            sb_put(b, wc->attribute ? wc->attribute : "<null>"); sb_put(b, US);
            sb_put(b, wc->operator ? wc->operator : "<null>"); sb_put(b, US);
            sb_put(b, wc->value ? wc->value : "<null>"); sb_put(b, US);
            sb_int(b, wc->value_type);
        }
        sb_put(b, US);
        sb_put(b, wc->logical_op ? wc->logical_op : "<end>");
        sb_put(b, RS);
    }
    sb_put(b, "]");
}

/* Everything the parser hands to the engine for one statement:
 * command US table US select_all US ncols {US col} US nvalues {US value} US order_by US order_desc RS where-dump */
long long refh_parse(const char *sql, char *out, long long cap) {
    Token tokens[MAX_TOKENS];
    struct sbuf b = { out, 0, (size_t)cap };
    if (cap > 0) out[0] = '\0';
    if (tokenize(sql, tokens, MAX_TOKENS) <= 0) { sb_put(&b, "TOKENIZE_FAILED"); return (long long)b.len; }
    ParsedSQL parsed = parse_tokens(tokens);
    sb_int(&b, parsed.command); sb_put(&b, US);
    sb_put(&b, parsed.table); sb_put(&b, US);
    sb_int(&b, parsed.select_all); sb_put(&b, US);
    sb_int(&b, parsed.num_columns);
    for (int i = 0; i < parsed.num_columns && i < 10; i++) { sb_put(&b, US); sb_put(&b, parsed.columns[i]); }
    sb_put(&b, US);
    sb_int(&b, parsed.num_values);
    for (int i = 0; i < parsed.num_values && i < 15; i++) { sb_put(&b, US); sb_put(&b, parsed.insert_values[i]); }
    sb_put(&b, US);
    sb_put(&b, parsed.order_by); sb_put(&b, US);
    sb_int(&b, parsed.order_desc); sb_put(&b, RS);
    struct whereClauseS *wc = convert_conditions(&parsed);
    dump_where(&b, wc);
    free_where_clause_list(wc);
    free_parsed_sql(&parsed);
    return (long long)b.len;
}

/* Leaf order of index `idx` of the engine as row numbers: the full
 * findRange(min..max) walk, pointers mapped back to their position in
 * all_records.  rows_out needs num_records ints.  Returns the count. */
static int ptr_cmp(const void *a, const void *b) {
    uintptr_t x = (uintptr_t)((void *const *)a)[0], y = (uintptr_t)((void *const *)b)[0];
    return x < y ? -1 : x > y;
}

int refh_index_order(void *ev, int idx, int *rows_out) {
    struct engineS *e = (struct engineS *)ev;
    int n = e->num_records;
    if (idx < 0 || idx >= e->num_indexes || n == 0) return 0;
    KEY_T lo, hi;
    switch (e->attribute_types[idx]) {
    case FIELD_UINT64: lo.type = hi.type = KEY_UINT64; lo.v.u64 = 0; hi.v.u64 = UINT64_MAX; break;
    case FIELD_INT: lo.type = hi.type = KEY_INT; lo.v.i32 = INT_MIN; hi.v.i32 = INT_MAX; break;
    case FIELD_BOOL: lo.type = hi.type = KEY_BOOL; lo.v.b = false; hi.v.b = true; break;
    default: lo.type = hi.type = KEY_STRING; lo.v.str = ""; hi.v.str = "\xff\xff\xff\xff"; break;
    }
    KEY_T *keys = malloc((size_t)n * sizeof *keys);
    ROW_PTR *ptrs = malloc((size_t)n * sizeof *ptrs);
    int found = findRange(e->bplus_tree_roots[idx], lo, hi, false, keys, ptrs);
    void *(*map)[2] = malloc((size_t)n * sizeof *map);
    for (int i = 0; i < n; i++) { map[i][0] = e->all_records[i]; map[i][1] = (void *)(intptr_t)i; }
    qsort(map, (size_t)n, sizeof *map, ptr_cmp);
    for (int k = 0; k < found; k++) {
        int l = 0, r = n;
        while (l < r) { int m = (l + r) / 2; if ((uintptr_t)map[m][0] < (uintptr_t)ptrs[k]) l = m + 1; else r = m; }
        rows_out[k] = (int)(intptr_t)map[l][1];
    }
    free(map); free(keys); free(ptrs);
    return found;
}

/* Inclusive-window probe on index `idx` with an integer key pair, rows as
 * positions in all_records (findRange itself, bplus.c:282). */
int refh_find_range(void *ev, int idx, long long key_lo, long long key_hi, int *rows_out) {
    struct engineS *e = (struct engineS *)ev;
    int n = e->num_records;
    KEY_T lo, hi;
    if (e->attribute_types[idx] == FIELD_UINT64) {
        lo.type = hi.type = KEY_UINT64; lo.v.u64 = (uint64_t)key_lo; hi.v.u64 = (uint64_t)key_hi;
    } else if (e->attribute_types[idx] == FIELD_INT) {
        lo.type = hi.type = KEY_INT; lo.v.i32 = (int)key_lo; hi.v.i32 = (int)key_hi;
    } else {
        lo.type = hi.type = KEY_BOOL; lo.v.b = key_lo != 0; hi.v.b = key_hi != 0;
    }
    KEY_T *keys = malloc((size_t)(n + 1) * sizeof *keys);
    ROW_PTR *ptrs = malloc((size_t)(n + 1) * sizeof *ptrs);
    int found = findRange(e->bplus_tree_roots[idx], lo, hi, false, keys, ptrs);
    for (int k = 0; k < found; k++) {
        rows_out[k] = -1;
        for (int i = 0; i < n; i++) if (e->all_records[i] == ptrs[k]) { rows_out[k] = i; break; }
    }
    free(keys); free(ptrs);
    return found;
}
