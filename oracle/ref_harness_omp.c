/* ref_harness_omp.c -- OUR glue around the reference's OpenMP engine (engine/omp), for pinning the BOOL-index
 * probing variant of the SELECT path (executeEngine-omp.c:362-494; engine/mpi has the same block).
 *
 * Compiled only by oracle/Makefile, only when /root/reference exists, together with the reference's own sources
 * (taken where they lie, never copied) into oracle/_ref/libqpeomp_ref.so.  With -fopenmp, as the reference's makefile
 * builds QPEOMP; tests/golden/make_golden.py runs it with OMP_NUM_THREADS=1: the append order of the candidates of
 * several indexes is a race in the reference (omp:366,481) and only the one-thread order is a well-defined target.
 * Below the parser: executeQuerySelectOMP on a caller-built whereClauseS list, serialised like refh_select_where.
 *
 * TEST INFRASTRUCTURE ONLY.  Not part of the product, not shipped.
 */
#define _POSIX_C_SOURCE 200809L
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bplus.h"
#include "executeEngine-omp.h"

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

void *refo_open(const char *csv, int num_idx, const char **names, const int *types) {
    return initializeEngineOMP(num_idx, names, types, csv, "commands");
}

void refo_close(void *e) { destroyEngineOMP((struct engineS *)e); }

int refo_num_records(void *e) { return ((struct engineS *)e)->num_records; }

long long refo_select_where(void *e, const char **items, int n_items, struct whereClauseS *wc, char *out, long long cap) {
    struct resultSetS *rs = executeQuerySelectOMP((struct engineS *)e, items, n_items, "commands", wc);
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
