/* hipPredicate.c -- WHERE list -> pqps_predicate (see include/hipPredicate.h).
 *
 * Semantics follow the reference's serial evaluator
 * (engine/serial/executeEngine-serial.c, "S" below); nothing here touches the
 * GPU and nothing here evaluates rows: it only decides, per leaf, which
 * unsigned window of column values makes the leaf true.
 */
#define _POSIX_C_SOURCE 200809L
#include "hipPredicate.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <strings.h>

#define T_ACCEPT (-1)
#define T_REJECT (-2)
#define MAX_RAW_STEPS 512

enum { ST_LEAF = 0, ST_FALSE = 1, ST_TRUE = 2 };

struct step {
    int kind;
    int col;                 /* HIPCOL id */
    uint64_t lo, span;
    int neg;
    int t, f;                /* raw step index, T_ACCEPT or T_REJECT */
};

static const char *const k_names[PQPS_MAX_COLUMNS] = {
    "command_id", "raw_command", "base_command", "shell_type", "exit_code", "timestamp",
    "sudo_used", "working_directory", "user_id", "user_name", "host_name", "risk_level"
};

int hipColumnId(const char *name) {
    if (!name) return -1;
    for (int i = 0; i < PQPS_MAX_COLUMNS; i++)
        if (strcmp(name, k_names[i]) == 0) return i;
    return -1;
}

/* S:131-136 */
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

/* Window of an ordered unsigned domain [0, max] for `x OP v`. */
static void window_unsigned(struct step *s, int op, uint64_t v, uint64_t max) {
    s->kind = ST_LEAF;
    s->neg = 0;
    switch (op) {
    case 0: s->lo = v; s->span = 0; break;
    case 1: s->lo = v; s->span = 0; s->neg = 1; break;
    case 2: if (v >= max) s->kind = ST_FALSE; else { s->lo = v + 1; s->span = max - (v + 1); } break;
    case 3: if (v == 0) s->kind = ST_FALSE; else { s->lo = 0; s->span = v - 1; } break;
    case 4: if (v > max) s->kind = ST_FALSE; else { s->lo = v; s->span = max - v; } break;
    default: s->lo = 0; s->span = v > max ? max : v; break;
    }
}

/* Signed 32-bit column: windows on the two's-complement bit pattern, all
 * arithmetic mod 2^32 (the kernel tests (x - lo) <= span in u32). */
static void window_i32(struct step *s, int op, int v) {
    const uint32_t uv = (uint32_t)v, umin = 0x80000000u, umax = 0x7FFFFFFFu;
    s->kind = ST_LEAF;
    s->neg = 0;
    switch (op) {
    case 0: s->lo = uv; s->span = 0; break;
    case 1: s->lo = uv; s->span = 0; s->neg = 1; break;
    case 2: if (v == INT_MAX) s->kind = ST_FALSE; else { s->lo = uv + 1u; s->span = (uint32_t)(umax - (uv + 1u)); } break;
    case 3: if (v == INT_MIN) s->kind = ST_FALSE; else { s->lo = umin; s->span = (uint32_t)((uv - 1u) - umin); } break;
    case 4: s->lo = uv; s->span = (uint32_t)(umax - uv); break;
    default: s->lo = umin; s->span = (uint32_t)(uv - umin); break;
    }
    s->lo &= 0xFFFFFFFFull;
    s->span &= 0xFFFFFFFFull;
}

/* String column as order-preserving codes: strcmp(field, lit) OP 0 (S:23-26)
 * rewritten on ranks.  lb = #values < lit, ub = #values <= lit. */
static void window_dict(struct step *s, int op, const char *lit, const struct hipColumnInfo *ci) {
    int l = 0, r = ci->dict_count;
    while (l < r) { int m = l + (r - l) / 2; if (strcmp(ci->dict[m], lit) < 0) l = m + 1; else r = m; }
    const int lb = l;
    const int present = (lb < ci->dict_count && strcmp(ci->dict[lb], lit) == 0);
    const int ub = lb + present;
    const uint64_t top = 0xFFFFFFFFull;
    s->kind = ST_LEAF;
    s->neg = 0;
    switch (op) {
    case 0: if (!present) s->kind = ST_FALSE; else { s->lo = (uint64_t)lb; s->span = 0; } break;
    case 1: if (!present) s->kind = ST_TRUE; else { s->lo = (uint64_t)lb; s->span = 0; s->neg = 1; } break;
    case 2: s->lo = (uint64_t)ub; s->span = top - (uint64_t)ub; break;                 /* code >= ub */
    case 3: if (lb == 0) s->kind = ST_FALSE; else { s->lo = 0; s->span = (uint64_t)lb - 1; } break;   /* code < lb */
    case 4: s->lo = (uint64_t)lb; s->span = top - (uint64_t)lb; break;                 /* code >= lb */
    default: if (ub == 0) s->kind = ST_FALSE; else { s->lo = 0; s->span = (uint64_t)ub - 1; } break;  /* code < ub */
    }
}

/* Result flags of an earlier pass of a multi-pass plan, bound like a 1-byte column: its column id is
 * PQPS_MAX_COLUMNS + pass number, its leaf is `flags = 1`.  The planner names it "\x01<pass>". */
#define FLAG_ATTRIBUTE '\x01'
static const struct hipColumnInfo k_flag_column = { 1, HIPKIND_BOOL, 1, 0, NULL };

static const struct hipColumnInfo *col_info(const struct hipSchema *schema, int col) {
    return col >= PQPS_MAX_COLUMNS ? &k_flag_column : &schema->col[col];
}

/* Flag leaves exist only inside hipCompileWherePlan: the planner makes them (make_flag_leaf: the name "\x01<pass>" AND
 * the mark in value_type), and only passes that exist can be named -- `flag_passes` = the passes emitted so far
 * while the planner runs on the calling thread, 0 otherwise.  A caller-supplied list that names "\x01<n>" (the engine
 * API takes whereClauseS lists from anybody) is an unknown attribute like any other, never an index into flag
 * buffers that do not exist. */
#define FLAG_MARK 0x7F1A6
static _Thread_local int flag_passes;

static int node_column(const struct whereClauseS *c) {
    if (c->attribute && c->attribute[0] == FLAG_ATTRIBUTE) {
        const int pass = atoi(c->attribute + 1);
        return c->value_type == FLAG_MARK && pass >= 0 && pass < flag_passes ? PQPS_MAX_COLUMNS + pass : -1;
    }
    return hipColumnId(c->attribute);
}

struct builder {
    const struct hipSchema *schema;
    struct step st[MAX_RAW_STEPS];
    int n;
    char *err;
    size_t errlen;
    int failed;
};

static int count_leaves(const struct whereClauseS *wc) {
    int n = 0;
    for (; wc; wc = wc->next) n += wc->sub ? count_leaves(wc->sub) : 1;
    return n;
}

/* checkCondition S:251-289 + create_where_condition S:129-213 as a window. */
static void make_leaf(struct builder *b, struct step *s, const struct whereClauseS *c) {
    s->kind = ST_FALSE;
    s->col = -1;
    const int col = node_column(c);
    const int op = op_code(c->operator);
    if (col < 0 || op < 0 || c->value == NULL) return;          /* S:212, S:279: never true */
    const struct hipColumnInfo *ci = col_info(b->schema, col);
    s->col = col;
    if (col >= PQPS_MAX_COLUMNS) { s->kind = ST_LEAF; s->lo = 1; s->span = 0; s->neg = 0; return; }
    /* the literal is typed by the column, not by the token (S:256-276) */
    if (strcmp(c->attribute, "command_id") == 0) {
        window_unsigned(s, op, strtoull(c->value, NULL, 10), UINT64_MAX);
    } else if (strcmp(c->attribute, "risk_level") == 0 || strcmp(c->attribute, "exit_code") == 0 ||
               strcmp(c->attribute, "user_id") == 0) {
        window_i32(s, op, atoi(c->value));
    } else if (strcmp(c->attribute, "sudo_used") == 0) {
        const int lit = (strcasecmp(c->value, "true") == 0 || strcmp(c->value, "1") == 0);
        if (op == 0) { s->kind = ST_LEAF; s->lo = (uint64_t)lit; s->span = 0; s->neg = 0; }
        else if (op == 1) { s->kind = ST_LEAF; s->lo = (uint64_t)lit; s->span = 0; s->neg = 1; }
        else return;                                             /* S:207-210: no ordering comparators */
    } else {
        if (ci->present && ci->kind == HIPKIND_DICT) {
            window_dict(s, op, c->value, ci);
            /* a column with ONE value: every row carries code 0, the comparison is decided here and now (such a column
             * may have no device buffer at all: include/buildEngine-hip.h) */
            if (s->kind == ST_LEAF && ci->dict_count == 1) s->kind = ((s->lo == 0) != (s->neg != 0)) ? ST_TRUE : ST_FALSE;
        } else s->kind = ST_LEAF;                                /* reported as absent below */
    }
    if (s->kind == ST_LEAF && !ci->present) {
        b->failed = 1;
        if (b->err) snprintf(b->err, b->errlen, "column '%s' is not materialised on the device", c->attribute);
    }
}

/* Emits the steps of one chain starting at raw index `start`; T / F are the
 * targets of the whole chain.  evaluateWhereClause S:292-316:
 *   cur OR  rest  -> true: T,    false: rest
 *   cur AND rest  -> true: rest, false: F      (also any other / NULL logical_op, S:315)
 *   last          -> true: T,    false: F                                               */
static void emit_chain(struct builder *b, const struct whereClauseS *wc, int start, int T, int F) {
    for (; wc; wc = wc->next) {
        const int size = wc->sub ? count_leaves(wc->sub) : 1;
        const int next_start = start + size;
        int t = T, f = F;
        if (wc->next) {
            if (wc->logical_op && strcmp(wc->logical_op, "OR") == 0) f = next_start;
            else t = next_start;
        }
        if (wc->sub) {
            emit_chain(b, wc->sub, start, t, f);
        } else {
            struct step *s = &b->st[start];
            if (wc->attribute == NULL) { s->kind = ST_FALSE; s->col = -1; }   /* reference would crash; never true */
            else make_leaf(b, s, wc);
            s->t = t;
            s->f = f;
        }
        start = next_start;
    }
}

static int cmp_int(const void *a, const void *b) { return *(const int *)a - *(const int *)b; }

int hipCompileWhere(const struct hipSchema *schema, const struct whereClauseS *where,
                    pqps_predicate *pred, int column_ids[PQPS_MAX_COLUMNS],
                    char *err, size_t errlen) {
    memset(pred, 0, sizeof *pred);
    if (err && errlen) err[0] = '\0';
    for (int i = 0; i < PQPS_MAX_COLUMNS; i++) column_ids[i] = -1;
    if (where == NULL) { pred->truth = 1; return 0; }            /* S:864: no clause keeps every row */

    const int raw = count_leaves(where);
    if (raw > MAX_RAW_STEPS) {
        if (err) snprintf(err, errlen, "WHERE has %d leaves (limit %d)", raw, MAX_RAW_STEPS);
        return -1;
    }
    struct builder *b = calloc(1, sizeof *b);
    if (!b) { if (err) snprintf(err, errlen, "out of memory"); return -1; }
    b->schema = schema; b->n = raw; b->err = err; b->errlen = errlen;
    emit_chain(b, where, 0, T_ACCEPT, T_REJECT);

    /* fold constant leaves: alias[s] = where control really goes when it reaches s */
    int alias[MAX_RAW_STEPS];
    for (int s = raw - 1; s >= 0; s--) {
        struct step *st = &b->st[s];
        if (st->t >= 0) st->t = alias[st->t];
        if (st->f >= 0) st->f = alias[st->f];
        alias[s] = st->kind == ST_LEAF ? s : (st->kind == ST_TRUE ? st->t : st->f);
    }
    const int entry = alias[0];
    /* reachability (jumps only go forward) */
    char reach[MAX_RAW_STEPS];
    memset(reach, 0, sizeof reach);
    if (entry >= 0) reach[entry] = 1;
    for (int s = 0; s < raw; s++) {
        if (!reach[s]) continue;
        if (b->st[s].t >= 0) reach[b->st[s].t] = 1;
        if (b->st[s].f >= 0) reach[b->st[s].f] = 1;
    }
    int newidx[MAX_RAW_STEPS], n = 0;
    for (int s = 0; s < raw; s++) newidx[s] = reach[s] ? n++ : -1;

    /* only leaves that can actually be evaluated may name an absent column */
    int absent = 0;
    for (int s = 0; s < raw; s++)
        if (reach[s] && !col_info(schema, b->st[s].col)->present) absent = 1;
    if (b->failed && !absent) { b->failed = 0; if (err && errlen) err[0] = '\0'; }
    if (b->failed) { free(b); return -1; }

    if (entry < 0) {                                             /* constant predicate */
        pred->truth = (entry == T_ACCEPT) ? 1 : 0;
        free(b);
        return 0;
    }
    if (n > PQPS_MAX_LEAVES) {
        if (err) snprintf(err, errlen, "WHERE needs %d leaf comparisons (limit %d)", n, PQPS_MAX_LEAVES);
        free(b);
        return -1;
    }

    /* column slots: distinct columns of the reachable leaves, widest first (then
     * ascending HIPCOL id) -- the order the width-specialised kernels expect */
    int cols[PQPS_MAX_LEAVES], nc = 0;
    for (int s = 0; s < raw; s++) if (reach[s]) cols[nc++] = b->st[s].col;
    qsort(cols, (size_t)nc, sizeof cols[0], cmp_int);
    int n_cols = 0;
    for (int i = 0; i < nc; i++) {
        if (i != 0 && cols[i] == cols[i - 1]) continue;
        if (n_cols == PQPS_MAX_COLUMNS) {
            if (err) snprintf(err, errlen, "WHERE reads more than %d columns in one pass", PQPS_MAX_COLUMNS);
            free(b);
            return -1;
        }
        column_ids[n_cols++] = cols[i];
    }
    for (int i = 1; i < n_cols; i++) {                           /* stable insertion sort by width desc */
        const int c = column_ids[i];
        int j = i;
        while (j > 0 && col_info(schema, column_ids[j - 1])->width < col_info(schema, c)->width) { column_ids[j] = column_ids[j - 1]; j--; }
        column_ids[j] = c;
    }

    /* leaf slots sorted by column slot (stable in evaluation order) */
    int slot_of_step[PQPS_MAX_LEAVES];
    int k = 0;
    for (int c = 0; c < n_cols; c++) {
        for (int s = 0; s < raw; s++) {
            if (!reach[s] || b->st[s].col != column_ids[c]) continue;
            const struct step *st = &b->st[s];
            pred->leaf[k].column = (uint32_t)c;
            pred->leaf[k].negate = (uint32_t)st->neg;
            pred->leaf[k].lo = st->lo;
            pred->leaf[k].span = st->span;
            slot_of_step[newidx[s]] = k++;
        }
    }
    for (int s = 0; s < raw; s++) {
        if (!reach[s]) continue;
        const int i = newidx[s];
        const struct step *st = &b->st[s];
        pred->order[i] = (uint8_t)slot_of_step[i];
        pred->on_true[i] = st->t == T_ACCEPT ? PQPS_ACCEPT : st->t == T_REJECT ? PQPS_REJECT : (uint8_t)newidx[st->t];
        pred->on_false[i] = st->f == T_ACCEPT ? PQPS_ACCEPT : st->f == T_REJECT ? PQPS_REJECT : (uint8_t)newidx[st->f];
    }
    pred->n_leaves = (uint32_t)n;
    pred->n_columns = (uint32_t)n_cols;

    /* truth table over leaf SLOTS by walking the jump table for every assignment */
    if (n <= PQPS_TT_LEAVES) {
        uint64_t tt = 0;
        for (uint32_t m = 0; m < (1u << n); m++) {
            int s = 0;
            while (s < n) {
                const int r = (m >> pred->order[s]) & 1u;
                const uint8_t nx = r ? pred->on_true[s] : pred->on_false[s];
                if (nx >= PQPS_ACCEPT) { s = nx; break; }
                s = nx;
            }
            if (s == PQPS_ACCEPT) tt |= 1ull << m;
        }
        pred->truth = tt;
    }
    free(b);
    return 0;
}

/* ---- multi-pass plans ----------------------------------------------------------------------------- */

struct planner {
    const struct hipSchema *schema;
    struct hipPlan *plan;
    int capacity;
    void **arena; int n_arena, cap_arena;
    char *err; size_t errlen;
};

static void *plan_alloc(struct planner *P, size_t bytes) {
    if (P->n_arena == P->cap_arena) {
        const int cap = P->cap_arena ? 2 * P->cap_arena : 16;
        void **grown = realloc(P->arena, (size_t)cap * sizeof *grown);
        if (!grown) return NULL;
        P->arena = grown; P->cap_arena = cap;
    }
    void *p = calloc(1, bytes ? bytes : 1);
    if (p) P->arena[P->n_arena++] = p;
    return p;
}

/* distinct columns a chain would bind: table columns once each, every flag column */
static void chain_columns(const struct whereClauseS *wc, uint32_t *table_mask, int *flag_columns) {
    for (; wc; wc = wc->next) {
        if (wc->sub) { chain_columns(wc->sub, table_mask, flag_columns); continue; }
        const int col = node_column(wc);
        if (col >= PQPS_MAX_COLUMNS) *flag_columns += 1;
        else if (col >= 0) *table_mask |= 1u << col;
    }
}

static int element_leaves(const struct whereClauseS *e) { return e->sub ? count_leaves(e->sub) : 1; }

/* does [first, last] (elements of one array) fit one pass? */
static int span_fits(const struct whereClauseS *first, const struct whereClauseS *last) {
    int leaves = 0, flags = 0;
    uint32_t mask = 0;
    for (const struct whereClauseS *e = first; e <= last; e++) {
        leaves += element_leaves(e);
        if (e->sub) chain_columns(e->sub, &mask, &flags);
        else {
            const int col = node_column(e);
            if (col >= PQPS_MAX_COLUMNS) flags++; else if (col >= 0) mask |= 1u << col;
        }
    }
    return leaves <= PQPS_MAX_LEAVES && __builtin_popcount(mask) + flags <= PQPS_MAX_COLUMNS;
}

/* Compiles `chain` as the next pass; returns its number, -1 on failure. */
static int emit_pass(struct planner *P, const struct whereClauseS *chain) {
    struct hipPlan *plan = P->plan;
    if (plan->n_passes == P->capacity) {
        const int cap = P->capacity ? 2 * P->capacity : 4;
        struct hipPass *grown = realloc(plan->pass, (size_t)cap * sizeof *grown);
        if (!grown) { if (P->err) snprintf(P->err, P->errlen, "out of memory"); return -1; }
        plan->pass = grown; P->capacity = cap;
    }
    struct hipPass *pass = &plan->pass[plan->n_passes];
    flag_passes = plan->n_passes;                                  /* this pass may read the flags of the passes before it */
    if (hipCompileWhere(P->schema, chain, &pass->pred, pass->column_ids, P->err, P->errlen) != 0) return -1;
    flag_passes = plan->n_passes + 1;                              /* the planner may now name this pass's flags */
    return plan->n_passes++;
}

static int make_flag_leaf(struct planner *P, struct whereClauseS *e, int pass) {
    char *name = plan_alloc(P, 16);
    if (!name) { if (P->err) snprintf(P->err, P->errlen, "out of memory"); return -1; }
    snprintf(name, 16, "%c%d", FLAG_ATTRIBUTE, pass);
    e->attribute = name;
    e->operator = "=";
    e->value = "1";
    e->value_type = FLAG_MARK;
    e->sub = NULL;
    return 0;
}

/* A chain equivalent to `chain` that fits one pass; what does not fit is evaluated by earlier passes and
 * read back as flag leaves.  evaluateWhereClause (S:292-316) is right-recursive: `e AND rest` / `e OR rest`
 * only ever continue at the START of rest, so any suffix of the element list can be evaluated first and
 * replaced by one leaf; a parenthesised element is a value of its own and can be replaced likewise. */
static struct whereClauseS *fit_chain(struct planner *P, const struct whereClauseS *chain) {
    int n = 0;
    for (const struct whereClauseS *wc = chain; wc; wc = wc->next) n++;
    struct whereClauseS *el = plan_alloc(P, (size_t)n * sizeof *el);
    if (!el) { if (P->err) snprintf(P->err, P->errlen, "out of memory"); return NULL; }
    int i = 0;
    for (const struct whereClauseS *wc = chain; wc; wc = wc->next, i++) {
        el[i] = *wc;
        el[i].next = i + 1 < n ? &el[i + 1] : NULL;
    }
    for (i = 0; i < n; i++) {
        if (!el[i].sub || (span_fits(&el[i], &el[i]) && count_leaves(el[i].sub) < PQPS_MAX_LEAVES)) continue;
        struct whereClauseS *inner = fit_chain(P, el[i].sub);        /* too large for one pass even alone */
        const int pass = inner ? emit_pass(P, inner) : -1;
        if (pass < 0 || make_flag_leaf(P, &el[i], pass) != 0) return NULL;
    }
    while (!span_fits(&el[0], &el[n - 1])) {
        int first = n - 1;                                            /* longest suffix that fits a pass */
        while (first > 0 && span_fits(&el[first - 1], &el[n - 1])) first--;
        if (first == n - 1 && element_leaves(&el[n - 1]) == 1) {
            /* a lone leaf: replacing it gains nothing.  The element before it (a parenthesised one, or the
             * two would fit together) becomes a leaf first. */
            if (n < 2 || !el[n - 2].sub) { if (P->err) snprintf(P->err, P->errlen, "WHERE cannot be split into passes"); return NULL; }
            const int pass = emit_pass(P, el[n - 2].sub);
            if (pass < 0 || make_flag_leaf(P, &el[n - 2], pass) != 0) return NULL;
            continue;
        }
        const int pass = emit_pass(P, &el[first]);
        if (pass < 0 || make_flag_leaf(P, &el[first], pass) != 0) return NULL;
        el[first].next = NULL;
        el[first].logical_op = NULL;
        n = first + 1;
    }
    return el;
}

int hipCompileWherePlan(const struct hipSchema *schema, const struct whereClauseS *where,
                        struct hipPlan *plan, char *err, size_t errlen) {
    memset(plan, 0, sizeof *plan);
    struct planner P;
    memset(&P, 0, sizeof P);
    P.schema = schema; P.plan = plan; P.err = err; P.errlen = errlen;
    /* one pass whenever the clause allows it (constant leaves are folded away first) */
    flag_passes = 0;
    if (count_leaves(where) <= MAX_RAW_STEPS && emit_pass(&P, where) == 0) return 0;
    if (err && errlen) err[0] = '\0';
    plan->n_passes = 0;
    struct whereClauseS *fitted = fit_chain(&P, where);
    const int last = fitted ? emit_pass(&P, fitted) : -1;
    flag_passes = 0;
    for (int i = 0; i < P.n_arena; i++) free(P.arena[i]);
    free(P.arena);
    if (last < 0) { hipPlanFree(plan); return -1; }
    return 0;
}

void hipPlanFree(struct hipPlan *plan) {
    if (!plan) return;
    free(plan->pass);
    plan->pass = NULL;
    plan->n_passes = 0;
}
