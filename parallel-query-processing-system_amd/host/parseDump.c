/* parseDump.c -- text dumps of what the SQL front end hands to the engine.
 * Debug / test aid exported from libpqps_hip.so: the format is the one the
 * tests' harness around the reference build prints, so the same golden
 * strings (tests/golden/parse_golden.json) pin both front ends. */
#include <stdio.h>
#include <string.h>

#include "connectEngine.h"
#include "sql.h"

#define US "\x1f"
#define RS "\x1e"

struct out { char *p; size_t len, cap; };

static void put(struct out *o, const char *s) {
    const size_t n = strlen(s);
    if (o->len + n + 1 <= o->cap) { memcpy(o->p + o->len, s, n); o->p[o->len + n] = '\0'; }
    o->len += n;
}

static void put_int(struct out *o, long long v) {
    char t[32];
    snprintf(t, sizeof t, "%lld", v);
    put(o, t);
}

long long hipDumpTokens(const char *sql, char *buf, long long cap) {
    Token tokens[MAX_TOKENS];
    const int n = tokenize(sql, tokens, MAX_TOKENS);
    struct out o = { buf, 0, (size_t)cap };
    if (cap > 0) buf[0] = '\0';
    for (int i = 0; i <= n && i < MAX_TOKENS; i++) {
        put_int(&o, tokens[i].type); put(&o, US); put(&o, tokens[i].value); put(&o, RS);
    }
    return (long long)o.len;
}

static void dump_where(struct out *o, const struct whereClauseS *wc) {
    put(o, "[");
    for (; wc; wc = wc->next) {
        if (wc->sub) {
            put(o, "(");
            dump_where(o, wc->sub);
            put(o, ")");
        } else {
            put(o, wc->attribute ? wc->attribute : "<null>"); put(o, US);
            put(o, wc->operator ? wc->operator : "<null>"); put(o, US);
            put(o, wc->value ? wc->value : "<null>"); put(o, US);
            put_int(o, wc->value_type);
        }
        put(o, US);
        put(o, wc->logical_op ? wc->logical_op : "<end>");
        put(o, RS);
    }
    put(o, "]");
}

long long hipDumpParse(const char *sql, char *buf, long long cap) {
    Token tokens[MAX_TOKENS];
    struct out o = { buf, 0, (size_t)cap };
    if (cap > 0) buf[0] = '\0';
    if (tokenize(sql, tokens, MAX_TOKENS) <= 0) { put(&o, "TOKENIZE_FAILED"); return (long long)o.len; }
    ParsedSQL p = parse_tokens(tokens);
    put_int(&o, p.command); put(&o, US);
    put(&o, p.table); put(&o, US);
    put_int(&o, p.select_all); put(&o, US);
    put_int(&o, p.num_columns);
    for (int i = 0; i < p.num_columns && i < 10; i++) { put(&o, US); put(&o, p.columns[i]); }
    put(&o, US);
    put_int(&o, p.num_values);
    for (int i = 0; i < p.num_values && i < 15; i++) { put(&o, US); put(&o, p.insert_values[i]); }
    put(&o, US);
    put(&o, p.order_by); put(&o, US);
    put_int(&o, p.order_desc); put(&o, RS);
    struct whereClauseS *wc = convert_conditions(&p);
    dump_where(&o, wc);
    free_where_clause_list(wc);
    free_parsed_sql(&p);
    return (long long)o.len;
}
