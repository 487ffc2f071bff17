/* tokenizer.c -- SQL lexer + parser behind include/sql.h.
 *
 * The north star keeps the reference's tokenizer / sql.h API unchanged; when
 * the HIP engine is dropped into the reference tree, the reference's own
 * tokenizer/src/tokenizer.c is what gets linked.  This file is a fresh body
 * with the same OBSERVABLE behaviour (token streams, ParsedSQL contents),
 * pinned against the compiled reference by tests/golden/parse_golden.json,
 * so that this repository runs stand-alone (the GPU box has no reference).
 * Behaviour notes (reference tokenizer.c:8-303):
 *   - keywords are case-insensitive and stored upper-case; AND / ASC are NOT
 *     keywords (matched case-sensitively as identifiers by the parser);
 *   - numbers are digit runs only; strings take '...' or "..." without escapes;
 *   - at most 5 conditions per nesting level.  Writing the logic operator of
 *     the 5th condition lands on num_conditions in the reference's struct
 *     layout (logic_ops[4] == num_conditions); the compiled reference really
 *     behaves that way (a WHERE with exactly five conditions ends up with
 *     zero), so the same well-defined store is made here explicitly.
 * Deliberate differences, only where the reference reads uninitialised
 * memory or never terminates: the cursor never moves past the EOF token,
 * SELECT lists that cannot make progress stop instead of looping forever,
 * and tokens / column lists / VALUES lists are clipped to their arrays.
 */
#include "sql.h"

#include <ctype.h>
#include <stdlib.h>
#include <string.h>

/* ---- lexer ------------------------------------------------------------------ */

static const char *const k_keywords[] = {
    "SELECT", "FROM", "WHERE", "ORDER", "BY", "DESC", "OR", "TRUE", "FALSE",
    "DESCRIBE", "INSERT", "INTO", "VALUES", "DELETE"
};

static void put_token(Token *t, TokenType type, const char *src, size_t len) {
    if (len > sizeof t->value - 1) len = sizeof t->value - 1;
    memcpy(t->value, src, len);
    t->value[len] = '\0';
    t->type = type;
}

static int is_word(unsigned char c) { return isalnum(c) || c == '_'; }

int tokenize(const char *input, Token tokens[], int max_tokens) {
    int n = 0;
    size_t pos = 0;
    while (input[pos] && n < max_tokens - 1) {
        const unsigned char c = (unsigned char)input[pos];
        if (isspace(c)) { pos++; continue; }
        if (c == '-' && input[pos + 1] == '-') {                 /* comment to end of line */
            while (input[pos] && input[pos] != '\n') pos++;
            continue;
        }
        if (strchr(";,()*=", c)) {
            put_token(&tokens[n++], TOKEN_SYMBOL, input + pos, 1);
            pos++;
            continue;
        }
        if (c == '>' || c == '<' || c == '!') {
            const size_t len = input[pos + 1] == '=' ? 2 : 1;
            put_token(&tokens[n++], TOKEN_SYMBOL, input + pos, len);
            pos += len;
            continue;
        }
        if (c == '"' || c == '\'') {
            const size_t start = ++pos;
            while (input[pos] && (unsigned char)input[pos] != c) pos++;
            put_token(&tokens[n++], TOKEN_STRING, input + start, pos - start);
            if (input[pos]) pos++;
            continue;
        }
        if (is_word(c)) {
            const size_t start = pos;
            if (isdigit(c)) {
                while (isdigit((unsigned char)input[pos])) pos++;
                if (!isalpha((unsigned char)input[pos])) {
                    put_token(&tokens[n++], TOKEN_NUMBER, input + start, pos - start);
                    continue;
                }
            }
            while (is_word((unsigned char)input[pos])) pos++;
            Token *t = &tokens[n++];
            put_token(t, TOKEN_IDENTIFIER, input + start, pos - start);
            char upper[sizeof t->value];
            size_t k = 0;
            for (; t->value[k]; k++) upper[k] = (char)toupper((unsigned char)t->value[k]);
            upper[k] = '\0';
            for (size_t w = 0; w < sizeof k_keywords / sizeof k_keywords[0]; w++) {
                if (strcmp(upper, k_keywords[w]) == 0) {
                    t->type = TOKEN_KEYWORD;
                    strcpy(t->value, upper);
                    break;
                }
            }
            continue;
        }
        pos++;                                                    /* anything else is skipped */
    }
    tokens[n].type = TOKEN_EOF;
    tokens[n].value[0] = '\0';
    return n;
}

/* ---- parser ----------------------------------------------------------------- */

static int is_val(const Token *t, const char *s) { return strcmp(t->value, s) == 0; }

/* advance, but never beyond the EOF token */
static void step(Token tokens[], int *i) {
    if (tokens[*i].type != TOKEN_EOF) (*i)++;
}

static OperatorType operator_of(const Token *t) {
    if (is_val(t, "=")) return OP_EQ;
    if (is_val(t, "!=")) return OP_NEQ;
    if (is_val(t, ">")) return OP_GT;
    if (is_val(t, "<")) return OP_LT;
    if (is_val(t, ">=")) return OP_GTE;
    if (is_val(t, "<=")) return OP_LTE;
    return OP_NONE;
}

static void parse_conditions(Token tokens[], int *i, ParsedSQL *sql) {
    while (tokens[*i].type != TOKEN_EOF && !is_val(&tokens[*i], "ORDER") &&
           !is_val(&tokens[*i], ";") && !is_val(&tokens[*i], ")")) {
        if (sql->num_conditions >= 5) break;
        if (sql->num_conditions < 0) break;
        Condition *cond = &sql->conditions[sql->num_conditions];
        cond->is_nested = false;
        cond->nested_sql = NULL;

        if (is_val(&tokens[*i], "(")) {
            step(tokens, i);
            cond->is_nested = true;
            cond->nested_sql = calloc(1, sizeof(ParsedSQL));
            if (cond->nested_sql) parse_conditions(tokens, i, cond->nested_sql);
            if (is_val(&tokens[*i], ")")) step(tokens, i);
        } else {
            if (tokens[*i].type == TOKEN_IDENTIFIER) {
                strncpy(cond->column, tokens[*i].value, sizeof cond->column - 1);
                cond->column[sizeof cond->column - 1] = '\0';
                step(tokens, i);
            }
            cond->op = operator_of(&tokens[*i]);
            step(tokens, i);                                       /* consumed even when it was no operator */
            const Token *v = &tokens[*i];
            if (v->type == TOKEN_STRING || v->type == TOKEN_NUMBER ||
                (v->type == TOKEN_KEYWORD && (is_val(v, "TRUE") || is_val(v, "FALSE")))) {
                strcpy(cond->value, v->value);
                cond->is_numeric = v->type == TOKEN_NUMBER;
                step(tokens, i);
            }
        }
        sql->num_conditions++;

        LogicOperator lg = LOGIC_NONE;
        if (is_val(&tokens[*i], "AND")) { lg = LOGIC_AND; step(tokens, i); }
        else if (is_val(&tokens[*i], "OR")) { lg = LOGIC_OR; step(tokens, i); }
        const int slot = sql->num_conditions - 1;
        if (slot < 4) sql->logic_ops[slot] = lg;
        else sql->num_conditions = (int)lg;        /* reference layout: logic_ops[4] IS num_conditions */
    }
}

static void copy_name(char *dst, size_t cap, const char *src) {
    strncpy(dst, src, cap - 1);
    dst[cap - 1] = '\0';
}

ParsedSQL parse_tokens(Token tokens[]) {
    ParsedSQL sql;
    memset(&sql, 0, sizeof sql);
    sql.command = CMD_NONE;
    int i = 0;
    if (tokens[0].type != TOKEN_KEYWORD) return sql;

    if (is_val(&tokens[0], "DESCRIBE")) {
        sql.command = CMD_DESCRIBE;
        i = 1;
        if (tokens[i].type == TOKEN_IDENTIFIER) copy_name(sql.table, sizeof sql.table, tokens[i].value);
    } else if (is_val(&tokens[0], "SELECT")) {
        sql.command = CMD_SELECT;
        i = 1;
        while (tokens[i].type != TOKEN_EOF) {
            const int before = i;
            if (is_val(&tokens[i], "*")) {
                sql.select_all = true;
                i++;
            } else if (tokens[i].type == TOKEN_IDENTIFIER) {
                if (sql.num_columns < 10) copy_name(sql.columns[sql.num_columns++], 64, tokens[i].value);
                i++;
            }
            if (is_val(&tokens[i], ",")) { i++; continue; }
            if (is_val(&tokens[i], "FROM")) break;
            if (tokens[i].type == TOKEN_EOF) break;
            if (i == before) break;                               /* reference would spin forever here */
        }
        if (is_val(&tokens[i], "FROM")) {
            i++;
            if (tokens[i].type == TOKEN_IDENTIFIER) { copy_name(sql.table, sizeof sql.table, tokens[i].value); i++; }
        }
        if (is_val(&tokens[i], "WHERE")) { i++; parse_conditions(tokens, &i, &sql); }
        if (is_val(&tokens[i], "ORDER")) {
            i++;
            if (is_val(&tokens[i], "BY")) {
                i++;
                if (tokens[i].type == TOKEN_IDENTIFIER) { copy_name(sql.order_by, sizeof sql.order_by, tokens[i].value); i++; }
                if (is_val(&tokens[i], "DESC")) sql.order_desc = true;
                else if (is_val(&tokens[i], "ASC")) sql.order_desc = false;
            }
        }
    } else if (is_val(&tokens[0], "INSERT")) {
        sql.command = CMD_INSERT;
        i = 1;
        if (is_val(&tokens[i], "INTO")) i++;
        if (tokens[i].type == TOKEN_IDENTIFIER) { copy_name(sql.table, sizeof sql.table, tokens[i].value); i++; }
        if (is_val(&tokens[i], "VALUES")) i++;
        if (is_val(&tokens[i], "(")) i++;
        while (tokens[i].type != TOKEN_EOF && !is_val(&tokens[i], ")")) {
            if (!is_val(&tokens[i], ",") && sql.num_values < 15)
                strcpy(sql.insert_values[sql.num_values++], tokens[i].value);
            i++;
        }
    } else if (is_val(&tokens[0], "DELETE")) {
        sql.command = CMD_DELETE;
        i = 1;
        if (is_val(&tokens[i], "FROM")) i++;
        if (tokens[i].type == TOKEN_IDENTIFIER) { copy_name(sql.table, sizeof sql.table, tokens[i].value); i++; }
        if (is_val(&tokens[i], "WHERE")) { i++; parse_conditions(tokens, &i, &sql); }
    } else {
        sql.command = CMD_UNKNOWN;
    }
    return sql;
}

void free_parsed_sql(ParsedSQL *sql) {
    for (int k = 0; k < sql->num_conditions && k < 5; k++) {
        Condition *c = &sql->conditions[k];
        if (c->is_nested && c->nested_sql) {
            free_parsed_sql(c->nested_sql);
            free(c->nested_sql);
            c->nested_sql = NULL;
        }
    }
}
