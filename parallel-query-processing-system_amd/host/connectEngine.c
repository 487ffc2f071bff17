/* connectEngine.c -- parser -> HIP engine bridge and per-query dispatcher.
 *
 * API of the reference's connectEngine.c (convert_conditions :65-113,
 * free_where_clause_list :116-122, run_test_query :125-233, the optimalIndexes
 * tables :48-62) kept unchanged; the body dispatches SELECT / INSERT / DELETE
 * to the *HIP entry points (include/executeEngine-hip.h).  Messages printed
 * to stdout are the same text the reference prints.
 */
#define _POSIX_C_SOURCE 200809L
#include "connectEngine.h"
#include "executeEngine-hip.h"
#include "printHelper.h"

#include <stdlib.h>
#include <strings.h>
#include <time.h>

const char *optimalIndexes[] = { "command_id", "user_id", "risk_level", "exit_code", "sudo_used" };
const FieldType optimalIndexTypes[] = { FIELD_UINT64, FIELD_INT, FIELD_INT, FIELD_INT, FIELD_BOOL };
const int numOptimalIndexes = 5;

const char *get_operator_string(OperatorType op) {
    static const char *const text[] = { "=", "=", "!=", ">", "<", ">=", "<=" };   /* OP_NONE reads as "=" */
    return (op >= OP_NONE && op <= OP_LTE) ? text[op] : "=";
}

const char *get_logic_op_string(LogicOperator op) {
    return op == LOGIC_OR ? "OR" : "AND";                        /* LOGIC_NONE joins as AND */
}

struct whereClauseS *convert_conditions(ParsedSQL *parsed) {
    struct whereClauseS *head = NULL, **link = &head;
    for (int i = 0; i < parsed->num_conditions; i++) {
        Condition *c = &parsed->conditions[i];
        struct whereClauseS *n = calloc(1, sizeof *n);
        if (!n) break;
        if (c->is_nested && c->nested_sql) {
            n->sub = convert_conditions(c->nested_sql);          /* strings alias the caller's ParsedSQL */
        } else {
            n->attribute = c->column;
            n->operator = get_operator_string(c->op);
            n->value = c->value;
            n->value_type = c->is_numeric ? 0 : 1;
        }
        n->logical_op = (i < parsed->num_conditions - 1) ? get_logic_op_string(parsed->logic_ops[i]) : NULL;
        *link = n;
        link = &n->next;
    }
    return head;
}

void free_where_clause_list(struct whereClauseS *head) {
    while (head) {
        struct whereClauseS *next = head->next;
        free(head);
        head = next;
    }
}

static void bounded_copy(char *dst, size_t cap, const char *src) {
    snprintf(dst, cap, "%.*s", (int)cap - 1, src);
}

static double cpu_seconds_since(clock_t t0) { return (double)(clock() - t0) / CLOCKS_PER_SEC; }

void run_test_query(struct engineS *engine, const char *query, int max_rows) {
    printf("Executing Query: %s\n", query);

    Token tokens[MAX_TOKENS];
    if (tokenize(query, tokens, MAX_TOKENS) <= 0) {
        printf("Tokenization failed.\n");
        return;
    }
    ParsedSQL parsed = parse_tokens(tokens);

    if (parsed.command == CMD_INSERT) {
        if (parsed.num_values != 12) {
            printf("Error: INSERT requires exactly 12 values.\n");
            return;
        }
        record r;
        memset(&r, 0, sizeof r);
        char (*v)[256] = parsed.insert_values;
        r.command_id = strtoull(v[0], NULL, 10);
        bounded_copy(r.raw_command, sizeof r.raw_command, v[1]);
        bounded_copy(r.base_command, sizeof r.base_command, v[2]);
        bounded_copy(r.shell_type, sizeof r.shell_type, v[3]);
        r.exit_code = atoi(v[4]);
        bounded_copy(r.timestamp, sizeof r.timestamp, v[5]);
        r.sudo_used = (strcasecmp(v[6], "true") == 0 || strcmp(v[6], "1") == 0);
        bounded_copy(r.working_directory, sizeof r.working_directory, v[7]);
        r.user_id = atoi(v[8]);
        bounded_copy(r.user_name, sizeof r.user_name, v[9]);
        bounded_copy(r.host_name, sizeof r.host_name, v[10]);
        r.risk_level = atoi(v[11]);
        const clock_t t0 = clock();
        const bool ok = executeQueryInsertHIP(engine, parsed.table, &r);
        printf("Insert %s. Execution Time: %.6f\n\n", ok ? "successful" : "failed", cpu_seconds_since(t0));
        return;
    }

    if (parsed.command == CMD_DELETE) {
        struct whereClauseS *where = convert_conditions(&parsed);
        const clock_t t0 = clock();
        struct resultSetS *res = executeQueryDeleteHIP(engine, parsed.table, where);
        const double dt = cpu_seconds_since(t0);
        if (res) {
            printf("Delete successful. Rows affected: %d. Execution Time: %.6f\n\n", res->numRecords, dt);
            freeResultSet(res);
        } else {
            printf("Delete failed. Execution Time: %.6f\n\n", dt);
        }
        free_where_clause_list(where);
        free_parsed_sql(&parsed);
        return;
    }

    if (parsed.command == CMD_SELECT) {
        const char *items[10];
        int n_items = 0;
        if (!parsed.select_all)
            for (; n_items < parsed.num_columns && n_items < 10; n_items++) items[n_items] = parsed.columns[n_items];
        struct whereClauseS *where = convert_conditions(&parsed);
        struct resultSetS *res = executeQuerySelectHIP(engine, items, n_items, parsed.table, where);
        printTable(NULL, res, max_rows);
        if (res) freeResultSet(res);
        free_where_clause_list(where);
        free_parsed_sql(&parsed);
        printf("\n");
        return;
    }

    if (parsed.command == CMD_NONE) {
        printf("No command detected.\n");
        return;
    }
    fprintf(stderr, "Unsupported command.\n");
}
