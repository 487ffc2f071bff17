/* connectEngine.h -- parser -> engine bridge and per-query dispatcher.
 *
 * Contract header: API of the reference's include/connectEngine.h:11-41
 * (macros, run_test_query, convert_conditions, free_where_clause_list, the
 * optimalIndexes tables) kept unchanged; the body in host/connectEngine.c
 * dispatches to the *HIP entry points instead of *Serial.
 */
#ifndef CONNECT_ENGINE_H
#define CONNECT_ENGINE_H

#include <ctype.h>
#include <stdio.h>
#include <string.h>
#include "executeEngine-serial.h"
#include "sql.h"

#define DATA_FILE "data-generation/commands_50k.csv"
#define TABLE_NAME "commands"
#define MAX_TOKENS 100
#define ROW_LIMIT 20

static inline char *trim(char *s) {
    while (*s && isspace((unsigned char)*s)) s++;
    return s;
}

const char *get_operator_string(OperatorType op);
const char *get_logic_op_string(LogicOperator op);
struct whereClauseS *convert_conditions(ParsedSQL *parsed);
void free_where_clause_list(struct whereClauseS *head);
void run_test_query(struct engineS *engine, const char *query, int max_rows);

extern const char *optimalIndexes[];
extern const FieldType optimalIndexTypes[];
extern const int numOptimalIndexes;

#endif /* CONNECT_ENGINE_H */
