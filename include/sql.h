/* sql.h -- lexer/parser API of the SQL front end.
 *
 * Contract header: the north star keeps the reference's tokenizer / sql.h
 * API unchanged, so the type layouts (Token 260 B, Condition 336 B,
 * ParsedSQL 6336 B) and the three entry points mirror the reference's
 * include/sql.h:7-84.  When the HIP engine is dropped into the reference
 * tree the reference's own tokenizer is linked; host/tokenizer.c here is a
 * fresh body with the same observable behaviour so this
 * repository is runnable on its own.
 */
#ifndef SQL_H
#define SQL_H

#include <stdbool.h>

typedef enum {
    TOKEN_KEYWORD, TOKEN_IDENTIFIER, TOKEN_SYMBOL, TOKEN_STRING, TOKEN_NUMBER, TOKEN_EOF
} TokenType;

typedef enum {
    CMD_NONE, CMD_DESCRIBE, CMD_SELECT, CMD_INSERT, CMD_DELETE, CMD_UNKNOWN
} CommandType;

typedef enum { OP_NONE, OP_EQ, OP_NEQ, OP_GT, OP_LT, OP_GTE, OP_LTE } OperatorType;

typedef enum { LOGIC_NONE, LOGIC_AND, LOGIC_OR } LogicOperator;

typedef struct {
    TokenType type;
    char value[256];
} Token;

typedef struct ParsedSQL ParsedSQL;

typedef struct {
    char column[64];
    OperatorType op;
    char value[256];
    bool is_numeric;
    bool is_nested;
    ParsedSQL *nested_sql;
} Condition;

struct ParsedSQL {
    CommandType command;
    char table[64];
    char columns[10][64];
    int num_columns;
    bool select_all;

    Condition conditions[5];
    LogicOperator logic_ops[4];
    int num_conditions;

    char insert_values[15][256];
    int num_values;

    char order_by[64];
    bool order_desc;
};

int tokenize(const char *input, Token tokens[], int max_tokens);
ParsedSQL parse_tokens(Token tokens[]);
void free_parsed_sql(ParsedSQL *sql);

#endif /* SQL_H */
