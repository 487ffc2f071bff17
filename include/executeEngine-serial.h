/* executeEngine-serial.h -- the three structs every engine of the reference
 * shares (engineS 72 B, resultSetS 48 B, whereClauseS 56 B) and the helper
 * prototypes the dispatcher relies on.
 *
 * Contract header: layouts follow the reference's
 * include/executeEngine-serial.h:15-56 bit for bit (asserted in
 * tests/test_abi_layout.py); the HIP backend (executeEngine-hip.h) includes
 * this file exactly the way the reference's executeEngine-omp.h:6 does.
 * There is no serial engine in this repository: the *Serial entry points of
 * the reference are not declared here.
 */
#ifndef EXECUTE_ENGINE_SERIAL_H
#define EXECUTE_ENGINE_SERIAL_H

#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "logType.h"
#include "recordSchema.h"

struct engineS {
    char *tableName;
    node **bplus_tree_roots;      /* HIP engine: array of NULLs, one per index          */
    int num_indexes;
    char **indexed_attributes;
    FieldType *attribute_types;
    record **all_records;         /* host row store (projection / INSERT / DELETE)      */
    int num_records;
    char *datafile;
    void *record_block;           /* engine-private slot; HIP engine: struct hipTable * */
};

struct resultSetS {
    int numRecords;
    int numColumns;
    char **columnNames;
    FieldType *columnTypes;       /* zero-filled placeholder, as in the reference        */
    char ***data;                 /* data[row][col], every cell heap allocated           */
    double queryTime;             /* seconds spent in index probe + filter               */
    bool success;
};

struct whereClauseS {
    const char *attribute;        /* NULL on a parenthesised node (see `sub`)            */
    const char *operator;         /* "=", "!=", "<", ">", "<=", ">="                     */
    const char *value;            /* literal as text; typed by the COLUMN, not the token */
    int value_type;               /* never read by any evaluator                         */
    struct whereClauseS *next;
    const char *logical_op;       /* "AND" / "OR" joining this node to `next`; NULL last */
    struct whereClauseS *sub;     /* nested chain                                        */
};

typedef bool (*compare_func_t)(const char *, const char *);
typedef bool (*compare_func_int_t)(const bool, const bool);

/* Shared helpers, reference names (executeEngine-serial.h:41,137-151).
 * In this repository they are implemented by the HIP engine
 * (engine/hip/executeEngine-hip.c): the row predicate always runs on the GPU. */
void freeResultSet(struct resultSetS *result);
int isAttributeIndexed(struct engineS *engine, const char *attributeName);
record **linearSearchRecords(record **records, int num_records,
                             struct whereClauseS *whereClause, int *matchingRecords);
bool evaluateWhereClause(record *r, struct whereClauseS *wc);

#endif /* EXECUTE_ENGINE_SERIAL_H */
