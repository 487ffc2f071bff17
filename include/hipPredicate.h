/* hipPredicate.h -- WHERE list -> device predicate program.
 *
 * Host-side compiler of the HIP engine.  It turns the reference's
 * `struct whereClauseS` list (include/executeEngine-serial.h) into the
 * pqps_predicate the filter kernel executes, with exactly the semantics of
 * the serial engine:
 *   evaluateWhereClause  engine/serial/executeEngine-serial.c:292-316
 *                        (right-recursive AND/OR, no precedence, nesting via sub)
 *   checkCondition       :251-289  (literal typed by the COLUMN: strtoull / atoi /
 *                        "true"|"1" / raw text; unknown attribute or operator = false)
 *   CMP_NUM / CMP_STR    :18-123   (strcmp byte order for the 7 string columns)
 * Strings never reach the GPU: a string column is stored as order-preserving
 * dictionary codes (rank in strcmp order), so `col OP "literal"` becomes a
 * window on the code -- decided on the host by two binary searches.
 */
#ifndef HIP_PREDICATE_H
#define HIP_PREDICATE_H

#include <stddef.h>
#include <stdint.h>
#include "executeEngine-serial.h"
#include "pqps_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

enum { HIPKIND_U64 = 0, HIPKIND_I32 = 1, HIPKIND_BOOL = 2, HIPKIND_DICT = 3 };

/* What the compiler needs to know about one column of `record`. */
struct hipColumnInfo {
    int present;                 /* 0: column not materialised on the device       */
    int kind;                    /* HIPKIND_*                                      */
    uint32_t width;              /* bytes per row on the device: 1, 2, 4 or 8      */
    int dict_count;              /* HIPKIND_DICT: number of distinct values        */
    const char *const *dict;     /* ... ascending in strcmp order                  */
};

/* Indexed by HIPCOL_* (buildEngine-hip.h): the 12 columns in `record` order. */
struct hipSchema {
    struct hipColumnInfo col[PQPS_MAX_COLUMNS];
};

/* Compiles `where` (NULL = every row).  On success returns 0, fills *pred and
 * column_ids[0 .. pred->n_columns) with the HIPCOL_* id bound to each column
 * slot of the predicate.  Returns -1 and a message in err when the clause
 * cannot be expressed (more than PQPS_MAX_LEAVES reachable leaves, or a leaf
 * on a column that is not materialised). */
int hipCompileWhere(const struct hipSchema *schema, const struct whereClauseS *where,
                    pqps_predicate *pred, int column_ids[PQPS_MAX_COLUMNS],
                    char *err, size_t errlen);

/* A WHERE of any size as a sequence of passes.  Pass k < n_passes - 1 is evaluated into one flag byte per row
 * (pqps_filter_flags); a later pass reads those flags as a 1-byte column: column id PQPS_MAX_COLUMNS + k in its
 * column_ids.  The last pass yields the query's result.  Nearly every clause is a single pass; a clause with
 * more than PQPS_MAX_LEAVES reachable comparisons (or more than PQPS_MAX_COLUMNS columns once flags are
 * counted) is split along the reference's own evaluation order (evaluateWhereClause, serial:292-316). */
struct hipPass {
    pqps_predicate pred;
    int column_ids[PQPS_MAX_COLUMNS];
};
struct hipPlan {
    int n_passes;
    struct hipPass *pass;        /* malloc'd; hipPlanFree */
};
int hipCompileWherePlan(const struct hipSchema *schema, const struct whereClauseS *where,
                        struct hipPlan *plan, char *err, size_t errlen);
void hipPlanFree(struct hipPlan *plan);

/* Column name -> HIPCOL_* id, -1 if unknown. */
int hipColumnId(const char *name);

#ifdef __cplusplus
}
#endif
#endif /* HIP_PREDICATE_H */
