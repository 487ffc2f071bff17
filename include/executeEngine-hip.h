/* executeEngine-hip.h -- MI355X (gfx950) execute engine.
 *
 * Drop-in for the reference's OpenMP / MPI engines: same shape as
 * include/executeEngine-omp.h:8-51 with the suffix HIP, same structs
 * (executeEngine-serial.h), same ownership and error behaviour as the serial
 * engine that defines the results (engine/serial/executeEngine-serial.c):
 *
 *   executeQuerySelectHIP   replaces executeQuerySelectSerial :328-528
 *                           (and executeQuerySelectOMP omp:333, ...MPI mpi:332)
 *   initializeEngineHIP     replaces initializeEngineSerial :727-771
 *   destroyEngineHIP        replaces destroyEngineSerial :774-814
 *   addAttributeIndexHIP    replaces addAttributeIndexSerial :825-841
 *   executeQueryDeleteHIP   replaces executeQueryDeleteSerial :627-715
 *   executeQueryInsertHIP   replaces executeQueryInsertSerial :538-617
 *
 * Results are bit-exact with QPESeq (the serial engine): scan mode returns
 * ascending row order; index mode returns (key asc, row desc) per probed
 * top-level condition, concatenated, then re-filtered.
 * The engine fails loudly (stderr + exit) if no gfx950 device / HIP runtime
 * is available: there is no CPU fallback.
 */
#ifndef EXECUTE_ENGINE_HIP_H
#define EXECUTE_ENGINE_HIP_H

#include "executeEngine-serial.h"

#ifdef __cplusplus
extern "C" {
#endif

struct resultSetS *executeQuerySelectHIP(
    struct engineS *engine,
    const char **selectItems,        /* NULL / 0 items: all 12 columns            */
    int numSelectItems,
    const char *tableName,           /* ignored, as in the reference              */
    struct whereClauseS *whereClause /* NULL: every row                           */
);

bool executeQueryInsertHIP(struct engineS *engine, const char *tableName, const record *r);

struct resultSetS *executeQueryDeleteHIP(
    struct engineS *engine, const char *tableName, struct whereClauseS *whereClause);

struct engineS *initializeEngineHIP(
    int num_indexes,
    const char *indexed_attributes[],
    const int attribute_types[],     /* 0 = u64, 1 = int, 2 = string, 3 = bool    */
    const char *datafile,
    const char *tableName
);

void destroyEngineHIP(struct engineS *engine);

bool addAttributeIndexHIP(struct engineS *engine, const char *tableName,
                          const char *attributeName, int attributeType);

/* ---- HIP-engine extensions (not in the reference API) ------------------- */

/* The filter without the string projection: matching row numbers (the index
 * into engine->all_records) in QPESeq result order.  Returns the number of
 * matches, or -1 on error; *ids is malloc'd (caller frees).  This is what
 * executeQuerySelectHIP projects from, and what COUNT(*) reads. */
long long executeQuerySelectIdsHIP(struct engineS *engine,
                                   struct whereClauseS *whereClause,
                                   unsigned int **ids, double *queryTime);

/* COUNT(*) through the backend API (the reference parser cannot express it,
 * SURVEY.md fact 10): scan-mode count of matching rows, no ID list. */
long long executeQueryCountHIP(struct engineS *engine, struct whereClauseS *whereClause);

#ifdef __cplusplus
}
#endif
#endif /* EXECUTE_ENGINE_HIP_H */
