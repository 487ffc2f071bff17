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
 * There is no CPU fallback: without a gfx950 device / the HIP runtime
 * initializeEngineHIP prints the reason and exits, like the reference's engines on a failed start-up
 * allocation.  A query that fails later (a WHERE that cannot be compiled, a device error) prints the reason
 * and reports failure -- success = false / -1 -- and the engine stays usable.  (A device step that fails in
 * the middle of an INSERT / DELETE -- after the CSV and the host rows have changed -- is still fatal: the device
 * table could not be trusted afterwards.)
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

/* Columnar SELECT: the scalable form of resultSetS (executeEngine-serial.h:30-38, whose rows x columns
 * heap strings are the right shape for 50 k rows, not for 10^8).  Same row selection and order as
 * executeQuerySelectHIP; the selected columns are gathered ON THE DEVICE for the result rows
 * (pqps_project_column) and come back as typed arrays; text exists only for the cells someone asks for
 * (hipColumnarCellText, hipColumnarHead -> printTable), formatted exactly as get_attribute_string_value
 * (serial:216-248) would. */
struct hipColumnarResult {
    int numRecords;
    int numColumns;
    char **columnNames;
    int *columnKinds;                  /* HIPKIND_U64 / _I32 / _BOOL / _DICT (hipPredicate.h); -1 = unknown column ("NULL" cells) */
    void **values;                     /* per column, numRecords entries: uint64_t / int32_t / uint8_t / uint32_t dictionary code */
    const char *const **dictionaries;  /* per column: code -> C string for _DICT columns, else NULL.  Owned by the result: the
                                          codes are numbered 0 .. dictionarySizes[col]-1 over the distinct values this result
                                          holds, ascending in strcmp order */
    int *dictionarySizes;
    double queryTime;                  /* selection + device gather + download */
    bool success;
};
struct hipColumnarResult *executeQuerySelectColumnarHIP(struct engineS *engine, const char **selectItems, int numSelectItems,
                                                        struct whereClauseS *whereClause);
void freeColumnarResultHIP(struct hipColumnarResult *result);
/* malloc'd text of one cell. */
char *hipColumnarCellText(const struct hipColumnarResult *result, int row, int column);
/* The first `limit` rows (all if limit <= 0, printTable's own convention) as an ordinary result set, e.g. for printTable; numRecords of the
 * returned set is the FULL count, as printTable's footer reports it, and only `limit` rows of data exist --
 * free it with freeResultSetHead. */
struct resultSetS *hipColumnarHead(const struct hipColumnarResult *result, int limit);
void freeResultSetHead(struct resultSetS *head, int rows);

/* Number of device shards the engine's table is split into (1 unless PQPS_DEVICES names several devices);
 * `rows` (may be NULL, room for that many entries) receives the rows each shard holds. */
int hipEngineShards(struct engineS *engine, unsigned long long *rows, int capacity);

/* COUNT(*) through the backend API (the reference parser cannot express it,
 * SURVEY.md fact 10): scan-mode count of matching rows, no ID list. */
long long executeQueryCountHIP(struct engineS *engine, struct whereClauseS *whereClause);

#ifdef __cplusplus
}
#endif
#endif /* EXECUTE_ENGINE_HIP_H */
