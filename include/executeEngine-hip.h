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

/* ---- engines over device-resident columns: tables of 10^8 - 10^9 rows behind struct engineS --------------------
 * initializeEngineHIP builds the device table from 1040-byte host rows (the reference's `record`); a 1 G-row table
 * is 1.04 TB in that form and 26 GB as columns.  These two constructors build the SAME engine -- same query API,
 * same results -- without host rows: engine->all_records is NULL, engine->datafile is "" (INSERT / DELETE change
 * the device table only, no CSV is kept in step), and executeQuerySelectHIP produces its strings from values
 * gathered on the device.
 *
 * initializeEngineColumnsHIP: the caller hands over the 12 columns of `record` (include/logType.h:11-24) as arrays:
 *   numeric columns   command_id u64, exit_code / user_id / risk_level i32, sudo_used u8 (0 / 1)
 *   string columns    order-preserving dictionary codes (u8 / u16 / u32 by `width`) + the dictionary, ascending in
 *                     strcmp order, no duplicates; a column whose dictionary has ONE value needs no array
 *   values            host memory, or (on_device != 0) device memory of the engine's device -- copied either way,
 *                     the engine owns padded buffers with head-room for INSERT.
 * initializeEngineSyntheticHIP: the seeded synthetic table of the commands_* schema (pqps_synth_generate,
 * SURVEY.md App. B distributions), generated in place on the device(s); raw_command, timestamp and
 * working_directory are single-valued columns.  With PQPS_DEVICES=0,1,... the rows are sharded like any engine's. */
struct hipColumnData {
    const void *values;                /* num_rows entries of `width` bytes; NULL for a single-valued string column */
    unsigned int width;                /* bytes per entry: 8 / 4 / 1 as the column's type says; codes: 1, 2 or 4      */
    int on_device;                     /* values is device memory (of the engine's first device)                      */
    const char *const *dictionary;     /* string columns: `dictionary_count` C strings, ascending strcmp order         */
    int dictionary_count;
};
struct engineS *initializeEngineColumnsHIP(unsigned long long num_rows, const struct hipColumnData columns[12],
                                           int num_indexes, const char *indexed_attributes[], const int attribute_types[],
                                           const char *tableName);
struct engineS *initializeEngineSyntheticHIP(unsigned long long num_rows, unsigned long long seed,
                                             int num_indexes, const char *indexed_attributes[], const int attribute_types[],
                                             const char *tableName);

/* ---- asynchronous queries: several in flight, results left on the device -----------------------------------------
 * The engine's table has LANES (default 4, PQPS_ENGINE_LANES): result buffers + a slot of the table's query stream
 * (pqps_qstream: two launches in flight on two HIP streams, one for tables of 537 M rows and more).  Every SELECT /
 * COUNT takes a lane for its device phase -- concurrent callers of the synchronous functions (the reference's OpenMP
 * driver, QPEOMP.c:234-291) therefore overlap on the device -- and a caller can keep several queries in flight itself:
 *   t = executeQuerySelectAsyncHIP(engine, where)     enqueues the query (blocks only while every lane is taken)
 *   n = awaitQueryHIP(t, &result)                     waits for it: number of matching rows, -1 on error
 *   ...                                               result.ids_dev: the row numbers in QPESeq order, ON THE DEVICE
 *   releaseQueryHIP(t)                                the lane is free again, result.ids_dev is no longer valid
 * With several shards the shards' lists are gathered on shard 0's device by peer copies (the one-process form of
 * MPI_Allgather of the sizes + MPI_Allgatherv of the payload, engine/mpi/executeEngine-mpi.c:753-765; index mode:
 * merged by key on the device).  executeQueryCountAsyncHIP: COUNT(*), no list.  A ticket must be released. */
struct hipQueryTicket;
struct hipDeviceResult {
    long long count;                   /* matching rows (COUNT: the only field that means anything)     */
    const unsigned int *ids_dev;       /* `count` row numbers, device memory of device `device`         */
    int device;
    int n_shards;
    unsigned long long shard_count[16];/* matches found by each shard                                    */
};
/* Limits, so that no caller can wait for itself (the engine refuses instead of hanging a process that holds the GPU):
 *   - a thread may hold at most hipEngineLanes(engine) unreleased tickets of one engine; asking for one more returns NULL at
 *     once (reason on stderr).  Tickets held by SEVERAL threads can still add up to all lanes: a further request then waits
 *     for a release, at most PQPS_LANE_WAIT_MS (default 10 000), and returns NULL after that;
 *   - INSERT / DELETE / addAttributeIndexHIP / hipEngineProbeBoolIndexes / hipEngineKernelTiming wait until every ticket is
 *     released; called from a thread that holds a ticket itself they are refused (false / success = false / -1).  While such
 *     a call waits, threads that hold no ticket wait behind it with new queries; a thread that holds one may take more. */
int hipEngineLanes(struct engineS *engine);
struct hipQueryTicket *executeQuerySelectAsyncHIP(struct engineS *engine, struct whereClauseS *whereClause);
struct hipQueryTicket *executeQueryCountAsyncHIP(struct engineS *engine, struct whereClauseS *whereClause);
long long awaitQueryHIP(struct hipQueryTicket *ticket, struct hipDeviceResult *result /* may be NULL */);
void releaseQueryHIP(struct hipQueryTicket *ticket);
/* out[0] = sum of the answer's row numbers, out[1] = sum of id[i] * (2 i + 1), mod 2^64 -- computed where the list lies,
 * on the device; awaits the ticket first.  0, or -1 (a failed query, a COUNT ticket). */
int hipQueryChecksumHIP(struct hipQueryTicket *ticket, unsigned long long out[2]);

/* ---- one process per GPU: the reference's QPEMPI shape (QPEMPI.c:145-155: a C driver, one process per rank; the row
 * partition and the exchange of engine/mpi/executeEngine-mpi.c:703-768) ----------------------------------------------
 * Every process builds ITS rows of the table -- initializeEngineSyntheticRankHIP: rows [start, start + count) of the
 * seeded table by the reference's block partition, row numbers table-wide -- on its own GPU (PQPS_DEVICE), then joins the
 * others: rank 0 makes a 128-byte RCCL id (hipEngineRcclIdHIP) and hands it to the other ranks by whatever the host has (a
 * file, MPI_Bcast, torch.distributed), every rank calls hipEngineJoinRanksHIP -- or its two halves with an agreement of
 * the ranks in between, so that a rank that fails locally cannot leave the others inside the communicator's bring-up.
 * From then on the engine's SELECT / COUNT are the TABLE's: executeQuerySelectAsyncHIP enqueues the shard's scan and the
 * all-gatherv of the matching row numbers over RCCL (sizes, then exactly-sized payload at displacements, compact on the
 * wire: pqps_exchange_select), awaitQueryHIP hands EVERY rank the whole ascending list on its device (count = matches in
 * the table, shard_count[0] = this rank's); COUNT is the all-reduced count (mpi:745).  Rules: every rank issues the same
 * queries in the same order (one issuing thread per process, or an order the host guarantees); scan-mode queries of one
 * pass only (no index probes: the rank engines have no indexes); INSERT / DELETE are not exchanged.  `rccl_library`: the
 * librccl.so to load (e.g. /opt/rocm/lib/librccl.so).  Every host wait of the exchange is bounded
 * (PQPS_EXCHANGE_TIMEOUT_S): a query that cannot finish fails -- awaitQueryHIP returns -1 -- it never hangs. */
struct engineS *initializeEngineSyntheticRankHIP(unsigned long long rows_total, unsigned long long seed, int world, int rank,
                                                 const char *tableName);
int hipEngineRcclIdHIP(const char *rccl_library, void *id128);
int hipEngineJoinRanksHIP(struct engineS *engine, const char *rccl_library, const void *id128);
int hipEngineJoinPrepareHIP(struct engineS *engine, const char *rccl_library);
int hipEngineJoinConnectHIP(struct engineS *engine, const void *id128);
int hipEngineLeaveRanksHIP(struct engineS *engine);
/* payload bytes this rank has received: out[0] as they travelled, out[1] as u32 row numbers would have */
int hipEngineWireBytesHIP(struct engineS *engine, unsigned long long out[2], int reset);
/* out[0] = SELECTs whose answer arrived with the sizes (one collective, pqps_exchange_eager), out[1] = SELECTs finished,
 * out[2] = row numbers a rank's block has room for (0: off) */
int hipEngineEagerQueriesHIP(struct engineS *engine, unsigned long long out[3], int reset);

/* Device time of the engine's queries AS THEY RUN on the lanes (several in flight): the recorders of the shards' query
 * streams (pqps_qstream_set_timing), events on the dispatch packets.  hipEngineKernelTime sums over the launches
 * recorded since the last call: scan_ms = the filter launches alone, query_ms = whole queries (COUNT: + the
 * one-workgroup reduction).  Up to 4096 launches per lane between two calls. */
int hipEngineKernelTiming(struct engineS *engine, int enable);
int hipEngineKernelTime(struct engineS *engine, double *scan_ms, double *query_ms, int *launches);

/* Which of the reference's two SELECT row selections index mode follows.  Off (the default): QPESeq's -- only
 * u64 / int indexes are probed (engine/serial/executeEngine-serial.c:377-433).  On: QPEOMP's / QPEMPI's -- BOOL indexes
 * are probed as well (engine/omp/executeEngine-omp.c:424-459, same block in engine/mpi), which changes the answers of
 * queries with a top-level condition on an indexed BOOL column: rows come back in the index's order (key ascending, row
 * descending), a row once per probe that finds it, and an OR beside the probed condition loses its other side.  The
 * one-thread order of that engine is reproduced (its append order across threads is a race).  Also switched on for a
 * new engine by PQPS_PROBE_BOOL=1.  Returns the previous setting, -1 on a NULL engine. */
int hipEngineProbeBoolIndexes(struct engineS *engine, int enable);

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
