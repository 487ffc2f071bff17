/* buildEngine-hip.h -- load step of the HIP engine: CSV -> rows -> per-column
 * device buffers (+ order-preserving dictionaries for the 7 string columns).
 *
 * Same role and naming as the reference's include/buildEngine-omp.h:30-35;
 * CSV rules follow engine/serial/buildEngine-serial.c:70-221 exactly
 * (SURVEY.md App. A.5): header line skipped, 1024-byte fgets window, quoted
 * fields with "" escape, missing trailing fields stay zero.
 */
#ifndef BUILDENGINE_HIP_H
#define BUILDENGINE_HIP_H

#include "bplus.h"
#include "executeEngine-hip.h"
#include "logType.h"
#include "recordSchema.h"
#include "pqps_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Column order of `record` / the CSV. */
enum {
    HIPCOL_COMMAND_ID = 0, HIPCOL_RAW_COMMAND, HIPCOL_BASE_COMMAND, HIPCOL_SHELL_TYPE,
    HIPCOL_EXIT_CODE, HIPCOL_TIMESTAMP, HIPCOL_SUDO_USED, HIPCOL_WORKING_DIRECTORY,
    HIPCOL_USER_ID, HIPCOL_USER_NAME, HIPCOL_HOST_NAME, HIPCOL_RISK_LEVEL,
    HIPCOL_COUNT
};

/* Sorted (strcmp byte order) distinct values of one string column. */
struct hipDictionary {
    int count;
    const char **values;     /* `count` strings, ascending; into `storage` or (after INSERT) strdup'd */
    char *storage;
    size_t storage_bytes;
};

/* One B+-tree replacement: device permutation + sorted keys. */
struct hipIndex {
    int column;              /* HIPCOL_*                                   */
    int key_kind;            /* 0 unsigned, 1 signed                       */
    uint32_t *perm_dev;      /* n rows, (key asc, row desc)                */
    void *keys_dev;          /* n keys, column width                       */
};

/* Locks of one engine (engine/hip/buildEngine-hip.c).  The OpenMP driver of the reference calls the engine
 * from several threads at once (QPEOMP.c:234-291).  SELECT / COUNT are READERS: each takes one of the table's
 * query LANES (result buffers of its own on every shard + a slot of the shards' query streams, pqps_qstream),
 * so several of them are on the device at once and none waits for another's result download; INSERT / DELETE /
 * index creation are WRITERS and run alone.  A reader may be finished by another thread than the one that began
 * it (asynchronous tickets), hence a counting gate and not a pthread rwlock. */
struct hipLocks;

/* Result buffers of one query in flight on one shard. */
struct hipLane {
    uint32_t *ids_dev;                           /* row numbers of the result, capacity_ids u32                 */
    uint64_t capacity_ids;
    uint64_t *count_dev;                         /* 8 x u64: count, spare, range[2], flag-pass count, spare ... */
    volatile uint64_t *count_host;               /* the same words as the host sees them (pinned, mapped): read after the
                                                    query's completion event, no download; NULL: count_dev is device memory */
    pqps_ctx *copy;                              /* context (stream) for this lane's downloads / gathers; NULL: the table's */
    uint32_t *merged_dev;                        /* shard 0 only, several shards: the gathered list of all shards */
    uint64_t merged_cap;
};

#define HIP_MAX_LANES 8

/* Device-resident table; hangs off engineS.record_block (same address for the engine's life). */
struct hipTable {
    pqps_ctx *ctx;
    uint64_t n_rows;
    uint64_t capacity_rows;                      /* allocation, multiple of PQPS_TILE_ROWS */
    pqps_column col[HIPCOL_COUNT];               /* device buffers; a dictionary column with ONE value may have none
                                                    (data NULL, width 0): every row carries code 0                  */
    struct hipDictionary dict[HIPCOL_COUNT];     /* string columns only                    */
    struct hipIndex *index;                      /* engine->num_indexes entries            */
    struct hipLane own;                          /* the table's own result buffers: writers, ad-hoc tables          */
    pqps_qstream *qs;                            /* engine tables: this shard's query stream, n_lanes slots         */
    int n_lanes;
    struct hipLane lane[HIP_MAX_LANES];
    record *row_block;                           /* contiguous host rows (all_records[i] point in); NULL: device only */
    size_t row_capacity;                         /* rows row_block / all_records have room for      */
    int device_only;                             /* the engine has no host rows (initializeEngineColumnsHIP / SyntheticHIP) */
    int probe_bool;                              /* index mode probes BOOL indexes too (hipEngineProbeBoolIndexes) */
    struct hipLocks *locks;                      /* engine tables only; NULL for ad-hoc tables      */
    /* Several devices in one process (PQPS_DEVICES=0,1,...): the engine's rows are split into contiguous
     * shards by the reference's block partition (executeEngine-mpi.c:703-715), one device table each.
     * shard[0] is this table; it alone owns the dictionaries (codes are global), the host rows and the
     * locks.  A table without shards (n_shards 0) is its own single shard. */
    int n_shards;
    struct hipTable **shard;
    uint64_t row0;                               /* engine row number of this shard's first row    */
    /* One process per GPU (hipEngineJoinRanksHIP): this process holds rows [row0, row0 + n_rows) of a table that `world`
     * processes hold between them; scan-mode SELECT / COUNT go through the exchange (pqps_exchange_*: the shard's scan +
     * all-gatherv of the row numbers / all-reduce of the counts over RCCL, engine/mpi/executeEngine-mpi.c:745-765), whose
     * ring slots are the table's lanes. */
    pqps_exchange *xch;
    int world, rank;
    uint64_t rows_total;
};

#define HIP_MAX_SHARDS 16
static inline int hipTableShards(const struct hipTable *t) { return t->n_shards > 1 ? t->n_shards : 1; }
static inline struct hipTable *hipTableShard(struct hipTable *t, int s) { return t->n_shards > 1 ? t->shard[s] : t; }

/* No-ops on a table without locks.  hipTableLockExclusive returns -1 (and takes nothing) when the calling thread holds a
 * query lane -- a ticket of its own it has not released: the writer would wait for that ticket for ever. */
void hipTableLockShared(struct hipTable *t);
void hipTableUnlockShared(struct hipTable *t);
int  hipTableLockExclusive(struct hipTable *t);
void hipTableUnlockExclusive(struct hipTable *t);
/* A free query lane of the table; -1 on a table without lanes (the query then uses the table's own buffers).  Waits while
 * all lanes are taken, but never for ever: HIP_LANE_REFUSED at once when the calling thread itself holds every lane, or
 * after PQPS_LANE_WAIT_MS (default 10 000) without a lane coming free; the reason is on stderr. */
#define HIP_LANE_REFUSED (-2)
int  hipTableAcquireLane(struct hipTable *t);
void hipTableReleaseLane(struct hipTable *t, int lane);
int  hipTableLaneCount(const struct hipTable *t);
/* test hooks: the gate alone on a table that has nothing else (tests/c/locks_test.c) */
void hipTableLocksCreate(struct hipTable *t, int n_lanes);
void hipTableLocksDestroy(struct hipTable *t);
/* Issuing calls on the shards' query streams are serialised (they take microseconds). */
void hipTableLockIssue(struct hipTable *t);
void hipTableUnlockIssue(struct hipTable *t);

/* CSV -> contiguous block of records + pointer array (reference signature of
 * getAllRecordsFromFileOMP, buildEngine-omp.h:31). */
record **getAllRecordsFromFileHIP(const char *filepath, int *num_records, void **record_block_out);
/* Parses one CSV line into a freshly calloc'd record (buildEngine-serial.c:159). */
record *getRecordFromLineHIP(char *line);
/* Same, into caller storage (zeroed first). */
void fillRecordFromLineHIP(record *dst, const char *line);
FieldType mapAttributeTypeHIP(int attributeType);
/* Builds the device index for one attribute and appends it to the engine. */
bool makeIndexHIP(struct engineS *engine, const char *indexName, int attributeType);

/* rows -> columns -> device.  Replaces engine->record_block (the row block
 * returned by getAllRecordsFromFileHIP) by the struct hipTable that owns it. */
bool buildDeviceTableHIP(struct engineS *engine);
/* Same, with the device context coming from a start-up begun earlier (hipBeginContextHIP runs
 * pqps_ctx_create on a background thread: the HIP runtime needs ~0.2 s, the CSV parse can use them). */
struct hipContextFuture;
struct hipContextFuture *hipBeginContextHIP(void);
bool buildDeviceTableOnHIP(struct engineS *engine, struct hipContextFuture *future);
/* Re-creates columns, dictionaries and indexes from engine->all_records
 * (after INSERT / DELETE changed the host rows). */
void rebuildDeviceTableHIP(struct engineS *engine);
/* INSERT: appends `r` (row number engine->num_records - 1) to the device table (its last shard) in place; false
 * when that needs a rebuild which a device-only engine cannot do (a dictionary outgrowing its code width). */
bool appendRowDeviceTableHIP(struct engineS *engine, const record *r);
/* DELETE: `delete_flags_dev[s]` (1 = row goes, from pqps_filter_flags on shard s) compacts that shard's
 * device columns in place; `expected_rows` = survivors counted on the host over all shards (cross-check). */
void compactDeviceTableHIP(struct engineS *engine, uint8_t *const *delete_flags_dev, size_t expected_rows);
void destroyDeviceTableHIP(struct engineS *engine);

/* Engines over device-resident columns (no host rows): see initializeEngineColumnsHIP / initializeEngineSyntheticHIP
 * in executeEngine-hip.h.  `columns` describes all 12 columns of `record`. */
struct hipColumnData;
bool buildDeviceTableFromColumnsHIP(struct engineS *engine, unsigned long long num_rows, const struct hipColumnData *columns);
bool buildSyntheticDeviceTableHIP(struct engineS *engine, unsigned long long num_rows, unsigned long long seed);
/* rows [first_row, first_row + num_rows) of the seeded table, as ONE shard whose row numbers start at first_row (a rank's
 * part of a table spread over several processes); `min_lanes`: at least that many query lanes */
bool buildSyntheticShardDeviceTableHIP(struct engineS *engine, unsigned long long num_rows, unsigned long long seed,
                                       unsigned long long first_row, int min_lanes);
/* Dictionary of a string column of the synthetic table (ascending strcmp order): the values behind the codes
 * pqps_synth_generate produces; NULL for a numeric column. */
const char *const *hipSyntheticDictionary(int column, int *count);

/* Lower-level pieces (also used for ad-hoc tables over caller-supplied rows,
 * see linearSearchRecords / evaluateWhereClause in executeEngine-hip.c). */
struct hipSchema;
struct hipTable *hipTableFromRows(pqps_ctx *ctx, record *const *rows, size_t n);
void hipTableFree(struct hipTable *t, int n_indexes);
void hipSchemaOfTable(const struct hipTable *t, struct hipSchema *s);

#ifdef __cplusplus
}
#endif
#endif /* BUILDENGINE_HIP_H */
