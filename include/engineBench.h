/* engineBench.h -- a stream of queries through the ENGINE API, timed on the host: what bench.py --level engine and
 * the QPEBENCH driver report.  Host code above the drop-in boundary: it uses nothing but include/executeEngine-hip.h
 * (initializeEngine*HIP, executeQuery{Select,Count}AsyncHIP, awaitQueryHIP, releaseQueryHIP), the way the reference's
 * drivers use nothing but their engine header (QPEOMP.c:234-291 issues its queries from several threads at once). */
#ifndef ENGINE_BENCH_H
#define ENGINE_BENCH_H

#include "executeEngine-hip.h"

#ifdef __cplusplus
extern "C" {
#endif

struct hipBenchResult {
    double seconds;                 /* wall clock over the timed queries: first issue .. last result awaited and released */
    long long queries;              /* timed queries (all threads)                                                       */
    long long matches;              /* matching rows of the LAST timed query of thread 0 (a check value)                */
    long long mismatches;           /* timed queries whose count differed from the first query's on the same engine      */
    double issue_seconds;           /* host time inside the issuing calls, summed over the threads                        */
    double await_seconds;           /* ... inside awaitQueryHIP (mostly: waiting for the device)                          */
    int want_checksum;              /* IN: checksum the ID list of thread 0's last timed query on the device (hipQueryChecksumHIP),
                                       after the clock has stopped                                                          */
    int have_checksum;              /* OUT                                                                                  */
    unsigned long long checksum[2]; /* OUT: sum of the row numbers, sum of id[i] * (2 i + 1), mod 2^64                      */
};

/* `queries` queries per thread after `warmup` untimed ones, each thread keeping `in_flight` tickets outstanding
 * (1 = one blocking query at a time); query k of a thread goes to engines[k % n_engines] -- several engines over
 * copies of one table keep the Infinity Cache out of the measurement.  count_only: COUNT(*) instead of the ID list.
 * The results stay on the device.  Returns 0; -1 when a query failed (every thread then winds down: nothing new is issued,
 * every ticket is awaited and released, nobody is left at a barrier); -2 without running anything when the tickets asked for
 * cannot be outstanding at once: a ticket holds one of its engine's lanes until it is released, so threads *
 * ceil(in_flight / n_engines) must not exceed hipEngineLanes(engine). */
int hipEngineBench(struct engineS **engines, int n_engines, struct whereClauseS *whereClause, int count_only,
                   int threads, int in_flight, int warmup, int queries, struct hipBenchResult *out);

#ifdef __cplusplus
}
#endif
#endif /* ENGINE_BENCH_H */
