/* engineBench.c -- see include/engineBench.h.  Host-side driver code: only the public engine API is used.
 * Built into libpqps_bench.so and QPEBENCH, not into the product library. */
#define _POSIX_C_SOURCE 200809L
#include "engineBench.h"

#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

static double wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

struct bench_thread {
    struct engineS **engines;
    int n_engines;
    struct whereClauseS *where;
    int count_only, in_flight, warmup, queries, id, checksum;
    pthread_barrier_t *start, *timed;
    volatile int *abort_all;             /* a thread that cannot go on tells the others; everybody still reaches the barriers */
    double t_first, t_last, issue_s, await_s;
    long long matches, mismatches;
    unsigned long long sums[2];
    int have_sums;
    int failed;
};

/* awaits and releases the oldest `n` tickets of the ring */
static void drain(struct bench_thread *b, struct hipQueryTicket **ring, const int *ring_engine, long long *expect, int depth,
                  int *head, int *held, int n, int timed, int last_of_run) {
    while (n-- > 0 && *held > 0) {
        struct hipQueryTicket *tk = ring[*head];
        const double t0 = wall();
        const long long m = awaitQueryHIP(tk, NULL);
        if (timed) b->await_s += wall() - t0;
        if (m < 0) { b->failed = 1; *b->abort_all = 1; }
        else if (timed) {
            const int e = ring_engine[*head] & 63;
            if (expect[e] < 0) expect[e] = m; else if (expect[e] != m) b->mismatches++;
            b->matches = m;
            /* the very last timed ticket of thread 0: its list is checksummed where it lies (after the clock has stopped) */
            if (last_of_run && *held == 1 && b->checksum && !b->count_only) {
                b->t_last = wall();
                b->have_sums = hipQueryChecksumHIP(tk, b->sums) == 0;
            }
        }
        releaseQueryHIP(tk);
        *head = (*head + 1) % depth;
        (*held)--;
    }
}

static void *bench_main(void *arg) {
    struct bench_thread *b = arg;
    enum { kMaxInFlight = 16 };
    struct hipQueryTicket *ring[kMaxInFlight];
    int ring_engine[kMaxInFlight];
    long long expect[64];
    for (int e = 0; e < 64; e++) expect[e] = -1;
    const int depth = b->in_flight < 1 ? 1 : (b->in_flight > kMaxInFlight ? kMaxInFlight : b->in_flight);
    int head = 0, held = 0;
    const int total = b->warmup + b->queries;
    pthread_barrier_wait(b->start);
    for (int k = 0; k <= total; k++) {
        if (k == b->warmup) {
            /* the warm-up queries are finished before the clock starts; EVERY thread arrives here, also after a failure */
            drain(b, ring, ring_engine, expect, depth, &head, &held, held, 0, 0);
            pthread_barrier_wait(b->timed);
            b->t_first = wall();
        }
        if (*b->abort_all) {                                     /* somebody failed: nothing new is issued */
            if (k < b->warmup) { k = b->warmup - 1; continue; }  /* (straight to the barrier) */
            break;
        }
        if (held == depth || k == total) {
            drain(b, ring, ring_engine, expect, depth, &head, &held, k == total ? held : 1, k >= b->warmup, k == total && b->id == 0);
            if (k == total) break;
        }
        const int e = (k + b->id) % b->n_engines;
        const double t0 = wall();
        struct hipQueryTicket *tk = b->count_only ? executeQueryCountAsyncHIP(b->engines[e], b->where)
                                                  : executeQuerySelectAsyncHIP(b->engines[e], b->where);
        if (k >= b->warmup) b->issue_s += wall() - t0;
        if (!tk) { b->failed = 1; *b->abort_all = 1; continue; } /* (the engine said why; the loop above winds down) */
        const int slot = (head + held) % depth;
        ring[slot] = tk;
        ring_engine[slot] = e;
        held++;
    }
    drain(b, ring, ring_engine, expect, depth, &head, &held, held, 0, 0);      /* nothing stays behind, whatever happened */
    if (b->t_last == 0.0) b->t_last = wall();
    return NULL;
}

int hipEngineBench(struct engineS **engines, int n_engines, struct whereClauseS *whereClause, int count_only,
                   int threads, int in_flight, int warmup, int queries, struct hipBenchResult *out) {
    if (!engines || n_engines < 1 || n_engines > 64 || !out || threads < 1 || threads > 64 || queries < 1 || warmup < 0) return -1;
    const int want_checksum = out->want_checksum;
    memset(out, 0, sizeof *out);
    /* Tickets keep their lane until they are released: `threads` threads with `in_flight` outstanding tickets spread over
     * n_engines engines hold up to threads * ceil(in_flight / n_engines) lanes of one engine.  More than it has would leave
     * every thread waiting for a lane that only the waiting threads could free -- refused here. */
    const int depth = in_flight < 1 ? 1 : (in_flight > 16 ? 16 : in_flight);
    const int per_engine = threads * ((depth + n_engines - 1) / n_engines);
    for (int e = 0; e < n_engines; e++) {
        const int lanes = hipEngineLanes(engines[e]);
        if (lanes > 0 && per_engine > lanes) {
            fprintf(stderr, "hipEngineBench: %d thread(s) x %d ticket(s) in flight over %d engine(s) need %d lanes of an engine that has %d "
                            "(PQPS_ENGINE_LANES)\n", threads, depth, n_engines, per_engine, lanes);
            return -2;
        }
    }
    struct bench_thread *bt = calloc((size_t)threads, sizeof *bt);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    pthread_barrier_t start, timed;
    volatile int abort_all = 0;
    if (!bt || !tid || pthread_barrier_init(&start, NULL, (unsigned)threads) != 0 || pthread_barrier_init(&timed, NULL, (unsigned)threads) != 0) {
        free(bt); free(tid);
        return -1;
    }
    for (int i = 0; i < threads; i++) {
        bt[i] = (struct bench_thread){ .engines = engines, .n_engines = n_engines, .where = whereClause, .count_only = count_only,
                                       .in_flight = depth, .warmup = warmup, .queries = queries, .id = i, .checksum = want_checksum && i == 0,
                                       .start = &start, .timed = &timed, .abort_all = &abort_all };
        if (i > 0 && pthread_create(&tid[i], NULL, bench_main, &bt[i]) != 0) { fprintf(stderr, "hipEngineBench: cannot start thread %d\n", i); exit(EXIT_FAILURE); }
    }
    bench_main(&bt[0]);
    for (int i = 1; i < threads; i++) pthread_join(tid[i], NULL);
    double first = bt[0].t_first, last = bt[0].t_last;
    int failed = 0;
    for (int i = 0; i < threads; i++) {
        if (bt[i].t_first < first) first = bt[i].t_first;
        if (bt[i].t_last > last) last = bt[i].t_last;
        out->issue_seconds += bt[i].issue_s;
        out->await_seconds += bt[i].await_s;
        out->mismatches += bt[i].mismatches;
        failed |= bt[i].failed;
    }
    out->seconds = last - first;
    out->queries = (long long)queries * threads;
    out->matches = bt[0].matches;
    out->have_checksum = bt[0].have_sums;
    out->checksum[0] = bt[0].sums[0];
    out->checksum[1] = bt[0].sums[1];
    pthread_barrier_destroy(&start);
    pthread_barrier_destroy(&timed);
    free(bt); free(tid);
    return failed ? -1 : 0;
}
