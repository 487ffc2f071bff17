/* engineBench.c -- see include/engineBench.h.  Host-side driver code: only the public engine API is used. */
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
    int count_only, in_flight, warmup, queries, id;
    pthread_barrier_t *start, *timed;
    double t_first, t_last, issue_s, await_s;
    long long matches, mismatches;
    int failed;
};

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
            /* the warm-up queries are finished before the clock starts */
            while (held) {
                struct hipQueryTicket *tk = ring[head];
                if (awaitQueryHIP(tk, NULL) < 0) b->failed = 1;
                releaseQueryHIP(tk);
                head = (head + 1) % depth;
                held--;
            }
            pthread_barrier_wait(b->timed);
            b->t_first = wall();
        }
        if (held == depth || k == total) {
            /* the oldest ticket (at the end: all of them) */
            do {
                struct hipQueryTicket *tk = ring[head];
                const double t0 = wall();
                const long long n = awaitQueryHIP(tk, NULL);
                b->await_s += wall() - t0;
                if (n < 0) b->failed = 1;
                const int e = ring_engine[head] & 63;
                if (expect[e] < 0) expect[e] = n; else if (expect[e] != n) b->mismatches++;
                b->matches = n;
                releaseQueryHIP(tk);
                head = (head + 1) % depth;
                held--;
            } while (k == total && held);
            if (k == total) break;
        }
        const int e = (k + b->id) % b->n_engines;
        const double t0 = wall();
        struct hipQueryTicket *tk = b->count_only ? executeQueryCountAsyncHIP(b->engines[e], b->where)
                                                  : executeQuerySelectAsyncHIP(b->engines[e], b->where);
        if (k >= b->warmup) b->issue_s += wall() - t0;
        if (!tk) { b->failed = 1; break; }
        const int slot = (head + held) % depth;
        ring[slot] = tk;
        ring_engine[slot] = e;
        held++;
    }
    b->t_last = wall();
    return NULL;
}

int hipEngineBench(struct engineS **engines, int n_engines, struct whereClauseS *whereClause, int count_only,
                   int threads, int in_flight, int warmup, int queries, struct hipBenchResult *out) {
    if (!engines || n_engines < 1 || n_engines > 64 || !out || threads < 1 || threads > 64 || queries < 1 || warmup < 0) return -1;
    memset(out, 0, sizeof *out);
    struct bench_thread *bt = calloc((size_t)threads, sizeof *bt);
    pthread_t *tid = calloc((size_t)threads, sizeof *tid);
    pthread_barrier_t start, timed;
    if (!bt || !tid || pthread_barrier_init(&start, NULL, (unsigned)threads) != 0 || pthread_barrier_init(&timed, NULL, (unsigned)threads) != 0) {
        free(bt); free(tid);
        return -1;
    }
    for (int i = 0; i < threads; i++) {
        bt[i] = (struct bench_thread){ engines, n_engines, whereClause, count_only, in_flight, warmup, queries, i, &start, &timed, 0, 0, 0, 0, 0, 0, 0 };
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
    pthread_barrier_destroy(&start);
    pthread_barrier_destroy(&timed);
    free(bt); free(tid);
    return failed ? -1 : 0;
}
