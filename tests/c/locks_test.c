/* locks_test.c -- the engine's reader / writer / lane gate on a table that has nothing but locks (no device).
 * Scenarios of the round-3 review: a thread that asks for more tickets than the engine has lanes; threads that hold all
 * lanes between them; a writer waiting while a ticket holder wants another ticket; a writer call from a ticket holder.
 * Every scenario must END (refusal or progress) -- the harness gives each a wall-clock budget and fails otherwise. */
#define _DEFAULT_SOURCE
#define _POSIX_C_SOURCE 200809L
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "buildEngine-hip.h"

static double now(void) { struct timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }
static int failures = 0;
#define CHECK(cond, what) do { if (!(cond)) { printf("FAIL %s (%s:%d)\n", what, __FILE__, __LINE__); failures++; } else printf("ok   %s\n", what); } while (0)

static struct hipTable T;

/* 1: one thread, more tickets than lanes -> refused at once */
static void one_thread_too_many(void) {
    int got[HIP_MAX_LANES + 1], n = hipTableLaneCount(&T);
    const double t0 = now();
    for (int i = 0; i < n; i++) { hipTableLockShared(&T); got[i] = hipTableAcquireLane(&T); }
    hipTableLockShared(&T);
    const int extra = hipTableAcquireLane(&T);
    const double dt = now() - t0;
    CHECK(extra == HIP_LANE_REFUSED, "a thread holding every lane is refused one more");
    CHECK(dt < 0.5, "... at once, not after the lane timeout");
    hipTableUnlockShared(&T);
    for (int i = 0; i < n; i++) { CHECK(got[i] >= 0 && got[i] < n, "lane handed out"); hipTableReleaseLane(&T, got[i]); hipTableUnlockShared(&T); }
}

/* 2: two threads hold two lanes each (4 lanes) and both ask for a third: both are refused after the bounded wait */
struct two_arg { pthread_barrier_t *b; int result; double waited; };
static void *two_main(void *p) {
    struct two_arg *a = p;
    hipTableLockShared(&T); const int l0 = hipTableAcquireLane(&T);
    hipTableLockShared(&T); const int l1 = hipTableAcquireLane(&T);
    pthread_barrier_wait(a->b);
    const double t0 = now();
    hipTableLockShared(&T);
    a->result = hipTableAcquireLane(&T);
    a->waited = now() - t0;
    if (a->result >= 0) hipTableReleaseLane(&T, a->result);
    hipTableUnlockShared(&T);
    pthread_barrier_wait(a->b);                                  /* nobody releases before both have been refused */
    hipTableReleaseLane(&T, l0); hipTableUnlockShared(&T);
    hipTableReleaseLane(&T, l1); hipTableUnlockShared(&T);
    return NULL;
}
static void two_threads_all_lanes(void) {
    pthread_barrier_t b;
    pthread_barrier_init(&b, NULL, 2);
    struct two_arg a[2] = { { &b, 0, 0 }, { &b, 0, 0 } };
    pthread_t t[2];
    for (int i = 0; i < 2; i++) pthread_create(&t[i], NULL, two_main, &a[i]);
    for (int i = 0; i < 2; i++) pthread_join(t[i], NULL);
    CHECK(a[0].result == HIP_LANE_REFUSED && a[1].result == HIP_LANE_REFUSED, "two threads holding all lanes between them: both refused");
    CHECK(a[0].waited < 5.0 && a[1].waited < 5.0, "... after the bounded wait (PQPS_LANE_WAIT_MS)");
    pthread_barrier_destroy(&b);
}

/* 3: thread A holds a ticket, a writer waits, A takes a second ticket (must not wait behind the writer), releases both,
 *    the writer gets in; a thread WITHOUT a ticket waits behind the writer meanwhile */
static volatile int writer_in = 0, writer_waiting = 0, bystander_in = 0;
static void *writer_main(void *p) {
    (void)p;
    writer_waiting = 1;
    const int rc = hipTableLockExclusive(&T);
    writer_in = rc == 0 ? 1 : -1;
    usleep(50000);
    hipTableUnlockExclusive(&T);
    return NULL;
}
static void *bystander_main(void *p) {
    (void)p;
    hipTableLockShared(&T);
    bystander_in = writer_in ? 1 : -1;                          /* -1: got in FRONT of the waiting writer */
    hipTableUnlockShared(&T);
    return NULL;
}
static void holder_passes_a_waiting_writer(void) {
    pthread_t w, by;
    hipTableLockShared(&T);
    const int l0 = hipTableAcquireLane(&T);
    pthread_create(&w, NULL, writer_main, NULL);
    while (!writer_waiting) usleep(1000);
    usleep(100000);                                              /* the writer sits in its wait now */
    pthread_create(&by, NULL, bystander_main, NULL);
    usleep(50000);
    const double t0 = now();
    hipTableLockShared(&T);                                      /* second ticket of the same thread */
    const int l1 = hipTableAcquireLane(&T);
    CHECK(now() - t0 < 0.5 && l1 >= 0, "a ticket holder takes another ticket while a writer waits");
    CHECK(writer_in == 0, "... and the writer is still waiting for the holder's tickets");
    CHECK(hipTableLockExclusive(&T) == -1, "a writer call from a ticket holder is refused");
    hipTableReleaseLane(&T, l1); hipTableUnlockShared(&T);
    hipTableReleaseLane(&T, l0); hipTableUnlockShared(&T);
    pthread_join(w, NULL);
    pthread_join(by, NULL);
    CHECK(writer_in == 1, "the writer got in once the tickets were released");
    CHECK(bystander_in == 1, "a thread without a ticket waited behind the writer");
}

static void *watchdog(void *p) { (void)p; sleep(60); printf("FAIL watchdog: a scenario hung\n"); fflush(stdout); _exit(3); }

int main(void) {
    setenv("PQPS_LANE_WAIT_MS", "300", 1);
    pthread_t wd;
    pthread_create(&wd, NULL, watchdog, NULL);
    memset(&T, 0, sizeof T);
    hipTableLocksCreate(&T, 4);
    CHECK(hipTableLaneCount(&T) == 4, "four lanes");
    one_thread_too_many();
    two_threads_all_lanes();
    holder_passes_a_waiting_writer();
    /* the gate is whole afterwards */
    hipTableLockShared(&T);
    const int l = hipTableAcquireLane(&T);
    CHECK(l == 0, "all lanes free again");
    hipTableReleaseLane(&T, l);
    hipTableUnlockShared(&T);
    CHECK(hipTableLockExclusive(&T) == 0, "a writer without tickets gets in");
    hipTableUnlockExclusive(&T);
    hipTableLocksDestroy(&T);
    printf(failures ? "FAILED %d\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
