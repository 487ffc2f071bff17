/* QPEBENCH.c -- C driver: a stream of one SELECT through the HIP engine over the seeded synthetic table, no Python,
 * no torch.  Same front end as QPEHIP / the reference's drivers: the statement goes through the tokenizer, the sql.h
 * parser and convert_conditions (connectEngine.c:65-113); the engine API does the rest.
 *
 *   QPEBENCH [rows] [queries] [threads] [in_flight] [copies] ["SELECT ... WHERE ..."] [count]
 *
 * prints rows/s of whole queries (results left on the device) and the device time per launch as it ran in the stream.
 *
 * One process per GPU -- the reference's QPEMPI shape (QPEMPI.c:145-155 is a C driver started once per rank): with
 * WORLD_SIZE > 1 in the environment (RANK, LOCAL_RANK as torchrun / mpirun-style launchers set them) `rows` is the TABLE's
 * size, every process builds its rows of it on GPU LOCAL_RANK (initializeEngineSyntheticRankHIP) and joins the others over
 * RCCL (hipEngineJoinRanksHIP; PQPS_RCCL_LIBRARY names the librccl.so, default /opt/rocm/lib/librccl.so); rank 0 writes the
 * 128-byte RCCL id to the file PQPS_ID_FILE (default /tmp/pqps_rccl_id.<MASTER_PORT or 0>.<copy>) and the other ranks wait
 * for it to appear (60 s).  Every ticket's answer is then the all-gathered list of the whole table; rank 0 prints. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "connectEngine.h"
#include "engineBench.h"
#include "sql.h"

/* the RCCL id from rank 0 to the others through a file (a C host has no other bootstrap here): written under a temporary
 * name and renamed, so that a reader never sees half of it */
static int share_id(const char *path, int rank, void *id128) {
    if (rank == 0) {
        char tmp[600];
        snprintf(tmp, sizeof tmp, "%s.tmp", path);
        FILE *f = fopen(tmp, "wb");
        if (!f || fwrite(id128, 1, 128, f) != 128) { perror("QPEBENCH: RCCL id file"); if (f) fclose(f); return -1; }
        fclose(f);
        return rename(tmp, path);
    }
    for (int tries = 0; tries < 6000; tries++) {
        FILE *f = fopen(path, "rb");
        if (f) {
            const size_t got = fread(id128, 1, 128, f);
            fclose(f);
            if (got == 128) return 0;
        }
        struct timespec ts = { 0, 10000000 };
        nanosleep(&ts, NULL);
    }
    fprintf(stderr, "QPEBENCH: rank %d: no RCCL id in %s after 60 s\n", rank, path);
    return -1;
}

int main(int argc, char *argv[]) {
    const unsigned long long rows = argc > 1 ? strtoull(argv[1], NULL, 10) : 100000000ull;
    const int queries = argc > 2 ? atoi(argv[2]) : 200;
    const int threads = argc > 3 ? atoi(argv[3]) : 1;
    const int in_flight = argc > 4 ? atoi(argv[4]) : 3;
    const int copies = argc > 5 ? atoi(argv[5]) : 2;
    const char *sql = argc > 6 ? argv[6] : "SELECT command_id FROM commands WHERE sudo_used = FALSE AND user_name = \"student1030\"";
    const int count_only = argc > 7 && strcmp(argv[7], "count") == 0;
    if (copies < 1 || copies > 8) { fprintf(stderr, "copies must be 1..8\n"); return EXIT_FAILURE; }

    Token tokens[MAX_TOKENS];
    const int n_tokens = tokenize(sql, tokens, MAX_TOKENS);
    ParsedSQL parsed = parse_tokens(tokens);
    (void)n_tokens;
    struct whereClauseS *where = convert_conditions(&parsed);

    const int world = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
    const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0;
    if (world > 1 && getenv("LOCAL_RANK") && !getenv("PQPS_DEVICE")) setenv("PQPS_DEVICE", getenv("LOCAL_RANK"), 1);
    const char *rccl = getenv("PQPS_RCCL_LIBRARY") ? getenv("PQPS_RCCL_LIBRARY") : "/opt/rocm/lib/librccl.so";

    struct engineS *engines[8];
    for (int c = 0; c < copies; c++) {
        if (world > 1) {
            engines[c] = initializeEngineSyntheticRankHIP(rows, 0x5EED, world, rank, TABLE_NAME);
            if (!engines[c]) return EXIT_FAILURE;
            char path[512], id[128];
            if (getenv("PQPS_ID_FILE")) snprintf(path, sizeof path, "%s.%d", getenv("PQPS_ID_FILE"), c);
            else snprintf(path, sizeof path, "/tmp/pqps_rccl_id.%s.%d", getenv("MASTER_PORT") ? getenv("MASTER_PORT") : "0", c);
            if (rank == 0 && hipEngineRcclIdHIP(rccl, id) != 0) return EXIT_FAILURE;
            if (share_id(path, rank, id) != 0 || hipEngineJoinRanksHIP(engines[c], rccl, id) != 0) return EXIT_FAILURE;
        } else {
            engines[c] = initializeEngineSyntheticHIP(rows, 0x5EED, 0, NULL, NULL, TABLE_NAME);
            if (!engines[c]) return EXIT_FAILURE;
        }
        hipEngineKernelTiming(engines[c], 1);
    }
    struct hipBenchResult r;
    /* a fresh process starts with the GPU's clocks down: a few hundred untimed queries (tens of milliseconds) first */
    const int rc = hipEngineBench(engines, copies, where, count_only, threads, in_flight, 400 / threads + 20, queries, &r);
    double scan_ms = 0.0, query_ms = 0.0;
    int launches = 0;
    for (int c = 0; c < copies; c++) {
        double e = 0.0, q = 0.0;
        int k = 0;
        if (hipEngineKernelTime(engines[c], &e, &q, &k) == 0) { scan_ms += e; query_ms += q; launches += k; }
    }
    if (rc != 0) fprintf(stderr, "QPEBENCH: a query failed\n");
    if (rank == 0) printf("{\"rows\": %llu, \"queries\": %lld, \"threads\": %d, \"in_flight\": %d, \"copies\": %d, \"mode\": \"%s\", \"matches\": %lld, "
           "\"mismatches\": %lld, \"seconds\": %.6f, \"us_per_query\": %.2f, \"rows_per_s\": %.4e, \"host_issue_us_per_query\": %.2f, "
           "\"host_await_us_per_query\": %.2f, \"in_stream_kernel_us\": %.2f, \"launches_timed\": %d}\n",
           rows, r.queries, threads, in_flight, copies, count_only ? "count" : "ids", r.matches, r.mismatches, r.seconds,
           r.seconds / (double)r.queries * 1e6, (double)rows * (double)r.queries / r.seconds, r.issue_seconds / (double)r.queries * 1e6,
           r.await_seconds / (double)r.queries * 1e6, launches ? query_ms / launches * 1e3 : 0.0, launches);
    free_where_clause_list(where);
    free_parsed_sql(&parsed);
    for (int c = 0; c < copies; c++) destroyEngineHIP(engines[c]);
    return rc == 0 ? EXIT_SUCCESS : EXIT_FAILURE;
}
