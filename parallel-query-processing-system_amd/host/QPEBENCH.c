/* QPEBENCH.c -- C driver: a stream of one SELECT through the HIP engine over the seeded synthetic table, no Python,
 * no torch.  Same front end as QPEHIP / the reference's drivers: the statement goes through the tokenizer, the sql.h
 * parser and convert_conditions (connectEngine.c:65-113); the engine API does the rest.
 *
 *   QPEBENCH [rows] [queries] [threads] [in_flight] [copies] ["SELECT ... WHERE ..."] [count]
 *
 * prints rows/s of whole queries (results left on the device) and the device time per launch as it ran in the stream. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "connectEngine.h"
#include "engineBench.h"
#include "sql.h"

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

    struct engineS *engines[8];
    for (int c = 0; c < copies; c++) {
        engines[c] = initializeEngineSyntheticHIP(rows, 0x5EED, 0, NULL, NULL, TABLE_NAME);
        if (!engines[c]) return EXIT_FAILURE;
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
    printf("{\"rows\": %llu, \"queries\": %lld, \"threads\": %d, \"in_flight\": %d, \"copies\": %d, \"mode\": \"%s\", \"matches\": %lld, "
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
