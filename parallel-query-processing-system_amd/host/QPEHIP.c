/* QPEHIP.c -- driver for the HIP engine: runs every statement of
 * sample-queries.txt (or argv[2]) against the CSV in argv[1].
 * Same flow and summary as the reference's QPESeq.c:16-96, engine swapped
 * for initializeEngineHIP / destroyEngineHIP. */
#define _POSIX_C_SOURCE 200809L
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "connectEngine.h"
#include "executeEngine-hip.h"

#define CYAN "\x1b[36m"
#define YELLOW "\x1b[33m"
#define BOLD "\x1b[1m"
#define RESET "\x1b[0m"

static double wall(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static char *slurp(const char *path) {
    FILE *fp = fopen(path, "r");
    if (!fp) return NULL;
    fseek(fp, 0, SEEK_END);
    const long size = ftell(fp);
    fseek(fp, 0, SEEK_SET);
    char *buf = malloc((size_t)size + 1);
    if (buf && fread(buf, 1, (size_t)size, fp) != (size_t)size) { free(buf); buf = NULL; }
    if (buf) buf[size] = '\0';
    fclose(fp);
    return buf;
}

int main(int argc, char *argv[]) {
    const char *data_file = argc > 1 ? argv[1] : DATA_FILE;
    const char *query_file = argc > 2 ? argv[2] : "sample-queries.txt";

    const double t_start = wall();
    struct engineS *engine = initializeEngineHIP(numOptimalIndexes, optimalIndexes,
                                                 (const int *)optimalIndexTypes, data_file, TABLE_NAME);
    const double t_init = wall();

    char *text = slurp(query_file);
    if (!text) {
        perror("Failed to open query file");
        destroyEngineHIP(engine);
        return EXIT_FAILURE;
    }
    const double t_load = wall();

    for (char *q = strtok(text, ";"); q; q = strtok(NULL, ";")) {
        q = trim(q);
        if (*q) run_test_query(engine, q, ROW_LIMIT);
    }
    free(text);
    destroyEngineHIP(engine);
    const double t_end = wall();

    printf(CYAN "======= HIP Execution Summary =======" RESET "\n");
    printf(CYAN "Engine Initialization Time: " RESET YELLOW "%.4f seconds\n" RESET, t_init - t_start);
    printf(CYAN "Query Loading Time: " RESET YELLOW "%.4f seconds\n" RESET, t_load - t_init);
    printf(CYAN "Query Execution Time: " RESET YELLOW "%.4f seconds\n" RESET, t_end - t_load);
    printf(BOLD CYAN "Total Execution Time: " RESET BOLD YELLOW "%.4f seconds" RESET "\n", t_end - t_start);
    printf(CYAN "=====================================" RESET "\n");
    return EXIT_SUCCESS;
}
