/* printHelper.c -- ASCII table output of a resultSetS.
 * Fresh body producing byte-identical output to the reference's
 * engine/printHelper.c:9-130 (the text format is the interface; pinned by
 * tests/golden/print_golden.json). */
#include "printHelper.h"

static void rule(FILE *out, const struct resultSetS *rs, const int *w) {
    fputc('+', out);
    for (int j = 0; j < rs->numColumns; j++) {
        for (int k = 0; k < w[j] + 2; k++) fputc('-', out);
        fputc('+', out);
    }
    fputc('\n', out);
}

void printHeader(FILE *output, struct resultSetS *result, int *colWidths) {
    if (!result || !result->columnNames) return;
    if (!output) output = stdout;
    fputc('|', output);
    for (int j = 0; j < result->numColumns; j++)
        fprintf(output, " %-*s |", colWidths[j], result->columnNames[j]);
    fputc('\n', output);
}

void printTable(FILE *output, struct resultSetS *result, int limit) {
    if (!output) output = stdout;
    if (!result || !result->data) {
        fprintf(output, "No data found.\n");
        return;
    }
    int shown = result->numRecords;
    if (limit > 0 && limit < shown) shown = limit;

    /* widths come from the header and from the rows that will be printed */
    int *w = malloc((size_t)(result->numColumns > 0 ? result->numColumns : 1) * sizeof *w);
    for (int j = 0; j < result->numColumns; j++) w[j] = (int)strlen(result->columnNames[j]);
    for (int i = 0; i < shown; i++) {
        if (!result->data[i]) continue;
        for (int j = 0; j < result->numColumns; j++) {
            const char *cell = result->data[i][j];
            if (cell && (int)strlen(cell) > w[j]) w[j] = (int)strlen(cell);
        }
    }

    rule(output, result, w);
    printHeader(output, result, w);
    rule(output, result, w);
    for (int i = 0; i < shown; i++) {
        fputc('|', output);
        if (!result->data[i]) {
            fprintf(output, " NULL ROW |\n");
            continue;
        }
        for (int j = 0; j < result->numColumns; j++)
            fprintf(output, " %-*s |", w[j], result->data[i][j] ? result->data[i][j] : "NULL");
        fputc('\n', output);
    }
    rule(output, result, w);
    if (limit > 0 && result->numRecords > limit)
        fprintf(output, "... (%d more records) ...\n", result->numRecords - limit);
    fprintf(output, "Total Records: %d | Query Time: %.4f seconds\n\n", result->numRecords, result->queryTime);
    free(w);
}
