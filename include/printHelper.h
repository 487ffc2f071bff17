/* printHelper.h -- ASCII table printer for a resultSetS.
 * Contract header, API of the reference's include/printHelper.h:6-10
 * (output format of engine/printHelper.c:35-130). */
#ifndef PRINT_HELPER_H
#define PRINT_HELPER_H

#include <stdio.h>
#include "executeEngine-serial.h"

void printHeader(FILE *output, struct resultSetS *result, int *colWidths);
/* limit <= 0 prints every row; output == NULL means stdout. */
void printTable(FILE *output, struct resultSetS *result, int limit);

#endif /* PRINT_HELPER_H */
