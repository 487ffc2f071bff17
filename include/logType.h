/* logType.h -- the one row type of the `commands` table.
 *
 * Contract header: the HIP backend is a drop-in behind the reference's own
 * headers, so this struct must stay layout-identical (sizeof == 1040, x86-64)
 * to the reference's include/logType.h:11-24.  Offsets are asserted in
 * tests/test_abi_layout.py.  Only the layout is shared; the text is ours.
 */
#ifndef LOGTYPE_H
#define LOGTYPE_H

#include <stdbool.h>
#include <stddef.h>
#include <string.h>

typedef struct record {
    unsigned long long command_id;   /* @0    u64 key, file order in generated data */
    char raw_command[512];           /* @8    */
    char base_command[100];          /* @520  */
    char shell_type[20];             /* @620  */
    int exit_code;                   /* @640  */
    char timestamp[30];              /* @644  */
    bool sudo_used;                  /* @674  */
    char working_directory[200];     /* @675  */
    int user_id;                     /* @876  */
    char user_name[50];              /* @880  */
    char host_name[100];             /* @930  */
    int risk_level;                  /* @1032 */
} record;

#endif /* LOGTYPE_H */
