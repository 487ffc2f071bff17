/* recordSchema.h -- name -> (offset, type) table of `record`, key helpers.
 *
 * Contract header; same public names as the reference's
 * include/recordSchema.h:9-29, bodies in host/recordSchema.c (written fresh).
 */
#ifndef RECORD_SCHEMA_H
#define RECORD_SCHEMA_H

#include <stddef.h>
#include <stdint.h>
#include "logType.h"
#include "bplus.h"

typedef enum { FIELD_UINT64, FIELD_INT, FIELD_STRING, FIELD_BOOL } FieldType;

typedef struct {
    const char *name;
    size_t offset;
    FieldType type;
} FieldInfo;

/* NULL when `name` is not a column of `record`. */
const FieldInfo *get_field_info(const char *name);
/* Exits the process on an unknown attribute, like the reference (recordSchema.c:44-47). */
KEY_T extract_key_from_record(const record *rec, const char *attr_name);
/* <0, 0, >0; differing key types order by enum value; false<true; strcmp for strings. */
int compare_key(KEY_T key1, KEY_T key2);

#endif /* RECORD_SCHEMA_H */
