/* recordSchema.c -- column table of `record` and typed key helpers.
 * Fresh body for the API of the reference's engine/recordSchema.c:30-127
 * (same names, same results). */
#include "recordSchema.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define COL(name, kind) { #name, offsetof(record, name), kind }
static const FieldInfo k_fields[] = {
    COL(command_id, FIELD_UINT64), COL(raw_command, FIELD_STRING), COL(base_command, FIELD_STRING),
    COL(shell_type, FIELD_STRING), COL(exit_code, FIELD_INT), COL(timestamp, FIELD_STRING),
    COL(sudo_used, FIELD_BOOL), COL(working_directory, FIELD_STRING), COL(user_id, FIELD_INT),
    COL(user_name, FIELD_STRING), COL(host_name, FIELD_STRING), COL(risk_level, FIELD_INT),
};
#undef COL

const FieldInfo *get_field_info(const char *name) {
    if (!name) return NULL;
    for (size_t i = 0; i < sizeof k_fields / sizeof k_fields[0]; i++)
        if (strcmp(k_fields[i].name, name) == 0) return &k_fields[i];
    return NULL;
}

KEY_T extract_key_from_record(const record *rec, const char *attr_name) {
    const FieldInfo *fi = get_field_info(attr_name);
    if (!fi) {
        fprintf(stderr, "Unknown index attribute: %s\n", attr_name);
        exit(EXIT_FAILURE);
    }
    const char *p = (const char *)rec + fi->offset;
    KEY_T k;
    memset(&k, 0, sizeof k);
    switch (fi->type) {
    case FIELD_UINT64: k.type = KEY_UINT64; memcpy(&k.v.u64, p, sizeof k.v.u64); break;
    case FIELD_INT: k.type = KEY_INT; memcpy(&k.v.i32, p, sizeof k.v.i32); break;
    case FIELD_BOOL: k.type = KEY_BOOL; k.v.b = *(const bool *)p; break;
    default: k.type = KEY_STRING; k.v.str = p; break;
    }
    return k;
}

int compare_key(KEY_T a, KEY_T b) {
    if (a.type != b.type) return (int)a.type - (int)b.type;
    switch (a.type) {
    case KEY_UINT64: return a.v.u64 < b.v.u64 ? -1 : (a.v.u64 > b.v.u64);
    case KEY_INT: return a.v.i32 < b.v.i32 ? -1 : (a.v.i32 > b.v.i32);
    case KEY_BOOL: return a.v.b == b.v.b ? 0 : (a.v.b ? 1 : -1);
    case KEY_STRING:
        if (!a.v.str || !b.v.str) return a.v.str ? 1 : (b.v.str ? -1 : 0);
        return strcmp(a.v.str, b.v.str);
    default:
        fprintf(stderr, "Unknown KEY_T type in compare_key\n");
        return 0;
    }
}
