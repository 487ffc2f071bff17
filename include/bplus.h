/* bplus.h -- key/row types of the reference's B+ tree index.
 *
 * Contract header (types only).  The HIP backend does NOT rebuild the
 * pointer-based tree of the reference (engine/bplus.c); an index is a device
 * resident permutation sorted by (key ascending, row descending), which
 * reproduces the leaf order findRange() walks (reference engine/bplus.c:282-314,
 * duplicate placement :471-490, :511-517).  `node` is kept so that
 * `struct engineS` (executeEngine-serial.h) keeps its 72-byte layout; the HIP
 * engine leaves `bplus_tree_roots[i]` NULL.
 * Layout reference: include/bplus.h:22-44 (KEY_T 16 B, node 40 B).
 */
#ifndef BPLUS_SERIAL_H
#define BPLUS_SERIAL_H

#include <stdbool.h>
#include <stdint.h>
#include "logType.h"

#define ORDER 3   /* reference fan-out; informational only for the HIP engine */

typedef enum { KEY_INT, KEY_UINT64, KEY_BOOL, KEY_STRING } KeyType;

typedef struct {
    KeyType type;
    union {
        uint64_t u64;
        int i32;
        bool b;
        const char *str;
    } v;
} KEY_T;

typedef void *ROW_PTR;

typedef struct node node;
struct node {
    void **pointers;
    KEY_T *keys;
    struct node *parent;
    bool is_leaf;
    int num_keys;
    struct node *next;
};

#endif /* BPLUS_SERIAL_H */
