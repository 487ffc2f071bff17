"""numpy model of what filter_kernel computes from a pqps_predicate (window
leaves + truth table / jump table).  Test-only: lets the CPU suite check the
host-side predicate compiler against the oracle without a GPU."""
import numpy as np

import qpelib as q

pq = q.pq


def evaluate(pred, col_arrays):
    """col_arrays[slot] = numpy array (uint8/16/32/64 or int32) of one column. -> bool mask."""
    n = len(col_arrays[0]) if col_arrays else None
    L = pred.n_leaves
    if L == 0:
        return bool(pred.truth & 1)
    idx = np.zeros(n, dtype=np.uint64)
    for k in range(L):
        leaf = pred.leaf[k]
        v = col_arrays[leaf.column]
        if v.dtype.itemsize == 8:
            x = v.astype(np.uint64)
            hit = (x - np.uint64(leaf.lo)) <= np.uint64(leaf.span)
        else:
            x = v.astype(np.int64).astype(np.uint32) if v.dtype == np.int32 else v.astype(np.uint32)
            hit = (x - np.uint32(leaf.lo & 0xFFFFFFFF)) <= np.uint32(leaf.span & 0xFFFFFFFF)
        if leaf.negate:
            hit = ~hit
        idx |= hit.astype(np.uint64) << np.uint64(k)
    if L <= pq.TT_LEAVES:
        return ((np.uint64(pred.truth) >> idx) & np.uint64(1)).astype(bool)
    state = np.zeros(n, dtype=np.int64)
    for s in range(L):
        r = ((idx >> np.uint64(pred.order[s])) & np.uint64(1)).astype(bool)
        nxt = np.where(r, pred.on_true[s], pred.on_false[s])
        state = np.where(state == s, nxt, state)
    return state == pq.ACCEPT


def columns_from_records(rows, n):
    """Oracle-loaded `record` array -> (SchemaSpec, {column name: numpy array}) with
    order-preserving dictionaries built exactly like buildEngine-hip.c does."""
    import ctypes as C
    spec = pq.SchemaSpec()
    arrays = {}
    raw = np.frombuffer((C.c_char * (1040 * max(n, 1))).from_address(C.addressof(rows.contents)), dtype=np.uint8)
    raw = raw[:1040 * n].reshape(n, 1040) if n else raw[:0].reshape(0, 1040)
    offs = dict(command_id=0, raw_command=8, base_command=520, shell_type=620, exit_code=640, timestamp=644,
                sudo_used=674, working_directory=675, user_id=876, user_name=880, host_name=930, risk_level=1032)
    arrays["command_id"] = raw[:, 0:8].copy().view(np.uint64).reshape(n)
    for name in ("exit_code", "user_id", "risk_level"):
        arrays[name] = raw[:, offs[name]:offs[name] + 4].copy().view(np.int32).reshape(n)
    arrays["sudo_used"] = (raw[:, 674] != 0).astype(np.uint8)
    for name, w in (("command_id", 8), ("exit_code", 4), ("user_id", 4), ("risk_level", 4), ("sudo_used", 1)):
        spec.set_numeric(name, w)
    flat = raw.tobytes() + b"\0"
    for name in pq.COLUMNS:
        if pq.COLUMN_KIND[pq.COL[name]] != pq.KIND_DICT:
            continue
        vals = []
        for i in range(n):
            start = i * 1040 + offs[name]
            end = flat.index(b"\0", start)          # C string from the field start (may run on)
            vals.append(flat[start:end])
        dic = sorted(set(vals))
        width = 1 if len(dic) <= 256 else 2 if len(dic) <= 65536 else 4
        lut = {v: i for i, v in enumerate(dic)}
        arrays[name] = np.array([lut[v] for v in vals], dtype={1: np.uint8, 2: np.uint16, 4: np.uint32}[width])
        spec.set_dict(name, width, dic)
    return spec, arrays
