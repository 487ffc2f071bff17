"""The compact wire form by itself (include/pqps_hip.h: pqps_wire_pack / pqps_wire_expand / pqps_wire_bytes / pqps_wire_pays):
the two kernels against their numpy twins (merge.wire_pack_numpy / wire_expand_numpy), byte for byte, over the shapes an
exchange meets -- an empty list, one entry, every row of the shard, a shard whose rows are no multiple of 65 536, whole
empty groups in the middle, a first row that is not 0, a slot that overflowed (holds `capacity` IDs, reports more), and
the floor below which a list travels as it is.  The CPU half (round trip of the twins, sizes) runs without a GPU."""
import ctypes as C

import numpy as np
import pytest

import qpelib as q

pq = q.pq
HDR = pq.SLOT_HEADER_WORDS          # u32 words in front of a slot's IDs: [u64 count][u64]


def shapes():
    rng = np.random.default_rng(77)
    out = []
    for name, rows, base, pick in (
        ("empty", 200_000, 0, lambda r: np.zeros(0, np.uint32)),
        ("one", 200_000, 5, lambda r: np.array([r - 1], np.uint32)),
        ("every_row", 70_001, 1_000_000, lambda r: np.arange(r, dtype=np.uint32)),
        ("ragged_groups", 3 * 65536 + 17, 123, lambda r: np.sort(rng.choice(r, size=r // 23, replace=False)).astype(np.uint32)),
        ("hole_in_the_middle", 5 * 65536, 0, lambda r: np.concatenate([np.arange(0, 40_000, 3), np.arange(4 * 65536 + 5, 5 * 65536, 7)]).astype(np.uint32)),
        ("group_edges", 4 * 65536, 4_000_000_000 - 4 * 65536, lambda r: np.array([0, 65535, 65536, 131071, 131072, 4 * 65536 - 1], np.uint32)),
        ("dense_tail", 65536 + 1, 0, lambda r: np.arange(60_000, r, dtype=np.uint32)),
    ):
        out.append((name, rows, base, pick(rows)))
    return out


@pytest.mark.parametrize("name,rows,base,rel", shapes(), ids=[s[0] for s in shapes()])
def test_numpy_twins_round_trip(name, rows, base, rel):
    mg = q.pq_merge()
    ids = (rel.astype(np.uint64) + base).astype(np.uint32)
    wire = mg.wire_pack_numpy(ids, rows, base)
    assert len(wire) == mg.wire_bytes(rows, len(ids)) == mg.wire_goff_bytes(rows) + ((2 * len(ids) + 3) & ~3)
    assert mg.wire_goff_bytes(rows) % 16 == 0 and mg.wire_groups(rows) == -(-rows // 65536)
    back = mg.wire_expand_numpy(wire, rows, base, len(ids))
    assert np.array_equal(back, ids)


def test_the_floor_and_the_break_even(monkeypatch):
    mg = q.pq_merge()
    monkeypatch.setenv("PQPS_WIRE_MIN_IDS", "0")
    rows = 10 * 65536
    # bytes: 4 per ID against 2 per ID + 4 per group (+ padding): pays from a little over 2 IDs per group
    assert not mg.wire_pays(rows, 20) and mg.wire_pays(rows, 40)
    monkeypatch.delenv("PQPS_WIRE_MIN_IDS")
    assert not mg.wire_pays(rows, 32767) and mg.wire_pays(rows, 32768)


@pytest.mark.gpu
@pytest.mark.parametrize("floor", ["0", None])
def test_kernels_equal_their_twins(monkeypatch, floor):
    if floor is None:
        monkeypatch.delenv("PQPS_WIRE_MIN_IDS", raising=False)
    else:
        monkeypatch.setenv("PQPS_WIRE_MIN_IDS", floor)
    mg = q.pq_merge()
    L = pq.lib()
    ctx = pq.Context(0)
    for name, rows, base, rel in shapes() + [("big", 40 * 65536, 7, np.arange(0, 40 * 65536, 13, dtype=np.uint32))]:
        ids = (rel.astype(np.uint64) + base).astype(np.uint32)
        for cap, reported in ((max(len(ids), 2) + 6, len(ids)),) + (((len(ids) // 2) & ~1, len(ids)),) * (len(ids) > 8):
            held = min(cap, reported)
            slot = np.zeros(HDR + cap, dtype=np.uint32)
            slot[:2] = np.array([reported], dtype=np.uint64).view(np.uint32)
            slot[HDR:HDR + held] = ids[:held]
            slot_dev = ctx.malloc(slot.nbytes)
            ctx.upload(slot_dev, slot.ctypes.data, slot.nbytes)
            hdr_dev = ctx.malloc(64)
            room = int(L.pqps_wire_bytes(rows, min(cap, rows)))
            assert room == mg.wire_bytes(rows, min(cap, rows))
            wire_dev = ctx.malloc(room + 64)
            ctx.memset(wire_dev, 0xEE, room + 64)
            for enabled in (1, 0):
                pq.check(L.pqps_wire_pack(ctx.h, slot_dev, cap, rows, base, enabled, hdr_dev, wire_dev, None), "pqps_wire_pack")
                ctx.sync()
                hdr = np.zeros(4, dtype=np.uint64)
                ctx.download(hdr.ctypes.data, hdr_dev, 32)
                pays = bool(L.pqps_wire_pays(rows, held))
                assert pays == bool(mg.wire_pays(rows, held)), (name, held)
                assert hdr.tolist() == [reported, rows, base, 1 if (enabled and pays) else 0], (name, cap, enabled, hdr)
                if not (enabled and pays):
                    continue
                want = mg.wire_pack_numpy(ids[:held], rows, base)
                got = np.zeros(len(want) + 16, dtype=np.uint8)
                ctx.download(got.ctypes.data, wire_dev, len(got))
                g0 = mg.wire_goff_bytes(rows)
                ngo = 4 * (mg.wire_groups(rows) + 1)
                assert np.array_equal(got[:ngo], want[:ngo]), (name, "group offsets")
                assert np.array_equal(got[g0:g0 + 2 * held], want[g0:g0 + 2 * held]), (name, "low halves")
                assert (got[len(want):] == 0xEE).all(), (name, "wrote past the payload")
                out_dev = ctx.malloc(4 * held + 64)
                ctx.memset(out_dev, 0xAB, 4 * held + 64)
                pq.check(L.pqps_wire_expand(ctx.h, wire_dev, rows, base, out_dev, None), "pqps_wire_expand")
                ctx.sync()
                back = np.zeros(held + 16, dtype=np.uint32)
                ctx.download(back.ctypes.data, out_dev, back.nbytes)
                assert np.array_equal(back[:held], ids[:held]), name
                assert (back[held:] == 0xABABABAB).all(), (name, "wrote past the list")
                ctx.free(out_dev)
            for p in (slot_dev, hdr_dev, wire_dev):
                ctx.free(p)
    ctx.close()
