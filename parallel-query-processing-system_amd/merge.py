"""Row-range sharding + all-gatherv merge of matching row IDs over torch.distributed.

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The
exchange mirrors the only row-range data-parallel path of the reference,
executeQueryDeleteMPI (engine/mpi/executeEngine-mpi.c):

    :703-715  block partition of the rows           -> shard_rows()
    :753      MPI_Allgather of the per-rank sizes   -> an 8-byte-per-rank all_gather
    :758-762  displacements = exclusive prefix      -> displacements()
    :765      MPI_Allgatherv of the payload         -> exactly count[r] IDs from every peer, point to
              point, landing at its displacement (RCCL / torch.distributed have no all-gatherv)

Rank-order concatenation of ascending per-shard lists IS the ascending global
list, so scan-mode results stay bit-exact with the single-GPU / QPESeq answer.
Nothing is padded on the wire and no receive buffer can be too small: the sizes
are known before the payload moves (the gathered list is grown to them).

Two implementations of the same step: ShardExchange (the shim calls RCCL itself,
include/pqps_hip.h pqps_exchange_*) and IdMerger (torch.distributed collectives:
the fallback, and the form the CPU tests run under gloo).

torch is plumbing here (device memory, streams, process group).
"""
from __future__ import annotations


def shard_rows(n_rows: int, world: int, rank: int):
    """(start, count) of rank's contiguous row range -- mpi:703-715."""
    base, rem = divmod(n_rows, world)
    if rank < rem:
        return rank * (base + 1), base + 1
    return rem * (base + 1) + (rank - rem) * base, base


def displacements(sizes, capacities=None):
    """mpi:758-762: (what each rank sends, where it lands, total).  A rank whose own buffer was too small
    (size > its capacity) sends what it holds; the caller sees reported > total."""
    held = [min(int(s), int(capacities[r])) if capacities is not None else int(s) for r, s in enumerate(sizes)]
    displ, at = [], 0
    for h in held:
        displ.append(at)
        at += h
    return held, displ, at


SLOT_HEADER_WORDS = 4     # include/pqps_hip.h: [u64 count][u64 reserved] in front of the IDs

# ---- the compact wire form (csrc/pqps_hip.hip: wire_pack_kernel / wire_expand_kernel, restated with numpy) ----------
# A shard's ascending list travels as the low 16 bits of every row number relative to the shard's first row + one u32 per
# 65 536-row group saying where the group's entries begin: [u32 goff[groups + 1], padded to 16 bytes][u16 low[n]].
WIRE_GROUP_ROWS = 1 << 16
WIRE_HEADER_WORDS = 4     # per rank in the sizes all-gather: reported count, rows, first row, form (1 = compact)


def wire_groups(n_rows):
    return (n_rows + WIRE_GROUP_ROWS - 1) // WIRE_GROUP_ROWS


def wire_goff_bytes(n_rows):
    return ((wire_groups(n_rows) + 1) * 4 + 15) & ~15


def wire_bytes(n_rows, n_ids):
    return wire_goff_bytes(n_rows) + ((n_ids * 2 + 3) & ~3)


WIRE_MIN_IDS = 32768      # below this many IDs a list travels as it is: latency, not bytes, is what it costs (PQPS_WIRE_MIN_IDS: tests set 0)


def wire_pays(n_rows, n_ids):
    import os
    floor = int(os.environ.get("PQPS_WIRE_MIN_IDS", WIRE_MIN_IDS))
    return n_ids >= floor and wire_bytes(n_rows, n_ids) < 4 * n_ids


def wire_pack_numpy(ids_u32, n_rows, id_base):
    """uint8 numpy payload of an ascending uint32 list (the CPU twin of wire_pack_kernel)."""
    import numpy as np
    rel = (ids_u32.astype(np.uint64) - np.uint64(id_base)).astype(np.uint64)
    groups = wire_groups(n_rows)
    goff = np.searchsorted(rel, np.arange(groups + 1, dtype=np.uint64) * np.uint64(WIRE_GROUP_ROWS), side="left").astype(np.uint32)
    out = np.zeros(wire_bytes(n_rows, len(ids_u32)), dtype=np.uint8)
    out[:4 * (groups + 1)] = goff.view(np.uint8)
    low = (rel & np.uint64(0xFFFF)).astype(np.uint16)
    g0 = wire_goff_bytes(n_rows)
    out[g0:g0 + 2 * len(low)] = low.view(np.uint8)
    return out


def wire_expand_numpy(wire_u8, n_rows, id_base, n_ids):
    """The uint32 list a payload stands for (the CPU twin of wire_expand_kernel)."""
    import numpy as np
    groups = wire_groups(n_rows)
    goff = wire_u8[:4 * (groups + 1)].view(np.uint32).astype(np.int64)
    g0 = wire_goff_bytes(n_rows)
    low = wire_u8[g0:g0 + 2 * n_ids].view(np.uint16).astype(np.uint64)
    per_group = np.diff(goff)
    assert goff[0] == 0 and goff[-1] == n_ids and (per_group >= 0).all(), "malformed compact payload"
    high = np.repeat(np.arange(groups, dtype=np.uint64) * np.uint64(WIRE_GROUP_ROWS), per_group)
    return (np.uint64(id_base) + high + low).astype(np.uint32)


class IdMerger:
    """Buffers + the exchange for one (world, local capacity), over torch.distributed.

    The filter writes its count to `count_ptr` and its IDs to `ids_ptr`; begin() gathers the counts,
    finish() moves the payload; after that `merged[:totals[0]]` on EVERY rank holds the global ascending
    ID list.  merge() = begin() + finish().  A caller with several queries in flight calls begin(k) and only
    then finish(k-1): the host waits for the sizes of k-1 while the device already has k to run."""

    def __init__(self, torch, dist, world, rank, slot_capacity, device, ctx=None, pq=None, host_staged=False,
                 always_collective=False, shard=None, compact=True):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        # shard = (rows, first row) of this rank's row range: with it, answers that gain from it travel in the compact
        # wire form (2 bytes per match + 4 per 65 536-row group, see wire_pack_numpy); without it, u32 IDs
        self.shard = shard
        self.compact = bool(compact) and shard is not None
        self.wire_bytes_in = 0             # payload bytes received from the peers as they travelled ...
        self.u32_bytes_in = 0              # ... and what the same lists are as u32 IDs
        # host_staged: the collectives run on CPU copies (gloo rehearsal of the GPU control flow on a
        # box where RCCL cannot be used, e.g. several ranks sharing one device); never the fast path
        self.host_staged = host_staged
        self.always_collective = always_collective      # world == 1 still goes through the collectives (rehearsal)
        self.cap = int(slot_capacity)
        self.stride = self.cap + SLOT_HEADER_WORDS            # u32 words per slot (even)
        if self.stride % 2:
            self.cap += 1
            self.stride += 1
        self.device = device
        self.ctx, self.pq = ctx, pq
        t = torch
        self.slot_local = t.zeros(self.stride, dtype=t.int32, device=device)     # u32 payload, int32 container
        self.sizes = t.zeros(world * WIRE_HEADER_WORDS, dtype=t.int64, device=device)       # per rank: count, rows, first row, form
        self.header = t.zeros(WIRE_HEADER_WORDS, dtype=t.int64, device=device)
        self.wire_out = None               # this rank's payload in compact form (uint8), made in begin()
        self.merged = t.zeros(min(self.cap, 1 << 20), dtype=t.int32, device=device)   # grown to what a query needs
        self.totals = [0, 0]                                                      # merged, reported
        self._pending = False
        self._issued = 0                  # serial number of the last begin() (callers finish mergers oldest first)
        self._sizes_pinned = None
        # every rank's capacity, once (shards differ by a row, so may the capacities): constructing a merger is
        # a collective step, like every later call
        self.caps = [self.cap] * world
        if self._collective() and world > 1:
            mine = t.tensor([self.cap], dtype=t.int64)
            parts = [t.zeros_like(mine) for _ in range(world)]
            if host_staged or device.type != "cuda":
                dist.all_gather(parts, mine)
            else:
                mine = mine.to(device)
                parts = [p.to(device) for p in parts]
                dist.all_gather(parts, mine)
            self.caps = [int(p.item()) for p in parts]

    @property
    def count_ptr(self):
        return self.slot_local.data_ptr()

    @property
    def ids_ptr(self):
        return self.slot_local.data_ptr() + 4 * SLOT_HEADER_WORDS

    def set_local(self, ids_u32, count=None):
        """Fills this rank's slot from a numpy uint32 array (CPU tests)."""
        import numpy as np
        k = min(len(ids_u32), self.cap)
        self.slot_local[SLOT_HEADER_WORDS:SLOT_HEADER_WORDS + k] = self.torch.from_numpy(ids_u32[:k].view(np.int32).copy())
        c = len(ids_u32) if count is None else count
        self.slot_local[0:2] = self.torch.from_numpy(np.array([c], dtype=np.uint64).view(np.int32).copy())

    def local_count(self):
        return int(self.slot_local[0:2].cpu().numpy().view("uint64")[0])

    def _collective(self):
        return self.world > 1 or self.always_collective

    def _pack(self, stream_ptr=None):
        """This rank's four header words and, if it pays, its payload in compact form."""
        t = self.torch
        rows, base = self.shard if self.shard is not None else (0, 0)
        if self.device.type == "cuda" and self.compact:
            # the shim's own kernel (the same one pqps_exchange_select runs)
            need = wire_bytes(rows, min(self.cap, rows))
            if self.wire_out is None or self.wire_out.numel() < need:
                self.wire_out = t.zeros(need + 64, dtype=t.uint8, device=self.device)
            self.pq.check(self.pq.lib().pqps_wire_pack(self.ctx.h, self.slot_local.data_ptr(), self.cap, rows, base, 1,
                                                       self.header.data_ptr(), self.wire_out.data_ptr(), stream_ptr), "pqps_wire_pack")
            return
        count = self.local_count()
        held = min(count, self.cap)
        form = 1 if (self.compact and wire_pays(rows, held)) else 0
        self.header.copy_(t.tensor([count, rows, base, form], dtype=t.int64))
        if form:
            import numpy as np
            ids = self.slot_local[SLOT_HEADER_WORDS:SLOT_HEADER_WORDS + held].cpu().numpy().view(np.uint32)
            self.wire_out = t.from_numpy(wire_pack_numpy(ids, rows, base)).to(self.device)

    def begin(self, stream_ptr=None):
        """mpi:753 -- the sizes (with the shard's rows, first row and the form the sender chose), enqueued on the current torch stream."""
        dist, t = self.dist, self.torch
        self._pack(stream_ptr)
        mine = self.header
        if not self._collective():
            self.sizes.copy_(mine)
        elif self.host_staged or self.device.type != "cuda":
            loc = mine.cpu()
            parts = [t.zeros_like(loc) for _ in range(self.world)]
            dist.all_gather(parts, loc)
            self.sizes.copy_(t.cat(parts))
        else:
            dist.all_gather_into_tensor(self.sizes, mine)
        if self.device.type == "cuda":
            if self._sizes_pinned is None:
                self._sizes_pinned = t.empty(self.world * WIRE_HEADER_WORDS, dtype=t.int64).pin_memory()
                self._sizes_ready = t.cuda.Event()
            self._sizes_pinned.copy_(self.sizes, non_blocking=True)
            self._sizes_ready.record()
            self._sizes_host = self._sizes_pinned
        else:
            self._sizes_host = self.sizes
        self._pending = True
        IdMerger._serial = getattr(IdMerger, "_serial", 0) + 1
        self._issued = IdMerger._serial

    def finish(self):
        """mpi:758-765 -- displacements on the host, then every peer's list: exactly count[r] IDs, or its compact form
        (rebuilt to IDs at the displacement)."""
        if not self._pending:
            return
        self._pending = False
        dist, t = self.dist, self.torch
        if self.device.type == "cuda":
            self._sizes_ready.synchronize()
        hdr = [int(v) for v in self._sizes_host.tolist()]
        sizes = hdr[0::WIRE_HEADER_WORDS]
        rows_of, base_of, form_of = hdr[1::WIRE_HEADER_WORDS], hdr[2::WIRE_HEADER_WORDS], hdr[3::WIRE_HEADER_WORDS]
        held, displ, total = displacements(sizes, self.caps)
        if total > self.merged.numel():
            self.merged = t.zeros(total + total // 4 + 4096, dtype=t.int32, device=self.device)
        mine = self.slot_local[SLOT_HEADER_WORDS:SLOT_HEADER_WORDS + held[self.rank]]
        self.merged[displ[self.rank]:displ[self.rank] + held[self.rank]] = mine
        if self._collective() and self.world > 1:
            staged = self.host_staged and self.device.type == "cuda"
            me_compact = bool(form_of[self.rank])
            out = self.wire_out[:wire_bytes(rows_of[self.rank], held[self.rank])] if me_compact else mine
            src = out.cpu() if staged else out
            into = t.zeros(total, dtype=t.int32) if staged else self.merged
            where = t.device("cpu") if staged else self.device
            ops, wires = [], {}
            for r in range(self.world):
                if r == self.rank:
                    continue
                if held[self.rank]:
                    ops.append(dist.P2POp(dist.isend, src, r))
                if held[r]:
                    if form_of[r]:
                        wires[r] = t.zeros(wire_bytes(rows_of[r], held[r]), dtype=t.uint8, device=where)
                        ops.append(dist.P2POp(dist.irecv, wires[r], r))
                        self.wire_bytes_in += wires[r].numel()
                    else:
                        ops.append(dist.P2POp(dist.irecv, into[displ[r]:displ[r] + held[r]], r))
                        self.wire_bytes_in += 4 * held[r]
                    self.u32_bytes_in += 4 * held[r]
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for r, w in wires.items():
                if self.device.type == "cuda" and not staged:
                    self.pq.check(self.pq.lib().pqps_wire_expand(self.ctx.h, w.data_ptr(), rows_of[r], base_of[r],
                                                                 self.merged.data_ptr() + 4 * displ[r], None), "pqps_wire_expand")
                    self.ctx.sync()                              # (w is a temporary)
                else:
                    import numpy as np
                    ids = wire_expand_numpy(w.cpu().numpy(), rows_of[r], base_of[r], held[r])
                    into[displ[r]:displ[r] + held[r]] = t.from_numpy(ids.view(np.int32).copy())
            if staged:
                for r in range(self.world):
                    if r != self.rank and held[r]:
                        self.merged[displ[r]:displ[r] + held[r]] = into[displ[r]:displ[r] + held[r]].to(self.device)
        self.totals = [total, sum(sizes)]

    def merge(self, stream_ptr=None):
        self.begin(stream_ptr)
        self.finish()

    def result(self):
        """Host copy of the merged IDs as uint32 numpy (synchronises)."""
        self.finish()
        if self.totals[1] > self.totals[0]:
            raise RuntimeError(f"local ID buffer overflow: {self.totals[1]} IDs reported, capacity {self.cap} per rank")
        return self.merged[:self.totals[0]].cpu().numpy().view("uint32")


class IndexMerger(IdMerger):
    """Index mode across shards: every rank holds its shard's index-mode result, ordered
    (key asc, row desc) inside the shard; the table-wide leaf order is the sort of the union by
    (key asc, row desc).  Two all-gathers ([count | ids] slots, key slots) + pqps_merge_index_slots
    (device sort) / numpy lexsort (CPU tensors).  One probed condition per merge."""

    def __init__(self, torch, dist, world, rank, slot_capacity, device, ctx=None, pq=None, host_staged=False):
        super().__init__(torch, dist, world, rank, slot_capacity, device, ctx=ctx, pq=pq, host_staged=host_staged)
        t = torch
        self.slots = t.zeros(world * self.stride, dtype=t.int32, device=device)     # equal-size [count | IDs] slots
        self.merged = t.zeros(world * self.cap, dtype=t.int32, device=device)
        self.totals_dev = t.zeros(2, dtype=t.int64, device=device)
        self.keys_local = t.zeros(self.cap, dtype=t.int64, device=device)          # u64 payload
        self.key_slots = t.zeros(world * self.cap, dtype=t.int64, device=device)

    def set_local(self, ids_u32, keys_u64, count=None):
        import numpy as np
        super().set_local(ids_u32, count)
        k = min(len(keys_u64), self.cap)
        self.keys_local[:k] = self.torch.from_numpy(np.asarray(keys_u64[:k], dtype=np.uint64).view(np.int64).copy())

    def gather_keys(self, column_array, key_kind, id_base, stream_ptr=None):
        """Fills keys_local from the shard's key column for the IDs the filter left in the slot."""
        self.pq.check(self.pq.lib().pqps_gather_keys(self.ctx.h, column_array, key_kind, self.ids_ptr, self.count_ptr,
                                                     self.cap, id_base, self.keys_local.data_ptr(), stream_ptr), "pqps_gather_keys")

    def merge(self, stream_ptr=None):
        dist, t = self.dist, self.torch
        if self.world == 1:
            self.slots.copy_(self.slot_local)
            self.key_slots.copy_(self.keys_local)
        elif self.host_staged or self.device.type != "cuda":
            loc, kloc = self.slot_local.cpu(), self.keys_local.cpu()
            parts, kparts = [t.zeros_like(loc) for _ in range(self.world)], [t.zeros_like(kloc) for _ in range(self.world)]
            dist.all_gather(parts, loc)
            dist.all_gather(kparts, kloc)
            self.slots.copy_(t.cat(parts))
            self.key_slots.copy_(t.cat(kparts))
        else:
            dist.all_gather_into_tensor(self.slots, self.slot_local)
            dist.all_gather_into_tensor(self.key_slots, self.keys_local)
        if self.device.type == "cuda":
            self.pq.check(self.pq.lib().pqps_merge_index_slots(
                self.ctx.h, self.slots.data_ptr(), self.key_slots.data_ptr(), self.world, self.stride,
                self.merged.data_ptr(), self.merged.numel(), self.totals_dev.data_ptr(), stream_ptr), "pqps_merge_index_slots")
        else:
            import numpy as np
            ids, keys, raw = [], [], 0
            for r in range(self.world):
                slot = self.slots[r * self.stride:(r + 1) * self.stride]
                reported = int(slot[0:2].numpy().view("uint64")[0])
                c = min(reported, self.cap)
                ids.append(slot[SLOT_HEADER_WORDS:SLOT_HEADER_WORDS + c].numpy().view("uint32"))
                keys.append(self.key_slots[r * self.cap:r * self.cap + c].numpy().view("uint64"))
                raw += reported
            ids, keys = np.concatenate(ids), np.concatenate(keys)
            order = np.lexsort((-(ids.astype(np.int64)), keys))                 # key asc, then row desc
            out = ids[order]
            self.merged[:len(out)] = t.from_numpy(out.view(np.int32).copy())
            self.totals = [len(out), raw]

    def result(self):
        if self.device.type == "cuda":
            self.totals = [int(v) for v in self.totals_dev.cpu().tolist()]
        if self.totals[1] > self.totals[0]:
            raise RuntimeError(f"merge slot overflow: {self.totals[1]} IDs reported, capacity {self.cap} per rank")
        return self.merged[:self.totals[0]].cpu().numpy().view("uint32")


def default_rccl_library(torch=None):
    """The RCCL that goes with the HIP runtime this process runs on: under torch (which loads its
    own bundled libamdhip64 + librccl) the bundled one, otherwise the system one."""
    import os
    if os.environ.get("PQPS_RCCL_LIBRARY"):
        return os.environ["PQPS_RCCL_LIBRARY"]
    if torch is not None:
        p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
        if os.path.exists(p):
            return p
    return "/opt/rocm/lib/librccl.so"


def open_exchange(dist, world, rank, make_id, prepare, connect, close, control_device="cpu", torch=None):
    """Brings a shim-driven exchange up on every rank or on none -- never on some.

    make_id()      rank 0 only: the RCCL id (bytes), raises on failure
    prepare()      everything local to a rank (library, streams, buffers), returns a handle, raises on failure
    connect(h, id) the collective part (ncclCommInitRank blocks until every rank of the world calls it)
    close(h)       undoes prepare / connect

    A rank that fails locally must not leave the others blocked inside connect(): the ranks agree
    (all_reduce MIN over torch.distributed) after prepare() and again after connect().  Returns the handle,
    or None on EVERY rank when any of them failed (the caller then takes the torch.distributed path)."""
    def agreed(ok):
        if world == 1:
            return bool(ok)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int64, device=control_device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag.item()) == 1

    def note(stage, e):
        import sys
        print(f"[exchange] rank {rank}: {stage} failed ({e!r}); all ranks fall back to torch.distributed", file=sys.stderr, flush=True)

    box = [None, None]
    if rank == 0:
        try:
            box[0] = make_id()
        except Exception as e:                                   # noqa: BLE001
            box[1] = repr(e)
    if world > 1:
        dist.broadcast_object_list(box, src=0)                   # rank 0's failure travels with the id
    if box[1] is not None:
        if rank == 0:
            note("RCCL id", box[1])
        return None
    handle = None
    try:
        handle = prepare()
    except Exception as e:                                       # noqa: BLE001
        note("local preparation", e)
    if not agreed(handle is not None):
        if handle is not None:
            close(handle)
        return None
    ok = True
    try:
        connect(handle, box[0])
    except Exception as e:                                       # noqa: BLE001
        note("communicator", e)
        ok = False
    if not agreed(ok):
        close(handle)
        return None
    return handle


class ShardExchange:
    """Native exchange (include/pqps_hip.h: pqps_exchange_*): ONE host call per query enqueues the
    shard scan and the all-gatherv of its matching IDs (sizes, then exactly-sized point-to-point payload:
    mpi:717-768) on the shim's own stream.

    torch.distributed is only the bootstrap: rank 0's RCCL id reaches the other ranks through
    broadcast_object_list, and the ranks agree through it that all of them came up (open())."""

    def __init__(self, pq, ctx, handle, world, rank, ring):
        self.pq, self.ctx, self.h, self.world, self.rank, self.ring = pq, ctx, handle, world, rank, ring

    @classmethod
    def open(cls, pq, ctx, torch, dist, world, rank, slot_capacity, ring=4, rccl_library=None, control_device="cpu"):
        """The exchange, or None on every rank if any rank could not set it up."""
        import ctypes as C
        L = pq.lib()
        path = (rccl_library or default_rccl_library(torch)).encode()

        def make_id():
            ident = C.create_string_buffer(128)
            pq.check(L.pqps_exchange_unique_id(path, ident), "pqps_exchange_unique_id")
            return ident.raw

        def prepare():
            h = C.c_void_p()
            pq.check(L.pqps_exchange_prepare(ctx.h, path, world, rank, int(slot_capacity), ring, C.byref(h)), "pqps_exchange_prepare")
            return h

        def connect(h, ident):
            pq.check(L.pqps_exchange_connect(h, C.create_string_buffer(ident, 128)), "pqps_exchange_connect")

        def close(h):
            L.pqps_exchange_destroy(h)

        h = open_exchange(dist, world, rank, make_id, prepare, connect, close, control_device=control_device, torch=torch)
        return cls(pq, ctx, h, world, rank, ring) if h is not None else None

    def select(self, cols, n_cols, n_rows, id_base, pred_ref, slot, stream_ptr):
        self.pq.check(self.pq.lib().pqps_exchange_select(self.h, cols, n_cols, n_rows, id_base, pred_ref, slot, stream_ptr),
                      "pqps_exchange_select")

    def count(self, cols, n_cols, n_rows, pred_ref, slot, stream_ptr):
        """COUNT(*): local count kernel + all-reduce (mpi:745); read it back with count_result()."""
        self.pq.check(self.pq.lib().pqps_exchange_count(self.h, cols, n_cols, n_rows, pred_ref, slot, stream_ptr),
                      "pqps_exchange_count")

    def sync(self):
        """Every query handed in so far has been exchanged (enqueues what was still held back, then waits)."""
        self.pq.check(self.pq.lib().pqps_exchange_sync(self.h), "pqps_exchange_sync")

    def count_result(self, slot):
        """(global count, this rank's count) of a count() slot."""
        import ctypes as C
        local, totals = C.c_uint64(), (C.c_uint64 * 2)()
        self.pq.check(self.pq.lib().pqps_exchange_result(self.h, slot, None, C.byref(local), totals), "pqps_exchange_result")
        return int(totals[0]), int(local.value)

    def result(self, slot):
        """(merged uint32 numpy array, this rank's own match count); waits for the slot's exchange."""
        import ctypes as C
        import numpy as np
        ptr, local, totals = C.c_void_p(), C.c_uint64(), (C.c_uint64 * 2)()
        self.pq.check(self.pq.lib().pqps_exchange_result(self.h, slot, C.byref(ptr), C.byref(local), totals),
                      "pqps_exchange_result")
        out = np.empty(int(totals[0]), dtype=np.uint32)
        if len(out):
            self.ctx.download(out.ctypes.data, ptr.value, out.nbytes)
        return out, int(local.value)

    def close(self):
        if self.h:
            self.pq.check(self.pq.lib().pqps_exchange_destroy(self.h), "pqps_exchange_destroy")
            self.h = None
