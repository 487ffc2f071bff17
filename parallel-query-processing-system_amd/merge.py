"""Row-range sharding + all-gatherv merge of matching row IDs over torch.distributed.

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The
exchange mirrors the only row-range data-parallel path of the reference,
executeQueryDeleteMPI (engine/mpi/executeEngine-mpi.c):

    :703-715  block partition of the rows           -> shard_rows()
    :753      MPI_Allgather of the per-rank sizes   -> the count rides in the slot header
    :758-762  displacements = exclusive prefix      -> done on the device
    :765      MPI_Allgatherv of the payload         -> ONE equal-size all_gather of
              [count | IDs] slots + pqps_merge_slots (device) / torch indexing (CPU)

Rank-order concatenation of ascending per-shard lists IS the ascending global
list, so scan-mode results stay bit-exact with the single-GPU / QPESeq answer.
RCCL has no all-gatherv; an equal-size all-gather of slots padded to a common
capacity keeps the whole step free of host round trips (counts never leave the
device).  A slot overflow is reported in totals[1] > totals[0], never silent.

torch is plumbing here (device memory, streams, process group).
"""
from __future__ import annotations


def shard_rows(n_rows: int, world: int, rank: int):
    """(start, count) of rank's contiguous row range -- mpi:703-715."""
    base, rem = divmod(n_rows, world)
    if rank < rem:
        return rank * (base + 1), base + 1
    return rem * (base + 1) + (rank - rem) * base, base


SLOT_HEADER_WORDS = 4     # include/pqps_hip.h: [u64 count][u64 reserved] in front of the IDs


class IdMerger:
    """Buffers + the merge for one (world, slot capacity).

    The filter writes its count to `count_ptr` and its IDs to `ids_ptr` -- both inside this
    rank's slot -- so ONE equal-size all-gather moves count and payload together; after
    merge(), `merged[:totals[0]]` on EVERY rank holds the global ascending ID list."""

    def __init__(self, torch, dist, world, rank, slot_capacity, device, ctx=None, pq=None, host_staged=False,
                 always_collective=False):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        # host_staged: the collective runs on CPU copies (gloo rehearsal of the GPU control flow on a
        # box where RCCL cannot be used, e.g. several ranks sharing one device); never the fast path
        self.host_staged = host_staged
        self.always_collective = always_collective      # world == 1 still goes through the collective (rehearsal)
        self.cap = int(slot_capacity)
        self.stride = self.cap + SLOT_HEADER_WORDS            # u32 words per slot (even)
        if self.stride % 2:
            self.cap += 1
            self.stride += 1
        self.device = device
        self.ctx, self.pq = ctx, pq
        t = torch
        self.slot_local = t.zeros(self.stride, dtype=t.int32, device=device)     # u32 payload, int32 container
        self.slots = t.zeros(world * self.stride, dtype=t.int32, device=device)
        self.merged = t.zeros(world * self.cap, dtype=t.int32, device=device)
        self.totals = t.zeros(2, dtype=t.int64, device=device)

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

    def merge(self, stream_ptr=None):
        """One collective + compaction, enqueued on the current torch stream."""
        dist, t = self.dist, self.torch
        if self.world == 1 and not self.always_collective:
            self.slots.copy_(self.slot_local)
        elif self.host_staged:
            loc = self.slot_local.cpu()
            parts = [t.zeros_like(loc) for _ in range(self.world)]
            dist.all_gather(parts, loc)
            self.slots.copy_(t.cat(parts))
        else:
            dist.all_gather_into_tensor(self.slots, self.slot_local)       # mpi:753 + mpi:765 in one collective
        if self.device.type == "cuda":
            self.pq.check(self.pq.lib().pqps_merge_slots(
                self.ctx.h, self.slots.data_ptr(), self.world, self.stride,
                self.merged.data_ptr(), self.merged.numel(), self.totals.data_ptr(), stream_ptr), "pqps_merge_slots")
        else:
            # CPU tensors (gloo unit tests): same layout arithmetic in torch
            displ = raw = 0
            for r in range(self.world):
                slot = self.slots[r * self.stride:(r + 1) * self.stride]
                reported = int(slot[0:2].numpy().view("uint64")[0])
                c = min(reported, self.cap)
                self.merged[displ:displ + c] = slot[SLOT_HEADER_WORDS:SLOT_HEADER_WORDS + c]
                displ += c
                raw += reported
            self.totals[0] = displ
            self.totals[1] = raw

    def result(self):
        """Host copy of the merged IDs as uint32 numpy (synchronises)."""
        tot = self.totals.cpu()
        if int(tot[1]) > int(tot[0]):
            raise RuntimeError(f"merge slot overflow: {int(tot[1])} IDs reported, capacity {self.cap} per rank")
        return self.merged[:int(tot[0])].cpu().numpy().view("uint32")


class IndexMerger(IdMerger):
    """Index mode across shards: every rank holds its shard's index-mode result, ordered
    (key asc, row desc) inside the shard; the table-wide leaf order is the sort of the union by
    (key asc, row desc).  Two all-gathers ([count | ids] slots, key slots) + pqps_merge_index_slots
    (device sort) / numpy lexsort (CPU tensors).  One probed condition per merge."""

    def __init__(self, torch, dist, world, rank, slot_capacity, device, ctx=None, pq=None, host_staged=False):
        super().__init__(torch, dist, world, rank, slot_capacity, device, ctx=ctx, pq=pq, host_staged=host_staged)
        t = torch
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
                self.merged.data_ptr(), self.merged.numel(), self.totals.data_ptr(), stream_ptr), "pqps_merge_index_slots")
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
            self.totals[0] = len(out)
            self.totals[1] = raw


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


class ShardExchange:
    """Native exchange (include/pqps_hip.h: pqps_exchange_*): ONE host call per query enqueues the
    shard scan, the RCCL all-gather of the [count | IDs] slot and the device merge (mpi:717-768).

    torch.distributed is only the bootstrap: rank 0's RCCL id reaches the other ranks through
    broadcast_object_list; the data path is ncclAllGather called from the shim on its own stream."""

    def __init__(self, pq, ctx, torch, dist, world, rank, slot_capacity, ring=4, rccl_library=None):
        import ctypes as C
        self.pq, self.ctx, self.world, self.rank, self.ring = pq, ctx, world, rank, ring
        L = pq.lib()
        path = (rccl_library or default_rccl_library(torch)).encode()
        ident = C.create_string_buffer(128)
        err = None
        if rank == 0 and L.pqps_exchange_unique_id(path, ident) != 0:
            err = "pqps_exchange_unique_id failed: " + L.pqps_last_error().decode()
        box = [ident.raw, err]                       # rank 0's failure travels with the broadcast: every rank raises
        if world > 1:
            dist.broadcast_object_list(box, src=0)
        if box[1] is not None:
            raise pq.PqpsError(box[1])
        ident = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        pq.check(L.pqps_exchange_create(ctx.h, path, ident, world, rank, int(slot_capacity), ring, C.byref(h)),
                 "pqps_exchange_create")
        self.h = h

    def select(self, cols, n_cols, n_rows, id_base, pred_ref, slot, stream_ptr):
        self.pq.check(self.pq.lib().pqps_exchange_select(self.h, cols, n_cols, n_rows, id_base, pred_ref, slot, stream_ptr),
                      "pqps_exchange_select")

    def count(self, cols, n_cols, n_rows, pred_ref, slot, stream_ptr):
        """COUNT(*): local count kernel + all-reduce (mpi:745); read it back with count_result()."""
        self.pq.check(self.pq.lib().pqps_exchange_count(self.h, cols, n_cols, n_rows, pred_ref, slot, stream_ptr),
                      "pqps_exchange_count")

    def count_result(self, slot):
        """(global count, this rank's count) of a count() slot."""
        import ctypes as C
        local, totals = C.c_uint64(), (C.c_uint64 * 2)()
        self.pq.check(self.pq.lib().pqps_exchange_result(self.h, slot, None, C.byref(local), totals), "pqps_exchange_result")
        return int(totals[0]), int(local.value)

    def result(self, slot):
        """(merged uint32 numpy array, this rank's own match count); waits for the slot's merge."""
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
