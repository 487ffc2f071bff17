"""Row-range sharding + all-gatherv merge of matching row IDs over torch.distributed.

One process per GPU; backend "nccl" is RCCL on ROCm (xGMI inside a node).  The
exchange mirrors the only row-range data-parallel path of the reference,
executeQueryDeleteMPI (engine/mpi/executeEngine-mpi.c):

    :703-715  block partition of the rows           -> shard_rows()
    :753      MPI_Allgather of the per-rank sizes   -> all_gather of match counts
    :758-762  displacements = exclusive prefix      -> done on the device
    :765      MPI_Allgatherv of the payload         -> equal-size all_gather of
              ID slots + pqps_merge_segments (device) / torch indexing (CPU)

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


class IdMerger:
    """Buffers + the three-step merge for one (world, slot capacity).

    ids_local / count_local are written by pqps_filter_scan; after merge(),
    `merged[:totals[0]]` on EVERY rank holds the global ascending ID list."""

    def __init__(self, torch, dist, world, rank, slot_capacity, device, ctx=None, pq=None, host_staged=False):
        self.torch, self.dist, self.world, self.rank = torch, dist, world, rank
        # host_staged: collectives run on CPU copies (gloo rehearsal of the GPU control flow on a
        # box where RCCL cannot be used, e.g. several ranks sharing one device); never the fast path
        self.host_staged = host_staged
        self.cap = int(slot_capacity)
        self.device = device
        self.ctx, self.pq = ctx, pq
        t = torch
        self.ids_local = t.zeros(self.cap, dtype=t.int32, device=device)       # u32 payload, int32 container
        self.count_local = t.zeros(1, dtype=t.int64, device=device)
        self.counts = t.zeros(world, dtype=t.int64, device=device)
        self.slots = t.zeros(world * self.cap, dtype=t.int32, device=device)
        self.merged = t.zeros(world * self.cap, dtype=t.int32, device=device)
        self.totals = t.zeros(2, dtype=t.int64, device=device)

    def merge(self, stream_ptr=None):
        """Collectives + compaction, all enqueued on the current torch stream."""
        dist, t = self.dist, self.torch
        if self.world == 1:
            self.counts.copy_(self.count_local)
            self.slots.copy_(self.ids_local)
        elif self.host_staged:
            c_loc, i_loc = self.count_local.cpu(), self.ids_local.cpu()
            c_all = [t.zeros_like(c_loc) for _ in range(self.world)]
            i_all = [t.zeros_like(i_loc) for _ in range(self.world)]
            dist.all_gather(c_all, c_loc)
            dist.all_gather(i_all, i_loc)
            self.counts.copy_(t.cat(c_all))
            self.slots.copy_(t.cat(i_all))
        else:
            dist.all_gather_into_tensor(self.counts, self.count_local)           # mpi:753
            dist.all_gather_into_tensor(self.slots, self.ids_local)              # mpi:765 (equal-size slots)
        if self.device.type == "cuda":
            self.pq.check(self.pq.lib().pqps_merge_segments(
                self.ctx.h, self.slots.data_ptr(), self.counts.data_ptr(), self.world, self.cap,
                self.merged.data_ptr(), self.merged.numel(), self.totals.data_ptr(), stream_ptr), "pqps_merge_segments")
        else:
            # CPU tensors (gloo rehearsal / unit tests): same layout arithmetic in torch
            counts = t.clamp(self.counts, max=self.cap)
            displ = 0
            for r in range(self.world):
                c = int(counts[r])
                self.merged[displ:displ + c] = self.slots[r * self.cap:r * self.cap + c]
                displ += c
            self.totals[0] = displ
            self.totals[1] = int(self.counts.sum())

    def result(self):
        """Host copy of the merged IDs as uint32 numpy (synchronises)."""
        tot = self.totals.cpu()
        if int(tot[1]) > int(tot[0]):
            raise RuntimeError(f"merge slot overflow: {int(tot[1])} IDs reported, capacity {self.cap} per rank")
        return self.merged[:int(tot[0])].cpu().numpy().view("uint32")
