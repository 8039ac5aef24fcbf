"""Row sharding of one corpus over the GPUs of a node.

Rows are independent for Flat and PQ-Flat, so shard s holds the contiguous block
[s*ceil(N/S), min(N, (s+1)*ceil(N/S))) and reports GLOBAL ids (vdb_index_set_id_offset).
Every rank computes its local top-k for all queries; ONE all-gather of a single byte buffer per rank
([nq, k] i64 ids + [nq, k] f32 distances + [nq] i64 counts = 128 KB at nq = 1000, k = 10) over RCCL/xGMI follows, then each rank merges the S sorted lists per query by the
CandidatePair order (distance, index) -- top-k under a total order is decomposable, so the result equals
the unsharded one exactly.  One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm,
"gloo" is used for the CPU rehearsal in tests/.
"""
from __future__ import annotations

import numpy as np


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    per = -(-n // world) if world > 0 else n
    return min(n, rank * per), min(n, (rank + 1) * per)


def allgather_merge(local_idx, local_dist, local_cnt, k: int, group=None, gpu_index=None):
    """local_*: torch tensors [nq,k] int64 / [nq,k] float32 / [nq] int64 with GLOBAL ids.

    Returns merged (idx [nq,k] int64, dist [nq,k] float32, cnt [nq] int64) on every rank.
    On CUDA tensors the merge runs on the GPU (vdb_merge_topk_device, needs `gpu_index`);
    on CPU tensors (gloo rehearsal) it uses the host merge utility of the same library.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_idx, local_dist, local_cnt
    nq = local_idx.shape[0]
    # ONE collective: ids (i64), distances (f32) and counts (i64) travel in a single byte buffer per rank
    parts = (local_idx.contiguous().view(torch.uint8).reshape(-1), local_dist.contiguous().view(torch.uint8).reshape(-1),
             local_cnt.contiguous().view(torch.uint8).reshape(-1))
    sizes = [p.numel() for p in parts]
    mine = torch.cat(parts)
    gathered = torch.empty((world, mine.numel()), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered.view(-1), mine, group=group)
    o0, o1 = sizes[0], sizes[0] + sizes[1]
    g_idx = gathered[:, :o0].contiguous().view(local_idx.dtype).view(world, nq, k)
    g_dist = gathered[:, o0:o1].contiguous().view(local_dist.dtype).view(world, nq, k)
    g_cnt = gathered[:, o1:].contiguous().view(local_cnt.dtype).view(world, nq)
    if local_idx.is_cuda:
        if gpu_index is None:
            raise ValueError("allgather_merge on CUDA tensors needs the rank's GpuIndex")
        o_idx = torch.empty((nq, k), dtype=torch.int64, device=local_idx.device)
        o_dist = torch.empty((nq, k), dtype=torch.float32, device=local_idx.device)
        o_cnt = torch.empty((nq,), dtype=torch.int64, device=local_idx.device)
        torch.cuda.current_stream().synchronize()
        gpu_index.merge_topk_device(g_dist.data_ptr(), g_idx.data_ptr(), g_cnt.data_ptr(), world, nq, k,
                                    o_idx.data_ptr(), o_dist.data_ptr(), o_cnt.data_ptr())
        return o_idx, o_dist, o_cnt
    from .index import merge_topk

    oi, od, oc = merge_topk(g_dist.numpy(), g_idx.numpy().astype(np.uint64), g_cnt.numpy().astype(np.uint64), k)
    return (torch.from_numpy(oi.astype(np.int64)), torch.from_numpy(od), torch.from_numpy(oc.astype(np.int64)))


def allgather_merge_pq(adc_keys, exact_keys, k: int, group=None, gpu_index=None):
    """Row-sharded FlatIndex::knn_pq (flat_index.rs:84-104): adc_keys / exact_keys are this rank's
    [nq, max(ef,k)] pair-key rows (GpuIndex.knn_pq_shard[_device]; torch int64 views of the u64 keys).

    pq_resort replays ResultSet::add in the GLOBAL (ADC, id) order (candidate_pair.rs:102-108), so the exact
    re-sort cannot finish per shard: ONE all-gather moves both key rows ([nq, efk] x 2 x 8 B per rank), then every
    rank merges to the global ADC top-efk and replays the re-sort.  Returns (idx, dist, cnt) like allgather_merge.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    nq, efk = adc_keys.shape
    mine = torch.stack((adc_keys.contiguous(), exact_keys.contiguous()))  # [2, nq, efk] int64
    if world > 1:
        gathered = torch.empty((world,) + tuple(mine.shape), dtype=mine.dtype, device=mine.device)
        dist.all_gather_into_tensor(gathered.view(-1), mine.view(-1), group=group)
    else:
        gathered = mine.unsqueeze(0)
    g_adc = gathered[:, 0].contiguous()
    g_ex = gathered[:, 1].contiguous()
    if adc_keys.is_cuda:
        if gpu_index is None:
            raise ValueError("allgather_merge_pq on CUDA tensors needs the rank's GpuIndex")
        o_idx = torch.empty((nq, k), dtype=torch.int64, device=adc_keys.device)
        o_dist = torch.empty((nq, k), dtype=torch.float32, device=adc_keys.device)
        o_cnt = torch.empty((nq,), dtype=torch.int64, device=adc_keys.device)
        torch.cuda.current_stream().synchronize()
        gpu_index.pq_merge_resort_device(g_adc.data_ptr(), g_ex.data_ptr(), world, nq, efk, k, o_idx.data_ptr(),
                                         o_dist.data_ptr(), o_cnt.data_ptr())
        return o_idx, o_dist, o_cnt
    from .index import pq_merge_resort

    oi, od, oc = pq_merge_resort(g_adc.numpy().view(np.uint64), g_ex.numpy().view(np.uint64), k)
    return (torch.from_numpy(oi.astype(np.int64)), torch.from_numpy(od), torch.from_numpy(oc.astype(np.int64)))


def replica_query_slice(nq: int, world: int, rank: int) -> tuple[int, int]:
    """HNSW does not shard (edges cross any row partition): every GPU holds a full replica and serves the
    contiguous query block [rank*ceil(nq/world), ...) -- no data-path collective."""
    return shard_bounds(nq, world, rank)


def allgather_concat(local_idx, local_dist, local_cnt, nq: int, group=None):
    """Replica mode: rank r computed rows replica_query_slice(nq, world, r) of the answer; every rank receives all nq
    rows.  The blocks are padded to ceil(nq/world) rows so one fixed-size all-gather suffices."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return local_idx, local_dist, local_cnt
    per = -(-nq // world)
    k = local_idx.shape[1]

    def pad(t, shape):
        out = torch.zeros(shape, dtype=t.dtype, device=t.device)
        out[: t.shape[0]] = t
        return out

    parts = (pad(local_idx, (per, k)).view(torch.uint8).reshape(-1), pad(local_dist, (per, k)).view(torch.uint8).reshape(-1),
             pad(local_cnt, (per,)).view(torch.uint8).reshape(-1))
    sizes = [p.numel() for p in parts]
    mine = torch.cat(parts)
    gathered = torch.empty((world, mine.numel()), dtype=torch.uint8, device=mine.device)
    dist.all_gather_into_tensor(gathered.view(-1), mine, group=group)
    o0, o1 = sizes[0], sizes[0] + sizes[1]
    g_idx = gathered[:, :o0].contiguous().view(local_idx.dtype).view(world * per, k)[:nq]
    g_dist = gathered[:, o0:o1].contiguous().view(local_dist.dtype).view(world * per, k)[:nq]
    g_cnt = gathered[:, o1:].contiguous().view(local_cnt.dtype).view(world * per)[:nq]
    return g_idx, g_dist, g_cnt


class ShardExchange:
    """Persistent buffers for the per-step exchange of row-sharded Flat results on the GPU.

    One rank's block = [nq*k ids i64 | nq*k distances f32 | pad | nq counts i64] in ONE uint8 tensor: the search
    writes straight into typed views of it (no packing kernels), `all_gather_into_tensor` moves the block (one RCCL
    collective per step), and the merge reads the S blocks of the receive buffer in place.

    The exchange is pipelined: there are two sets of buffers, and for k <= 64 the collective and the merge of a step are only
    ENQUEUED on torch's current stream (`vdb_merge_topk_gathered_async`, no host synchronisation), so they run under the next
    step's search, which the library issues on its own stream.  `begin_step()` hands out the buffers of the step and first
    waits until the exchange that used them two steps ago is done; the merged results of a step are valid once torch's stream
    has passed its exchange (the caller's fence, or `wait()`).
    """

    def __init__(self, nq: int, k: int, device, world: int, force: bool = False, min_depth: int = 1):
        """force: run the collective and the gathered merge even for one rank (a 1-rank RCCL all-gather: the only way to
        execute this path's nccl calls on a one-GPU box); min_depth: buffer sets to rotate at least (a host that keeps two
        searches in flight -- vdb_flat_knn_device_begin / _end -- needs three: one being exchanged, two being written)"""
        import torch

        self.nq, self.k, self.world = nq, k, world
        self.force = force
        self.off_ids = 0
        self.off_dists = nq * k * 8
        self.off_counts = (nq * k * 12 + 7) // 8 * 8
        self.block = self.off_counts + nq * 8
        self.active = world > 1 or force
        self.depth = max(2 if self.active else 1, int(min_depth))
        self._bufs = []
        for _ in range(self.depth):
            send = torch.zeros(self.block, dtype=torch.uint8, device=device)
            b = {"send": send,
                 "idx": send[self.off_ids:self.off_dists].view(torch.int64).view(nq, k),
                 "dist": send[self.off_dists:self.off_dists + nq * k * 4].view(torch.float32).view(nq, k),
                 "cnt": send[self.off_counts:self.block].view(torch.int64), "event": None}
            if self.active:
                b["recv"] = torch.empty(world * self.block, dtype=torch.uint8, device=device)
                b["m_idx"] = torch.empty((nq, k), dtype=torch.int64, device=device)
                b["m_dist"] = torch.empty((nq, k), dtype=torch.float32, device=device)
                b["m_cnt"] = torch.empty((nq,), dtype=torch.int64, device=device)
            self._bufs.append(b)
        self._cur = 0
        self._set_views()

    def _set_views(self):
        b = self._bufs[self._cur]
        self.send, self.idx, self.dist, self.cnt = b["send"], b["idx"], b["dist"], b["cnt"]

    def begin_step(self):
        """buffers of the next step: (idx, dist, cnt) views the local search writes into"""
        if self.depth > 1:
            self._cur = (self._cur + 1) % self.depth
            ev = self._bufs[self._cur]["event"]
            if ev is not None:
                ev.synchronize()  # the exchange that last read / wrote these buffers
            self._set_views()
        return self.idx, self.dist, self.cnt

    @property
    def slot(self) -> int:
        """the buffer set begin_step() handed out last (pass it to exchange_merge when later steps have begun since)"""
        return self._cur

    def exchange_merge(self, gpu_index, group=None, slot=None):
        """After the local search has filled idx / dist / cnt (of buffer set `slot`, default: the current one): the merged
        (idx, dist, cnt), the same on every rank.  For k <= 64 they are valid once torch's current stream has passed this point
        (see the class comment)."""
        b = self._bufs[self._cur if slot is None else slot]
        if not self.active:
            return b["idx"], b["dist"], b["cnt"]
        import torch
        import torch.distributed as dist

        dist.all_gather_into_tensor(b["recv"], b["send"], group=group)
        stream = torch.cuda.current_stream()
        args = (b["recv"].data_ptr(), self.block, self.off_ids, self.off_dists, self.off_counts, self.world, self.nq, self.k,
                b["m_idx"].data_ptr(), b["m_dist"].data_ptr(), b["m_cnt"].data_ptr())
        if self.k <= 64:
            gpu_index.merge_topk_gathered_async(*args, stream=stream.cuda_stream)
            if b["event"] is None:
                b["event"] = torch.cuda.Event()
            b["event"].record(stream)
        else:  # the scratch-list merge: waits for the collective's stream itself and returns synchronised
            gpu_index.merge_topk_gathered(*args, stream=stream.cuda_stream)
        return b["m_idx"], b["m_dist"], b["m_cnt"]

    def wait(self):
        """host wait for every enqueued exchange"""
        for b in self._bufs:
            if b["event"] is not None:
                b["event"].synchronize()
