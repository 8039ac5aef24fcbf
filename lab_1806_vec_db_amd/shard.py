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
