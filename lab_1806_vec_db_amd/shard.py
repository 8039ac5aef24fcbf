"""Row sharding of one corpus over the GPUs of a node.

Rows are independent for Flat and PQ-Flat, so shard s holds the contiguous block
[s*ceil(N/S), min(N, (s+1)*ceil(N/S))) and reports GLOBAL ids (vdb_index_set_id_offset).
Every rank computes its local top-k for all queries; ONE all-gather of [nq, k] (f32 distance, i64 id,
i64 count) per rank over RCCL/xGMI follows, then each rank merges the S sorted lists per query by the
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
    # all_gather_into_tensor concatenates along dim 0: [world*nq, k] viewed as [world, nq, k]
    g_idx = torch.empty((world * nq, k), dtype=local_idx.dtype, device=local_idx.device)
    g_dist = torch.empty((world * nq, k), dtype=local_dist.dtype, device=local_dist.device)
    g_cnt = torch.empty((world * nq,), dtype=local_cnt.dtype, device=local_cnt.device)
    dist.all_gather_into_tensor(g_idx, local_idx.contiguous(), group=group)
    dist.all_gather_into_tensor(g_dist, local_dist.contiguous(), group=group)
    dist.all_gather_into_tensor(g_cnt, local_cnt.contiguous(), group=group)
    g_idx, g_dist, g_cnt = g_idx.view(world, nq, k), g_dist.view(world, nq, k), g_cnt.view(world, nq)
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
