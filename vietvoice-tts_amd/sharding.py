"""Data-parallel plumbing for one node: which rank synthesises which unit, and how the model gets there.

The hot path has no cross-rank dependence: each (reference clip, text chunk) unit is
synthesised independently; only the host-side cross-fade couples chunks of one text.  So a job on N
GPUs = one process per GPU, ONE collective at start-up (broadcast of the flat weight buffer, RCCL
over xGMI on GPUs / gloo in the CPU tests), then each rank works through its own shard.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import pack
from .model_spec import ModelSpec


def shard_units(costs: Sequence[float], world: int) -> List[List[int]]:
    """Longest-processing-time assignment of unit indices to ranks.  cost ~ N * (8 D^2 + 2 N D) per
    unit (frames N); equal-cost units degrade to round-robin.  Every unit appears exactly once."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def plan_batches(frames: Sequence[int], max_units: int, pad_frac: float = 0.08, min_units: int = 8) -> List[List[int]]:
    """Length-bucketed GPU batches for ragged units.  Every kernel except attention works on B x N_max rows, so a batch
    pays for its padding; units are sorted by frames and cut into batches of at most ``max_units`` such that the padded
    share of a batch stays below ``pad_frac`` once it holds ``min_units`` (small batches under-fill the GEMM tiles).
    Returns lists of unit indices, longest batch first; every unit appears exactly once."""
    order = sorted(range(len(frames)), key=lambda i: (-int(frames[i]), i))
    out: List[List[int]] = []
    cur: List[int] = []
    tot = 0
    for i in order:
        f = int(frames[i])
        if cur:
            n_max = int(frames[cur[0]])
            waste = 1.0 - (tot + f) / float((len(cur) + 1) * n_max)
            if len(cur) >= max_units or (len(cur) >= min_units and waste > pad_frac):
                out.append(cur)
                cur, tot = [], 0
        cur.append(i)
        tot += f
    if cur:
        if out and len(cur) < min_units and len(out[-1]) + len(cur) <= max_units:
            out[-1].extend(cur)                  # a short tail rides with the previous batch rather than alone
        else:
            out.append(cur)
    return out


def unit_cost(frames: int, dim: int = 1024) -> float:
    return float(frames) * (8.0 * dim * dim + 2.0 * frames * dim)


def broadcast_weights(spec: ModelSpec, acoustic_dtype: torch.dtype, weights: Optional[Dict[str, torch.Tensor]],
                      device: torch.device, src: int = 0) -> Tuple[torch.Tensor, list]:
    """Collective C1.  Rank `src` packs the fp32 weights into the flat device layout; every rank
    allocates the same number of bytes (the layout is a function of shapes only) and receives it with
    ONE broadcast.  Returns (flat uint8 tensor on `device`, [(name, offset, nbytes)])."""
    import torch.distributed as dist
    table, total = pack.plan(spec, acoustic_dtype)
    rank = dist.get_rank() if dist.is_initialized() else 0
    if rank == src:
        if weights is None:
            raise ValueError("the source rank needs the weights")
        cpu = torch.zeros(total, dtype=torch.uint8)
        pack.fill(spec, acoustic_dtype, weights, cpu)
        flat = cpu.to(device)
    else:
        flat = torch.empty(total, dtype=torch.uint8, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src=src)
    return flat, table


def gather_pcm(pcm: torch.Tensor, pcm_len: torch.Tensor, unit_ids: Sequence[int], n_units: int) -> Optional[List[torch.Tensor]]:
    """Collective C2 (optional): bring every rank's PCM to all ranks in global unit order.
    pcm int16 [B_r][L_r] and pcm_len int32 [B_r] on this rank's device, unit_ids = the global indices of its rows
    (`shard_units`).  Two fixed-shape all_gathers (lengths + ids, then samples padded to the job-wide maximum):
    no variable-size exchange, so it maps onto RCCL's ring all_gather; 32 units of 11 s are 17 MB per rank.
    Returns a list of n_units 1-D int16 CPU tensors (every rank gets it; rank 0 is the usual consumer)."""
    import torch.distributed as dist
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        out: List[Optional[torch.Tensor]] = [None] * n_units
        for j, u in enumerate(unit_ids):
            out[u] = pcm[j, : int(pcm_len[j])].cpu()
        return out
    world = dist.get_world_size()
    dev = pcm.device
    B = torch.tensor([pcm.shape[0], pcm.shape[1]], dtype=torch.int64, device=dev)
    shapes = [torch.zeros_like(B) for _ in range(world)]
    dist.all_gather(shapes, B)
    Bm, Lm = int(max(int(t[0]) for t in shapes)), int(max(int(t[1]) for t in shapes))
    meta = torch.full((Bm, 2), -1, dtype=torch.int32, device=dev)          # (unit id, length) per row, -1 = padding row
    meta[: pcm.shape[0], 0] = torch.as_tensor(list(unit_ids), dtype=torch.int32, device=dev)
    meta[: pcm.shape[0], 1] = pcm_len.to(torch.int32)
    pad = torch.zeros((Bm, Lm), dtype=torch.int16, device=dev)
    pad[: pcm.shape[0], : pcm.shape[1]] = pcm
    metas = [torch.empty_like(meta) for _ in range(world)]
    pad_b = pad.view(torch.uint8)                          # int16 is not a collective dtype (gloo / RCCL): move the bytes
    pads_b = [torch.empty_like(pad_b) for _ in range(world)]
    dist.all_gather(metas, meta)
    dist.all_gather(pads_b, pad_b)
    pads = [t.view(torch.int16) for t in pads_b]
    out = [None] * n_units
    for mt, pd in zip(metas, pads):
        mt, pd = mt.cpu(), pd.cpu()
        for j in range(Bm):
            u, n = int(mt[j, 0]), int(mt[j, 1])
            if u >= 0:
                out[u] = pd[j, :n].clone()
    if any(o is None for o in out):
        raise RuntimeError("gather_pcm: some units were produced by no rank")
    return out
