"""Sentence sharding for multi-GPU runs: one process per GPU, each tokenizes an independent
contiguous range of sentences (a Worker holds only per-sentence state, worker.rs:13-19, and the
dictionary is immutable), so there is no data-path collective; only per-rank totals (and, if the
caller wants them on one rank, token records) are gathered at the end over RCCL/xGMI."""
import numpy as np


def shard_bounds(offsets, world_size):
    """Split n sentences into `world_size` contiguous ranges balanced by byte count (cost is
    proportional to characters to first order). Returns world_size+1 sentence indices."""
    offsets = np.asarray(offsets, dtype=np.uint64)
    n = len(offsets) - 1
    total = int(offsets[-1] - offsets[0])
    bounds = [0]
    for r in range(1, world_size):
        target = int(offsets[0]) + (total * r) // world_size
        bounds.append(int(np.searchsorted(offsets, target, side="left")))
    bounds.append(n)
    for i in range(1, len(bounds)):  # monotone, never beyond n
        bounds[i] = min(max(bounds[i], bounds[i - 1]), n)
    return bounds


def local_shard(text, offsets, rank, world_size):
    """(text, offsets) of this rank's range, offsets rebased to 0."""
    b = shard_bounds(offsets, world_size)
    lo, hi = b[rank], b[rank + 1]
    offs = np.asarray(offsets[lo:hi + 1], dtype=np.uint64)
    return np.asarray(text)[int(offs[0]):int(offs[-1])], offs - offs[0], (lo, hi)


def gather_totals(local_sentences, local_tokens, device=None):
    """all_gather of (sentences, tokens) per rank (torch.distributed must be initialised:
    backend "nccl" = RCCL on ROCm, or "gloo" in CPU tests). Returns an int64 array [world, 2]."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    mine = torch.tensor([local_sentences, local_tokens], dtype=torch.int64, device=device)
    out = torch.zeros(world * 2, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(out, mine)
    return out.view(world, 2).cpu().numpy()


def gather_token_records(tokens_np, device=None):
    """Variable-length gather of token records to every rank (padded all_gather; RCCL has no
    gatherv). tokens_np: structured array of 24-byte records. Returns a list of arrays by rank."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size()
    raw = torch.from_numpy(np.ascontiguousarray(tokens_np).view(np.uint8).copy())
    n = torch.tensor([raw.numel()], dtype=torch.int64, device=device)
    sizes = torch.zeros(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(sizes, n)
    cap = int(sizes.max().item())
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[:raw.numel()] = raw.to(buf.device)
    out = torch.zeros(world * cap, dtype=torch.uint8, device=device)
    dist.all_gather_into_tensor(out, buf)
    out = out.cpu().numpy().reshape(world, cap)
    return [out[r, :int(sizes[r].item())].view(tokens_np.dtype) for r in range(world)]
