"""Sentence sharding for multi-GPU runs: one process per GPU, each tokenizes an independent
contiguous range of sentences (a Worker holds only per-sentence state, worker.rs:13-19, and the
dictionary is immutable), so there is no data-path collective.  The only exchange is the final
gather of the per-rank results, which stays device-resident: every rank's padded slot -- its totals,
per-sentence token ranges and 24-byte token records -- goes to the root (`gather_to_root`: grouped
send/recv under RCCL over xGMI when the backend is "nccl"; RCCL has no gatherv) or, on request, to
every rank (`gather_packed`: one `all_gather_into_tensor`).  On the GPU the kernels write a rank's slot themselves
(`Workspace.set_packed_output`: `vbt_workspace_set_packed_output`); `pack_results` lays the same slot out with tensor copies for
the CPU (gloo) tests and for callers that already hold results in a workspace's own buffers."""
import numpy as np

TOKEN_BYTES = 24
_HEADER = 32  # bytes: n_sentences (u64), n_tokens (u32), padding: keeps the payload 16-byte aligned


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


class _DevicePtr:
    """Raw device memory as a `__cuda_array_interface__` object (uint8), so torch can alias it
    without a copy.  `keep` pins whatever owns the memory."""

    def __init__(self, ptr, nbytes, keep=None):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
        self._keep = keep


def device_view(ptr, nbytes, keep=None):
    """torch.uint8 tensor aliasing `nbytes` of device memory at `ptr` (plumbing: no copy, no compute)."""
    import torch
    if nbytes == 0:
        return torch.empty(0, dtype=torch.uint8, device="cuda")
    return torch.as_tensor(_DevicePtr(ptr, nbytes, keep), device="cuda")


def packed_bytes(max_sentences, max_tokens):
    """Size of one rank's slot in the gather buffer."""
    return _HEADER + 8 * int(max_sentences) + TOKEN_BYTES * int(max_tokens)


def pack_results(out, n_sentences, n_tokens, total_dev, tok_off, tok_cnt, tokens, max_sentences):
    """Lays one rank's results out in `out` (uint8 tensor of packed_bytes(...) bytes, 8-byte aligned):
    header {n_sentences u64, n_tokens u32 (copied from the device counter `total_dev`), error flags u32 = 0, 0 x 16 bytes}, tok_off[u32 x
    max_sentences], tok_cnt[u32 x max_sentences], then the first `n_tokens` token records.  tok_off / tok_cnt /
    tokens / total_dev are uint8 views of the workspace's result buffers (device_view); only device-side copies and
    fills are issued, on the current stream -- nothing passes through host memory."""
    import torch
    out[:8].view(torch.int64).fill_(int(n_sentences))
    out[8:_HEADER].zero_()
    out[8:12].copy_(total_dev[:4], non_blocking=True)
    base = _HEADER
    out[base:base + 4 * n_sentences].copy_(tok_off[:4 * n_sentences], non_blocking=True)
    base += 4 * max_sentences
    out[base:base + 4 * n_sentences].copy_(tok_cnt[:4 * n_sentences], non_blocking=True)
    base += 4 * max_sentences
    out[base:base + TOKEN_BYTES * n_tokens].copy_(tokens[:TOKEN_BYTES * n_tokens], non_blocking=True)
    return out


def gather_packed(send, world_out=None, group=None, async_op=False):
    """One collective: every rank contributes its packed slot (same size on every rank) and receives all of
    them, device-resident.  Returns (out tensor [world, slot_bytes], work or None)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    if world_out is None:
        world_out = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
    try:
        work = dist.all_gather_into_tensor(world_out, send, group=group, async_op=async_op)
    except (RuntimeError, NotImplementedError):  # a backend without the flat form (old gloo builds)
        parts = list(world_out.view(world, -1).unbind(0))
        work = dist.all_gather(parts, send, group=group, async_op=async_op)
    return world_out.view(world, -1), work


def gather_to_root(send, root_out=None, root=0, group=None, async_op=False):
    """The path's final exchange as a true GATHER (north star: "RCCL over xGMI only for the final gather"): every rank sends its
    packed slot to `root` only -- grouped send/recv under RCCL, 1/world of the all-gather's inbound traffic per non-root rank, and
    nothing at all lands on them.  `root` is a GLOBAL rank (what torch's `dst` means, also with a sub-group).  `root_out`
    ([world * slot_bytes] uint8, device-resident) is needed on the root only.
    Returns (out tensor [world, slot_bytes] on the root, None elsewhere; work or None)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    is_root = dist.get_rank() == root
    parts = None
    if is_root:
        if root_out is None:
            root_out = torch.empty(world * send.numel(), dtype=torch.uint8, device=send.device)
        parts = list(root_out.view(world, -1).unbind(0))
    work = dist.gather(send, parts, dst=root, group=group, async_op=async_op)
    return (root_out.view(world, -1) if is_root else None), work


def unpack_results(slot, max_sentences):
    """Host view of one rank's slot: (n_sentences, n_tokens, tok_off, tok_cnt, tokens[structured]).  Raises if the rank that wrote
    the slot flagged it (header word 3: device error flags, e.g. 1 = the slot was too small for the rank's tokens)."""
    from .api import TOKEN_DTYPE
    raw = slot.cpu().numpy()
    n_s, n_t = int(raw[:8].view(np.int64)[0]), int(raw[8:12].view(np.uint32)[0])
    flags = int(raw[12:16].view(np.uint32)[0])
    if flags:
        raise RuntimeError(f"packed result slot carries device error flags {flags} (1 = slot too small for the batch's tokens)")
    base = _HEADER
    off = raw[base:base + 4 * n_s].view(np.uint32)
    base += 4 * max_sentences
    cnt = raw[base:base + 4 * n_s].view(np.uint32)
    base += 4 * max_sentences
    toks = raw[base:base + TOKEN_BYTES * n_t].view(TOKEN_DTYPE)
    return n_s, n_t, off, cnt, toks


def tokens_in_sentence_order(off, cnt, toks):
    """Token records of one rank re-ordered by sentence (the device hands out token ranges in arrival order)."""
    ends = np.zeros(len(cnt) + 1, dtype=np.int64)
    ends[1:] = np.cumsum(cnt, dtype=np.int64)
    idx = np.repeat(off.astype(np.int64) - ends[:-1], cnt) + np.arange(int(ends[-1]), dtype=np.int64)
    return toks[idx], ends


def agree_max(value, device=None, group=None):
    """all_reduce(MAX) of one integer (slot sizes must be equal on every rank)."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(value)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return int(t.item())


def workspace_views(ws, n_sentences, max_tokens):
    """uint8 device views (no copies) of a Workspace's result buffers, ready for pack_results."""
    p = ws.result_ptrs()
    return {"tokens": device_view(p["tokens"], TOKEN_BYTES * int(max_tokens), ws), "tok_off": device_view(p["tok_off"], 4 * int(n_sentences), ws),
            "tok_cnt": device_view(p["tok_cnt"], 4 * int(n_sentences), ws), "total": device_view(p["total"], 4, ws)}
