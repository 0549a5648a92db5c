"""N>1 path without a GPU: two gloo ranks shard a batch, tokenize their ranges independently, pack
their results and exchange them with the one collective the path has (sharding.gather_to_root, or gather_packed to every rank); the
union must equal the single-process result.  The compute stand-in on CPU is the oracle (the HIP path
needs a GPU -- tests/test_distributed_gpu.py runs the same exchange over the HIP workspace); the
sharding, packing and collective code under test is exactly what bench.py --gpus N uses."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from oracle import oracle as ora
    from tools import synth
    from vibrato_amd import sharding
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sd = synth.SynthDict("tiny")
    d = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    w = ora.Tokenizer(d).new_worker()
    text, offs = sd.sentences(600, "lognormal_40")
    ltext, loffs, (lo, hi) = sharding.local_shard(text, offs, rank, world)
    toks, toff = w.tokenize_batch(ltext, loffs)
    n_local = hi - lo
    max_s = sharding.agree_max(n_local)
    max_t = sharding.agree_max(len(toks))
    as_u8 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy())
    send = torch.zeros(sharding.packed_bytes(max_s, max_t), dtype=torch.uint8)
    sharding.pack_results(send, n_local, len(toks), as_u8(np.array([len(toks)], dtype=np.uint32)),
                          as_u8(toff[:-1].astype(np.uint32)), as_u8(np.diff(toff).astype(np.uint32)), as_u8(toks), max_s)
    out, _ = sharding.gather_packed(send)
    # the same slots as a true gather (bench.py's default): the root holds every shard, the other ranks receive nothing
    out_root, _ = sharding.gather_to_root(send, root=0)
    assert (out_root is None) == (rank != 0)
    if rank == 0:
        assert bytes(out_root.numpy().tobytes()) == bytes(out.numpy().tobytes())
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        parts, totals = [], []
        for r in range(world):
            n_s, n_t, off, cnt, tk = sharding.unpack_results(out[r], max_s)
            ordered, _ = sharding.tokens_in_sentence_order(off, cnt, tk)
            parts.append(ordered.tobytes())
            totals.append((n_s, n_t))
        q.put((totals, parts, sharding.shard_bounds(offs, world)))


def test_two_rank_sharding_and_gather():
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    from tools import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    totals, parts, bounds = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sd = synth.SynthDict("tiny")
    d = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    text, offs = sd.sentences(600, "lognormal_40")
    toks, toff = ora.Tokenizer(d).new_worker().tokenize_batch(text, offs)
    assert bounds[0] == 0 and bounds[-1] == 600 and 0 < bounds[1] < 600
    assert sum(t[0] for t in totals) == 600
    assert sum(t[1] for t in totals) == len(toks)
    assert b"".join(parts) == toks.tobytes()  # rank order = sentence order: records are sentence-relative


def test_shard_bounds_edge_cases():
    from vibrato_amd import sharding
    assert sharding.shard_bounds(np.array([0], dtype=np.uint64), 4) == [0, 0, 0, 0, 0]
    assert sharding.shard_bounds(np.array([0, 10], dtype=np.uint64), 2)[-1] == 1
    offs = np.cumsum([0] + [5] * 100).astype(np.uint64)
    b = sharding.shard_bounds(offs, 8)
    assert b[0] == 0 and b[-1] == 100 and all(b[i] <= b[i + 1] for i in range(8))
    sizes = [b[i + 1] - b[i] for i in range(8)]
    assert max(sizes) - min(sizes) <= 1
    offs = np.array([0, 1000, 1001, 1002, 1003], dtype=np.uint64)  # one huge sentence
    b = sharding.shard_bounds(offs, 2)
    assert b == sorted(b) and b[-1] == 4


def test_pack_unpack_round_trip():
    import torch
    from vibrato_amd import sharding, TOKEN_DTYPE
    toks = np.zeros(5, dtype=TOKEN_DTYPE)
    toks["start_char"] = np.arange(5)
    off = np.array([3, 0, 3], dtype=np.uint32)   # sentence 0 -> records 3,4; sentence 1 -> records 0..2; sentence 2 empty
    cnt = np.array([2, 3, 0], dtype=np.uint32)
    as_u8 = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).copy())
    buf = torch.zeros(sharding.packed_bytes(7, 9), dtype=torch.uint8)
    sharding.pack_results(buf, 3, 5, as_u8(np.array([5], dtype=np.uint32)), as_u8(off), as_u8(cnt), as_u8(toks), 7)
    n_s, n_t, o, c, t = sharding.unpack_results(buf, 7)
    assert (n_s, n_t) == (3, 5) and o.tolist() == [3, 0, 3] and c.tolist() == [2, 3, 0]
    ordered, ends = sharding.tokens_in_sentence_order(o, c, t)
    assert ordered["start_char"].tolist() == [3, 4, 0, 1, 2] and ends.tolist() == [0, 2, 5, 5]


def test_bench_py_self_launches_ranks_when_no_launcher_started_it():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py starts its own two ranks under torch.distributed.run (here, without a
    GPU, every rank then stops at "needs an MI355X" -- not at the old "launch with torch.distributed.run" refusal)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dict", "tiny", "--sentences", "100"], env=env, cwd=root,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    # (the launcher tears the other rank down as soon as one has failed: one or both get as far as the message)
    assert 1 <= p.stderr.count("bench.py needs an MI355X") <= 2, p.stderr[-3000:]
    assert "launch with torch.distributed.run" not in p.stderr
