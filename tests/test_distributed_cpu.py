"""N>1 path on CPU: two gloo ranks shard a batch, tokenize their ranges independently and
gather totals / token records; the union must equal the single-process result.  The compute
stand-in on CPU is the oracle (the HIP path needs a GPU); the sharding + collective code under
test is exactly what bench.py / a multi-GPU caller uses."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
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
    totals = sharding.gather_totals(hi - lo, len(toks))
    parts = sharding.gather_token_records(toks)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        q.put((totals.tolist(), [p.tobytes() for p in parts], sharding.shard_bounds(offs, world)))


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
