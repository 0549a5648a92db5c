"""N>1 path on the HIP kernels (run with -m gpu): two processes under torch.distributed share cuda:0 (the box has
one GPU; RCCL refuses two ranks on one device, so the process group is gloo -- the collective call is the same
`all_gather_into_tensor` bench.py issues over RCCL).  Each rank takes its byte-balanced shard, runs the HIP
workspace on it, packs its results on the device and joins the gather; the union must equal the oracle's
single-process result, bit for bit."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SENT = 6000


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    try:
        import torch
        import torch.distributed as dist
        import vibrato_amd as V
        from tools import synth
        from vibrato_amd import sharding
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        sd = synth.SynthDict("small")
        dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        tok = V.Tokenizer(dv, device=0).ignore_space(True).max_grouping_len(24)
        text, offs = sd.sentences(N_SENT, "mixed", space_p=0.05)
        ltext, loffs, (lo, hi) = sharding.local_shard(text, offs, rank, world)
        n_local, nbytes = hi - lo, len(ltext)
        d_text = torch.from_numpy(np.ascontiguousarray(ltext)).cuda()
        d_offs = torch.from_numpy(loffs.astype(np.int64)).cuda()
        ws = tok.workspace(n_local, nbytes)
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n_local, nbytes, torch.cuda.current_stream().cuda_stream)
        st = ws.stats()
        assert st["error_flags"] == 0
        max_s = sharding.agree_max(n_local, device="cuda")
        max_t = sharding.agree_max(st["n_tokens"], device="cuda")
        # the kernels write the rank's slot themselves (vbt_workspace_set_packed_output): no pack step in front of the collective
        send = torch.zeros(sharding.packed_bytes(max_s, max_t), dtype=torch.uint8, device="cuda")
        ws.set_packed_output(send.data_ptr(), send.numel(), max_s)
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n_local, nbytes, torch.cuda.current_stream().cuda_stream)
        assert ws.stats()["error_flags"] == 0
        out, _ = sharding.gather_packed(send)
        torch.cuda.synchronize()
        assert out.is_cuda
        dist.barrier()
        res = None
        if rank == 0:
            parts, totals = [], []
            for r in range(world):
                n_s, n_t, off, cnt, tk = sharding.unpack_results(out[r], max_s)
                ordered, _ = sharding.tokens_in_sentence_order(off, cnt, tk)
                parts.append(ordered.tobytes())
                totals.append((n_s, n_t))
            res = (totals, parts)
        dist.destroy_process_group()
        if rank == 0:
            q.put(("ok", res))
    except Exception as e:  # surface the failure in the parent instead of a timeout
        import traceback
        q.put(("error", f"rank {rank}: {e}\n{traceback.format_exc()}"))
        raise


def test_two_process_hip_shards_and_device_gather():
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    from tools import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    status, payload = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    assert status == "ok", payload
    assert all(p.exitcode == 0 for p in procs)
    totals, parts = payload
    sd = synth.SynthDict("small")
    d = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    text, offs = sd.sentences(N_SENT, "mixed", space_p=0.05)
    toks, _ = ora.Tokenizer(d, True, 24).new_worker().tokenize_batch(text, offs)
    assert sum(t[0] for t in totals) == N_SENT
    assert sum(t[1] for t in totals) == len(toks)
    assert b"".join(parts) == toks.tobytes()


def test_bench_py_multi_rank_path_runs_config4_shape():
    """bench.py --gpus 2 end to end (sharded corpus, HIP per rank, packed device-resident gather on the communication
    stream, parity of the gathered records) with both ranks on this box's single GPU and gloo instead of RCCL."""
    import json
    import subprocess
    env = dict(os.environ, VBT_BENCH_BACKEND="gloo", VBT_BENCH_SINGLE_DEVICE="1", MASTER_ADDR="127.0.0.1")
    port = 29600 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--dict", "small", "--sentences", "20000"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    _check_multi_rank_line(p.stdout)


def _check_multi_rank_line(stdout):
    import json
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines  # ONE JSON line, from rank 0
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "strong" and r["parity_vs_oracle_sample"] is True
    assert r["gather"]["delivered_all_shards"] is True and r["gather"]["device_resident"] is True
    assert r["value"] > 0 and r["tokens_per_step"] > 20000
    # rank 0's shard carries the roofline and a bounded CPU baseline at N > 1 as well
    assert r["roofline"]["frac"] > 0 and r["roofline"]["kernel_ms"] > 0
    assert r["cpu_baseline"]["value"] > 0 and r["cpu_baseline"]["cores"] == 1


def test_bench_py_launches_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a driver without torch.distributed.run would start
    it): bench.py re-executes itself under torch.distributed.run and rank 0 still prints exactly one JSON line."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(VBT_BENCH_BACKEND="gloo", VBT_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dict", "small", "--sentences", "20000"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    _check_multi_rank_line(p.stdout)


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    try:
        import torch
        import torch.distributed as dist
        import vibrato_amd as V
        from tools import synth
        from vibrato_amd import sharding
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(port)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(0)
        # the real backend: "nccl" IS RCCL on ROCm; one rank is all a one-GPU box admits
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        sd = synth.SynthDict("small")
        dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        tok = V.Tokenizer(dv, device=0).ignore_space(True).max_grouping_len(24)
        text, offs = sd.sentences(N_SENT, "mixed", space_p=0.05)
        ltext, loffs, (lo, hi) = sharding.local_shard(text, offs, 0, 1)
        n_local, nbytes = hi - lo, len(ltext)
        d_text = torch.from_numpy(np.ascontiguousarray(ltext)).cuda()
        d_offs = torch.from_numpy(loffs.astype(np.int64)).cuda()
        ws = tok.workspace(n_local, nbytes)
        stream = torch.cuda.current_stream().cuda_stream
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n_local, nbytes, stream)
        st = ws.stats()
        assert st["error_flags"] == 0
        ntok = int(st["n_tokens"])
        max_s = sharding.agree_max(n_local, device="cuda")   # all_reduce(MAX) over RCCL
        max_t = sharding.agree_max(ntok, device="cuda")
        assert (max_s, max_t) == (n_local, ntok)
        slot = sharding.packed_bytes(max_s, max_t)
        send = [torch.empty(slot, dtype=torch.uint8, device="cuda") for _ in range(2)]
        out = [torch.empty(slot, dtype=torch.uint8, device="cuda") for _ in range(2)]
        work = [None, None]
        comm = torch.cuda.Stream()
        # bench.py's step(): kernels on the launch stream, pack on the launch stream, the collective asynchronously on the
        # communication stream, double-buffered; three steps so that both buffers are reused once
        for k in range(4):
            b = k & 1
            if work[b] is not None:
                work[b].wait()
            send[b].zero_()
            ws.set_packed_output(send[b].data_ptr(), slot, max_s)  # the kernels write the slot: no pack step
            ws.run(d_text.data_ptr(), d_offs.data_ptr(), n_local, nbytes, stream)
            ready = torch.cuda.Event()
            ready.record()
            with torch.cuda.stream(comm):
                comm.wait_event(ready)
                # (both forms of the final exchange on the real backend: to every rank, and the gather to rank 0 that bench.py defaults to)
                if k < 2:
                    _, work[b] = sharding.gather_packed(send[b], out[b], async_op=True)
                else:
                    _, work[b] = sharding.gather_to_root(send[b], out[b], root=0, async_op=True)
        for wk in work:
            wk.wait()
        torch.cuda.synchronize()
        # Does a collective in flight slow the kernels down?  (RCCL's collectives are CU kernels; shader-driven copies were
        # measured to stall the other kernels in flight)  The same 20 steps with and without the gather of
        # the previous step running on the communication stream; the ratio goes to profiles/ via gpurun_out/.
        import time as _time
        def timed(with_gather):
            torch.cuda.synchronize()
            t0 = _time.perf_counter()
            pending = [None, None]
            for k in range(20):
                if with_gather and pending[k & 1] is not None:
                    pending[k & 1].wait()
                ws.set_packed_output(send[k & 1].data_ptr(), slot, max_s)
                ws.run(d_text.data_ptr(), d_offs.data_ptr(), n_local, nbytes, stream)
                if with_gather:
                    ready = torch.cuda.Event()
                    ready.record()
                    with torch.cuda.stream(comm):
                        comm.wait_event(ready)
                        _, pending[k & 1] = sharding.gather_to_root(send[k & 1], out[k & 1], root=0, async_op=True)
            for p_ in pending:
                if p_ is not None:
                    p_.wait()
            torch.cuda.synchronize()
            return (_time.perf_counter() - t0) / 20
        timed(False); timed(True)
        t_plain, t_gather = timed(False), timed(True)
        overlap = {"ms_per_step_kernels_only": round(t_plain * 1e3, 4), "ms_per_step_with_gather_in_flight": round(t_gather * 1e3, 4),
                   "ratio": round(t_gather / t_plain, 4), "slot_bytes": slot, "sentences": n_local, "world_size": 1, "backend": dist.get_backend()}
        tmax = torch.tensor([1.5], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        assert float(tmax.item()) == 1.5
        dist.barrier()
        res = []
        for b in range(2):
            n_s, n_t, off, cnt, tk = sharding.unpack_results(out[b].view(1, -1)[0], max_s)
            ordered, _ = sharding.tokens_in_sentence_order(off, cnt, tk)
            res.append((n_s, n_t, ordered.tobytes()))
        backend = dist.get_backend()
        dist.destroy_process_group()
        q.put(("ok", (backend, res, overlap)))
    except Exception as e:
        import traceback
        q.put(("error", f"{e}\n{traceback.format_exc()}"))
        raise


def test_rccl_backend_world_size_one_gather_on_the_communication_stream():
    """The collective calls bench.py issues at N > 1 -- init_process_group("nccl", device_id=...), all_reduce(MAX) for the slot
    sizes, pack_results, asynchronous all_gather_into_tensor on the communication stream with the double buffer, barrier --
    (the kernels write every rank's slot themselves: vbt_workspace_set_packed_output) executed on the real RCCL backend (world size 1: what one GPU admits); the gathered records equal the oracle's."""
    import torch.multiprocessing as mp
    from oracle import oracle as ora
    from tools import synth
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + (os.getpid() % 2000)
    p = ctx.Process(target=_rccl_worker, args=(port, q))
    p.start()
    status, payload = q.get(timeout=600)
    p.join(timeout=120)
    assert status == "ok", payload
    assert p.exitcode == 0
    backend, res, overlap = payload
    assert backend == "nccl"
    print("rccl world-size-1 step with / without the gather in flight:", overlap)
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        import json
        with open(os.path.join(out_dir, "rccl_ws1_overlap.json"), "w") as f:
            json.dump(overlap, f)
    sd = synth.SynthDict("small")
    d = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    text, offs = sd.sentences(N_SENT, "mixed", space_p=0.05)
    toks, _ = ora.Tokenizer(d, True, 24).new_worker().tokenize_batch(text, offs)
    for n_s, n_t, blob in res:
        assert n_s == N_SENT and n_t == len(toks)
        assert blob == toks.tobytes()
