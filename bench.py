#!/usr/bin/env python3
"""Headline benchmark: batched tokenization throughput (sentences/s) on MI355X.

A "step" is one pass of the tokenize() hot path over one device-resident batch of synthetic Japanese sentences.

  --gpus 1 (default)  BASELINE config 3: unidic-cwj-3.1.1-shaped dictionary (458.6 MiB connection matrix),
                      100k sentences, text and offsets resident in HBM, results left in HBM.
  --gpus N > 1        BASELINE config 4: ONE corpus of 1M sentences (same law, same seed on every rank), split into
                      N contiguous shards balanced by bytes (vibrato_amd.sharding); one process per GPU tokenizes
                      its shard -- the kernels write the results straight into the rank's slot of the gather
                      (vbt_workspace_set_packed_output) -- and joins the path's only collective, the final gather to
                      rank 0 over RCCL/xGMI (device-resident, issued on a communication stream so that it overlaps the
                      next step's kernels; --gather all: all_gather_into_tensor).  Fixed total work: "scaling": "strong".

Launching: `python bench.py --gpus N` with no WORLD_SIZE in the environment starts its own N ranks (it re-executes itself
under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`); started under
torch.distributed.run it uses the ranks it is given.  Either way rank 0 prints ONE JSON line.

The line (contract in the task statement): value = sentences of the whole job per second, `roofline` (dominant kernel,
algorithmic bytes from the oracle's event counters / hipEvent kernel time, measured on rank 0's shard), `cpu_baseline` (the C
restatement of vibrato's CPU path in oracle/, timed on this host: the whole batch at N = 1, a 20k-sentence sample of rank 0's
shard at N > 1) and, at N = 1, three legs that are never `value`: `suite` (BASELINE config 5 and the dense lexicon law through
the same timed loop), `worker_loop` (the reference's per-sentence 3-call loop through Worker), `host_to_host` and `format`
(text in -> MeCab-format text out, next to the oracle's tokenize + print loop).
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# one HIP stream per LDS tier of the lattice kernel: let the runtime use more than 4 hardware queues
# (read when the HIP runtime initialises, i.e. before the first torch.cuda call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")

HBM_PEAK_BPS = 8.0e12   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
N_SIMD = 256 * 4        # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9        # max shader clock
VALU_CYCLES = 4         # issue cycles one wave64 VALU instruction is charged on its SIMD (the unit SQ_ACTIVE_INST_VALU counts in)


def algorithmic_bytes(c):
    """B_alg from the oracle's event counters."""
    return (c["n_bytes"] + 4 * c["n_chars"] + 12 * c["n_trie_steps"] + 4 * (c["n_trie_hits"] + c["n_lex_matches"])
            + 8 * c["n_lex_matches"] + 8 * c["n_unk_nodes"] + 2 * c["n_pairs_dedup"] + 24 * c["n_tokens"])


def committed_counters(workload):
    """HBM bytes and wave-instruction counts per step from the newest committed PMC summary of the same workload
    (profiles/*_traffic.json, written by tools/summarize_profile.py from separate rocprofv3 --pmc passes)."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json"))):
        try:
            t = json.load(open(f))
        except Exception:
            continue
        if t.get("workload") == workload:
            best = t
    return best


def cpu_protocol(run, trials=3, runs=3):
    """benchmark/src/main.rs:53-91: per trial `runs` timed runs, drop the fastest and the slowest, average the rest;
    report mean [min, max] over the trials (seconds per run)."""
    means = []
    for _ in range(trials):
        ts = []
        for _ in range(runs):
            t = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t)
        ts.sort()
        kept = ts[1:-1] if len(ts) > 2 else ts
        means.append(sum(kept) / len(kept))
    return sum(means) / len(means), min(means), max(means)


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, as the driver's
    torch.distributed.run command does) and hand their output through; rank 0 prints the JSON line."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL between processes), see the task's environment notes
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.call(cmd, env=env, cwd=ROOT)


def timed_leg(tok, text, offs, steps, warmup, torch):
    """The timed loop of one single-GPU leg (device-resident input, results left in HBM): returns (seconds, stats)."""
    n, nbytes = len(offs) - 1, int(len(text))
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    # the tokenizer's renumbering of its connection ids by measured usage, done up front (vbt_tokenizer_calibrate): left to the first
    # large batch it runs on a background thread and would share the GPU with the timed steps
    tok.calibrate(text=text, offsets=offs)
    ws = tok.workspace(n, nbytes)
    ws.set_timing(True)
    stream = torch.cuda.current_stream().cuda_stream
    for _ in range(warmup):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return dt, ws.stats()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dict", default="unidic", choices=["tiny", "small", "small-dense", "ipadic", "unidic", "unidic-dense"])
    ap.add_argument("--sentences", type=int, default=0, help="sentences of the whole job (default: 100000 at --gpus 1 = config 3, "
                                                             "1000000 at --gpus N > 1 = config 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ignore-space", action="store_true")
    ap.add_argument("--max-grouping-len", type=int, default=0)
    ap.add_argument("--user-lexicon", type=int, default=0, help="attach a synthetic user.csv of N compounds (BASELINE config 5)")
    ap.add_argument("--law", default="lognormal_40", choices=["uniform_5_20", "lognormal_40", "mixed"])
    ap.add_argument("--reorder", action="store_true", help="map the connection ids by usage frequency first (the reference's "
                                                           "`reorder` + `map` tools; statistics from a training batch on the GPU)")
    ap.add_argument("--no-host-pipeline", action="store_true", help="skip the host-to-host leg (vbt_tokenize_batch from host buffers to host "
                                                                    "results: one call, and host threads streaming batches); never `value`")
    ap.add_argument("--gather", default="root", choices=["root", "all"], help="the final exchange at --gpus N > 1: every rank's packed results to rank 0 "
                                                                           "(a gather: grouped send/recv) or to every rank (all_gather_into_tensor)")
    ap.add_argument("--no-suite", action="store_true", help="skip the extra single-GPU legs (BASELINE config 5, dense lexicon law); never `value`")
    ap.add_argument("--no-worker-loop", action="store_true", help="skip the per-sentence Worker leg (the reference's 3-call loop); never `value`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus} "
                         f"(or with no launcher at all: bench.py then starts its own ranks)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    # test hooks (tests/test_distributed_gpu.py runs this file with two ranks on a one-GPU box): the collective backend
    # ("nccl" = RCCL unless overridden) and whether every rank uses device 0
    backend = os.environ.get("VBT_BENCH_BACKEND", "nccl")
    if os.environ.get("VBT_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"rank {rank}: local rank {local_rank} but this node has {torch.cuda.device_count()} GPU(s)")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    import vibrato_amd as V
    from tools import synth
    from vibrato_amd import sharding

    t_setup = time.time()
    sd = synth.SynthDict(args.dict)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    user_csv = sd.user_csv(args.user_lexicon) if args.user_lexicon else None
    if user_csv is not None:
        dv.reset_user_lexicon_from_reader(user_csv)
    space_p = 0.1 if args.ignore_space else 0.0
    n_total = args.sentences or (100000 if world == 1 else 1000000)
    # the corpus: identical on every rank (same seed); a rank keeps only its shard
    text_all, offs_all = sd.sentences(n_total, args.law, space_p=space_p, seed=synth.SEED)
    total_bytes_all = int(len(text_all))
    if world > 1:
        text, offs, (lo, hi) = sharding.local_shard(text_all, offs_all, rank, world)
        text = np.ascontiguousarray(text)
        del text_all
    else:
        text, offs, (lo, hi) = text_all, offs_all, (0, n_total)
    n = hi - lo
    nbytes = int(len(text))

    reorder_info = None
    if args.reorder:
        # docs/map.md workflow on the GPU: count connection-id usage on a training batch, sort ids by probability
        # (Worker::compute_connid_probs), re-map the dictionary (Dictionary::map_connection_ids_from_iter).  Pure
        # permutation: results stay bit-exact (tests/test_connid_mapping.py), the hot ids become neighbours.
        t_r = time.time()
        dt = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        tt = V.Tokenizer(dt, device=local_rank).ignore_space(args.ignore_space).max_grouping_len(args.max_grouping_len)
        tr_text, tr_offs = sd.sentences(20000, args.law, space_p=space_p, seed=synth.SEED + 7919)
        wst = tt.workspace(20000, len(tr_text))
        wst.count_connids(True)
        dtt = torch.from_numpy(tr_text).cuda()
        dto = torch.from_numpy(tr_offs.astype(np.int64)).cuda()
        wst.run(dtt.data_ptr(), dto.data_ptr(), 20000, len(tr_text), torch.cuda.current_stream().cuda_stream)
        lidc, ridc = wst.connid_counts()
        lp, rp = V.compute_connid_probs(lidc, ridc)
        lmap, rmap = [i for i, _ in lp], [i for i, _ in rp]
        dv.map_connection_ids_from_iter(lmap, rmap)
        reorder_info = {"training_sentences": 20000, "seconds": round(time.time() - t_r, 2)}
        del wst, tt, dt, dtt, dto

    tok = V.Tokenizer(dv, device=local_rank).ignore_space(args.ignore_space).max_grouping_len(args.max_grouping_len)
    # the tokenizer's internal renumbering of the connection ids by measured usage, done up front on this rank's own text
    # (vbt_tokenizer_calibrate; left alone it runs on a background thread behind the first large batch, next to the timed steps)
    tok.calibrate(text=text, offsets=offs)
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(n, nbytes)
    ws.set_timing(os.environ.get("VBT_BENCH_TIMING", "1") != "0")  # (0: developer A/B of what the four timing events per step cost)
    stream = torch.cuda.current_stream().cuda_stream
    t_setup = time.time() - t_setup

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---- multi-GPU: the final gather, device-resident, double-buffered on a communication stream -------------
    gather = None
    if world > 1:
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)  # sizing run (the workload is deterministic)
        st0 = ws.stats()
        if st0["error_flags"]:
            raise SystemExit(f"device error flags {st0['error_flags']}")
        ntok_local = int(st0["n_tokens"])
        max_s = sharding.agree_max(n, device="cuda")
        max_t = sharding.agree_max(ntok_local, device="cuda")
        slot = sharding.packed_bytes(max_s, max_t)
        to_all = args.gather == "all"
        gather = {"send": [torch.empty(slot, dtype=torch.uint8, device="cuda") for _ in range(2)],
                  "out": [torch.empty(world * slot, dtype=torch.uint8, device="cuda") if (to_all or rank == 0) else None for _ in range(2)],
                  "work": [None, None], "comm": torch.cuda.Stream(), "slot": slot, "max_s": max_s, "k": 0,
                  "ev": [[torch.cuda.Event(enable_timing=True) for _ in range(2)] for _ in range(2)]}

    def step():
        if gather is None:
            ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)
            return
        b = gather["k"] & 1
        gather["k"] += 1
        if gather["work"][b] is not None:
            gather["work"][b].wait()  # the collective that last read send[b] (two steps ago): orders this stream behind it
        # the kernels write this step's results straight into send[b], laid out as the rank's slot (no pack / copy kernels)
        ws.set_packed_output(gather["send"][b].data_ptr(), gather["slot"], gather["max_s"])
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)
        ready = torch.cuda.Event()
        ready.record()
        with torch.cuda.stream(gather["comm"]):
            gather["comm"].wait_event(ready)
            gather["ev"][b][0].record()  # (the collective of this step, timed on the communication stream: per_rank.gather_ms)
            if to_all:
                _, gather["work"][b] = sharding.gather_packed(gather["send"][b], gather["out"][b], async_op=True)
            else:
                _, gather["work"][b] = sharding.gather_to_root(gather["send"][b], gather["out"][b], root=0, async_op=True)
            gather["ev"][b][1].record()

    def drain():
        if gather is not None:
            for w in gather["work"]:
                if w is not None:
                    w.wait()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    st = ws.stats()
    if st["error_flags"]:
        raise SystemExit(f"device error flags {st['error_flags']}")
    total_tokens = int(st["n_tokens"])
    if reorder_info is None:
        # the tokenizer renumbers the connection ids of its device image by the usage it measures on a sample of its text
        # (tok.calibrate above): include/vibrato_hip.h, vbt_tokenizer_connid_reorder_info.  VBT_CONNID_REORDER=0: off (A/B).
        ri = tok.connid_reorder_info()
        reorder_info = ("internal" if ri["epoch"] else "off") if world > 1 else {"mode": "internal" if ri["epoch"] else "off", **ri}
    gathered_ok = None
    per_rank = None
    if world > 1:
        # what every rank did, so that the first run on real GPUs explains itself: its own wall time per step (the job's is the
        # maximum), the kernels of its last step (hipEvents on the launch stream), the gather of its last step on the communication
        # stream (under RCCL an asynchronous collective is "done" for a non-root rank once its send is), and its share of the corpus
        lastb = (gather["k"] - 1) & 1
        torch.cuda.synchronize()
        g_ms = gather["ev"][lastb][0].elapsed_time(gather["ev"][lastb][1])
        mine = torch.tensor([elapsed / args.steps * 1e3, st["ms_tier0"], st["ms_tier12"], st["ms_pack"], g_ms, float(n), float(nbytes), float(total_tokens)],
                            dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        rows = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(rows, mine)
        per_rank = [{"rank": r, "ms_per_step": round(float(x[0]), 4), "gen_ms": round(float(x[1]), 4), "lattice_ms": round(float(x[2]), 4),
                     "pack_ms": round(float(x[3]), 4), "gather_ms": round(float(x[4]), 4), "sentences": int(x[5]), "bytes": int(x[6]), "tokens": int(x[7])}
                    for r, x in enumerate(rows)]
        slowest = max(p_["ms_per_step"] for p_ in per_rank)
        for p_ in per_rank:
            p_["idle_ms_per_step"] = round(slowest - p_["ms_per_step"], 4)  # what the rank waits for the slowest one at the closing barrier
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        # what the last gather delivered (rank 0 holds all shards): totals per rank from the slot headers
        if rank == 0:
            last = gather["out"][(gather["k"] - 1) & 1].view(world, -1)
            heads = last[:, :16].cpu().numpy()
            sent_by_rank = [int(heads[r, :8].view(np.int64)[0]) for r in range(world)]
            tok_by_rank = [int(heads[r, 8:12].view(np.uint32)[0]) for r in range(world)]
            gathered_ok = sum(sent_by_rank) == n_total and tok_by_rank[rank] == total_tokens
            total_tokens = sum(tok_by_rank)

    kernel_ms = st["ms_tier0"] + st["ms_tier12"] + st["ms_pack"]

    # two batches in flight (two workspaces on two streams, the same resident input): what a caller that streams batches gets when the
    # tail of one batch's kernels runs under the head of the next one's.  Outside the timed region and never `value`: a step of
    # `value` is one batch at a time on one stream, which is also what the roofline's kernel durations are measured on.
    two_in_flight = None
    if world == 1 and not args.no_suite:
        wsp = [tok.workspace(n, nbytes) for _ in range(2)]
        stp = [torch.cuda.Stream() for _ in range(2)]
        for i in range(4):
            wsp[i & 1].run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stp[i & 1].cuda_stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for i in range(args.steps):
            wsp[i & 1].run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stp[i & 1].cuda_stream)
        torch.cuda.synchronize()
        t2 = time.perf_counter() - t2
        stx = [w.stats() for w in wsp]
        two_in_flight = {"sentences_per_s": round(n * args.steps / t2, 1), "ms_per_batch": round(t2 / args.steps * 1e3, 4), "batches": args.steps,
                         "same_tokens_as_the_timed_steps": all(int(x["n_tokens"]) == total_tokens and not x["error_flags"] for x in stx)}
        del wsp, stp

    # host-to-host leg, outside the timed region and never `value`: what a caller of the batched entry
    # point pays -- the copy of the text into the batch, H2D, kernels, D2H of the token records into host memory
    h2h = None
    if not args.no_host_pipeline and world == 1:
        h2h = {"one_call": tok.host_pipeline_benchmark(text, offs, threads=1, rounds=1, repeats=6),
               "pipelined": tok.host_pipeline_benchmark(text, offs, threads=4, rounds=4),
               "pipelined_8_threads": tok.host_pipeline_benchmark(text, offs, threads=8, rounds=4)}
        # one tokenizer over several replicas of the image (vbt_tokenizer_new_multi; this box has one GPU, so device 0 is listed 4 and 8
        # times): what the host side of a multi-device call costs -- one host thread per device -- before 8 real devices are
        for k_dev in (4, 8):
            dvm = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
            if user_csv is not None:
                dvm.reset_user_lexicon_from_reader(user_csv)
            tokm = V.Tokenizer(dvm, devices=[local_rank] * k_dev).ignore_space(args.ignore_space).max_grouping_len(args.max_grouping_len)
            tokm.calibrate(text=text, offsets=offs)
            h2h[f"one_call_device_list_0x{k_dev}"] = tokm.host_pipeline_benchmark(text, offs, threads=1, rounds=1)
            del tokm, dvm
        # what one more device costs a call on the host side (one host thread per device; both lists are replicas on ONE physical GPU, so
        # the kernels' time is the same and the difference is launches, copies' fixed costs and thread hand-offs): (T(x8) - T(x4)) / 4
        h2h["fixed_ms_per_additional_device"] = round((h2h["one_call_device_list_0x8"]["ms_per_batch"] - h2h["one_call_device_list_0x4"]["ms_per_batch"]) / 4, 4)

    result = None
    if rank == 0:
        from oracle import oracle as ora  # checker + cpu_baseline leg only
        import tempfile
        native_dir = tempfile.mkdtemp(prefix="vbt_oracle_")
        native_so = None if args.no_cpu_baseline else ora.build_native(native_dir)  # -march=native for the timed CPU leg
        if native_so:
            ora.use_library(native_so)
        do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        if user_csv is not None:
            do.reset_user_lexicon(user_csv)
        to = ora.Tokenizer(do, args.ignore_space, args.max_grouping_len)
        w = to.new_worker()
        # parity gate on a sample (outside the timed region): bit-exact vs the oracle.  (With --reorder the product's
        # dictionary is a permutation of the oracle's: token records do not contain connection ids, so they still agree.)
        ns = min(n, 5000)
        sub_offs = offs[:ns + 1]
        sub_text = text[:int(sub_offs[-1])]
        got, got_off = tok.tokenize_batch(text=sub_text, offsets=sub_offs).tokens_in_order()
        exp, exp_off = w.tokenize_batch(sub_text, sub_offs)
        parity = bool(np.array_equal(got_off, exp_off) and all(np.array_equal(got[f], exp[f]) for f in V.TOKEN_DTYPE.names))
        if world > 1:  # and the gathered copy of rank 0's own shard
            n_s, n_t, goff, gcnt, gtok = sharding.unpack_results(gather["out"][(gather["k"] - 1) & 1].view(world, -1)[0], gather["max_s"])
            ordered, ends = sharding.tokens_in_sentence_order(goff[:ns], gcnt[:ns], gtok)
            parity = parity and bool(gathered_ok) and ordered.tobytes() == exp.tobytes()
        # algorithmic bytes of one launch on this GPU (oracle event counters over rank 0's batch; at N > 1 extrapolated from the
        # first 100k sentences of the shard: the counters are sums over sentences drawn from one law)
        n_cnt = min(n, 100000)
        w.reset_counters()
        w.tokenize_batch(text[:int(offs[n_cnt])], offs[:n_cnt + 1], counted=True, want_tokens=False)
        cnt = w.counters()
        scale = nbytes / max(int(offs[n_cnt]), 1)
        b_alg = algorithmic_bytes(cnt) * scale
        # per-kernel split of the SURVEY 8(d) terms: text, char, trie, posting, param and unknown-entry bytes are
        # read by gen_candidates; matrix cells and token records belong to lattice_lds (the dominant kernel)
        b_lat = (2 * cnt["n_pairs_dedup"] + 24 * cnt["n_tokens"]) * scale
        b_gen = b_alg - b_lat
        ms_gen, ms_lat = st["ms_tier0"], st["ms_tier12"]
        achieved = b_lat / (ms_lat * 1e-3) if ms_lat > 0 else 0.0
        workload = (f"{sd.name} ({sd.n_words} words, {sd.num_right}x{sd.num_left} i16 matrix = "
                    f"{sd.num_right * sd.num_left * 2 / 2**20:.1f} MiB), {n_total} sentences {args.law} chars, "
                    f"{total_bytes_all} bytes, seed {synth.SEED}")
        tr = committed_counters(workload) if world == 1 else None
        trk = (tr or {}).get("hbm_bytes_by_kernel", {})
        ins = (tr or {}).get("wave_insts_by_kernel", {})
        sqk = (tr or {}).get("sq_by_kernel", {})

        SWEEP = ("lattice_lds", "lattice_slim", "lattice_lean")  # the sweep's instances: general (escape tiers), slim (segment tier), lean -- launched side by side

        def merged(table, kernels):
            """Sum of a per-kernel table of the committed PMC summary over several kernels."""
            out = {}
            for kname in kernels:
                for key, v in (table.get(kname) or {}).items():
                    out[key] = out.get(key, 0) + v
            return out or None

        def issue(kernel, ms):
            """Instruction-issue view of a kernel from the committed PMC pass: the VALU-only busy time (SALU has its own pipe) at
            VALU_CYCLES issue cycles per wave64 instruction on 1024 SIMDs, and where the waves' cycles went (SQ counters)."""
            kernels = kernel if isinstance(kernel, tuple) else (kernel,)
            k = merged(ins, kernels)
            if not k or ms <= 0:
                return None
            floor_ms = k["valu"] * VALU_CYCLES / (N_SIMD * CLOCK_HZ) * 1e3
            out = {"wave_insts": k, "cycles_per_wave64_valu_assumed": VALU_CYCLES, "valu_busy_ms": round(floor_ms, 4),
                   "valu_busy_frac_of_kernel_time": round(floor_ms / ms, 4)}
            q = merged(sqk, kernels)
            if q and q.get("wave_cycles"):
                out["sq_wave_cycle_shares"] = {"wait_any": round(q.get("wait_any", 0) / q["wave_cycles"], 4),
                                               "wait_inst_any": round(q.get("wait_inst_any", 0) / q["wave_cycles"], 4),
                                               "active_inst_any": round(q.get("active_inst_any", 0) / q["wave_cycles"], 4),
                                               "active_inst_valu": round(q.get("active_inst_valu", 0) / q["wave_cycles"], 4)}
                if q.get("lds_idx_active"):
                    out["lds_bank_conflict_share_of_lds_cycles"] = round(q.get("lds_bank_conflict", 0) / q["lds_idx_active"], 4)
            return out

        roofline = {"bound": "hbm",
                    "kernel": "lattice_lean + lattice_slim: the sweep, two instances of one loop launched side by side (lean: whole sentences that "
                              "arrive with the generator's pass records, 7.5 KiB tier, 80 VGPRs; slim: everything else, 8 KiB segments, 5 waves per SIMD -- "
                              "batches of dense lattices: 8 / 10 KiB, by the density the tokenizer measured on the batches before); both span the "
                              "same interval of a step (rocprofv3: either kernel's duration = this span).  duration = hipEvents on the launch stream from "
                              "the fork behind the generators to the join of every sweep; the fallback launch and the packing behind the join are pack_ms",
                    "achieved": round(achieved / 1e9, 3), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_BPS, 6),
                    "traffic": (sum(trk.get(kname, 0) for kname in SWEEP) or None), "traffic_source": tr["source"] if tr else None,
                    "traffic_from_committed_profile": True,  # (not measured in this run: separate rocprofv3 --pmc passes of the same command, tools/profile_round.sh)
                    "algorithmic_bytes_per_launch": int(b_lat), "kernel_ms": round(ms_lat, 4),
                    "pairs_per_launch": {"dedup_cells_in_the_numerator": int(cnt["n_pairs_dedup"] * scale), "reference_pairs_gathered_by_the_kernel": int(cnt["n_pairs_ref"] * scale),
                                         "note": "the kernel gathers one cell per (candidate, predecessor) pair of the reference's search_min_node; the "
                                                 "numerator counts one per distinct (start node, left id, predecessor)"},
                    "issue": issue(SWEEP, ms_lat),
                    "gen_candidates": {"achieved": round(b_gen / (ms_gen * 1e-3) / 1e9, 3) if ms_gen > 0 else None,
                                       "frac": round(b_gen / (ms_gen * 1e-3) / HBM_PEAK_BPS, 6) if ms_gen > 0 else None,
                                       "algorithmic_bytes_per_launch": int(b_gen), "kernel_ms": round(ms_gen, 4),
                                       "traffic": (trk.get("gen_candidates", 0) + trk.get("gen_candidates_large", 0)) or None,
                                       "issue": issue("gen_candidates", ms_gen)},
                    "whole_path": {"achieved": round(b_alg / (kernel_ms * 1e-3) / 1e9, 3) if kernel_ms > 0 else None,
                                   "frac": round(b_alg / (kernel_ms * 1e-3) / HBM_PEAK_BPS, 6) if kernel_ms > 0 else None,
                                   "algorithmic_bytes_per_step": int(b_alg), "ms": round(kernel_ms, 4), "pack_ms": round(st["ms_pack"], 4),
                                   "traffic": tr["hbm_bytes_per_step"] if tr else None},
                    "connector_GBps": round(2 * cnt["n_pairs_dedup"] * scale / (kernel_ms * 1e-3) / 1e9, 3) if kernel_ms > 0 else 0.0,
                    "lattice_density": {"nodes_per_char": round(cnt["n_nodes"] / max(cnt["n_chars"], 1), 2),
                                        "dedup_pairs_per_char": round(cnt["n_pairs_dedup"] / max(cnt["n_chars"], 1), 1)},
                    "tiers": [st["n_tier0"], st["n_tier1"], st["n_tier2"]],
                    "measured_on": f"rank 0 ({n} sentences, {nbytes} bytes" + (f"; oracle counters of its first {n_cnt} sentences scaled by bytes" if n_cnt < n else "") + ")"}
        cpu = cpu_all = None
        if not args.no_cpu_baseline:
            # vibrato's CPU path restated (oracle/): warm-up, then the protocol of benchmark/src/main.rs:53-91
            # (3 trials x 3 runs, fastest and slowest run of a trial dropped) on a bounded sample of the same batch
            n_cpu = min(n, 100000 if world == 1 else 20000)
            c_offs = offs[:n_cpu + 1]
            c_text = text[:int(c_offs[-1])]
            c_bytes = int(c_offs[-1])
            w.tokenize_batch(sub_text, sub_offs, want_tokens=False)
            mean, lo_t, hi_t = cpu_protocol(lambda: w.tokenize_batch(c_text, c_offs, want_tokens=False))
            sample = (f"first {n_cpu} sentences of {'the batch' if world == 1 else 'rank 0 shard'} ({c_bytes} bytes); 3 trials x 3 runs, min and max run of a trial "
                      f"dropped (benchmark/src/main.rs:53-91); C restatement of vibrato Worker::tokenize (oracle/, gcc -O3 "
                      f"{'-march=native, compiled on this host' if native_so else '-march=x86-64-v3, prebuilt library'}), host has {os.cpu_count()} cores")
            cpu = {"value": round(n_cpu / mean, 1), "unit": "sentences/s", "cores": 1, "kind": "port", "sample": sample,
                   "range": [round(n_cpu / hi_t, 1), round(n_cpu / lo_t, 1)], "MB_per_s": round(c_bytes / mean / 1e6, 3),
                   "us_per_sentence": round(mean / n_cpu * 1e6, 3)}
        if cpu and world == 1:
            # all host cores: one oracle worker per core on contiguous chunks balanced by bytes (ctypes releases the GIL)
            from concurrent.futures import ThreadPoolExecutor
            cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            bnd = sharding.shard_bounds(c_offs, cores)
            workers = [to.new_worker() for _ in range(cores)]
            chunks = []
            for i in range(cores):
                o = c_offs[bnd[i]:bnd[i + 1] + 1]
                chunks.append((np.ascontiguousarray(c_text[int(o[0]):int(o[-1])]), o - o[0]))
            with ThreadPoolExecutor(cores) as ex:
                def run_all():
                    list(ex.map(lambda a: a[0].tokenize_batch(a[1][0], a[1][1], want_tokens=False), zip(workers, chunks)))
                run_all()
                mean_a, lo_a, hi_a = cpu_protocol(run_all)
            cpu_all = {"value": round(n_cpu / mean_a, 1), "unit": "sentences/s", "cores": cores, "kind": "port",
                       "sample": "same sample and protocol, one worker thread per host core on contiguous chunks",
                       "range": [round(n_cpu / hi_a, 1), round(n_cpu / lo_a, 1)], "MB_per_s": round(c_bytes / mean_a / 1e6, 3)}
            del workers, chunks

        # ---- per-sentence leg: the reference's only calling pattern (tokenize/src/main.rs:78-82, benchmark/src/main.rs:57-61) ----
        worker_loop = None
        if not args.no_worker_loop and world == 1:
            n_w = min(n, 10000)
            w_offs = offs[:n_w + 1]
            w_text = text[:int(w_offs[-1])]
            wk = tok.new_worker()
            wk.loop_benchmark(w_text[:int(w_offs[min(500, n_w)])], w_offs[:min(500, n_w) + 1])  # warm-up: buffers, code, clocks
            worker_loop = wk.loop_benchmark(w_text, w_offs, rounds=1)
            fast, slow = wk.path_stats()
            worker_loop.update({"single_launch_sentences": fast, "batch_pipeline_sentences": slow,
                                "mean_chars_per_sentence": round(cnt["n_chars"] / max(n_cnt, 1), 1),
                                "pattern": "reset_sentence -> tokenize -> num_tokens -> token(i) for every token, one Worker, one host thread; "
                                           "tokenize = a doorbell to the worker's resident one-wavefront kernel (no launch in steady state), text and token records through the worker's pinned host block",
                                "cpu_us_per_sentence_same_host": cpu["us_per_sentence"] if cpu else None})
            del wk

        # ---- the output stage: text in -> MeCab-format text out (tokenize/src/main.rs:78-95), never `value` ----
        fmt = None
        if not args.no_host_pipeline and world == 1:
            for _ in range(3):  # warm: pooled workspaces and pinned blocks exist, and the calls just before this leg (several host
                bt = tok.tokenize_batch(text=text, offsets=offs)  # threads at once) no longer count as "concurrent callers": a lone call is pipelined
                bt.format_bytes("mecab")
                del bt
            t_e = time.perf_counter()
            bt = tok.tokenize_batch(text=text, offsets=offs)
            t_tok = time.perf_counter() - t_e
            out_b, t_fmt = bt.format_bytes("mecab")
            t_e2e = time.perf_counter() - t_e
            fmt_calls = [t_fmt] + [bt.format_bytes("mecab")[1] for _ in range(4)]  # (the same batch again: output buffer and threads are reused)
            t_fmt = sorted(fmt_calls)[len(fmt_calls) // 2]
            n_f = min(n, 20000)
            f_offs = offs[:n_f + 1]
            f_text = text[:int(f_offs[-1])]
            need, obuf = w.tokenize_format_batch(f_text, f_offs, "mecab")  # sizes the buffer (and warms the oracle)
            t_c = time.perf_counter()
            got_n, obuf = w.tokenize_format_batch(f_text, f_offs, "mecab", out=obuf)
            t_cpu = time.perf_counter() - t_c
            same = bool(out_b[:got_n] == obuf[:got_n].tobytes())  # the oracle's bytes for the first n_f sentences are a prefix of the product's
            parity = parity and same
            piped = tok.text_pipeline_benchmark(text, offs, batches=8, mode="mecab")
            fmt = {"mode": "mecab", "output_bytes": len(out_b), "format_ms": round(t_fmt * 1e3, 3), "format_ms_five_calls": [round(x * 1e3, 3) for x in fmt_calls], "format_MB_per_s": round(len(out_b) / t_fmt / 1e6, 1),
                   "tokenize_batch_ms": round(t_tok * 1e3, 3), "text_in_to_text_out_ms": round((t_tok + t_fmt) * 1e3, 3),
                   "text_in_to_text_out_sentences_per_s": piped["sentences_per_s"],
                   "text_in_to_text_out_pipelined": piped,
                   "text_in_to_text_out_one_unpipelined_batch_sentences_per_s": round(n / (t_tok + t_fmt), 1),
                   "python_wall_incl_copy_out_ms": round(t_e2e * 1e3, 3),
                   "what": "vbt_tokenize_batch (host text in, token records in pinned host memory) then vbt_batch_format(MECAB): sizes and "
                           "rendering on one set of threads over chunks of sentences, feature strings from a flat table; text_in_to_text_out = a stream "
                           "of batches, the formatter of batch k under the kernels of batch k + 1 (how vibrato_amd.cli runs); the unpipelined figure is the "
                           "two library calls back to back; python_wall adds the copy of the output bytes out of the library into a python object",
                   "cpu_port_1thread": {"sample_sentences": n_f, "sentences_per_s": round(n_f / t_cpu, 1), "output_MB_per_s": round(got_n / t_cpu / 1e6, 2),
                                        "what": "oracle: tokenize + print per line, the loop of tokenize/src/main.rs:78-95, one thread"},
                   "bytes_identical_to_oracle_sample": same}
            del bt, out_b, obuf

        # ---- the other single-GPU BASELINE workloads through the same timed loop (never `value`) ----
        suite = None
        if not args.no_suite and world == 1 and args.dict == "unidic" and not args.ignore_space and not args.user_lexicon and not args.reorder:
            suite = {}
            legs = [("cfg5", "BASELINE config 5: syn-unidic + user.csv (1000 compounds), -S -M 24, mixed lengths, injected spaces", sd,
                     dict(user=1000, ignore_space=True, mgl=24, law="mixed", space_p=0.1)),
                    ("dense", "dense lexicon law (syn-unidic-dense: the upper end of SURVEY 8a's nodes / pairs per character)", None,
                     dict(user=0, ignore_space=False, mgl=0, law="lognormal_40", space_p=0.0))]
            for key, what, sdx, o in legs:
                t_leg = time.time()
                sdx = sdx or synth.SynthDict("unidic-dense")
                dvx = V.SystemDictionaryBuilder.from_readers_binmatrix(sdx.lex, sdx.matrix, sdx.num_right, sdx.num_left, sdx.char_def, sdx.unk)
                dox = ora.Dictionary.from_sources_binmatrix(sdx.lex, sdx.matrix, sdx.num_right, sdx.num_left, sdx.char_def, sdx.unk)
                if o["user"]:
                    ucsv = sdx.user_csv(o["user"])
                    dvx.reset_user_lexicon_from_reader(ucsv)
                    dox.reset_user_lexicon(ucsv)
                tokx = V.Tokenizer(dvx, device=local_rank).ignore_space(o["ignore_space"]).max_grouping_len(o["mgl"])
                tx, ox = sdx.sentences(100000, o["law"], space_p=o["space_p"], seed=synth.SEED)
                dt_x, stx = timed_leg(tokx, tx, ox, steps=10, warmup=2, torch=torch)
                wx = ora.Tokenizer(dox, o["ignore_space"], o["mgl"]).new_worker()
                nsx = 3000
                g_t, g_o = tokx.tokenize_batch(text=tx[:int(ox[nsx])], offsets=ox[:nsx + 1]).tokens_in_order()
                wx.reset_counters()
                e_t, e_o = wx.tokenize_batch(tx[:int(ox[nsx])], ox[:nsx + 1], counted=True)
                cx = wx.counters()
                okx = bool(np.array_equal(g_o, e_o) and all(np.array_equal(g_t[f], e_t[f]) for f in V.TOKEN_DTYPE.names))
                parity = parity and okx
                suite[key] = {"workload": what, "sentences": 100000, "bytes": int(len(tx)), "steps": 10, "warmup": 2,
                              "value": round(100000 * 10 / dt_x, 1), "unit": "sentences/s", "ms_per_step": round(dt_x / 10 * 1e3, 4),
                              "gen_ms": round(stx["ms_tier0"], 4), "lattice_ms": round(stx["ms_tier12"], 4), "pack_ms": round(stx["ms_pack"], 4),
                              "tiers": [stx["n_tier0"], stx["n_tier1"], stx["n_tier2"]], "tokens_per_step": int(stx["n_tokens"]),
                              "error_flags": int(stx["error_flags"]),
                              "lattice_density_sample": {"nodes_per_char": round(cx["n_nodes"] / max(cx["n_chars"], 1), 2),
                                                         "dedup_pairs_per_char": round(cx["n_pairs_dedup"] / max(cx["n_chars"], 1), 1)},
                              "connection_ids_reordered": "internal" if tokx.connid_reorder_info()["epoch"] else "off",
                              "parity_vs_oracle_sample": okx, "parity_sample_sentences": nsx, "leg_s": round(time.time() - t_leg, 1)}
                del tokx, dvx, dox, wx, tx, ox

        value = n_total * args.steps / elapsed
        par = (f"dp{world}: {n_total}-sentence corpus in {world} contiguous shards balanced by bytes, one process per GPU, no data-path "
               f"collective; final device-resident gather of the packed results ({'to every rank: all_gather_into_tensor' if args.gather == 'all' else 'to rank 0: grouped send/recv'}, {backend}), overlapped with the next step"
               if world > 1 else "dp1")
        result = {
            "metric": "sentences/sec", "value": round(value, 1), "unit": "sentences/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if world > 1 else "weak", "vs_baseline": None,
            "dtype": "i32 costs / i16 matrix / u32 ids",
            "data": "synthetic", "input_MB_per_s": round(total_bytes_all * args.steps / elapsed / 1e6, 2),
            # what a drop-in's single-threaded caller sees (tokenize/src/main.rs:76-95): ONE vbt_tokenize_batch call, one host thread,
            # host text in -> token records in host memory (H2D, kernels and D2H pipelined in chunks inside the call); never `value`
            "value_host_to_host": (h2h["one_call"]["sentences_per_s"] if h2h else None),
            "value_text_in_to_mecab_text_out": (fmt["text_in_to_text_out_sentences_per_s"] if fmt else None),
            "config": {"workload": workload,
                       "value_is": "device-resident: text and offsets in HBM when the timed region starts, token records left in HBM (the PCIe-inclusive "
                                   "rates are value_host_to_host / host_to_host / format)",
                       "baseline_config": 4 if world > 1 else (5 if args.ignore_space and args.user_lexicon else 3 if args.dict == "unidic" else None),
                       "ignore_space": args.ignore_space, "max_grouping_len": args.max_grouping_len, "user_lexicon_words": args.user_lexicon,
                       "connection_ids_reordered": reorder_info, "parallelism": par},
            "parity_vs_oracle_sample": parity, "tokens_per_step": total_tokens, "lattice_density_candidates_per_byte": round(tok.lattice_density(), 3),
            "parity_gate": "this run compared a 5 000-sentence sample of the timed batch (3 000 per suite leg) and the formatter's bytes with the oracle; the "
                           "full-size comparisons (configs 2, 3, 5 and the dense law: every record of 100 000 sentences) are tests/test_gpu_parity.py",
            "per_rank": per_rank,
            "gather": ({"bytes_per_rank_slot": gather["slot"], "collective": "all_gather_into_tensor" if args.gather == "all" else "gather to rank 0 (grouped send/recv)", "device_resident": True,
                        "delivered_all_shards": bool(gathered_ok)} if world > 1 else None),
            "roofline": roofline, "cpu_baseline": cpu, "cpu_baseline_all_cores": cpu_all,
            "speedup_vs_cpu_1thread": round(value / cpu["value"], 1) if cpu else None,
            "suite": suite, "worker_loop": worker_loop, "two_batches_in_flight": two_in_flight,
            "host_to_host": h2h, "format": fmt,
            "setup_s": round(t_setup, 1),
        }
        print(json.dumps(result, ensure_ascii=False), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result and not result["parity_vs_oracle_sample"]:
        raise SystemExit("PARITY FAILURE vs oracle")


if __name__ == "__main__":
    main()
