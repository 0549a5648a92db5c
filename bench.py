#!/usr/bin/env python3
"""Headline benchmark: batched tokenization throughput (sentences/s) on MI355X.

A "step" is one pass of the tokenize() hot path over one device-resident batch of
synthetic Japanese sentences (BASELINE.json config: unidic-cwj-3.1.1-shaped dictionary with
the 458.6 MiB connection matrix, 100k sentences per GPU).  N>1: one process per GPU, every
rank tokenizes its own batch (weak scaling, sentences are independent); the only collective
is the final RCCL gather of per-rank totals.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# one HIP stream per LDS tier of the lattice kernel: let the runtime use more than 4 hardware queues
# (read when the HIP runtime initialises, i.e. before the first torch.cuda call)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "10")

HBM_PEAK_BPS = 8.0e12  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def algorithmic_bytes(c):
    """B_alg from the oracle's event counters."""
    return (c["n_bytes"] + 4 * c["n_chars"] + 12 * c["n_trie_steps"] + 4 * (c["n_trie_hits"] + c["n_lex_matches"])
            + 8 * c["n_lex_matches"] + 8 * c["n_unk_nodes"] + 2 * c["n_pairs_dedup"] + 24 * c["n_tokens"])


def measured_traffic(workload):
    """HBM bytes per step from the newest committed PMC summary of the same workload
    (profiles/*_traffic.json, written by tools/summarize_profile.py from separate --pmc passes)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--dict", default="unidic", choices=["tiny", "small", "ipadic", "unidic"])
    ap.add_argument("--sentences", type=int, default=100000, help="sentences per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ignore-space", action="store_true")
    ap.add_argument("--max-grouping-len", type=int, default=0)
    ap.add_argument("--user-lexicon", type=int, default=0, help="attach a synthetic user.csv of N compounds (BASELINE config 5)")
    ap.add_argument("--law", default="lognormal_40", choices=["uniform_5_20", "lognormal_40", "mixed"])
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import vibrato_amd as V
    from tools import synth

    t_setup = time.time()
    sd = synth.SynthDict(args.dict)
    dv = V.SystemDictionaryBuilder.from_readers_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
    user_csv = sd.user_csv(args.user_lexicon) if args.user_lexicon else None
    if user_csv is not None:
        dv.reset_user_lexicon_from_reader(user_csv)
    tok = V.Tokenizer(dv, device=local_rank).ignore_space(args.ignore_space).max_grouping_len(args.max_grouping_len)
    n = args.sentences
    space_p = 0.1 if args.ignore_space else 0.0
    text, offs = sd.sentences(n, args.law, space_p=space_p, seed=synth.SEED + rank)
    nbytes = int(len(text))
    d_text = torch.from_numpy(text).cuda()
    d_offs = torch.from_numpy(offs.astype(np.int64)).cuda()
    ws = tok.workspace(n, nbytes)
    ws.set_timing(True)
    stream = torch.cuda.current_stream().cuda_stream
    t_setup = time.time() - t_setup

    def step():
        ws.run(d_text.data_ptr(), d_offs.data_ptr(), n, nbytes, stream)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    kernel_ms = []
    for _ in range(args.steps):
        step()
    totals = torch.zeros(world, 2, dtype=torch.int64, device="cuda")
    if world > 1:  # final gather of per-rank (sentences, tokens) over RCCL/xGMI
        torch.cuda.synchronize()
        mine = torch.tensor([n, ws.stats()["n_tokens"]], dtype=torch.int64, device="cuda")
        dist.all_gather_into_tensor(totals.view(-1), mine)
    fence()
    elapsed = time.perf_counter() - t0
    st = ws.stats()
    if st["error_flags"]:
        raise SystemExit(f"device error flags {st['error_flags']}")
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        total_sentences = int(totals[:, 0].sum().item())
        bytes_all = torch.tensor([nbytes], dtype=torch.int64, device="cuda")
        dist.all_reduce(bytes_all)
        total_bytes = int(bytes_all.item())
    else:
        total_sentences, total_bytes = n, nbytes

    # dominant-kernel duration: hipEvents on the launch stream (last timed step)
    kernel_ms = st["ms_tier0"] + st["ms_tier12"]

    result = None
    if rank == 0:
        from oracle import oracle as ora  # checker + cpu_baseline leg only
        do = ora.Dictionary.from_sources_binmatrix(sd.lex, sd.matrix, sd.num_right, sd.num_left, sd.char_def, sd.unk)
        if user_csv is not None:
            do.reset_user_lexicon(user_csv)
        to = ora.Tokenizer(do, args.ignore_space, args.max_grouping_len)
        w = to.new_worker()
        # parity gate on a sample (outside the timed region): bit-exact vs the oracle
        ns = min(n, 5000)
        sub_offs = offs[:ns + 1]
        sub_text = text[:int(sub_offs[-1])]
        got, got_off = tok.tokenize_batch(text=sub_text, offsets=sub_offs).tokens_in_order()
        exp, exp_off = w.tokenize_batch(sub_text, sub_offs)
        parity = bool(np.array_equal(got_off, exp_off) and all(np.array_equal(got[f], exp[f]) for f in V.TOKEN_DTYPE.names))
        # algorithmic bytes of one launch (oracle event counters over the whole batch)
        w.reset_counters()
        w.tokenize_batch(text, offs, counted=True, want_tokens=False)
        cnt = w.counters()
        b_alg = algorithmic_bytes(cnt)
        # per-kernel split of the SURVEY 8(d) terms: text, char, trie, posting, param and unknown-entry bytes are
        # read by gen_candidates; matrix cells and token records belong to lattice_lds (the dominant kernel)
        b_lat = 2 * cnt["n_pairs_dedup"] + 24 * cnt["n_tokens"]
        b_gen = b_alg - b_lat
        ms_gen, ms_lat = st["ms_tier0"], st["ms_tier12"]
        achieved = b_lat / (ms_lat * 1e-3) if ms_lat > 0 else 0.0
        workload = (f"{sd.name} ({sd.n_words} words, {sd.num_right}x{sd.num_left} i16 matrix = "
                    f"{sd.num_right * sd.num_left * 2 / 2**20:.1f} MiB), {n} sentences/GPU {args.law} chars, "
                    f"{nbytes} bytes/GPU, seed {synth.SEED}")
        tr = measured_traffic(workload)
        trk = (tr or {}).get("hbm_bytes_by_kernel", {})
        roofline = {"bound": "hbm",
                    "kernel": "lattice_lds (the per-tier launches of one step run concurrently on side streams; duration = "
                              "hipEvents on the launch stream from the fork to the join)",
                    "achieved": round(achieved / 1e9, 3), "peak": HBM_PEAK_BPS / 1e9, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_BPS, 6),
                    "traffic": trk.get("lattice_lds"), "traffic_source": tr["source"] if tr else None,
                    "algorithmic_bytes_per_launch": int(b_lat), "kernel_ms": round(ms_lat, 4),
                    "gen_candidates": {"achieved": round(b_gen / (ms_gen * 1e-3) / 1e9, 3) if ms_gen > 0 else None,
                                       "frac": round(b_gen / (ms_gen * 1e-3) / HBM_PEAK_BPS, 6) if ms_gen > 0 else None,
                                       "algorithmic_bytes_per_launch": int(b_gen), "kernel_ms": round(ms_gen, 4),
                                       "traffic": (trk.get("gen_candidates", 0) + trk.get("gen_candidates_large", 0)) or None},
                    "whole_path": {"achieved": round(b_alg / (kernel_ms * 1e-3) / 1e9, 3) if kernel_ms > 0 else None,
                                   "frac": round(b_alg / (kernel_ms * 1e-3) / HBM_PEAK_BPS, 6) if kernel_ms > 0 else None,
                                   "algorithmic_bytes_per_step": int(b_alg), "ms": round(kernel_ms, 4),
                                   "traffic": tr["hbm_bytes_per_step"] if tr else None},
                    "connector_GBps": round(2 * cnt["n_pairs_dedup"] / (kernel_ms * 1e-3) / 1e9, 3) if kernel_ms > 0 else 0.0,
                    "tiers": [st["n_tier0"], st["n_tier1"], st["n_tier2"]]}
        cpu = None
        if not args.no_cpu_baseline:
            # vibrato's CPU path restated (oracle, 1 thread): warm-up + 3 passes over the same batch
            w.tokenize_batch(sub_text, sub_offs, want_tokens=False)
            ts = []
            for _ in range(3):
                t = time.perf_counter()
                w.tokenize_batch(text, offs, want_tokens=False)
                ts.append(time.perf_counter() - t)
            cpu_t = sorted(ts)[1]
            cpu = {"value": round(n / cpu_t, 1), "unit": "sentences/s", "cores": 1, "kind": "port",
                   "sample": f"rank-0 batch ({n} sentences, {nbytes} bytes), median of 3 passes after warm-up, "
                             f"C restatement of vibrato Worker::tokenize (oracle/), host has {os.cpu_count()} cores",
                   "MB_per_s": round(nbytes / cpu_t / 1e6, 3)}
        value = total_sentences * args.steps / elapsed
        result = {
            "metric": "sentences/sec", "value": round(value, 1), "unit": "sentences/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "i32 costs / i16 matrix / u32 ids",
            "data": "synthetic", "input_MB_per_s": round(total_bytes * args.steps / elapsed / 1e6, 2),
            "config": {"workload": workload,
                       "ignore_space": args.ignore_space, "max_grouping_len": args.max_grouping_len, "user_lexicon_words": args.user_lexicon,
                       "parallelism": f"dp{world} (independent sentence shards, final RCCL gather of totals)"},
            "parity_vs_oracle_sample": parity, "tokens_per_step": int(st["n_tokens"]) if world == 1 else int(totals[:, 1].sum().item()),
            "roofline": roofline, "cpu_baseline": cpu,
            "speedup_vs_cpu_1thread": round(value / cpu["value"], 1) if cpu else None,
            "setup_s": round(t_setup, 1),
        }
        print(json.dumps(result, ensure_ascii=False))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and result and not result["parity_vs_oracle_sample"]:
        raise SystemExit("PARITY FAILURE vs oracle")


if __name__ == "__main__":
    main()
