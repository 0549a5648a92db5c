#!/bin/bash
# round 4: canary, the parity suite, the default bench line (all legs)
ulimit -c 0
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python tools/dbg/first_light.py 2>&1 | grep "mismatching" | cut -c1-150
bash tools/ab_variants.sh "base" --no-host-pipeline --no-worker-loop 2>&1 | grep -v amdgpu | tee gpurun_out/r04_call6_ab.txt
grep -q "True" gpurun_out/r04_call6_ab.txt || { echo "canary failed: skipping the rest"; exit 1; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r04_call6_pytest.txt
timeout 600 python bench.py 2>gpurun_out/r04_call6_bench.err | tail -1 > gpurun_out/r04_call6_bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_call6_bench.json"))
print("value",d["value"],"ms",d["ms_per_step"],"parity",d["parity_vs_oracle_sample"])
print("roofline frac",d["roofline"]["frac"],"lat ms",d["roofline"]["kernel_ms"],"gen ms",d["roofline"]["gen_candidates"]["kernel_ms"])
print("cpu",d["cpu_baseline"]["value"],"x",d["speedup_vs_cpu_1thread"])
print("suite",{k:(v["value"],v["lattice_ms"],v["tiers"]) for k,v in d["suite"].items()})
print("worker",d["worker_loop"])
print("h2h",d["host_to_host"])
print("format",d["format"])
PY
