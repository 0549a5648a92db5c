import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
if os.environ.get("WITH_TORCH") == "1":
    import torch
    torch.cuda.init(); torch.zeros(1).cuda()
exec(open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools/h2h_bench.py")).read())
