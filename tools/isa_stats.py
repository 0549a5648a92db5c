#!/usr/bin/env python3
"""Static instruction statistics of one kernel from hipcc's --save-temps assembly: per kernel the
instruction mix, and per natural loop (backward branch) its body size by class.  CPU-only feedback
for the issue-bound kernels: python tools/isa_stats.py file.s lattice_lds
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_"):
        return "VALU"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "SMEM"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"):
        return "WAIT"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_setpc") or op.startswith("s_swappc"):
        return "BRANCH"
    if op.startswith("s_"):
        return "SALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("scratch_"):
        return "VMEM"
    return "OTHER"


def main():
    path, pat = sys.argv[1], sys.argv[2]
    min_body = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w*:", l) and pat in l:
            start = i
            name = l[:-1]
            break
    if start is None:
        sys.exit("kernel not found")
    end = next(i for i in range(start, len(lines)) if ".end_amdhsa_kernel" in lines[i] or lines[i].startswith(".Lfunc_end"))
    insts = []  # (line no, op, text)
    labels = {}
    for i in range(start + 1, end):
        l = lines[i]
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        t = l.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.startswith("//"):
            continue
        op = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            insts.append((i + 1, op, t))
    mix = Counter(classify(op) for _, op, _ in insts)
    print(f"{name}: {len(insts)} instructions  " + "  ".join(f"{k}={v}" for k, v in sorted(mix.items())))
    spills = sum(1 for _, op, t in insts if op.startswith("scratch_"))
    print(f"  scratch (spill) instructions: {spills}")
    # the compiler's own metadata for this kernel (.amdgpu_metadata at the end of the file): registers and spill counts
    meta = re.search(r"\.name:\s+" + re.escape(name.split(":")[0].strip()) + r"\n(.*?)(?=\n  - \.|\Z)", "\n".join(lines), re.S)
    if meta:
        g = lambda k: (re.search(r"\." + k + r":\s+(\d+)", meta.group(1)) or [None, "?"])[1]
        print(f"  compiler metadata: vgpr_count {g('vgpr_count')}  vgpr_spill_count {g('vgpr_spill_count')}  sgpr_count {g('sgpr_count')}  "
              f"sgpr_spill_count {g('sgpr_spill_count')}  private_segment_fixed_size {g('private_segment_fixed_size')} bytes")
    # where the scratch instructions sit relative to the inline-assembly sweep loop (the compiler cannot spill inside an asm block)
    asm_lines = [i for i, l in enumerate(lines[start:end], start) if "ASMSTART" in l or "ASMEND" in l]
    big = None
    for a, b in zip(asm_lines[0::2], asm_lines[1::2]):
        if big is None or b - a > big[1] - big[0]:
            big = (a, b)
    if big:
        inside = sum(1 for ln_, op, _ in insts if op.startswith("scratch_") and big[0] < ln_ - 1 < big[1])
        wl = sum(1 for ln_, op, _ in insts if op.startswith("v_writelane") and big[0] < ln_ - 1 < big[1])
        print(f"  largest inline-asm block (the sweep loop): lines {big[0] + 1}-{big[1] + 1}; scratch instructions inside it: {inside}; v_writelane (SGPR spills) inside it: {wl}")
    loops = []
    for idx, (ln, op, t) in enumerate(insts):
        if op.startswith("s_cbranch") or op == "s_branch":
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= idx:
                loops.append((labels[tgt], idx, tgt))
    loops.sort(key=lambda x: (x[0], -x[1]))
    print(f"  loops (backward branches), body >= {min_body} instructions:")
    for a, b, tgt in loops:
        if b - a + 1 < min_body:
            continue
        body = insts[a:b + 1]
        m = Counter(classify(op) for _, op, _ in body)
        dpp = sum(1 for _, op, t in body if "dpp" in t or "permlane" in op)
        rfl = sum(1 for _, op, _ in body if op.startswith("v_readfirstlane") or op.startswith("v_readlane"))
        print(f"    {tgt:>14} lines {body[0][0]}-{body[-1][0]}: {len(body):5d}  " + "  ".join(f"{k}={v}" for k, v in sorted(m.items()))
              + f"  (dpp/permlane={dpp} readlane={rfl})")


if __name__ == "__main__":
    main()
