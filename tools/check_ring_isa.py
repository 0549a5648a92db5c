#!/usr/bin/env python3
"""Static check of the sweep kernel's gather ring in the compiled ISA (CPU only; also run by the CPU test suite).

lattice_sentence issues its connection-cost gathers as inline assembly (lattice.hip: `issue_gathers`) and waits for them with a
hand-placed `s_waitcnt vmcnt(N)`, so the compiler does not know that the destination registers are written asynchronously.
That is only sound if, between a gather and the instruction that consumes its result, nothing touches the destination
register: no copy (register allocation splitting a live range), no spill, no reuse.  This script proves exactly that on
the assembly hipcc emits:

  for every `buffer_load_{sshort,dword}` / `global_load_dwordx2` G inside an inline-asm block of a kernel and every path through the control-flow
  graph behind it, the first instruction that mentions G's destination register lies behind an inline `s_waitcnt vmcnt(N)`
  at which G has provably landed (at least N inline loads were issued behind G: loads return in order), or is the next
  load into the same ring slot;

  and the same for every inline `ds_read_*` (the loop's LDS reads and, in the default build, the fetch of a pass record out of
  LDS): the first mention of its destination lies behind an inline `s_waitcnt lgkmcnt(0)`, or is the next read into the same
  register.  (Inside the loop nothing but LDS operations counts in lgkmcnt, and those complete in order.)

usage: python tools/check_ring_isa.py [lattice.s]      (without an argument: compiles lattice.hip to assembly first)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = ("lattice_lds", "lattice_lean", "lattice_slim", "gen_sweep", "tokenize_serve")


def mentions(text, reg):
    """does the instruction text mention VGPR number `reg` (alone or inside a range v[a:b])?"""
    for m in re.finditer(r"\bv(\d+)\b", text):
        if int(m.group(1)) == reg:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


def kernel_bodies(asm):
    lines = asm.split("\n")
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):", lines[i])
        if m and any(k in m.group(1) for k in KERNELS):
            j = i
            while j < len(lines) and ".end_amdhsa_kernel" not in lines[j] and not lines[j].startswith(".Lfunc_end"):
                j += 1
            yield m.group(1), lines[i + 1:j]
            i = j
        i += 1


def check_kernel(name, body):
    # instruction list (text, inside an inline-asm block?) and label positions
    insts, labels, in_asm = [], {}, False
    for l in body:
        t = l.strip()
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            labels[m.group(1)] = len(insts)
            continue
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        insts.append((t.split(";")[0].strip(), in_asm))

    def successors_of(i):
        t = insts[i][0]
        op = t.split()[0]
        if op == "s_endpgm":
            return ()
        if op == "s_branch":
            return (labels[t.split()[1]],)
        if op.startswith("s_cbranch"):
            return (labels[t.split()[1]], i + 1)
        return (i + 1,) if i + 1 < len(insts) else ()

    # the loads of the ring: the gathers (one register) and the broadcast fetch of an 8-byte pass record (two)
    load_re = re.compile(r"^(?:buffer_load_(?:sshort|dword)|global_load_dwordx[23])\s+v\[?(\d+)(?::(\d+)\])?,")

    def dests(text):
        m = load_re.match(text)
        if not m:
            return ()
        lo = int(m.group(1))
        return tuple(range(lo, int(m.group(2)) + 1)) if m.group(2) else (lo,)

    # every inline-asm block that holds ring loads (the sweep loop) must BEGIN by draining what the compiler has in flight: a
    # load nobody went on to use is never waited for by compiled code and could land in a register the block owns
    entry_errors = []
    blk = None
    for l in body:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            blk = []
        elif t.startswith(";;#ASMEND"):
            if blk and any(load_re.match(x) for x in blk) and len(blk) > 20:
                if not re.match(r"^s_waitcnt\s+vmcnt\(0\)\s+lgkmcnt\(0\)", blk[0]):
                    entry_errors.append(f"an inline-asm block with ring loads begins with `{blk[0]}`, not with s_waitcnt vmcnt(0) lgkmcnt(0)")
            blk = None
        elif blk is not None and t and not t.startswith(";") and not t.startswith(".") and not re.match(r"^\.?\w+:", t):
            blk.append(t.split(";")[0].strip())

    lds_re = re.compile(r"^ds_read\w*\s+v\[?(\d+)(?::(\d+)\])?,")

    def lds_dests(text):
        m = lds_re.match(text)
        if not m:
            return ()
        lo = int(m.group(1))
        return tuple(range(lo, int(m.group(2)) + 1)) if m.group(2) else (lo,)

    # per instruction, once: successors, VGPRs mentioned, registers loaded by an inline load, N of an inline s_waitcnt vmcnt(N);
    # the same for inline LDS reads and s_waitcnt lgkmcnt(0)
    succ = [successors_of(i) for i in range(len(insts))]
    used, loaded, waits, lds_loaded, lds_wait = [], [], [], [], []
    for t, a in insts:
        regs = set(int(m.group(1)) for m in re.finditer(r"\bv(\d+)\b", t))
        for m in re.finditer(r"\bv\[(\d+):(\d+)\]", t):
            regs.update(range(int(m.group(1)), int(m.group(2)) + 1))
        used.append(regs)
        loaded.append(dests(t) if a else ())
        mw = re.match(r"^s_waitcnt\s+vmcnt\((\d+)\)", t) if a else None
        waits.append(int(mw.group(1)) if mw else None)
        lds_loaded.append(lds_dests(t) if a else ())
        lds_wait.append(bool(a and re.match(r"^s_waitcnt\s+(?:vmcnt\(\d+\)\s+)?lgkmcnt\(0\)", t)))
    cap = max([w for w in waits if w is not None] + [1])  # (more loads behind G than any wait asks for: no need to count on)

    errors, n_loads = list(entry_errors), 0
    targets = [(idx, reg) for idx in range(len(insts)) for reg in loaded[idx]]
    for idx, reg in targets:
        n_loads += 1
        # Along every path from behind the load G: `after` counts the inline loads issued behind G.  An inline
        # `s_waitcnt vmcnt(N)` retires all but the N most recent loads (loads return in order), so G has landed there iff
        # after >= N.  The first instruction that mentions G's register must come behind such a wait -- or be another inline
        # load into the same register (an unconsumed slot of a pass without that unit: in-order write after write).
        stack = [(j, 0, False) for j in succ[idx]]
        seen = set()
        verdict = None
        while stack and not verdict:
            st = stack.pop()
            if st in seen:
                continue
            seen.add(st)
            j, after, landed = st
            if waits[j] is not None and after >= waits[j]:
                landed, after = True, cap
            if reg in used[j]:
                if reg in loaded[j]:
                    continue  # overwritten by the next load into this ring slot
                if not landed:
                    verdict = f"v{reg}: touched by `{insts[j][0]}` (#{j}) while the load at #{idx} may still be in flight ({after} loads behind it)"
                continue
            if loaded[j]:
                after = min(after + 1, cap)
            stack.extend((k, after, landed) for k in succ[j])
        if verdict:
            errors.append(verdict)
    # the inline LDS reads: landed at the first inline s_waitcnt lgkmcnt(0) behind them
    for idx in range(len(insts)):
        for reg in lds_loaded[idx]:
            n_loads += 1
            stack = list(succ[idx])
            seen = set()
            verdict = None
            while stack and not verdict:
                j = stack.pop()
                if j in seen:
                    continue
                seen.add(j)
                if lds_wait[j]:
                    continue  # landed on this path
                if reg in used[j]:
                    if reg in lds_loaded[j]:
                        continue  # the next read into the same register (in order)
                    verdict = f"v{reg}: touched by `{insts[j][0]}` (#{j}) before an s_waitcnt lgkmcnt(0) behind the LDS read at #{idx}"
                    continue
                stack.extend(succ[j])
            if verdict:
                errors.append(verdict)
    return n_loads, errors


def main():
    if len(sys.argv) > 1:
        asm = open(sys.argv[1]).read()
    else:
        with tempfile.TemporaryDirectory() as d:
            out = os.path.join(d, "engine.s")
            src = os.path.join(ROOT, "vibrato_amd", "csrc", "lattice.hip")
            subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-S",
                                   "--cuda-device-only", "-o", out, "-x", "hip", src], stderr=subprocess.DEVNULL)
            asm = open(out).read()
    bad = 0
    total = 0
    for name, body in kernel_bodies(asm):
        n, errs = check_kernel(name, body)
        total += n
        print(f"{name[:90]}: {n} inline loads / LDS reads, {len(errs)} violations")
        for e in errs[:20]:
            print("   ", e)
        bad += len(errs)
    if total == 0:
        print("no inline-asm gathers found: the check does not apply to this build")
        return 2
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
