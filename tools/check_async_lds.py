#!/usr/bin/env python3
"""Static check of the hand-ordered asynchronous LDS reads in the Hessian kernels.

The kernels issue `ds_read_b64_tr_b16` through inline asm and order them with hand-placed
`s_waitcnt lgkmcnt(N)` (hipcc would otherwise drain the in-flight LDS-DMA with vmcnt(0), see
gptq_amd/csrc/hessian.hip).  The compiler does not know those reads are asynchronous, so a register
copy or any other use it places between a read and the wait that covers it would consume stale data.
This walks the control-flow graph of the generated ISA and reports every instruction that touches the
destination of an inline-asm read before a wait has retired that read.

    python3 tools/check_async_lds.py [file.s ...]      (no file: compiles hessian.hip for gfx950)
Exit status 1 on a violation."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse_kernels(path):
    """{kernel: [(label or None, text, in_asm)]} for every function of the file."""
    kernels, cur, name, in_asm = {}, None, None, False
    for line in open(path):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            kernels[name] = cur
            continue
        if cur is None:
            continue
        if s.startswith(".Lfunc_end"):
            cur = None
            continue
        if "#ASMSTART" in s:
            in_asm = True
            continue
        if "#ASMEND" in s:
            in_asm = False
            continue
        if not s or s.startswith(";") or s.startswith(".") and not s.startswith(".LBB"):
            continue
        lm = re.match(r"^(\.LBB\w+):", s)
        if lm:
            cur.append((lm.group(1), "", False))
            continue
        cur.append((None, s.split(";")[0].strip(), in_asm))
    return kernels


def check_kernel(name, ins):
    labels = {lab: i for i, (lab, _, _) in enumerate(ins) if lab}
    bad = []
    reads = [i for i, (_, t, a) in enumerate(ins) if a and t.startswith("ds_read_b64_tr_b16")]
    for i in reads:
        dst = regs(ins[i][1].split(",")[0])
        # depth-first over the CFG from the instruction after the read; state = reads issued after it
        stack, seen = [(i + 1, 0)], set()
        while stack:
            pc, later = stack.pop()
            while pc < len(ins):
                if (pc, min(later, 64)) in seen:
                    break
                seen.add((pc, min(later, 64)))
                lab, t, a = ins[pc]
                if lab:
                    pc += 1
                    continue
                op = t.split()[0] if t else ""
                if op == "s_waitcnt":
                    m = re.search(r"lgkmcnt\((\d+)\)", t)
                    if m and later >= int(m.group(1)):
                        break                                   # retired on this path
                    pc += 1
                    continue
                if a and t.startswith("ds_read_b64_tr_b16"):
                    if regs(t.split(",")[0]) & dst:
                        bad.append((i, pc, "destination overwritten by a later read before a wait"))
                        break
                    later += 1
                    pc += 1
                    continue
                if op == "s_endpgm":
                    break
                if op == "s_branch":
                    pc = labels[t.split()[1]]
                    continue
                if op.startswith("s_cbranch"):
                    stack.append((labels[t.split()[1]], later))
                    pc += 1
                    continue
                if regs(t) & dst:
                    bad.append((i, pc, "touched before the covering s_waitcnt"))
                    break
                pc += 1
    return len(reads), bad


def compile_hessian():
    out = os.path.join(tempfile.mkdtemp(prefix="gptq_isa_"), "hessian.s")
    from gptq_amd import build as B
    cmd = [B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get("hessian.hip", []), "--cuda-device-only", "-S",
           os.path.join(ROOT, "gptq_amd", "csrc", "hessian.hip"), "-o", out]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def main(paths):
    if not paths:
        sys.path.insert(0, ROOT)
        paths = [compile_hessian()]
    status = 0
    for p in paths:
        for name, ins in parse_kernels(p).items():
            n, bad = check_kernel(name, ins)
            if n:
                print(f"{name[:70]}: {n} async LDS reads, {len(bad)} violations")
            for i, pc, why in bad[:10]:
                status = 1
                print(f"   read `{ins[i][1]}`  <-  `{ins[pc][1]}`: {why}")
    return status


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
