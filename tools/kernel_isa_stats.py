"""Static instruction mix and register counts of every kernel of one HIP source (no GPU needed): compiles it for gfx950
with -save-temps (the Makefile's flags) in a scratch directory and reads the .s file.  How the round-3 work on k_chunk
started: SALU as large as VALU = per-lane branches the compiler made of a ternary.

    python tools/kernel_isa_stats.py collision_amd/csrc/lbvh.hip [name filter]      # -> one line per kernel
    python tools/kernel_isa_stats.py collision_amd/csrc/lbvh.hip k_chunk --loops    # + the loops of each kernel
"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fvisibility=hidden"]


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + names, capture_output=True, text=True, check=True).stdout
        return out.strip().splitlines()
    except Exception:
        return names


def main():
    src = os.path.abspath(sys.argv[1])
    want = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
    loops = "--loops" in sys.argv
    with tempfile.TemporaryDirectory() as tmp:
        subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src),
                        "-save-temps", "-c", src, "-o", "x.o"], cwd=tmp, check=True, stderr=subprocess.DEVNULL)
        asm = [f for f in os.listdir(tmp) if f.endswith("gfx950.s")][0]
        text = open(os.path.join(tmp, asm)).read()
    meta = {}
    for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size", text, re.S):       # one entry of amdhsa.kernels
        blk = m.group(0)
        nm = re.search(r"\.name:\s+(\S+)", blk)
        if not nm:
            continue
        g = lambda key: (re.search(r"\." + key + r":\s+(\d+)", blk) or [None, "?"])[1]
        meta[nm.group(1)] = (g("vgpr_count"), g("sgpr_count"), g("group_segment_fixed_size"), g("private_segment_fixed_size"))
    rows = []
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)^\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if name not in meta:
            continue
        ins = [l.strip().split()[0] for l in body.splitlines() if l.startswith("\t") and not l.strip().startswith((".", ";"))]
        c = Counter(ins)
        tot = lambda p: sum(v for k, v in c.items() if k.startswith(p))
        rows.append((name, len(ins), tot("v_"), tot("s_") - c.get("s_waitcnt", 0) - c.get("s_nop", 0), tot("ds_"),
                     tot("global_") + tot("buffer_") + tot("flat_") + tot("scratch_"), tot("s_cbranch") + c.get("s_branch", 0),
                     c.get("s_waitcnt", 0), c.get("s_nop", 0), body))
    nice = demangle([r[0] for r in rows])
    print("%-72s %6s %6s %6s %5s %5s %6s %5s %4s   vgpr sgpr    lds scratch" % ("kernel", "instr", "valu", "salu", "lds", "vmem", "branch", "wait", "nop"))
    for r, n in zip(rows, nice):
        n = re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0]
        if want and want not in n:
            continue
        v, s, l, p = meta[r[0]]
        print("%-72s %6d %6d %6d %5d %5d %6d %5d %4d   %4s %4s %6s %7s" % ((n[:72],) + r[1:9] + (v, s, l, p)))
        if loops:
            lines = r[9].splitlines()
            labels = {mm.group(1): i for i, l2 in enumerate(lines) for mm in [re.match(r"^(\.LBB\d+_\d+):", l2)] if mm}
            for i, l2 in enumerate(lines):
                mm = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l2)
                if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
                    body_l = [x for x in lines[labels[mm.group(1)]:i + 1] if x.startswith("\t") and not x.strip().startswith((".", ";"))]
                    print("    loop %-12s %4d instructions (%d vector, %d LDS, %d vmem)" % (
                        mm.group(1), len(body_l), sum(x.strip().startswith("v_") for x in body_l),
                        sum(x.strip().startswith("ds_") for x in body_l), sum(x.strip().startswith(("global_", "buffer_", "flat_")) for x in body_l)))


if __name__ == "__main__":
    main()
