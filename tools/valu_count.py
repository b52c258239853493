#!/usr/bin/env python3
"""VALU instruction census of the transform kernels: compiles one .hip file to gfx950 assembly and counts, per
kernel, the VALU instructions (all / those that cost a full ~4-cycle issue slot; v_mov, v_cndmask and the bitwise
ops cost about half, tools/ubench_valu.hip).

    python tools/valu_count.py aloha_amd/csrc/ntt_tile.hip [filter]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

FAST = ("v_mov", "v_xor", "v_and", "v_or", "v_cndmask", "v_not", "v_readfirstlane")


def main() -> int:
    src = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else "kernel"
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-S", "--cuda-device-only",
                               "-I" + os.path.dirname(os.path.abspath(src)), src, "-o", out] + sys.argv[3:], stderr=subprocess.DEVNULL)
        text = open(out).read()
    for f in re.split(r"\n(?=_Z\w+:)", text):
        name = f.split(":")[0]
        if not name.startswith("_Z") or flt not in name:
            continue
        body = f.split(".Lfunc_end")[0]
        ops = [ln.split()[0] for ln in (x.strip() for x in body.split("\n")) if ln.startswith("v_")]
        c = collections.Counter(ops)
        full = sum(v for k, v in c.items() if not k.startswith(FAST))
        nops = body.count("s_nop")
        try:
            dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        print(f"{dem[:90]:90s} valu {sum(c.values()):5d} full-slot {full:5d} slots~ {full + (sum(c.values()) - full) / 2:7.1f} s_nop {nops}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
