#!/usr/bin/env python3
"""Print registers / scratch / occupancy of every gfx950 kernel in a .hip source
(hipcc -Rpass-analysis=kernel-resource-usage), one line per kernel.

    python tools/kernel_resources.py aloha_amd/csrc/ntt_tile.hip [-DNAME=VALUE ...] [--filter row]
"""
from __future__ import annotations

import re
import subprocess
import sys
import tempfile


def main() -> int:
    args = sys.argv[1:]
    flt = None
    if "--filter" in args:
        i = args.index("--filter")
        flt = args[i + 1]
        del args[i:i + 2]
    src, extra = args[0], args[1:]
    with tempfile.TemporaryDirectory() as tmp:
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950",
               "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", f"{tmp}/x.o"] + extra
        p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        return p.returncode
    name = None
    rec: dict[str, str] = {}
    rows = []
    for line in p.stderr.splitlines():
        m = re.search(r"remark: [^ ]* +(?:Function )?Name: (\S+)", line) or re.search(r"Function Name: (\S+)", line)
        if m:
            if name:
                rows.append((name, rec))
            name, rec = m.group(1), {}
            continue
        m = re.search(r"(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
        if m and name:
            rec[m.group(1).split(" ")[0]] = m.group(2)
    if name:
        rows.append((name, rec))
    for name, rec in rows:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        dem = re.sub(r"\(anonymous namespace\)::|aloha::k::", "", dem)
        dem = re.sub(r"\(.*", "", dem)
        if flt and flt not in dem:
            continue
        print(f"{dem:60s} vgpr {rec.get('VGPRs','?'):>4} agpr {rec.get('AGPRs','?'):>3} sgpr {rec.get('TotalSGPRs','?'):>4} "
              f"scratch {rec.get('ScratchSize','?'):>4} occ {rec.get('Occupancy','?'):>2} lds {rec.get('LDS','?'):>6}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
