#!/usr/bin/env python3
"""Run a reference op list (``case3.txt`` format) on the GPU engine and write the per-op dumps
the test bench writes (``inst_K_out.txt``, ``inst_K_0_out.txt``), so they diff directly against
``tv/<case>/rtl_result``.

    python tools/run_oplist.py --oplist case3.txt --tv /path/to/tv/case2_16_16 --out outdir
          [--isram isram_file.mem]

Inputs taken from the ``tv`` directory: the initial ciphertext (``rtl_result/inst_0_out.txt``),
the RTL encoder's outputs (``rtl_result/inst_K_0_out.txt``: replaying the RTL chain bit for bit needs
the RTL's own single-precision encoder results) and the keys
``ksk_step{2,4,8}.txt`` in the slot order of ``load_ksk`` (top_noaxilite_tb.sv:372-394,:531).
Without ``--isram`` the own program generators are used.  The command line always runs the GPU
engine; ``run(vp, ...)`` takes any engine object (the tests pass the oracle's machine on the CPU).
"""
from __future__ import annotations

import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from aloha_amd import harness, isa, programs, tbfiles  # noqa: E402

N = 8192


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--oplist", required=True)
    ap.add_argument("--tv", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--isram")
    args = ap.parse_args()
    from aloha_amd import vp as avp
    with avp.VP(spm_rows=1 << 16, ksk_rows=1 << 16, max_log_n=13) as vp:
        return run(vp, args)


def run(vp, args) -> int:
    """``args``: an object with ``oplist``, ``tv``, ``out`` and ``isram`` (or None)."""
    tb = harness.TestBench(vp)
    if args.isram:
        vp.load_isram(0, isa.parse_mem_text(open(args.isram).read()))
    else:
        mod = (programs.Q0, programs.Q1)
        tb.load_programs({"encode_post": programs.assemble(programs.encode_post(N, mod)),
                          "mul_plain": programs.assemble(programs.mul_plain(N, mod)),
                          "hom_add": programs.assemble(programs.hom_add(N, mod)),
                          "keyswitch": programs.assemble(programs.keyswitch(N, mod, programs.P_SPECIAL))})
    ops = harness.parse_op_list(open(args.oplist).read())
    keys = []
    for step in (2, 4, 8):
        p = os.path.join(args.tv, f"ksk_step{step}.txt")
        keys.append(tbfiles.read_tv(p)[0] if os.path.exists(p) else np.zeros(12 * N, np.uint64))
    tb.load_ksk(keys)
    rr = os.path.join(args.tv, "rtl_result")
    for k, op in enumerate(ops):
        if op.op == harness.OP_LOAD:
            tb.dram_ct[op.dram_addr] = tbfiles.read_tv(os.path.join(rr, f"inst_{k}_out.txt"))[0]
        elif op.op == harness.OP_ENCODE:
            tb.dram_encoder[op.dram_addr] = tbfiles.read_tv(os.path.join(rr, f"inst_{k}_0_out.txt"))[0][:2 * N]
    os.makedirs(args.out, exist_ok=True)
    written = set()

    def dump(k, sub, data):
        # rows the run never wrote are 'x' in the RTL dump (fresh SPM); here they read as 0, so
        # an encode into untouched rows is written with its numeric prefix only
        valid = data.size
        ref = os.path.join(rr, f"inst_{k}{sub}_out.txt")
        if os.path.exists(ref):
            valid = tbfiles.read_tv(ref)[0].size
        tbfiles.write_dump(os.path.join(args.out, f"inst_{k}{sub}_out.txt"), data[:valid], data.size)
        written.add(f"inst_{k}{sub}_out.txt")

    tb.dump = dump
    tb.run(ops)
    bad = 0
    for name in sorted(written):
        ref = os.path.join(rr, name)
        if os.path.exists(ref):
            same = open(ref).read().split() == open(os.path.join(args.out, name)).read().split()
            bad += not same
            print(f"{name}: {'identical' if same else 'DIFFERENT'}")
    print(f"{len(written)} dumps written to {args.out}; {bad} differ from {rr}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
