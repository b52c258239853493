#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference's own golden vectors (``/root/reference/tv``).

Fixtures are DATA only: inputs and expected outputs of the reference's test vectors,
converted from one-decimal-per-line text to raw little-endian uint64 ``.bin``
files, plus SHA-256 digests of the larger expected outputs and of the reference's
instruction-ROM images (so the product's own assembler output can be compared against them
where the reference is absent).  No reference source text is copied.

Run here (the GPU box has no /root/reference):  python tests/golden/make_fixtures.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

REF = os.environ.get("ALOHA_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
N = 8192


def read_tv(path: str) -> np.ndarray:
    """Decimal text -> uint64 array of the numeric prefix ('x' = uninitialised SPM rows,
    which the test bench dumps after an encode: top_noaxilite_tb.sv:536-565)."""
    with open(path) as f:
        toks = f.read().split()
    n = len(toks)
    while n and toks[n - 1] == "x":
        n -= 1
    assert "x" not in toks[:n], f"{path}: x inside the numeric prefix"
    return np.array([int(t) for t in toks[:n]], dtype=np.uint64)


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def put(rel: str, a: np.ndarray) -> dict:
    p = os.path.join(OUT, rel)
    os.makedirs(os.path.dirname(p), exist_ok=True)
    np.ascontiguousarray(a, dtype="<u8").tofile(p)
    return {"file": rel, "words": int(a.size), "sha256": sha(a)}


def main() -> int:
    tv = os.path.join(REF, "tv")
    man: dict = {"source": "AntChainOpenLabs/ALOHA tv/ (SURVEY C.3-C.5)", "n": N}

    # ---- case0_4_4: complete 8-op RTL chain (op sequence recovered in SURVEY C.4) -------
    c0 = os.path.join(tv, "case0_4_4")
    r0 = os.path.join(c0, "rtl_result")
    m0: dict = {"ops": [
        # (op, dest_row, a, b) in the op-list encoding of top_noaxilite_tb.sv:249-298
        ["load", 0x000, 0, 0], ["rotate", 0x100, 2, 0x000], ["encode", 0x200, 0, 0],
        ["encode", 0x280, 0, 0], ["mul_plain", 0x300, 0x000, 0x200], ["mul_plain", 0x400, 0x100, 0x280],
        ["hom_add", 0x500, 0x300, 0x400], ["store", 0x500, 0, 0]]}
    m0["inst_0_out"] = put("case0/inst_0_out.bin", read_tv(os.path.join(r0, "inst_0_out.txt")))
    m0["ksk_step2"] = put("case0/ksk_step2.bin", read_tv(os.path.join(c0, "ksk_step2.txt")))
    for k in (2, 3):
        m0[f"inst_{k}_0_out"] = put(f"case0/inst_{k}_0_out.bin", read_tv(os.path.join(r0, f"inst_{k}_0_out.txt"))[:2 * N])
    for k in range(1, 8):
        a = read_tv(os.path.join(r0, f"inst_{k}_out.txt"))
        m0[f"inst_{k}_out"] = put(f"case0/inst_{k}_out.bin", a)
    # software-model kernel for the rotate (pins iNTT/aut/key-switch independently of the RTL dumps)
    m0["kernel1_ct_before_rotate"] = put("case0/kernel1_ct_before_rotate.bin", read_tv(os.path.join(c0, "kernel1_ct_before_rotate.txt")))
    m0["kernel1_ct_after_rotate"] = put("case0/kernel1_ct_after_rotate.bin", read_tv(os.path.join(c0, "kernel1_ct_after_rotate.txt")))
    man["case0"] = m0

    # ---- case1_8_8: one rotate by 4 (third prime / second key slot) and one encode NTT pair -
    c1 = os.path.join(tv, "case1_8_8")
    m1: dict = {}
    m1["ksk_step4"] = put("case1/ksk_step4.bin", read_tv(os.path.join(c1, "ksk_step4.txt")))
    m1["kernel12_ct_before_rotate"] = put("case1/kernel12_ct_before_rotate.bin", read_tv(os.path.join(c1, "kernel12_ct_before_rotate.txt")))
    m1["kernel12_ct_after_rotate"] = put("case1/kernel12_ct_after_rotate.bin", read_tv(os.path.join(c1, "kernel12_ct_after_rotate.txt")))
    m1["kernel1_pt_after_encode_fft_mod"] = put("case1/kernel1_pt_after_encode_fft_mod.bin", read_tv(os.path.join(c1, "kernel1_pt_after_encode_fft_mod.txt")))
    m1["kernel1_pt_after_encode"] = put("case1/kernel1_pt_after_encode.bin", read_tv(os.path.join(c1, "kernel1_pt_after_encode.txt")))
    man["case1"] = m1

    # ---- case2_16_16: the 29-op chain of case3.txt; inputs + digests of every dump ---------
    c2 = os.path.join(tv, "case2_16_16")
    r2 = os.path.join(c2, "rtl_result")
    oplist = os.path.join(REF, "vivado_prj/top_noaxilite.srcs/sources_1/imports/sim/model_run/case3.txt")
    m2: dict = {"oplist": [[int(x, 16) for x in l.strip().split(",")] for l in open(oplist) if l.strip()]}
    m2["inst_0_out"] = put("case2/inst_0_out.bin", read_tv(os.path.join(r2, "inst_0_out.txt")))
    m2["ksk_step2"] = put("case2/ksk_step2.bin", read_tv(os.path.join(c2, "ksk_step2.txt")))
    m2["ksk_step8"] = put("case2/ksk_step8.bin", read_tv(os.path.join(c2, "ksk_step8.txt")))
    m2["encoder_out"] = {}
    m2["expected_sha256"] = {}
    m2["expected_words"] = {}
    for k, op in enumerate(m2["oplist"]):
        a = read_tv(os.path.join(r2, f"inst_{k}_out.txt"))
        m2["expected_sha256"][str(k)] = sha(a)
        m2["expected_words"][str(k)] = int(a.size)
        if op[0] >> 28 == 3:
            m2["encoder_out"][str(k)] = put(f"case2/inst_{k}_0_out.bin", read_tv(os.path.join(r2, f"inst_{k}_0_out.txt"))[:2 * N])
    final = read_tv(os.path.join(REF, "vivado_prj/top_noaxilite.srcs/sources_1/new/case3_expected_result.txt"))
    m2["case3_expected_result"] = put("case2/case3_expected_result.bin", final)
    man["case2"] = m2

    # ---- CKKS encoder pairs of the software model (cleartext -> coefficient-form plaintext) -----
    # Cleartexts are N doubles ((re, im) interleaved slots) of which only a short prefix is non-zero:
    # kept in the manifest as [index, value] pairs.  Expected outputs: *_fft_mod (before the NTT);
    # case0 ships only the NTT-domain plaintext, kept as such.
    def cleartext(path):
        z = np.loadtxt(path)
        assert z.size == N
        nz = np.nonzero(z)[0]
        return {"n": N, "nonzero": [[int(i), float(z[i])] for i in nz]}
    enc: dict = {}
    for case, k in (("case1_8_8", 1), ("case1_8_8", 2), ("case2_16_16", 13), ("case2_16_16", 14)):
        d = os.path.join(tv, case)
        short = case.split("_")[0]
        key = f"{short}/kernel{k}"
        ent = {"cleartext": cleartext(os.path.join(d, f"kernel{k}_cleartext_before_encode_re_im.txt"))}
        rel = f"{short}/kernel{k}_pt_after_encode_fft_mod.bin"
        ent["fft_mod"] = put(rel, read_tv(os.path.join(d, f"kernel{k}_pt_after_encode_fft_mod.txt")))
        enc[key] = ent
    ent = {"cleartext": cleartext(os.path.join(c0, "kernel2_cleartext_before_encode_re_im.txt"))}
    ent["ntt"] = put("case0/kernel2_pt_after_encode.bin", read_tv(os.path.join(c0, "kernel2_pt_after_encode.txt")))
    enc["case0/kernel2"] = ent
    man["encode"] = enc

    # ---- instruction-ROM images the test bench loads ($readmemh data, inst_rom.v:32-35) -------
    # Stored as packed uint32 triplets (inst, imm_hi, imm_lo) -- the four micro-programs that
    # the system test runs (isram_file_generator.sv:27-30).  They are test INPUTS: the product
    # must execute them unchanged; its own program generators live in aloha_amd/programs.py.
    isd = os.path.join(REF, "sim/vp/isram_file_generator")
    man["isram"] = {}
    for name, pc in (("encode_post", 0), ("mul_plain", 64), ("hom_add", 160), ("keyswitch", 256)):
        rows = []
        for l in open(os.path.join(isd, name + ".mem")):
            l = l.strip()
            if l:
                v = int(l, 16)
                rows.append(((v >> 64) & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF, v & 0xFFFFFFFF))
        w = np.array(rows, dtype="<u4")
        path = os.path.join(OUT, "isram", name + ".u32")
        os.makedirs(os.path.dirname(path), exist_ok=True)
        w.tofile(path)
        man["isram"][name] = {"file": f"isram/{name}.u32", "pc": pc, "words": len(rows),
                              "sha256": hashlib.sha256(w.tobytes()).hexdigest()}

    with open(os.path.join(OUT, "manifest.json"), "w") as f:
        json.dump(man, f, indent=1, sort_keys=True)
    total = sum(os.path.getsize(os.path.join(dp, fn)) for dp, _, fns in os.walk(OUT) for fn in fns)
    print(f"wrote fixtures under {OUT}: {total / 1e6:.2f} MB")
    return 0


if __name__ == "__main__":
    sys.exit(main())
