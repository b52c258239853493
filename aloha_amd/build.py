"""Build libaloha_vp.so (HIP kernels + C-ABI host) for gfx950 with hipcc, in-tree.

    python -m aloha_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting aloha_amd/lib/libaloha_vp.so is git-ignored
but travels to the GPU box with the source snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "lib", "obj")
LIB = os.path.join(LIBDIR, "libaloha_vp.so")
SOURCES = ["vp_engine.cpp", "keyswitch_host.cpp", "numtheory.cpp", "ew_kernels.hip", "ntt_tile.hip", "ntt_ks.hip", "ntt_single.hip", "ntt_product.hip", "encode_kernels.hip"]
HEADERS = ["kernels.hpp", "modarith.hpp", "ntt_device.hpp", "ntt_tiles.hpp", "numtheory.hpp", os.path.join("..", "..", "include", "aloha_vp.h")]
# Variant builds (measured-and-rejected experiments, ablations): the two-pass tile kernels with every knob and
# the single-launch transforms replace the default tile file, and -DALOHA_EXPERIMENTS enables their switches in the engine
# (ALOHA_NTT_FUSED*, ALOHA_RUN_GRAPH).  The default library contains none of it.
VARIANT_SOURCES = [os.path.join("experiments", "ntt_tile_knobs.hip") if s == "ntt_tile.hip" else s for s in SOURCES]
EXPERIMENTS_LIB = "exp"          # lib/libaloha_vp_exp.so: the plain variant the experiments' tests load (tests/test_experiments.py)
ARCH = "gfx950"


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def _newest(paths) -> float:
    return max(os.path.getmtime(p) for p in paths)


def build_variant(name: str, defines=(), force: bool = True, experiments: bool = True) -> str:
    """A/B variant of the library with extra -D macros -> lib/libaloha_vp_<name>.so (select it with
    ALOHA_VP_LIB=...; bench.py then skips its oracle guard, variants may compute garbage on purpose).
    ``experiments=False``: the DEFAULT sources with the extra macros (A/B of a knob of the shipped kernels, e.g. ALOHA_ROW_BMAP)."""
    out = os.path.join(LIBDIR, f"libaloha_vp_{name}.so")
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in (VARIANT_SOURCES if experiments else SOURCES)]
    deps = srcs + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest(deps):
        return out
    cmd = [hipcc(), "-O3", "-std=c++17", "-fPIC", "-shared", f"--offload-arch={ARCH}"] + (["-DALOHA_EXPERIMENTS=1"] if experiments else []) + \
          [f"-D{d}" for d in defines if d] + ["-o", out] + srcs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl"]
    subprocess.check_call(cmd)
    return out


def build_ablation(level: int) -> str:
    """Timing-experiment variant of the library (ALOHA_ABLATE=level); results are invalid."""
    return build_variant(f"ablate{level}", [f"ALOHA_ABLATE={level}"])


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_time = _newest([os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)])
    objs, rebuilt = [], False
    flags = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, os.path.splitext(src)[0] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(sp), hdr_time):
            cmd = [hipcc()] + flags + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            rebuilt = True
    if rebuilt or not os.path.exists(LIB):
        cmd = [hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    for a in sys.argv[1:]:
        if a.startswith("--ablate="):
            print(build_ablation(int(a.split("=")[1])))
        if a.startswith("--variant="):                      # --variant=name:MACRO=1,MACRO2=0
            name, _, defs = a.split("=", 1)[1].partition(":")
            print(build_variant(name, defs.split(",")))
        if a.startswith("--plain="):                        # --plain=name:MACRO=1  (the default sources + macros)
            name, _, defs = a.split("=", 1)[1].partition(":")
            print(build_variant(name, defs.split(","), experiments=False))
