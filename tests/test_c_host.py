"""The C ABI from plain C: include/aloha_vp.h must compile as C99 (not just as C++), every declared symbol must link,
and a C program -- no Python, no torch in the process -- must reproduce the reference's golden rotate through the
key-switch host and through the reference's own instruction image (tests/c/ks_golden.c)."""
import os
import subprocess

import pytest

from aloha_amd import build as abuild
from util import GOLDEN, manifest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "ks_golden.c")


def _compile(tmp_path):
    abuild.build()
    exe = str(tmp_path / "ks_golden")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), SRC,
           "-L", abuild.LIBDIR, "-laloha_vp", f"-Wl,-rpath,{abuild.LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    return exe


def test_header_is_c99_and_the_example_links(tmp_path):
    exe = _compile(tmp_path)
    assert os.path.exists(exe)
    # without a GPU the program must fail at aloha_vp_create with the library's own status, not crash
    import torch
    if not torch.cuda.is_available():
        p = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=120)
        assert p.returncode == 1 and "aloha_vp_create" in p.stderr and "device" in p.stderr.lower(), p.stderr


@pytest.mark.gpu
def test_c_program_reproduces_the_golden_rotate(tmp_path):
    exe = _compile(tmp_path)
    image = os.path.join(GOLDEN, manifest()["isram"]["keyswitch"]["file"])
    p = subprocess.run([exe, GOLDEN, image], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "golden rotate reproduced (C host and reference image)" in p.stdout


def test_program_generator_is_clean_under_asan_and_ubsan(tmp_path):
    """The C host's generator (csrc/keyswitch_host.cpp, numtheory.cpp) compiled with g++ -fsanitize=address,undefined on the
    CPU: plans for 7 layouts x all ranks, every phase fetched, on-demand programs regenerated (tests/c/ks_plan_sanitize.cpp).
    (GPU AddressSanitizer is not available on the pool; the host code is what a sanitizer can see.)"""
    exe = str(tmp_path / "ks_plan_sanitize")
    cmd = ["g++", "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wno-subobject-linkage",
           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "c", "ks_plan_sanitize.cpp"), "-ldl", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "ks_plan_sanitize ok" in p.stdout and "ERROR" not in p.stderr


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    """The oracle's C restatement under gcc -fsanitize=address,undefined: the golden rotate through its ISA machine with the
    reference's keyswitch.mem image, transform round trips at N = 2^10 and 2^16, the vautn identity, the RTL Barrett form
    (tests/c/oracle_sanitize.c).  The checker is code too."""
    exe = str(tmp_path / "oracle_sanitize")
    cmd = ["gcc", "-std=gnu99", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fopenmp",
           os.path.join(ROOT, "tests", "c", "oracle_sanitize.c"), "-lm", "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    img = manifest()["isram"]["keyswitch"]
    p = subprocess.run([exe, GOLDEN, img["file"], str(img["words"])], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert "oracle_sanitize ok" in p.stdout and "ERROR" not in p.stderr
