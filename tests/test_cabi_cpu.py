"""CPU-side checks of the C-ABI library: it builds, loads, exports every symbol the header
declares, and refuses to run without a GPU (no CPU fallback).  Host-only parameter helpers
are cross-checked against the oracle."""
import ctypes
import os
import re

import pytest

import oracle
from aloha_amd import build as abuild
from aloha_amd import vp as avp
from util import P, Q0, Q1

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    abuild.build()
    return avp.load_library()


def _declared_functions():
    src = open(os.path.join(ROOT, "include", "aloha_vp.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(aloha_[a-z_]+)\s*\(", src)))


def test_header_symbols_all_exported(lib):
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/aloha_vp.h but not exported"
    from aloha_amd import ks
    assert sorted(set(avp.EXPORTS) | set(ks.EXPORTS)) == names          # the bindings (vp.py, ks.py) cover exactly the header


def test_every_exported_symbol_is_in_the_integration_table(lib):
    """INTEGRATION.md's table maps each entry point to the reference interface it replaces: every aloha_* symbol the LIBRARY
    exports (read from its dynamic symbol table, not from the header) must have a row, and every function named in the table
    must exist."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", avp.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if ln.split()[-1].startswith("aloha_")})
    assert exported == _declared_functions()                        # nothing exported beyond the header, nothing missing
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = "\n".join(ln for ln in text.splitlines() if ln.startswith("|"))
    cells = " ".join(ln.split("|")[1] for ln in table.splitlines() if ln.count("|") >= 3)   # first column only
    named = set()
    for tok in re.findall(r"`([^`]+)`", cells):
        # "aloha_vp_create / _destroy" names aloha_vp_create and aloha_vp_destroy
        parts = [t.strip() for t in re.split(r"[/,]", tok)]
        stem = None
        for t in parts:
            t = t.split("(")[0].strip()
            if t.startswith("aloha_"):
                named.add(t)
                stem = t.rsplit("_", 1)[0]
            elif t.startswith("_") and stem:
                named.add(stem + t)
    missing = [n for n in exported if n not in named]
    assert not missing, f"exported but absent from INTEGRATION.md's entry-point table: {missing}"
    ghosts = [n for n in named if n.startswith(("aloha_vp_", "aloha_ks_")) and n not in exported and "." not in n and n not in ("aloha_vp_stats", "aloha_vp_config")]
    assert not ghosts, f"named in the table but not exported: {ghosts}"


def test_abi_version_and_status_strings(lib):
    assert lib.aloha_vp_abi_version() == 4
    for s in range(12):
        assert lib.aloha_status_string(s)
    assert lib.aloha_status_string(0) == b"ok"


def test_no_cpu_fallback_without_gpu(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(avp.AlohaError) as e:
        avp.VP(spm_rows=64, ksk_rows=64)
    assert e.value.status == avp.ERR_DEVICE


def test_create_rejects_bad_config(lib):
    h = ctypes.c_void_p()
    cfg = avp._Config(abi_version=99, device=0, spm_rows=1, ksk_rows=1, max_log_n=13, flags=0, stream=None)
    assert lib.aloha_vp_create(ctypes.byref(cfg), ctypes.byref(h)) == avp.ERR_ARG
    cfg = avp._Config(abi_version=4, device=0, spm_rows=1, ksk_rows=1, max_log_n=17, flags=0, stream=None)
    assert lib.aloha_vp_create(ctypes.byref(cfg), ctypes.byref(h)) == avp.ERR_ARG
    assert lib.aloha_vp_create(None, ctypes.byref(h)) == avp.ERR_ARG


def test_host_parameter_helpers_match_oracle(lib):
    # the reference's hard-coded roots (tf_rom_generator.sv:75)
    assert avp.find_psi(Q0, 13) == 3825716582911
    assert avp.find_psi(Q1, 13) == 79932510954937
    assert avp.find_psi(P, 13) == 101017252977188
    for logn in (10, 12, 16):
        for idx in (0, 1, 5):
            p = avp.nth_prime(logn, idx)
            assert p == oracle.nth_prime(logn, idx)
            assert oracle.is_prime(p) and p % (2 << logn) == 1 and (1 << 59) < p < (1 << 60)
        q = avp.nth_prime(logn, 2)
        assert avp.find_psi(q, logn) == oracle.find_psi(q, logn)
    with pytest.raises(avp.AlohaError) as e:
        avp.find_psi(P, 14)                      # P - 1 has 2-adic valuation 14 < 15
    assert e.value.status == avp.ERR_ROOT
    with pytest.raises(avp.AlohaError) as e:
        avp.find_psi(Q0 + 2, 13)
    assert e.value.status == avp.ERR_MODULUS


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus N` with WORLD_SIZE unset starts N ranks itself (the first driver SCALE run would otherwise
    time one rank and print n_gpus: 1).  On this CPU-only host every rank stops at its GPU check, after saying what it
    was started with."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ALOHA_BENCH_ECHO"] = "1"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the ranks would run the benchmark")
    assert p.returncode == 2, p.stderr
    lines = [l for l in p.stderr.splitlines() if l.startswith("bench.py rank ")]
    assert sorted(l.split()[2] for l in lines) == ["0", "1", "2"]
    assert all(" of 3 " in l and "master 127.0.0.1:" in l for l in lines)
    assert len({l.split("master ")[1] for l in lines}) == 1          # one rendezvous for all ranks
    assert p.stderr.count("bench.py needs a GPU") == 3


def test_bench_self_launch_reports_a_rank_killed_by_a_signal():
    """A rank that dies by a signal (a GPU fault ends a process with SIGABRT) must fail the whole run -- its negative
    return code may not hide behind a sibling's 0 -- and its siblings must be ended instead of waiting for a timeout."""
    import signal
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ALOHA_BENCH_TEST_DIE"] = f"1:{int(signal.SIGABRT)}"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 128 + int(signal.SIGABRT), (p.returncode, p.stderr[-400:])
    assert time.monotonic() - t0 < 100                             # the sleeping siblings (120 s) were terminated


def test_bench_eight_rank_launch_and_the_three_rendezvous():
    """First contact with an 8-GPU node, rehearsed without one: `bench.py --gpus 8` starts 8 ranks with distinct
    RANK / LOCAL_RANK and ONE rendezvous; the two key-switch child groups each get a rendezvous port of their own, the same
    on every rank, different from the run's and from each other, with the elastic agent's variables removed."""
    import importlib.util
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["ALOHA_BENCH_ECHO"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if not torch.cuda.is_available():
        lines = [l for l in p.stderr.splitlines() if l.startswith("bench.py rank ")]
        assert sorted(int(l.split()[2]) for l in lines) == list(range(8))
        assert sorted(int(l.split(" local ")[1].split()[0]) for l in lines) == list(range(8))
        assert len({l.split("master ")[1] for l in lines}) == 1 and all(" of 8 " in l for l in lines)
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ports = set()
    for base in ("29500", "61234", "1024"):
        per_rank = []
        for r in range(8):
            run_env = {"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": "8", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": base,
                       "TORCHELASTIC_RUN_ID": "x", "TORCHELASTIC_USE_AGENT_STORE": "True"}
            envs = {m: bench.keyswitch_child_env(run_env, m, bench.KS_CHILD_PORT_SHIFTS[m]) for m in ("torch", "c_host")}
            for m, e in envs.items():
                assert (e["RANK"], e["LOCAL_RANK"], e["WORLD_SIZE"], e["ALOHA_KS_CHILD_MODE"]) == (str(r), str(r), "8", m)
                assert not any(k.startswith("TORCHELASTIC_") for k in e) and 2048 <= int(e["MASTER_PORT"]) < 62048
            per_rank.append((envs["torch"]["MASTER_PORT"], envs["c_host"]["MASTER_PORT"]))
        assert len(set(per_rank)) == 1                               # every rank derives the same two ports
        a, b = per_rank[0]
        assert len({a, b, base}) == 3
        ports.add((a, b))
    assert len(ports) == 3
