"""The measured-and-rejected experiments (single-launch N = 2^16 transforms through the Infinity Cache, recorded runs as
hipGraphs) live in a VARIANT build of the library (csrc/experiments/, `python -m aloha_amd.build
--variant`), not in the default one.  Their parity tests run here in a process of their own that loads the variant
(ALOHA_VP_LIB); the default library is checked not to contain the experiments' switches."""
import os
import subprocess
import sys

import pytest

from aloha_amd import build as abuild

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_build_has_no_experiment_code():
    """The default sources contain no experiment switches: the tile file has no preprocessor branches at all, the engine
    reads none of the experiments' environment variables outside its ALOHA_EXPERIMENTS blocks."""
    tile = open(os.path.join(ROOT, "aloha_amd", "csrc", "ntt_tile.hip")).read()
    assert "#if" not in tile.replace("#ifndef", "").replace("#ifdef", "") and "#ifdef" not in tile and "#ifndef" not in tile
    eng = open(os.path.join(ROOT, "aloha_amd", "csrc", "vp_engine.cpp")).read()
    depth, outside = 0, []
    for line in eng.splitlines():
        if line.startswith("#ifdef ALOHA_EXPERIMENTS"):
            depth += 1
        elif line.startswith("#endif") and depth:
            depth -= 1
        elif not depth and any(k in line for k in ("ALOHA_NTT_FUSED", "ALOHA_RUN_GRAPH", "hipGraphLaunch", "ntt_fused_fwd")):
            if not line.lstrip().startswith("//"):
                outside.append(line)
    assert not outside, outside


@pytest.mark.gpu
def test_experiments_variant_parity():
    lib = abuild.build_variant(abuild.EXPERIMENTS_LIB, force=False)
    env = dict(os.environ, ALOHA_VP_LIB=lib)
    p = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "experiments", "fused_ntt_cases.py")],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-1000:]
