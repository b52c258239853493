"""Test-bench file formats and the op-list runner."""
import os
import subprocess
import sys

import numpy as np
import pytest

from aloha_amd import harness, tbfiles
from util import HAVE_REFERENCE, REFERENCE, manifest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dump_roundtrip_with_x_rows(tmp_path):
    data = np.array([0, 1, (1 << 60) - 1, 576460825317867520], dtype=np.uint64)
    p = tmp_path / "inst_2_out.txt"
    tbfiles.write_dump(str(p), data, total_words=10)
    lines = p.read_text().split()
    assert lines[:4] == [str(int(v)) for v in data] and lines[4:] == ["x"] * 6
    back, total = tbfiles.read_tv(str(p))
    assert np.array_equal(back, data) and total == 10


def test_dram_image_roundtrip(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.integers(0, 1 << 60, 24, dtype=np.uint64)
    b = rng.integers(0, 1 << 60, 8, dtype=np.uint64)
    p = tmp_path / "dram.mem"
    tbfiles.write_dram_image(str(p), {0x40: a, 0x200: b}, size_bytes=0x400)
    lines = p.read_text().split()
    assert len(lines) == 0x400 // 64 and all(len(l) == 128 for l in lines)
    img = tbfiles.read_dram_image(str(p))
    assert np.array_equal(img[8:32], a) and np.array_equal(img[0x40:0x48], b) and img[0] == 0
    # element j of a 512-bit word sits at bit offset 64*j (top_noaxilite_tb.sv:556-561)
    assert int(lines[1], 16) & 0xFFFFFFFFFFFFFFFF == int(a[0])


def test_op_list_parser_matches_fixture():
    ops = [harness.parse_op(a) for a in manifest()["case2"]["oplist"]]
    assert len(ops) == 29
    assert ops[0].op == harness.OP_LOAD and ops[0].spm_addr == 0
    assert ops[5].op == harness.OP_ROTATE and ops[5].step == 2 and ops[5].src1 == 0 and ops[5].spm_addr == 0x400
    assert ops[26].op == harness.OP_ROTATE and ops[26].step == 8 and ops[26].src1 == 0x300
    assert ops[28].op == harness.OP_STORE and ops[28].spm_addr == 0x300
    text = "\n".join(",".join(f"{x:08x}" for x in a) for a in manifest()["case2"]["oplist"])
    assert [o.__dict__ for o in harness.parse_op_list(text)] == [o.__dict__ for o in ops]


@pytest.mark.reference
@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present")
def test_run_oplist_tool_reproduces_every_rtl_dump(tmp_path):
    """The reference's op list + its instruction ROM image, both unchanged, through the tool's flow
    (with the oracle's machine as the engine, on the CPU): every dump text-identical to
    tv/case2_16_16/rtl_result."""
    import argparse
    import contextlib
    import importlib.util
    import io
    import oracle
    spec = importlib.util.spec_from_file_location("run_oplist", os.path.join(ROOT, "tools", "run_oplist.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    args = argparse.Namespace(
        oplist=os.path.join(REFERENCE, "vivado_prj/top_noaxilite.srcs/sources_1/imports/sim/model_run/case3.txt"),
        tv=os.path.join(REFERENCE, "tv/case2_16_16"), out=str(tmp_path),
        isram=os.path.join(REFERENCE, "sim/vp/isram_file_generator/isram_file.mem"))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        rc = tool.run(oracle.OracleVP(), args)
    out = buf.getvalue()
    assert rc == 0, out[-2000:]
    assert "37 dumps written" in out and "0 differ" in out
    final = open(tmp_path / "inst_28_out.txt").read().split()
    exp = open(os.path.join(REFERENCE, "vivado_prj/top_noaxilite.srcs/sources_1/new/case3_expected_result.txt")).read().split()
    assert final == exp
