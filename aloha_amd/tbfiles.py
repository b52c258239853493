"""Text file formats of the reference test bench, so inputs and outputs interchange with ``tv/``
and with a Vivado run of ``sim/top/top_noaxilite_tb.sv``.

* ``tv`` vectors / per-op dumps: one unsigned decimal per line; rows of SPM that were never
  written are dumped as the literal ``x`` (``dump_poly``, top_noaxilite_tb.sv:536-565).
* DRAM image for ``$readmemh`` (``load_dram_data`` :341-346): one 512-bit word per line, 128 hex
  digits, word ``i`` holding bytes ``64*i .. 64*i+63`` with the lowest address in the least
  significant bits (the dump code reads element ``j`` of a word at bit offset ``64*j``, :556-561).
"""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import numpy as np

DDR_WORD_BYTES = 64          # DDR_DATA_WIDTH = 512 (top_noaxilite_tb.sv:38)
DRAM_ENCODER_BASE = 0        # :44
DRAM_VP_BASE = 10485760      # :45
DUMP_DRAM_ADDR = 0x100000    # :77


def read_tv(path: str):
    """-> (values uint64[valid], total_lines).  ``x`` lines may only trail the numbers."""
    with open(path) as f:
        toks = f.read().split()
    n = len(toks)
    while n and toks[n - 1] == "x":
        n -= 1
    if "x" in toks[:n]:
        raise ValueError(f"{path}: 'x' inside the numeric prefix")
    return np.array([int(t) for t in toks[:n]], dtype=np.uint64), len(toks)


def write_dump(path: str, data, total_words: Optional[int] = None) -> None:
    """Write a per-op dump as the test bench does; words beyond ``len(data)`` become ``x``."""
    data = np.ascontiguousarray(data, dtype=np.uint64)
    total = total_words if total_words is not None else data.size
    with open(path, "w") as f:
        f.write("\n".join(str(int(v)) for v in data))
        if data.size:
            f.write("\n")
        f.write("x\n" * (total - data.size))


def dram_image_lines(regions: Dict[int, np.ndarray], size_bytes: int) -> Iterable[str]:
    """``$readmemh`` text of a DRAM image: ``regions`` maps byte address -> uint64 array."""
    img = np.zeros(size_bytes // 8, dtype=np.uint64)
    for addr, arr in regions.items():
        if addr % 8:
            raise ValueError("region address must be 8-byte aligned")
        a = np.ascontiguousarray(arr, dtype=np.uint64)
        img[addr // 8: addr // 8 + a.size] = a
    words = img.reshape(-1, DDR_WORD_BYTES // 8)
    for w in words:
        yield "".join(f"{int(v):016x}" for v in w[::-1])


def write_dram_image(path: str, regions: Dict[int, np.ndarray], size_bytes: int) -> None:
    with open(path, "w") as f:
        for line in dram_image_lines(regions, size_bytes):
            f.write(line + "\n")


def read_dram_image(path: str) -> np.ndarray:
    """Inverse of ``write_dram_image``: -> uint64 array of the whole image."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            v = int(line, 16)
            out.extend((v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF for j in range(DDR_WORD_BYTES // 8))
    return np.array(out, dtype=np.uint64)
