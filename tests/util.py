"""Shared helpers for the test-suite: fixture loading and a pure-Python big-int model."""
from __future__ import annotations

import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
REFERENCE = os.environ.get("ALOHA_REFERENCE", "/root/reference")
HAVE_REFERENCE = os.path.isdir(os.path.join(REFERENCE, "tv"))

Q0 = 576460825317867521      # src/top/h2_top_no_axilite.sv:31
Q1 = 576460924102115329      # :32
P = 576462951330889729       # keyswitch.mem:11
PRIMES = (Q0, Q1, P)
N = 8192

_manifest = None


def manifest() -> dict:
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLDEN, "manifest.json")) as f:
            _manifest = json.load(f)
    return _manifest


def load(entry) -> np.ndarray:
    """Load a fixture by manifest entry (dict with 'file') or relative path."""
    rel = entry["file"] if isinstance(entry, dict) else entry
    a = np.fromfile(os.path.join(GOLDEN, rel), dtype="<u8")
    if isinstance(entry, dict):
        assert a.size == entry["words"]
    return a


def load_isram(name: str) -> np.ndarray:
    e = manifest()["isram"][name]
    return np.fromfile(os.path.join(GOLDEN, e["file"]), dtype="<u4").reshape(-1, 3)


def reference_images() -> dict:
    return {k: load_isram(k) for k in ("encode_post", "mul_plain", "hom_add", "keyswitch")}


def sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()


def read_tv(path: str) -> np.ndarray:
    with open(path) as f:
        toks = f.read().split()
    n = len(toks)
    while n and toks[n - 1] == "x":
        n -= 1
    return np.array([int(t) for t in toks[:n]], dtype=np.uint64)


def splitmix64(seed: int, count: int) -> np.ndarray:
    """Deterministic 64-bit stream (synthetic residues come from splitmix64)."""
    out = np.empty(count, dtype=np.uint64)
    x = seed & 0xFFFFFFFFFFFFFFFF
    M = 0xFFFFFFFFFFFFFFFF
    for i in range(count):
        x = (x + 0x9E3779B97F4A7C15) & M
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
        out[i] = z ^ (z >> 31)
    return out


def splitmix_residues(seed: int, n: int, q: int) -> np.ndarray:
    """The first n accepted candidates of splitmix64(seed): candidate = top 60 bits of an output, accepted when < q
    (SURVEY 8d's synthetic inputs; the rule aloha_vp_fill_synthetic implements on the device)."""
    out = np.empty(0, dtype=np.uint64)
    start = 0
    with np.errstate(over="ignore"):
        while out.size < n:
            m = max(1024, 2 * (n - out.size))
            idx = np.arange(start + 1, start + m + 1, dtype=np.uint64)
            z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
            z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
            z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
            z = z ^ (z >> np.uint64(31))
            c = z >> np.uint64(4)
            out = np.concatenate([out, c[c < np.uint64(q)]])
            start += m
    return out[:n]


def random_residues(rng: np.random.Generator, shape, q: int) -> np.ndarray:
    """Uniform canonical residues in [0, q)."""
    a = rng.integers(0, q, size=shape, dtype=np.uint64)
    return a
