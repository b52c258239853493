"""The row tile's LDS image (csrc/ntt_tiles.hpp: RowLanes, RP) under gfx950's banking rules (MI355X_MICROARCH.md, LDS):

    ds_write_b64: served 16 lanes at a time, bank = (a / 4) mod 32  -> the 16 lanes' 8-byte slots must differ mod 16
    ds_read_b64 : served 32 lanes at a time, bank = (a / 4) mod 64  -> the 32 lanes' 8-byte slots must differ mod 32

The tile has four access patterns: stores in the A mapping, loads in the B mapping (the first exchange), stores in the B mapping,
loads in the A mapping (the trip back).  VERDICT r3 found the shipped layout (ALOHA_ROW_BMAP = 0) conflicted on the trip back --
a third of the kernels' LDS cycles by the PMC counters; the layout of ALOHA_ROW_BMAP = 1 must be conflict-free on all four, and
this test also keeps the model of the OLD layout honest by reproducing the 2-way conflicts the counters showed."""
from __future__ import annotations

import os
import re

import pytest

HPP = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "aloha_amd", "csrc", "ntt_tiles.hpp")


def lanes(bmap: int):
    """(RP, per lane: ldsA, ldsB) exactly as RowLanes computes them."""
    rp = 272 if bmap else 280
    out = []
    for lane in range(64):
        lwB, khB = (lane >> 4, lane & 15) if bmap else (lane & 3, lane >> 2)
        out.append(((lane >> 4) * rp + (lane & 15), lwB * rp + 17 * khB))
    return rp, out


def worst(slots, group, mod):
    """largest number of distinct slots of one lane group that share a bank"""
    w = 0
    for g0 in range(0, 64, group):
        banks = {}
        for s in set(slots[g0:g0 + group]):                  # identical addresses broadcast
            banks.setdefault(s % mod, []).append(s)
        w = max(w, max(len(v) for v in banks.values()))
    return w


def patterns(bmap: int):
    rp, L = lanes(bmap)
    res = {}
    for r in range(16):
        a = [ldsA + 17 * r for ldsA, _ in L]
        b = [ldsB + r for _, ldsB in L]
        for name, slots, group, mod in (("A store", a, 16, 16), ("B load", b, 32, 32), ("B store", b, 16, 16), ("A load", a, 32, 32)):
            res[name] = max(res.get(name, 1), worst(slots, group, mod))
    return rp, res


def test_header_matches_the_model():
    src = open(HPP).read()
    assert "constexpr int RP = ALOHA_ROW_BMAP ? 272 : 280;" in src
    assert "lwB(ALOHA_ROW_BMAP ? lane >> 4 : lane & 3), khB(ALOHA_ROW_BMAP ? lane & 15 : lane >> 2)" in src
    assert "ldsA((lane >> 4) * RP + (lane & 15)), ldsB(lwB * RP + 17 * khB)" in src
    assert re.search(r"lds\[L\.ldsA \+ 17 \* kk\] = v\[kk\]", src) and re.search(r"v\[jj\] = lds\[L\.ldsB \+ jj\]", src)


@pytest.mark.parametrize("bmap", [0, 1])
def test_image_is_a_bijection_inside_the_wave_region(bmap):
    rp, L = lanes(bmap)
    a = sorted(ldsA + 17 * r for ldsA, _ in L for r in range(16))
    b = sorted(ldsB + r for _, ldsB in L for r in range(16))
    assert a == b and len(set(a)) == 1024 and a[-1] < 4 * rp      # both mappings address the same 1024 slots, inside 4*RP words


def test_bmap1_is_conflict_free_on_all_four_patterns():
    _, res = patterns(1)
    assert res == {"A store": 1, "B load": 1, "B store": 1, "A load": 1}


def test_bmap0_model_reproduces_the_measured_conflicts():
    """rounds 1-3: the first exchange clean, the trip back 2-way conflicted on both sides (r03_pmc_summary.json: 33.55 M conflict
    cycles of 100.66 M LDS cycles in ntt_row_kernel)"""
    _, res = patterns(0)
    assert res == {"A store": 1, "B load": 1, "B store": 2, "A load": 2}


def test_shipped_default():
    src = open(HPP).read()
    m = re.search(r"#ifndef ALOHA_ROW_BMAP\n#define ALOHA_ROW_BMAP (\d)", src)
    assert m, "default of ALOHA_ROW_BMAP not found"
    rp, res = patterns(int(m.group(1)))
    # whichever layout ships, the reduction of the inner-product kernel (R*8*64 words per wave, ntt_ks.hip) must fit the region
    assert 2 * 8 * 64 <= 4 * rp


def test_single_pass_n8192_closed_form_addresses():
    """csrc/ntt_single.hip at N = 2^13 addresses its LDS image through closed forms of swz(lay(t, r)) -- a per-lane base XOR a 5-bit
    constant of the register index, plus the register's own index bits as an immediate -- instead of deriving them per access
    (round 4: -9 % of the kernel's instructions).  The identities, exhaustively, against the definitions the other sizes use."""
    LOGN = 13
    N, T, E = 1 << LOGN, (1 << LOGN) // 16, LOGN - 12
    swz = lambda e: e ^ ((e >> 4) & 31) ^ ((e >> 5) & 31)

    def lay(K, t, r):
        P = LOGN - 4 - 4 * K
        return ((t >> P) << (P + 4)) | (r << P) | (t & ((1 << P) - 1))

    lay3 = lambda t, r: (r >> E) * ((1 << LOGN) >> (4 - E)) + (t << E) + (r & ((1 << E) - 1))
    src = open(os.path.join(os.path.dirname(HPP), "ntt_single.hip")).read()
    for text in ("cA0 = t ^ (t >> 4) ^ ((t >> 5) & 15)", "(cA0 ^ ((r & 1) << 4)) + (r << 9)", "(cB1 ^ (((r << 1) ^ r) & 31)) + (r << 5)",
                 "cC2 ^ ((r << 1) ^ (r >> 3))", "(cA3 ^ (r & 1)) + ((r >> 1) << 10)"):
        assert text in src, text
    for t in range(T):
        cA0 = t ^ (t >> 4) ^ ((t >> 5) & 15)
        cB1 = (((t >> 5) << 9) | (t & 31)) ^ ((t >> 4) & 1) ^ (((t >> 5) & 1) << 4)
        cC2 = (((t >> 1) << 5) | (t & 1)) ^ (((t >> 1) & 15) << 1) ^ ((t >> 1) & 31)
        cA3 = (t << 1) ^ ((t >> 3) & 31) ^ ((t >> 4) & 31)
        for r in range(16):
            assert swz(lay(0, t, r)) == (cA0 ^ ((r & 1) << 4)) + (r << 9)
            assert swz(lay(1, t, r)) == (cB1 ^ (((r << 1) ^ r) & 31)) + (r << 5)
            assert swz(lay(2, t, r)) == cC2 ^ ((r << 1) ^ (r >> 3))
            assert swz(lay3(t, r)) == (cA3 ^ (r & 1)) + ((r >> 1) << 10)


def test_half_row_closed_form_addresses():
    """csrc/ntt_device.hpp: sw1 / sw2 / sw3 are swh(lay1 / lay2 / lay3) in closed form (the radix-8 single-pass kernel at N = 2^12 and
    the one-launch polynomial product address their half rows through them)."""
    swh = lambda y: y ^ ((y >> 3) & 31)
    src = open(os.path.join(os.path.dirname(HPP), "ntt_device.hpp")).read()
    for text in ("return ((l ^ (l >> 3)) ^ ((r & 3) << 3)) + (r << 6);", "^ (r * 9);", "return ((l << 3) ^ (l & 31)) ^ r;"):
        assert text in src, text
    for l in range(64):
        for r in range(8):
            assert swh((r << 6) | l) == ((l ^ (l >> 3)) ^ ((r & 3) << 3)) + (r << 6)
            assert swh(((l >> 3) << 6) | (r << 3) | (l & 7)) == ((((l >> 3) << 6) | (l & 7)) ^ (((l >> 3) & 3) << 3)) ^ (r * 9)
            assert swh((l << 3) | r) == ((l << 3) ^ (l & 31)) ^ r
