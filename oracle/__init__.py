"""CPU oracle for the ALOHA VP lane datapath -- TEST INFRASTRUCTURE ONLY.

ctypes binding of ``oracle/vp_oracle.c`` (a plain-C restatement of the reference RTL's
arithmetic, every function citing the reference file:line it follows).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import this
package, and only as the checker.  The product package ``aloha_amd`` never imports it.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` replays the reference's own
``tv/`` golden chains (case0 8-op chain, case2 29-op chain ending in
``case3_expected_result.txt``, model-kernel rotates) bit-exactly through this oracle.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libvp_oracle.so")

OK, ERR_ARG, ERR_VL, ERR_MODULUS, ERR_ROOT, ERR_DECODE, ERR_BANK, ERR_ADDR, ERR_AUT, ERR_NOBREAK = range(10)
SPM, KSK = 0, 1
ROW_BYTES = 1024
LANES = 128


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (``make -C oracle``)."""
    src = os.path.join(_HERE, "vp_oracle.c")
    stale = (not os.path.exists(_SO)) or os.path.getmtime(_SO) < max(
        os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "vp_oracle.h")))
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _SO


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        # idle OpenMP workers must sleep, not spin: a team of one-per-logical-CPU threads spinning behind a one-limb ``ntt`` call
        # burns the container's CPU quota and the kernel then throttles the WHOLE process for tens of milliseconds -- seen as
        # 20-80 ms pauses between GPU launches in whatever is timed next (profiles/r03_z_launch_gaps.txt)
        os.environ.setdefault("OMP_WAIT_POLICY", "passive")
        L = C.CDLL(_SO)
        u64, u32, p = C.c_uint64, C.c_uint32, C.c_void_p
        L.orc_mulmod.restype = u64; L.orc_mulmod.argtypes = [u64, u64, u64]
        L.orc_powmod.restype = u64; L.orc_powmod.argtypes = [u64, u64, u64]
        L.orc_invmod.restype = u64; L.orc_invmod.argtypes = [u64, u64]
        L.orc_is_prime.restype = C.c_int; L.orc_is_prime.argtypes = [u64]
        L.orc_barrett_rtl.restype = u64; L.orc_barrett_rtl.argtypes = [u64, u64, u64, u64]
        L.orc_barrett_mu.restype = u64; L.orc_barrett_mu.argtypes = [u64]
        L.orc_half.restype = u64; L.orc_half.argtypes = [u64, u64]
        L.orc_alu.restype = C.c_int
        L.orc_alu.argtypes = [C.c_uint, u64, u64, u64, u64, C.POINTER(u64), C.POINTER(u64)]
        L.orc_find_psi.restype = C.c_int; L.orc_find_psi.argtypes = [u64, C.c_uint, C.POINTER(u64)]
        L.orc_nth_prime.restype = u64; L.orc_nth_prime.argtypes = [C.c_uint, C.c_uint]
        L.orc_make_tables.restype = None; L.orc_make_tables.argtypes = [u64, u64, C.c_uint, p, p]
        L.orc_ntt.restype = None; L.orc_ntt.argtypes = [p, C.c_uint, u64, p]
        L.orc_intt.restype = None; L.orc_intt.argtypes = [p, C.c_uint, u64, p]
        L.orc_vaut.restype = None; L.orc_vaut.argtypes = [p, p, C.c_uint, u64, u64]
        L.orc_vautn.restype = None; L.orc_vautn.argtypes = [p, p, C.c_uint, u64]
        L.orc_vautz.restype = None; L.orc_vautz.argtypes = [p, p, C.c_uint, u64, u64]
        L.orc_set_threads.restype = C.c_int; L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_threads(1)                                     # one thread unless a caller asks for more (set_threads)
        L.orc_vroli.restype = None; L.orc_vroli.argtypes = [p, p, C.c_uint, u64]
        L.orc_ntt_batch.restype = C.c_int; L.orc_ntt_batch.argtypes = [p, C.c_uint, u64, C.c_uint, C.c_int]
        L.orc_vp_create.restype = p; L.orc_vp_create.argtypes = [u64, u64]
        L.orc_vp_set_iram_depth.restype = C.c_int; L.orc_vp_set_iram_depth.argtypes = [p, C.c_uint32]
        L.orc_vp_destroy.restype = None; L.orc_vp_destroy.argtypes = [p]
        L.orc_vp_write.restype = C.c_int; L.orc_vp_write.argtypes = [p, C.c_int, u64, p, u64]
        L.orc_vp_read.restype = C.c_int; L.orc_vp_read.argtypes = [p, C.c_int, u64, p, u64]
        L.orc_vp_load_isram.restype = C.c_int; L.orc_vp_load_isram.argtypes = [p, u32, p, u32]
        L.orc_vp_run.restype = C.c_int; L.orc_vp_run.argtypes = [p, u32, u64, u64, u64, u64, u64]
        L.orc_vp_peek_vreg.restype = C.c_int; L.orc_vp_peek_vreg.argtypes = [p, C.c_uint, p, u64]
        L.orc_vp_instret.restype = u64; L.orc_vp_instret.argtypes = [p]
        _lib = L
    return _lib


def _u64(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


# ---- scalar helpers ------------------------------------------------------------------
def mulmod(a, b, q): return int(lib().orc_mulmod(a, b, q))
def powmod(a, e, q): return int(lib().orc_powmod(a, e, q))
def invmod(a, q): return int(lib().orc_invmod(a, q))
def is_prime(n): return bool(lib().orc_is_prime(n))
def barrett_rtl(a, b, q, mu): return int(lib().orc_barrett_rtl(a, b, q, mu))
def barrett_mu(q): return int(lib().orc_barrett_mu(q))
def nth_prime(logn, idx): return int(lib().orc_nth_prime(logn, idx))


def alu(opcode, a, b, s, q):
    r0, r1 = C.c_uint64(), C.c_uint64()
    rc = lib().orc_alu(opcode, a, b, s, q, C.byref(r0), C.byref(r1))
    if rc:
        raise ValueError(f"orc_alu rc={rc}")
    return int(r0.value), int(r1.value)


def find_psi(q, logn):
    psi = C.c_uint64()
    rc = lib().orc_find_psi(q, logn, C.byref(psi))
    if rc:
        raise ValueError(f"orc_find_psi rc={rc}")
    return int(psi.value)


def make_tables(q, psi, logn):
    n = 1 << logn
    tab = np.empty(n, np.uint64); itab = np.empty(n, np.uint64)
    lib().orc_make_tables(q, psi, logn, tab.ctypes.data, itab.ctypes.data)
    return tab, itab


# ---- vector kernels ------------------------------------------------------------------
def ntt(a, q, inverse=False):
    """Forward/inverse negacyclic NTT of one or more limbs (last axis = N)."""
    a = _u64(a).copy()
    n = a.shape[-1]
    logn = n.bit_length() - 1
    assert 1 << logn == n
    rc = lib().orc_ntt_batch(a.ctypes.data, logn, q, a.size // n, int(bool(inverse)))
    if rc:
        raise ValueError(f"orc_ntt_batch rc={rc}")
    return a


def intt(a, q): return ntt(a, q, inverse=True)


def set_threads(n: int = 0) -> int:
    """OpenMP threads used by batched ``ntt`` calls (limbs are independent); 0 = all cores."""
    return int(lib().orc_set_threads(n))


def vaut(a, q, k):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_vaut(out.ctypes.data, a.ctypes.data, a.size.bit_length() - 1, q, k)
    return out


def vautn(a, k):
    """Extension: the automorphism of ``vaut`` on an NTT-form limb (a permutation, no modulus)."""
    a = _u64(a); out = np.empty_like(a)
    lib().orc_vautn(out.ctypes.data, a.ctypes.data, a.size.bit_length() - 1, k)
    return out


def vautz(a, q, k):
    """Extension: a zero at a sign-flipped target of the automorphism k becomes q (what ``vaut`` stores there)."""
    a = _u64(a); out = np.empty_like(a)
    lib().orc_vautz(out.ctypes.data, a.ctypes.data, a.size.bit_length() - 1, q, k)
    return out


def vroli(a, s):
    a = _u64(a); out = np.empty_like(a)
    lib().orc_vroli(out.ctypes.data, a.ctypes.data, a.size.bit_length() - 1, s)
    return out


def elementwise(op, a, b, s, q):
    """Apply ALU opcode ``op`` (modalu.sv:22-37) element-wise; b may be None."""
    a = _u64(a)
    out = np.empty_like(a)
    bb = _u64(b) if b is not None else np.zeros_like(a)
    L = lib()
    r0 = C.c_uint64()
    for i in range(a.size):
        L.orc_alu(op, int(a.flat[i]), int(bb.flat[i]), s, q, C.byref(r0), None)
        out.flat[i] = r0.value
    return out


# ---- the VP machine ------------------------------------------------------------------
class OracleVP:
    """The VP as an ISA machine over a flat row-addressed memory (same call surface as
    ``aloha_amd.vp.VP`` so parity tests can drive both with identical code)."""

    def __init__(self, spm_rows=1 << 16, ksk_rows=1 << 16, iram_depth=4096):
        self._h = lib().orc_vp_create(spm_rows, ksk_rows)
        if not self._h:
            raise MemoryError("orc_vp_create")
        self.iram_depth = iram_depth
        if iram_depth != 4096:
            self._chk(lib().orc_vp_set_iram_depth(self._h, iram_depth))

    def close(self):
        if self._h:
            lib().orc_vp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _chk(rc):
        if rc:
            raise RuntimeError(f"oracle VP error {rc}")

    def write(self, space, row, data):
        data = _u64(data)
        self._chk(lib().orc_vp_write(self._h, space, row, data.ctypes.data, data.nbytes))

    def read(self, space, row, nwords):
        out = np.empty(nwords, np.uint64)
        self._chk(lib().orc_vp_read(self._h, space, row, out.ctypes.data, out.nbytes))
        return out

    def load_isram(self, base_pc, words):
        w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 3)
        self._chk(lib().orc_vp_load_isram(self._h, base_pc, w.ctypes.data, w.shape[0]))

    def run(self, pc, src0=0, src1=0, rslt=0, ksk=0, step=0):
        rc = lib().orc_vp_run(self._h, pc, src0, src1, rslt, ksk, step)
        return rc

    def sync(self):
        pass

    def encode(self, space, dst_row, re_im, log_n, moduli, scale=2.0 ** 38):
        """The encoder block (oracle/ckks_encode.py), same call as ``aloha_amd.vp.VP.encode``."""
        from . import ckks_encode
        z = np.ascontiguousarray(re_im, dtype=np.float64).reshape(-1, 1 << log_n)
        out = np.concatenate([ckks_encode.encode(zz, log_n, moduli, scale).reshape(-1) for zz in z])
        self.write(space, dst_row, out)

    def peek_vreg(self, r, n):
        out = np.empty(n, np.uint64)
        self._chk(lib().orc_vp_peek_vreg(self._h, r, out.ctypes.data, n))
        return out

    @property
    def instret(self):
        return int(lib().orc_vp_instret(self._h))
