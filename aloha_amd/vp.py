"""ctypes binding of libaloha_vp.so (the C ABI in include/aloha_vp.h).

``VP`` mirrors the reference's VP boundary -- instruction ROM, six CSRs + start/done, two
row-addressed memories (``src/vp/top/vp_top_full.sv:15-41``; test-bench tasks
``sim/top/top_noaxilite_tb.sv:372-520``) -- with the same call surface as the test oracle's
``OracleVP`` so parity tests drive both with identical code.

There is NO CPU fallback: if the HIP library is missing, or no gfx950 device is present,
construction raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ALOHA_VP_LIB") or os.path.join(_HERE, "lib", "libaloha_vp.so")

SPM, KSK = 0, 1
ROW_BYTES = 1024
LANES = 128
FLAG_NO_FUSION = 1
FLAG_RADIX2_ONLY = 2
FLAG_NTT_TWO_PASS = 4
FLAG_NTT_SINGLE_14 = 8

STATUS = ["OK", "ERR_ARG", "ERR_VL", "ERR_MODULUS", "ERR_ROOT", "ERR_DECODE", "ERR_BANK", "ERR_ADDR",
          "ERR_AUT", "ERR_NOBREAK", "ERR_DEVICE", "ERR_NOMEM"]
(OK, ERR_ARG, ERR_VL, ERR_MODULUS, ERR_ROOT, ERR_DECODE, ERR_BANK, ERR_ADDR, ERR_AUT, ERR_NOBREAK,
 ERR_DEVICE, ERR_NOMEM) = range(12)

# every symbol include/aloha_vp.h declares
EXPORTS = ["aloha_vp_abi_version", "aloha_status_string", "aloha_vp_create", "aloha_vp_destroy",
           "aloha_vp_last_error", "aloha_vp_write", "aloha_vp_read", "aloha_vp_device_ptr",
           "aloha_vp_load_isram", "aloha_vp_load_isram_hex", "aloha_vp_run", "aloha_vp_sync",
           "aloha_vp_ntt_rows", "aloha_find_psi", "aloha_nth_prime", "aloha_vp_get_stats",
           "aloha_vp_peek_vreg", "aloha_vp_encode", "aloha_vp_decode", "aloha_vp_fill_synthetic",
           "aloha_vp_valu_clock", "aloha_vp_get_config", "aloha_vp_set_scratch"]


class AlohaError(RuntimeError):
    def __init__(self, status: int, msg: str = ""):
        self.status = status
        name = STATUS[status] if 0 <= status < len(STATUS) else str(status)
        super().__init__(f"aloha_vp: {name}{': ' + msg if msg else ''}")


class _Config(C.Structure):
    _fields_ = [("abi_version", C.c_uint32), ("device", C.c_int32), ("spm_rows", C.c_uint64),
                ("ksk_rows", C.c_uint64), ("max_log_n", C.c_uint32), ("flags", C.c_uint32),
                ("stream", C.c_void_p), ("iram_depth", C.c_uint32), ("reserved", C.c_uint32)]


class _Stats(C.Structure):
    _fields_ = [("instructions", C.c_uint64), ("kernel_launches", C.c_uint64), ("twiddle_builds", C.c_uint64),
                ("fused_launches", C.c_uint64), ("graph_launches", C.c_uint64),
                ("twiddle_generations", C.c_uint64), ("ks_fused_launches", C.c_uint64), ("product_fallbacks", C.c_uint64), ("flow_runs", C.c_uint64)]


_lib: Optional[C.CDLL] = None


def load_library() -> C.CDLL:
    """Load libaloha_vp.so.  torch is imported first so that the process holds exactly one HIP
    runtime (torch bundles a libamdhip64.so.7 of its own; whichever is loaded first serves both)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AlohaError(ERR_DEVICE, f"{LIB_PATH} is missing -- run `python -m aloha_amd.build` "
                                     "(the engine has no CPU fallback)")
    try:
        import torch  # noqa: F401  (load order matters, see docstring)
    except Exception:  # torch is plumbing, not a dependency of the library itself
        pass
    L = C.CDLL(LIB_PATH)
    u64, u32, p, i = C.c_uint64, C.c_uint32, C.c_void_p, C.c_int
    L.aloha_vp_abi_version.restype = u32
    L.aloha_status_string.restype = C.c_char_p; L.aloha_status_string.argtypes = [i]
    L.aloha_vp_create.restype = i; L.aloha_vp_create.argtypes = [C.POINTER(_Config), C.POINTER(p)]
    L.aloha_vp_destroy.restype = None; L.aloha_vp_destroy.argtypes = [p]
    L.aloha_vp_last_error.restype = C.c_char_p; L.aloha_vp_last_error.argtypes = [p]
    L.aloha_vp_write.restype = i; L.aloha_vp_write.argtypes = [p, i, u64, p, u64]
    L.aloha_vp_read.restype = i; L.aloha_vp_read.argtypes = [p, i, u64, p, u64]
    L.aloha_vp_device_ptr.restype = p; L.aloha_vp_device_ptr.argtypes = [p, i, u64]
    L.aloha_vp_load_isram.restype = i; L.aloha_vp_load_isram.argtypes = [p, u32, p, u32]
    L.aloha_vp_load_isram_hex.restype = i; L.aloha_vp_load_isram_hex.argtypes = [p, u32, C.c_char_p]
    L.aloha_vp_run.restype = i; L.aloha_vp_run.argtypes = [p, u32, u64, u64, u64, u64, u64]
    L.aloha_vp_sync.restype = i; L.aloha_vp_sync.argtypes = [p]
    L.aloha_vp_ntt_rows.restype = i; L.aloha_vp_ntt_rows.argtypes = [p, i, u64, u64, u32, p, u32, i]
    L.aloha_find_psi.restype = i; L.aloha_find_psi.argtypes = [u64, u32, C.POINTER(u64)]
    L.aloha_nth_prime.restype = u64; L.aloha_nth_prime.argtypes = [u32, u32]
    L.aloha_vp_get_stats.restype = i; L.aloha_vp_get_stats.argtypes = [p, C.POINTER(_Stats)]
    L.aloha_vp_peek_vreg.restype = i; L.aloha_vp_peek_vreg.argtypes = [p, u32, p, u64]
    L.aloha_vp_encode.restype = i; L.aloha_vp_encode.argtypes = [p, i, u64, p, u32, u32, p, u32, C.c_double]
    L.aloha_vp_decode.restype = i; L.aloha_vp_decode.argtypes = [p, i, u64, u64, p, u32, u32, u64, C.c_double]
    L.aloha_vp_fill_synthetic.restype = i; L.aloha_vp_fill_synthetic.argtypes = [p, i, u64, u32, p, u32, u64, u64]
    L.aloha_vp_valu_clock.restype = i; L.aloha_vp_valu_clock.argtypes = [p, C.c_double, C.POINTER(C.c_double)]
    L.aloha_vp_get_config.restype = i; L.aloha_vp_get_config.argtypes = [p, C.POINTER(_Config)]
    L.aloha_vp_set_scratch.restype = i; L.aloha_vp_set_scratch.argtypes = [p, u64, u64]
    _lib = L
    return L


def find_psi(q: int, log_n: int) -> int:
    out = C.c_uint64()
    rc = load_library().aloha_find_psi(q, log_n, C.byref(out))
    if rc:
        raise AlohaError(rc)
    return int(out.value)


def nth_prime(log_n: int, idx: int) -> int:
    return int(load_library().aloha_nth_prime(log_n, idx))


class VP:
    """One VP engine on one GPU (one process per GPU; handles are single-threaded)."""

    def __init__(self, spm_rows: int = 1 << 16, ksk_rows: int = 1 << 16, max_log_n: int = 16,
                 device: int = 0, flags: int = 0, stream: int = 0, iram_depth: int = 4096):
        """``iram_depth``: instruction ROM words -- 4096 as in the reference (vp_defines.vh:31), up to 65536 for programs that
        do not fit (a whole key-switch phase at L = 32)."""
        L = load_library()
        cfg = _Config(abi_version=L.aloha_vp_abi_version(), device=device, spm_rows=spm_rows,
                      ksk_rows=ksk_rows, max_log_n=max_log_n, flags=flags, stream=stream or None, iram_depth=iram_depth)
        self.iram_depth = iram_depth
        h = C.c_void_p()
        rc = L.aloha_vp_create(C.byref(cfg), C.byref(h))
        if rc:
            raise AlohaError(rc, "aloha_vp_create failed (needs a gfx950 device; there is no CPU path)")
        self._h = h
        self._L = L
        self.spm_rows, self.ksk_rows = spm_rows, ksk_rows

    # -- lifecycle --------------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.aloha_vp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def last_error(self) -> str:
        return self._L.aloha_vp_last_error(self._h).decode()

    def _chk(self, rc: int) -> None:
        if rc:
            raise AlohaError(rc, self.last_error)

    # -- memory -----------------------------------------------------------------------------
    def write(self, space: int, row: int, data) -> None:
        a = np.ascontiguousarray(data, dtype=np.uint64)
        self._chk(self._L.aloha_vp_write(self._h, space, row, a.ctypes.data, a.nbytes))

    def read(self, space: int, row: int, nwords: int) -> np.ndarray:
        out = np.empty(nwords, dtype=np.uint64)
        self._chk(self._L.aloha_vp_read(self._h, space, row, out.ctypes.data, out.nbytes))
        return out

    def device_ptr(self, space: int, row: int) -> int:
        p = self._L.aloha_vp_device_ptr(self._h, space, row)
        if not p:
            raise AlohaError(ERR_ADDR, "row outside the space")
        return int(p)

    # -- program ----------------------------------------------------------------------------
    def load_isram(self, base_pc: int, words) -> None:
        w = np.ascontiguousarray(words, dtype=np.uint32).reshape(-1, 3)
        self._chk(self._L.aloha_vp_load_isram(self._h, base_pc, w.ctypes.data, w.shape[0]))

    def load_isram_hex(self, base_pc: int, text: str) -> None:
        self._chk(self._L.aloha_vp_load_isram_hex(self._h, base_pc, text.encode()))

    def run(self, pc: int, src0: int = 0, src1: int = 0, rslt: int = 0, ksk: int = 0, step: int = 0) -> int:
        """Start the program at ``pc`` with the six CSRs; returns the status code (0 = ok)
        like the oracle's ``run`` so error behaviour can be compared."""
        return int(self._L.aloha_vp_run(self._h, pc, src0, src1, rslt, ksk, step))

    def sync(self) -> None:
        self._chk(self._L.aloha_vp_sync(self._h))

    def set_scratch(self, row: int = 0, nrows: int = 0) -> None:
        """Declare SPM rows [row, row + nrows) as the caller's scratch: rows of the window that a run writes are
        undefined once that run has returned (``nrows = 0``: no window, the default)."""
        self._chk(self._L.aloha_vp_set_scratch(self._h, row, nrows))

    def config(self) -> dict:
        c = _Config()
        self._chk(self._L.aloha_vp_get_config(self._h, C.byref(c)))
        return {"device": int(c.device), "spm_rows": int(c.spm_rows), "ksk_rows": int(c.ksk_rows),
                "max_log_n": int(c.max_log_n), "flags": int(c.flags), "iram_depth": int(c.iram_depth)}

    # -- batched transforms -------------------------------------------------------------------
    def ntt_rows(self, space: int, dst_row: int, src_row: int, log_n: int, moduli: Sequence[int],
                 inverse: bool = False) -> None:
        m = np.ascontiguousarray(moduli, dtype=np.uint64)
        self._chk(self._L.aloha_vp_ntt_rows(self._h, space, dst_row, src_row, log_n, m.ctypes.data,
                                            m.size, int(bool(inverse))))

    # -- CKKS encoder block (src/encoder; run_encode's DMA command, top_noaxilite_tb.sv:419-448) -----
    def encode(self, space: int, dst_row: int, re_im, log_n: int, moduli: Sequence[int],
               scale: float = 2.0 ** 38) -> None:
        """``re_im``: one or more cleartexts of N doubles (N/2 slots, (re, im) interleaved) ->
        ``len(moduli)`` coefficient-form limbs per cleartext at ``dst_row``."""
        z = np.ascontiguousarray(re_im, dtype=np.float64).reshape(-1, 1 << log_n)
        m = np.ascontiguousarray(moduli, dtype=np.uint64)
        self._chk(self._L.aloha_vp_encode(self._h, space, dst_row, z.ctypes.data, z.shape[0], log_n,
                                          m.ctypes.data, m.size, float(scale)))

    def decode(self, space: int, src_row: int, log_n: int, modulus: int, nplain: int = 1, limb_stride: int = 1,
               scale: float = 2.0 ** 38) -> np.ndarray:
        """Slots ((re, im) interleaved, N doubles per plaintext) from one coefficient-form limb per plaintext."""
        out = np.empty((nplain, 1 << log_n), dtype=np.float64)
        self._chk(self._L.aloha_vp_decode(self._h, space, src_row, limb_stride, out.ctypes.data, nplain, log_n,
                                          modulus, float(scale)))
        return out

    # -- measurement helpers ---------------------------------------------------------------------
    def fill_synthetic(self, space: int, row: int, log_n: int, moduli: Sequence[int], seed_base: int = 0xA10A,
                       first_index: int = 0) -> None:
        """Limb i at row + i*(N/128): uniform residues below moduli[i] from splitmix64(seed_base + first_index + i)
        (SURVEY 8d), generated on the device."""
        m = np.ascontiguousarray(moduli, dtype=np.uint64)
        self._chk(self._L.aloha_vp_fill_synthetic(self._h, space, row, log_n, m.ctypes.data, len(m), seed_base, first_index))

    def valu_clock_ghz(self, ms: float = 60.0) -> float:
        out = C.c_double()
        self._chk(self._L.aloha_vp_valu_clock(self._h, float(ms), C.byref(out)))
        return float(out.value)

    # -- introspection ------------------------------------------------------------------------
    def stats(self) -> dict:
        s = _Stats()
        self._chk(self._L.aloha_vp_get_stats(self._h, C.byref(s)))
        return {"instructions": int(s.instructions), "kernel_launches": int(s.kernel_launches),
                "twiddle_builds": int(s.twiddle_builds), "fused_launches": int(s.fused_launches),
                "graph_launches": int(s.graph_launches), "twiddle_generations": int(s.twiddle_generations),
                "ks_fused_launches": int(s.ks_fused_launches), "product_fallbacks": int(s.product_fallbacks), "flow_runs": int(s.flow_runs)}

    def peek_vreg(self, reg: int, n: int) -> np.ndarray:
        out = np.empty(n, dtype=np.uint64)
        self._chk(self._L.aloha_vp_peek_vreg(self._h, reg, out.ctypes.data, n))
        return out
