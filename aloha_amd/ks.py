"""ctypes binding of the C host's key-switch (``aloha_ks_*``, include/aloha_vp.h; csrc/keyswitch_host.cpp).

The C++ host generates the phase programs, owns the slot layout and issues the two exchanges as ``ncclAllGather`` on
the engine's stream; this class only marshals arguments.  It is the ONE generator: ``aloha_amd/hybrid.py`` fetches the words of every phase
through ``program()`` below and drives the oracle's ISA machine (and the Python rehearsals of the exchanges) with them.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import numpy as np

from . import vp as avp


class _KsConfig(C.Structure):
    _fields_ = [("log_n", C.c_uint32), ("L", C.c_uint32), ("K", C.c_uint32), ("alpha", C.c_uint32), ("world", C.c_uint32),
                ("rank", C.c_uint32), ("batch", C.c_uint32), ("flags", C.c_uint32), ("moduli", C.POINTER(C.c_uint64)),
                ("specials", C.POINTER(C.c_uint64)), ("base_row", C.c_uint64), ("rom_depth", C.c_uint32), ("reserved", C.c_uint32)]


class _KsInfo(C.Structure):
    _fields_ = [("dnum", C.c_uint32), ("owned_ct", C.c_uint32), ("owned_all", C.c_uint32), ("rows_per_limb", C.c_uint32),
                ("resident", C.c_uint32), ("pad", C.c_uint32), ("spm_rows", C.c_uint64), ("ksk_rows", C.c_uint64),
                ("bytes_sent", C.c_uint64), ("bytes_received", C.c_uint64), ("key_rows", C.c_uint64)]


EXPORTS = ["aloha_ks_rows_needed", "aloha_ks_create", "aloha_ks_destroy", "aloha_ks_last_error", "aloha_ks_info", "aloha_ks_owned",
           "aloha_ks_program", "aloha_ks_nccl_unique_id", "aloha_ks_comm_init", "aloha_ks_set_comm", "aloha_ks_set_pairwise", "aloha_ks_load_key",
           "aloha_ks_load_ct", "aloha_ks_read_out", "aloha_ks_rotate", "aloha_ks_multiply", "aloha_ks_load_key_hoisted", "aloha_ks_hoist",
           "aloha_ks_rotate_hoisted", "aloha_ks_rotate_hoisted_many", "aloha_ks_rescale", "aloha_ks_stage", "aloha_ks_gather_buffer", "aloha_vp_stream", "aloha_vp_device"]
PHASES = ("rot1", "mul1", "mid", "terms", "fin_rot", "fin_mul", "hoist1", "modup", "inner_h", "fin_h", "rescale1", "rescale2")
_bound = False


def _lib():
    global _bound
    L = avp.load_library()
    if not _bound:
        p, i, u32, u64 = C.c_void_p, C.c_int, C.c_uint32, C.c_uint64
        L.aloha_ks_rows_needed.restype = i; L.aloha_ks_rows_needed.argtypes = [C.POINTER(_KsConfig), C.POINTER(u64), C.POINTER(u64)]
        L.aloha_ks_create.restype = i; L.aloha_ks_create.argtypes = [p, C.POINTER(_KsConfig), C.POINTER(p)]
        L.aloha_ks_destroy.restype = None; L.aloha_ks_destroy.argtypes = [p]
        L.aloha_ks_last_error.restype = C.c_char_p; L.aloha_ks_last_error.argtypes = [p]
        L.aloha_ks_info.restype = i; L.aloha_ks_info.argtypes = [p, C.POINTER(_KsInfo)]
        L.aloha_ks_owned.restype = i; L.aloha_ks_owned.argtypes = [p, p, p]
        L.aloha_ks_program.restype = i; L.aloha_ks_program.argtypes = [p, C.c_char_p, u32, p, C.POINTER(u32)]
        L.aloha_ks_nccl_unique_id.restype = i; L.aloha_ks_nccl_unique_id.argtypes = [p]
        L.aloha_ks_comm_init.restype = i; L.aloha_ks_comm_init.argtypes = [p, p]
        L.aloha_ks_set_comm.restype = i; L.aloha_ks_set_comm.argtypes = [p, p]
        L.aloha_ks_set_pairwise.restype = i; L.aloha_ks_set_pairwise.argtypes = [p, i]
        L.aloha_ks_load_key.restype = i; L.aloha_ks_load_key.argtypes = [p, p]
        L.aloha_ks_load_ct.restype = i; L.aloha_ks_load_ct.argtypes = [p, i, u32, p]
        L.aloha_ks_read_out.restype = i; L.aloha_ks_read_out.argtypes = [p, u32, p]
        L.aloha_ks_rotate.restype = i; L.aloha_ks_rotate.argtypes = [p, u64]
        L.aloha_ks_multiply.restype = i; L.aloha_ks_multiply.argtypes = [p]
        L.aloha_ks_load_key_hoisted.restype = i; L.aloha_ks_load_key_hoisted.argtypes = [p, p, u64, u32]
        L.aloha_ks_hoist.restype = i; L.aloha_ks_hoist.argtypes = [p, i]
        L.aloha_ks_rotate_hoisted.restype = i; L.aloha_ks_rotate_hoisted.argtypes = [p, u64, u32]
        L.aloha_ks_rotate_hoisted_many.restype = i; L.aloha_ks_rotate_hoisted_many.argtypes = [p, p, p, u32]
        L.aloha_ks_rescale.restype = i; L.aloha_ks_rescale.argtypes = [p]
        L.aloha_ks_stage.restype = i; L.aloha_ks_stage.argtypes = [p, i, i, u64]
        L.aloha_ks_gather_buffer.restype = i; L.aloha_ks_gather_buffer.argtypes = [p, i, C.POINTER(u64), C.POINTER(u64)]
        L.aloha_vp_stream.restype = p; L.aloha_vp_stream.argtypes = [p]
        L.aloha_vp_device.restype = i; L.aloha_vp_device.argtypes = [p]
        _bound = True
    return L


REFERENCE_PHASE1 = 1


def _config(log_n, moduli, specials, alpha, world, rank, batch, base_row, flags=0, rom_depth=0):
    m = (C.c_uint64 * len(moduli))(*[int(q) for q in moduli])
    s = (C.c_uint64 * len(specials))(*[int(q) for q in specials])
    cfg = _KsConfig(log_n=log_n, L=len(moduli), K=len(specials), alpha=alpha, world=world, rank=rank, batch=batch, flags=flags,
                    moduli=m, specials=s, base_row=base_row, rom_depth=rom_depth)
    return cfg, (m, s)


def rows_needed(log_n: int, moduli: Sequence[int], specials: Sequence[int], alpha: int, world: int = 1, rank: int = 0,
                batch: int = 1, base_row: int = 0):
    """(SPM rows, KSK rows) an ``aloha_vp`` must provide for this rank's share."""
    cfg, _keep = _config(log_n, moduli, specials, alpha, world, rank, batch, base_row)
    a, b = C.c_uint64(), C.c_uint64()
    rc = _lib().aloha_ks_rows_needed(C.byref(cfg), C.byref(a), C.byref(b))
    if rc:
        raise avp.AlohaError(rc, "aloha_ks_rows_needed")
    return int(a.value), int(b.value)


def nccl_unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    rc = _lib().aloha_ks_nccl_unique_id(buf)
    if rc:
        raise avp.AlohaError(rc, "ncclGetUniqueId (librccl not loadable?)")
    return buf.raw


class KeySwitch:
    """One rank's share of the L-limb key-switch, run by the C host.  ``vp=None`` gives a plan (programs only)."""

    def __init__(self, vp: Optional[avp.VP], log_n: int, moduli: Sequence[int], specials: Sequence[int], alpha: int,
                 world: int = 1, rank: int = 0, batch: int = 1, base_row: int = 0, flags: int = 0, rom_depth: int = 0):
        """``rom_depth``: instruction ROM words the phase programs may fill (0: the handle's ROM, or 4096 for a plan)."""
        self._L = _lib()
        cfg, self._keep = _config(log_n, moduli, specials, alpha, world, rank, batch, base_row, flags, rom_depth)
        h = C.c_void_p()
        rc = self._L.aloha_ks_create(vp._h if vp is not None else None, C.byref(cfg), C.byref(h))
        if rc:
            raise avp.AlohaError(rc, "aloha_ks_create")
        self._h, self.vp, self.n, self.L = h, vp, 1 << log_n, len(moduli)
        info = self.info()
        self.dnum, self.Lr, self.Mr = info["dnum"], info["owned_ct"], info["owned_all"]
        ct = (C.c_uint32 * max(1, self.Lr))()
        al = (C.c_uint32 * max(1, self.Mr))()
        self._chk(self._L.aloha_ks_owned(self._h, ct, al))
        self.owned_ct: List[int] = list(ct)[:self.Lr]
        self.owned_all: List[int] = list(al)[:self.Mr]

    def close(self) -> None:
        if getattr(self, "_h", None):
            self._L.aloha_ks_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        if rc:
            raise avp.AlohaError(rc, self._L.aloha_ks_last_error(self._h).decode())

    def info(self) -> dict:
        i = _KsInfo()
        self._chk(self._L.aloha_ks_info(self._h, C.byref(i)))
        return {k: int(getattr(i, k)) for k, _ in _KsInfo._fields_ if k != "pad"}

    def program(self, phase: str, index: int = 0) -> np.ndarray:
        """Instruction words (n, 3) of program ``index`` of a phase; shape (0, 3) when there is no such program."""
        n = C.c_uint32(0)
        self._chk(self._L.aloha_ks_program(self._h, phase.encode(), index, None, C.byref(n)))
        w = np.zeros((n.value, 3), dtype=np.uint32)
        if n.value:
            self._chk(self._L.aloha_ks_program(self._h, phase.encode(), index, w.ctypes.data, C.byref(n)))
        return w

    def comm_init(self, unique_id: bytes) -> None:
        self._chk(self._L.aloha_ks_comm_init(self._h, C.create_string_buffer(unique_id, 128)))

    def set_p2p(self, on: bool = True) -> None:
        self._chk(self._L.aloha_ks_set_pairwise(self._h, int(on)))

    def load_key(self, key_full) -> None:
        k = np.ascontiguousarray(key_full, dtype=np.uint64)
        self._chk(self._L.aloha_ks_load_key(self._h, k.ctypes.data))

    def load_ct(self, ct_full, slot: int = 0) -> None:
        c = np.ascontiguousarray(ct_full, dtype=np.uint64)
        self._chk(self._L.aloha_ks_load_ct(self._h, 0, slot, c.ctypes.data))

    def load_ct_b(self, ct_full, slot: int = 0) -> None:
        c = np.ascontiguousarray(ct_full, dtype=np.uint64)
        self._chk(self._L.aloha_ks_load_ct(self._h, 1, slot, c.ctypes.data))

    def read_out(self, slot: int = 0) -> np.ndarray:
        out = np.empty((2, self.Lr, self.n), dtype=np.uint64)
        self._chk(self._L.aloha_ks_read_out(self._h, slot, out.ctypes.data))
        return out

    def rotate(self, k: int) -> None:
        self._chk(self._L.aloha_ks_rotate(self._h, int(k)))

    def multiply(self) -> None:
        self._chk(self._L.aloha_ks_multiply(self._h))

    # -- hoisted rotations, rescale ---------------------------------------------------------------
    def load_key_hoisted(self, key_full, k: int, region: int) -> None:
        kf = np.ascontiguousarray(key_full, dtype=np.uint64)
        self._chk(self._L.aloha_ks_load_key_hoisted(self._h, kf.ctypes.data, int(k), region))

    def hoist(self, first_slot_only: bool = False) -> None:
        self._chk(self._L.aloha_ks_hoist(self._h, int(first_slot_only)))

    def rotate_hoisted(self, k: int, region: int) -> None:
        self._chk(self._L.aloha_ks_rotate_hoisted(self._h, int(k), region))

    def rotate_hoisted_many(self, steps: Sequence[int], regions: Optional[Sequence[int]] = None) -> None:
        st = (C.c_uint64 * len(steps))(*[int(x) for x in steps])
        rg = (C.c_uint32 * len(steps))(*[int(x) for x in regions]) if regions is not None else None
        self._chk(self._L.aloha_ks_rotate_hoisted_many(self._h, st, rg, len(steps)))

    def rescale(self) -> None:
        self._chk(self._L.aloha_ks_rescale(self._h))

    # -- stage by stage, for hosts that carry the exchanges themselves ------------------------------
    def stage(self, op: str, stage: int, k: int = 0) -> None:
        self._chk(self._L.aloha_ks_stage(self._h, {"rotate": 0, "multiply": 1}[op], stage, int(k)))

    def gather_buffer(self, which: int):
        """(first SPM row, words per rank's chunk) of gather buffer ``which`` (0: digits, 1: mod-down terms)."""
        r, c = C.c_uint64(), C.c_uint64()
        self._chk(self._L.aloha_ks_gather_buffer(self._h, which, C.byref(r), C.byref(c)))
        return int(r.value), int(c.value)
