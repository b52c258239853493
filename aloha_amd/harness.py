"""Host harness: the reference system test bench's tasks, restated over the C-ABI engine.

Mirrors ``sim/top/top_noaxilite_tb.sv``: the op-list format (``parse_op`` :249-298,
``parse_op_list`` :350-370), ``load_ksk`` (:372-394), ``run_vp`` (:396-417), ``run_encode``
(:419-448), ``run_load_cipher`` / ``run_store_cipher`` (:450-496), ``run_mul_plain`` /
``run_hom_add`` / ``run_rotate`` (:522-532) and the per-op dump (``run`` :596-638).

The AXI DMA and DDR model are replaced by host arrays.  ``encode`` runs the encoder block
(``vp.encode``: the reference's software-model convention in double precision, see
``include/aloha_vp.h``) on a cleartext registered in ``dram_cleartext``; to replay the RTL's
own chains -- whose single-precision encoder differs from the model in every coefficient -- it
instead takes the encoder's *output* from ``dram_encoder`` (two limbs, what the test bench
dumps as ``inst_K_0_out.txt``).  Either way ``encode_post`` follows, as in ``run_encode``
(:446-447).

``vp`` is any object with ``write/read/load_isram/run/sync`` (``aloha_amd.vp.VP``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence

import numpy as np

from . import isa

SPM, KSK = 0, 1
LANES = 128

OP_LOAD, OP_STORE, OP_ENCODE, OP_ENCODE_POST, OP_MUL_PLAIN, OP_HOM_ADD, OP_ROTATE = 1, 2, 3, 4, 5, 6, 7


@dataclass
class Op:
    """One op-list entry (top_noaxilite_tb.sv:249-298)."""
    op: int
    spm_addr: int = 0       # destination row, args[0][13:0]
    src1: int = 0
    src2: int = 0
    step: int = 0
    dram_addr: int = 0


def parse_op(args: Sequence[int]) -> Op:
    op = (args[0] >> 28) & 0xF
    spm = args[0] & 0x3FFF
    if op in (OP_LOAD, OP_STORE, OP_ENCODE):
        return Op(op, spm, dram_addr=(args[1] << 32) | args[2])
    if op in (OP_ENCODE_POST, OP_MUL_PLAIN, OP_HOM_ADD):
        return Op(op, spm, src1=args[1] & 0x3FFF, src2=args[2] & 0x3FFF)
    if op == OP_ROTATE:
        return Op(op, spm, step=args[1] & 0x3FFF, src1=args[2] & 0x3FFF)
    raise ValueError(f"unknown op {op} in op list")  # the TB leaves this as a TODO (:291-293)


def parse_op_list(text: str) -> List[Op]:
    """``%h,%h,%h`` per line (top_noaxilite_tb.sv:361)."""
    ops = []
    for line in text.splitlines():
        line = line.strip()
        if line:
            ops.append(parse_op([int(t, 16) for t in line.split(",")]))
    return ops


def clog2(x: int) -> int:
    return (x - 1).bit_length()


class TestBench:
    __test__ = False  # not a pytest class

    def __init__(self, vp, n: int = 8192, limbs: int = 2, specials: int = 1,
                 pcs: Optional[Dict[str, int]] = None, moduli: Optional[Sequence[int]] = None,
                 scale: float = 2.0 ** 38):
        self.vp = vp
        self.n = n
        self.limbs = limbs
        self.specials = specials
        self.rows_per_limb = n // LANES
        self.ct_words = 2 * limbs * n           # DMA_LOAD_POLY_NUM * POLY_ELEMENT_NUM (:43,:456)
        self.ksk_limbs = (limbs + specials) * limbs * 2
        self.pcs = pcs or {"encode_post": isa.PC_ENCODE_POST, "mul_plain": isa.PC_MUL_PLAIN,
                           "hom_add": isa.PC_HOM_ADD, "keyswitch": isa.PC_KEYSWITCH}
        self.dram_ct: Dict[int, np.ndarray] = {}        # DRAM_VP_BASE + dram_addr -> ciphertext
        self.dram_encoder: Dict[int, np.ndarray] = {}   # encode slot -> encoder output (2 limbs)
        self.dram_cleartext: Dict[int, np.ndarray] = {} # encode slot -> N doubles, (re, im) interleaved slots
        self.moduli = list(moduli) if moduli is not None else None   # ciphertext primes, for the encoder block
        self.scale = scale
        self.dump: Optional[Callable[[int, str, np.ndarray], None]] = None

    # -- init (:685-716) -------------------------------------------------------------------
    def load_programs(self, images: Dict[str, np.ndarray]) -> None:
        for name, words in images.items():
            self.vp.load_isram(self.pcs[name], words)

    def load_ksk(self, keys: Sequence[np.ndarray]) -> None:
        """DMA command 0: the keys for steps 2, 4, 8, ... back to back from KSK row 0."""
        row = 0
        for k in keys:
            k = np.ascontiguousarray(k, dtype=np.uint64)
            assert k.size == self.ksk_limbs * self.n
            self.vp.write(KSK, row, k)
            row += self.ksk_limbs * self.rows_per_limb

    # -- tasks ---------------------------------------------------------------------------
    def run_vp(self, dest: int, src1: int, src2: int, step: int, ksk_ptr: int, pc: int) -> None:
        rc = self.vp.run(pc, src0=src1, src1=src2, rslt=dest, ksk=ksk_ptr, step=step)
        if rc:
            raise RuntimeError(f"vp.run(pc={pc}) failed with status {rc}: {getattr(self.vp, 'last_error', '')}")

    def run_load_cipher(self, dest: int, dram_addr: int) -> None:
        self.vp.write(SPM, dest, self.dram_ct[dram_addr])

    def run_store_cipher(self, dram_addr: int, src: int) -> None:
        self.dram_ct[dram_addr] = self.copy_spm(src)

    def copy_spm(self, row: int) -> np.ndarray:
        self.vp.sync()
        return self.vp.read(SPM, row, self.ct_words)

    def run_encode(self, inst_id: int, dest: int, dram_addr: int) -> None:
        if dram_addr in self.dram_encoder:
            enc = np.ascontiguousarray(self.dram_encoder[dram_addr], dtype=np.uint64)
            assert enc.size == self.limbs * self.n
            self.vp.write(SPM, dest, enc)
        else:                                                                    # DMA_CMD_ENCODE (:424-431)
            if self.moduli is None:
                raise ValueError("TestBench(moduli=...) is needed to run the encoder block")
            z = np.ascontiguousarray(self.dram_cleartext[dram_addr], dtype=np.float64)
            assert z.size == self.n
            self.vp.encode(SPM, dest, z, self.n.bit_length() - 1, self.moduli[:self.limbs], self.scale)
        if self.dump:
            self.dump(inst_id, "_0", self.copy_spm(dest))                       # dump_sub_poly (:444-445)
        self.run_vp(dest, dest, 0, 0, 0, self.pcs["encode_post"])               # :447

    def run_mul_plain(self, dest: int, src1: int, src2: int) -> None:
        self.run_vp(dest, src1, src2, 0, 0, self.pcs["mul_plain"])

    def run_hom_add(self, dest: int, src1: int, src2: int) -> None:
        self.run_vp(dest, src1, src2, 0, 0, self.pcs["hom_add"])

    def run_rotate(self, dest: int, src: int, step: int) -> None:
        # k = 3^step mod 2N; key slot = clog2(step) - 1 (:531)
        slot = clog2(step) - 1
        if slot < 0:
            raise ValueError("rotate step must be >= 2 (key slots exist for steps 2, 4, 8, ...)")
        k = pow(3, step, 2 * self.n)
        self.run_vp(dest, src, 0, k, slot * self.ksk_limbs * self.rows_per_limb, self.pcs["keyswitch"])

    # -- run (:596-638) ---------------------------------------------------------------------
    def run(self, ops: Sequence[Op]) -> None:
        for i, op in enumerate(ops):
            if op.op == OP_LOAD:
                self.run_load_cipher(op.spm_addr, op.dram_addr)
            elif op.op == OP_STORE:
                self.run_store_cipher(op.dram_addr, op.spm_addr)
                if self.dump:
                    self.dump(i, "", self.dram_ct[op.dram_addr])
                continue
            elif op.op == OP_ENCODE:
                self.run_encode(i, op.spm_addr, op.dram_addr)
            elif op.op == OP_MUL_PLAIN:
                self.run_mul_plain(op.spm_addr, op.src1, op.src2)
            elif op.op == OP_HOM_ADD:
                self.run_hom_add(op.spm_addr, op.src1, op.src2)
            elif op.op == OP_ROTATE:
                self.run_rotate(op.spm_addr, op.src1, op.step)
            else:
                raise ValueError(f"op {op.op} not runnable")
            if self.dump:
                self.dump(i, "", self.copy_spm(op.spm_addr))
