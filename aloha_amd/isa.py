"""HE-ISA of the ALOHA vector processor: encoder, decoder, assembler, ``.mem`` text I/O.

Instruction word = 96 bit = ``inst[31:0] || imm[63:0]`` (reference
``src/vp/include/vp_defines.vh:20``, field split ``src/vp/sequncer/expander.v:123-130``):

    funct6 = inst[31:26]   mask = inst[25]   vs2 = inst[24:20]   vs1 = inst[19:15]
    funct3 = inst[14:12]   vd   = inst[11:7] opcode = inst[6:0] = 0x0B (custom-0)

The executing decoder is the C++ sequencer in ``csrc/vp_engine.cpp``; this module is the
authoring / inspection tool (assembler for new programs, disassembler, and reader/writer for
the ``$readmemh`` text format of the instruction ROM, ``src/vp/sequncer/inst_rom.v:32-35``:
one 24-hex-digit word per line).
"""
from __future__ import annotations

import re
from dataclasses import dataclass
from typing import Iterable, List, Sequence, Tuple

import numpy as np

OPCODE = 0x0B
IRAM_DEPTH = 4096  # vp_defines.vh:31

# funct6 values: expander.v:65-81
F6 = {
    "nop": 0b000000, "vsetvl": 0b000100, "vsetq": 0b001000, "vsetiq": 0b001100, "break": 0b010000,
    "vfqmul": 0b000001, "vfqadd": 0b000101, "vfqsub": 0b001001, "vfqmod": 0b001101,
    "vcp": 0b010001, "vaut": 0b010101, "vroli": 0b011001,
    "vautn": 0b011101,   # EXTENSION (not in the reference): vaut on an NTT-form limb, a pure permutation
    "vautz": 0b100001,   # EXTENSION: a zero at a sign-flipped target of the automorphism becomes q (vaut's flipped zero, vxu_lane.sv:598)
    "vntt": 0b000010, "vintt": 0b000110, "vle": 0b000011, "vse": 0b000111,
}
F6_NAME = {v: k for k, v in F6.items()}
# funct3: expander.v:83-87
F3_VV, F3_VS, F3_SV = 0, 1, 2
# pointer select in imm[63:48]: src/vp/top/vp_top_full.sv:105-117
PTR = {"src0": 0, "src1": 1, "rslt": 2, "ksk": 15, "abs": 3}
PTR_NAME = {0: "src0", 1: "src1", 2: "rslt", 15: "ksk"}

# Entry points of the shipped micro-programs: sim/top/top_noaxilite_tb.sv:63-66,
# sim/vp/isram_file_generator/isram_file_generator.sv:27-30
PC_ENCODE_POST, PC_MUL_PLAIN, PC_HOM_ADD, PC_KEYSWITCH = 0, 64, 160, 256


@dataclass(frozen=True)
class Inst:
    funct6: int
    vd: int = 0
    vs1: int = 0
    vs2: int = 0
    funct3: int = 0
    imm: int = 0
    mask: int = 1

    @property
    def word(self) -> int:
        return ((self.funct6 & 63) << 26) | ((self.mask & 1) << 25) | ((self.vs2 & 31) << 20) | \
               ((self.vs1 & 31) << 15) | ((self.funct3 & 7) << 12) | ((self.vd & 31) << 7) | OPCODE

    @property
    def name(self) -> str:
        return F6_NAME.get(self.funct6, f"?{self.funct6:06b}")


def decode(inst32: int, imm64: int) -> Inst:
    return Inst(funct6=inst32 >> 26, mask=(inst32 >> 25) & 1, vs2=(inst32 >> 20) & 31,
                vs1=(inst32 >> 15) & 31, funct3=(inst32 >> 12) & 7, vd=(inst32 >> 7) & 31, imm=imm64)


# ---- .mem text ($readmemh) ------------------------------------------------------------
def parse_mem_text(text: str) -> np.ndarray:
    """Hex text -> uint32 array of shape (n, 3) = (inst, imm_hi, imm_lo).  Accepts the
    reference's ``*.mem`` files unchanged (24 hex digits per line, ``//`` comments)."""
    rows = []
    for line in text.splitlines():
        line = line.split("//")[0].strip().replace("_", "")
        if not line:
            continue
        for tok in line.split():
            v = int(tok, 16)
            if v >> 96:
                raise ValueError(f"instruction word wider than 96 bit: {tok}")
            rows.append(((v >> 64) & 0xFFFFFFFF, (v >> 32) & 0xFFFFFFFF, v & 0xFFFFFFFF))
    return np.array(rows, dtype=np.uint32).reshape(-1, 3)


def to_mem_text(words: np.ndarray) -> str:
    w = np.asarray(words, dtype=np.uint32).reshape(-1, 3)
    return "".join(f"{int(a):08x}{int(b):08x}{int(c):08x}\n" for a, b, c in w)


def words_of(insts: Iterable[Inst]) -> np.ndarray:
    rows = [(i.word, (i.imm >> 32) & 0xFFFFFFFF, i.imm & 0xFFFFFFFF) for i in insts]
    return np.array(rows, dtype=np.uint32).reshape(-1, 3)


def insts_of(words: np.ndarray) -> List[Inst]:
    w = np.asarray(words, dtype=np.uint32).reshape(-1, 3)
    return [decode(int(a), (int(b) << 32) | int(c)) for a, b, c in w]


# ---- assembler --------------------------------------------------------------------------
_REG = re.compile(r"^v(\d+)$")


def _reg(tok: str) -> int:
    m = _REG.match(tok)
    if not m or not 0 <= int(m.group(1)) < 32:
        raise ValueError(f"bad vector register {tok!r}")
    return int(m.group(1))


def _int(tok: str) -> int:
    return int(tok.replace("_", ""), 0)


def _addr(tok: str) -> int:
    """``src0+0x20000`` / ``ksk+0`` / ``rslt`` -> imm with the pointer select in [63:48]."""
    base, _, off = tok.partition("+")
    if base not in PTR:
        raise ValueError(f"bad pointer name {base!r}")
    o = _int(off) if off else 0
    if o >> 48:
        raise ValueError("address offset wider than 48 bit")
    return (PTR[base] << 48) | o


def assemble_line(line: str) -> Inst | None:
    line = line.split("#")[0].split("//")[0].strip()
    if not line:
        return None
    mnem, _, rest = line.partition(" ")
    ops = [t.strip() for t in rest.split(",")] if rest.strip() else []
    base, _, suffix = mnem.partition(".")
    if base not in F6:
        raise ValueError(f"unknown mnemonic {mnem!r}")
    f6 = F6[base]
    # Unused fields are encoded the way the shipped programs encode them (funct3 = 010 on
    # config/break words, 001 on vle/vse) so assembled images are word-identical.
    if base in ("vsetvl", "vsetq", "vsetiq"):
        return Inst(f6, funct3=2, imm=_int(ops[0]))
    if base == "break":
        return Inst(f6, funct3=2)
    if base == "nop":
        return Inst(0, mask=0)
    if base in ("vfqmul", "vfqadd", "vfqsub"):
        if suffix == "vv":
            return Inst(f6, vd=_reg(ops[0]), vs1=_reg(ops[1]), vs2=_reg(ops[2]), funct3=F3_VV)
        if suffix == "vs":
            return Inst(f6, vd=_reg(ops[0]), vs1=_reg(ops[1]), funct3=F3_VS, imm=_int(ops[2]))
        if suffix == "sv" and base == "vfqsub":
            return Inst(f6, vd=_reg(ops[0]), vs2=_reg(ops[2]), funct3=F3_SV, imm=_int(ops[1]))
        raise ValueError(f"bad suffix on {mnem!r}")
    if base in ("vfqmod", "vcp", "vntt", "vintt"):
        return Inst(f6, vd=_reg(ops[0]), vs1=_reg(ops[1]))
    if base in ("vaut", "vroli", "vautn", "vautz"):
        # the shipped key-switch program encodes vaut with funct3 = 001 (keyswitch.mem:6)
        return Inst(f6, vd=_reg(ops[0]), vs1=_reg(ops[1]), funct3=1, imm=_int(ops[2]) if len(ops) > 2 else 0)
    if base == "vle":
        return Inst(f6, vd=_reg(ops[0]), funct3=1, imm=_addr(ops[1]))
    if base == "vse":
        return Inst(f6, vs1=_reg(ops[0]), funct3=1, imm=_addr(ops[1]))
    raise ValueError(mnem)


def assemble(text: str) -> np.ndarray:
    return words_of(i for i in (assemble_line(l) for l in text.splitlines()) if i is not None)


def disassemble_one(i: Inst) -> str:
    n = i.name
    if n in ("vsetvl", "vsetq", "vsetiq"):
        return f"{n} {i.imm:#x}"
    if n in ("break", "nop"):
        return n
    if n in ("vfqmul", "vfqadd", "vfqsub"):
        if i.funct3 == F3_VV:
            return f"{n}.vv v{i.vd}, v{i.vs1}, v{i.vs2}"
        if i.funct3 == F3_VS:
            return f"{n}.vs v{i.vd}, v{i.vs1}, {i.imm:#x}"
        if i.funct3 == F3_SV:
            return f"{n}.sv v{i.vd}, {i.imm:#x}, v{i.vs2}"
    if n in ("vfqmod", "vcp", "vntt", "vintt"):
        return f"{n} v{i.vd}, v{i.vs1}"
    if n in ("vaut", "vroli", "vautn", "vautz"):
        return f"{n} v{i.vd}, v{i.vs1}, {i.imm:#x}"
    if n in ("vle", "vse"):
        sel, off = i.imm >> 48, i.imm & ((1 << 48) - 1)
        ptr = PTR_NAME.get(sel, f"ptr{sel}")
        r = i.vd if n == "vle" else i.vs1
        return f"{n} v{r}, {ptr}+{off:#x}"
    return f".word {i.word:#010x}, {i.imm:#x}"


def disassemble(words: np.ndarray) -> str:
    return "\n".join(disassemble_one(i) for i in insts_of(words))
