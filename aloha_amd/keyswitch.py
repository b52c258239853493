"""Rotate + hybrid key-switch for any number of RNS limbs, sharded by limb over GPUs.

The reference ships one fixed program for two ciphertext primes and one special prime
(``sim/vp/isram_file_generator/keyswitch.mem:1-122``): digit = one ciphertext
prime, K = 1 special prime P, rounding mod-down.  This module issues the SAME algorithm, in the
same HE-ISA and operand layout, for L limbs, with the intermediates streamed through SPM rows
instead of the 32-entry register file, and with the limbs partitioned over ``world`` ranks
(one process per GPU):

    phase 1  per owned digit d :  D[d] = aut_k(iNTT(c1[d]));  R[d] = NTT(aut_k(iNTT(c0[d])))
    exchange ALL-GATHER of the coefficient-form digits D[0..L-1]           (keyswitch.mem:4-43)
    phase 2  per owned modulus j (ciphertext primes and P):
                 acc_c[j] = sum_d NTT_j(D[d] mod m_j) * KSK[j][d][c]        (keyswitch.mem:46-79)
             as two loops: ALL Mr*L transforms (one batched launch pair), then ALL 2*Mr inner
             products (one batched launch per program; the products are split over several
             programs when they exceed the 4096-word instruction ROM)
    phase 3  owner of P       :  t_c = iNTT(acc_c[P]) + floor(P/2)          (keyswitch.mem:80-83)
    exchange BROADCAST of t_0, t_1 from P's owner
    phase 4  per owned j < L  :  r_c = (acc_c[j] - NTT_j(t_c - floor(P/2))) * P^-1
                                 out_0[j] = R[j] + r_0,  out_1[j] = r_1      (keyswitch.mem:84-121)

The same phases 2-4 relinearise a ciphertext product (BASELINE config 4, "HMult + relinearisation";
no reference program exists): phase 1 becomes the tensor product

    phase 1m per owned prime j :  d0 = a0*b0,  d1 = a0*b1 + a1*b0,  d2 = a1*b1   (NTT domain),
                                  D[j] = iNTT(d2)                                  (digit j)

and phase 4 adds d0, d1 where the rotation adds the rotated c0: out = (d0 + r_0, d1 + r_1).

Every phase is an HE-ISA program run by the engine (``VP.run``); only the two exchanges go
through ``torch.distributed`` (RCCL over xGMI on GPUs, gloo in the CPU tests).  With
``world == 1`` this is the single-GPU key-switch for any L.

The programs are written loop by loop -- all transforms of a phase, then all permutations,
then the element-wise steps -- so that the engine's fusing sequencer turns each loop into ONE
batched launch (``vle; vntt; vse`` runs, ``vle; vaut; vse`` runs, ``vle; vle; op; vse`` runs)
and the digit-by-key inner product into one pass per accumulator (``vfqmul.vv`` /
``vfqadd.vv`` chains over memory operands).  The instruction semantics are the plain
sequential ones; an engine without fusion (or the oracle) computes the same bytes.

Where the reference writes ``vcp`` / ``vfqmod`` before a ``vntt`` to re-interpret a digit under
another modulus (keyswitch.mem:9,13,29,33) these programs transform the loaded digit directly:
the copy exists in the reference because ``vntt`` destroys its source register, and the
butterfly's operand normalisation (modalu.sv:44-46) performs the same single conditional
subtraction.

Rank-local SPM layout, in limb slots of N/128 rows from ``base_row``:
    CT_IN [2*Lr]  c0 of the owned primes, then c1        (Lr = owned ciphertext primes)
    DIG   [G*Dr]  all digits, grouped by owner           (Dr = ceil(L/G); all-gather target)
    ROT   [Lr]    rotated c0, NTT form  (product: d0)
    BASE1 [Lr]    product only: d1
    CT_B  [2*Lr]  product only: the second operand, c0 then c1
    ACC   [2*Mr]  inner products of the owned moduli     (Mr = owned moduli incl. P if owned)
    T     [2]     mod-down terms                         (broadcast target)
    OUT   [2*Lr]  result: c0 then c1 of the owned primes
    S1/S2 [2*Lr]  scratch (phase 1: coefficient forms; phase 4: mod-down intermediates)
    SD    [Mr*L]  scratch (phase 2: every digit in NTT form under every owned modulus)
Rank-local KSK layout: limb ((jl*L + d)*2 + c) for owned modulus jl, digit d, component c.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence

import numpy as np

from . import isa
from .programs import barrett_mu

SPM, KSK = 0, 1
LANES = 128


def owner(j: int, world: int) -> int:
    """Limb j (ciphertext primes 0..L-1, special prime = L) lives on rank j mod world."""
    return j % world


@dataclass
class Layout:
    n: int
    moduli: Sequence[int]            # ciphertext primes q_0..q_{L-1}
    special: int                     # P
    world: int = 1
    rank: int = 0
    base_row: int = 0
    owned_ct: List[int] = field(init=False)     # owned ciphertext-prime indices
    owned_all: List[int] = field(init=False)    # owned modulus indices (L = special)

    def __post_init__(self):
        L = len(self.moduli)
        self.owned_ct = [j for j in range(L) if owner(j, self.world) == self.rank]
        self.owned_all = [j for j in range(L + 1) if owner(j, self.world) == self.rank]
        self.L = L
        self.rows = self.n // LANES
        self.Lr = len(self.owned_ct)
        self.Mr = len(self.owned_all)
        self.Dr = -(-L // self.world)                       # digits per rank in the gather buffer
        s = 0
        self.ct_in = s; s += 2 * self.Lr
        self.dig = s; s += self.world * self.Dr
        self.rot = s; s += self.Lr
        self.base1 = s; s += self.Lr
        self.ct_b = s; s += 2 * self.Lr
        self.acc = s; s += 2 * self.Mr
        self.t = s; s += 2
        self.out = s; s += 2 * self.Lr
        self.s1 = s; s += 2 * self.Lr
        self.s2 = s; s += 2 * self.Lr
        self.sd = s; s += L * self.Mr
        self.slots = s

    # slot -> absolute row / byte offset helpers
    def row(self, slot: int) -> int:
        return self.base_row + slot * self.rows

    def off(self, slot: int) -> int:
        return self.row(slot) * 1024

    def dig_slot(self, d: int) -> int:
        return self.dig + owner(d, self.world) * self.Dr + d // self.world

    def modulus(self, j: int) -> int:
        return self.special if j == self.L else self.moduli[j]

    @property
    def spm_rows_needed(self) -> int:
        return self.base_row + self.slots * self.rows

    @property
    def ksk_limbs(self) -> int:
        return self.Mr * self.L * 2


def _setq(q: int) -> List[str]:
    return [f"vsetq {q:#x}", f"vsetiq {barrett_mu(q):#x}"]


def program_digits(lo: Layout, reference_shape: bool = False) -> List[str]:
    """Phase 1.  c0 limb i at CT_IN[i], c1 limb i at CT_IN[Lr + i] (i = local index).

    Default: the identity vntt(vaut(x, k)) == vautn(vntt(x), k) (tests) moves the automorphism onto the NTT-form limbs as a
    pure permutation (``vautn``): the rotated c0 needs no transform at all and the digits are iNTT(vautn(c1)) -- ONE transform
    per limb where the reference runs three and two coefficient-domain permutations.  The one place where the two differ in
    the WORDS is a zero coefficient at a sign-flipped position: the reference's ``vaut`` stores it as q_d (vxu_lane.sv:598),
    and the basis extension then lifts the digit as q_d, not 0, under the other moduli.  ``vautz`` (an extension, like
    ``vautn``) puts exactly those q_d back, so the digits -- and with them every output word -- are the reference's
    (tests: the crafted zero-corner ciphertext, the golden rotates).  The engine folds the ``vautz`` into the inverse
    transform's last pass.
    ``reference_shape=True``: the reference's own sequence (keyswitch.mem:4-7,:24-27,:100-121): iNTT of both components,
    ``vaut`` in the coefficient domain, NTT of the rotated c0."""
    out = [f"vsetvl {64 * lo.n:#x}"]
    if not reference_shape:
        for i, d in enumerate(lo.owned_ct):                     # the automorphism on both NTT-form components
            out += _setq(lo.moduli[d])
            out += [f"vle v0, abs+{lo.off(lo.ct_in + i):#x}", "vautn v2, v0, 0x0", f"vse v2, abs+{lo.off(lo.rot + i):#x}"]
            out += [f"vle v0, abs+{lo.off(lo.ct_in + lo.Lr + i):#x}", "vautn v2, v0, 0x0", f"vse v2, abs+{lo.off(lo.s1 + i):#x}"]
        for i, d in enumerate(lo.owned_ct):                     # digits: the rotated c1 in coefficient form
            out += _setq(lo.moduli[d])
            out += [f"vle v0, abs+{lo.off(lo.s1 + i):#x}", "vintt v2, v0", "vfqmul.vs v4, v2, 0x1", "vautz v6, v4, 0x0",
                    f"vse v6, abs+{lo.off(lo.dig_slot(d)):#x}"]
        return out + ["break"]
    for i, d in enumerate(lo.owned_ct):                         # all inverse transforms
        out += _setq(lo.moduli[d])
        for c in range(2):
            out += [f"vle v0, abs+{lo.off(lo.ct_in + c * lo.Lr + i):#x}", "vintt v2, v0",
                    f"vse v2, abs+{lo.off(lo.s1 + c * lo.Lr + i):#x}"]
    for i, d in enumerate(lo.owned_ct):                         # all automorphisms
        out += _setq(lo.moduli[d])
        out += [f"vle v0, abs+{lo.off(lo.s1 + lo.Lr + i):#x}", "vaut v2, v0, 0x0", f"vse v2, abs+{lo.off(lo.dig_slot(d)):#x}"]
        out += [f"vle v0, abs+{lo.off(lo.s1 + i):#x}", "vaut v2, v0, 0x0", f"vse v2, abs+{lo.off(lo.s2 + i):#x}"]
    for i, d in enumerate(lo.owned_ct):                         # rotated c0 back to NTT form
        out += _setq(lo.moduli[d])
        out += [f"vle v0, abs+{lo.off(lo.s2 + i):#x}", "vntt v2, v0", f"vse v2, abs+{lo.off(lo.rot + i):#x}"]
    return out + ["break"]


def program_tensor(lo: Layout) -> List[str]:
    """Phase 1 of the ciphertext product: a = CT_IN, b = CT_B (NTT form), both (c0, c1)."""
    out = [f"vsetvl {64 * lo.n:#x}"]
    a = lambda c, i: lo.off(lo.ct_in + c * lo.Lr + i)
    b = lambda c, i: lo.off(lo.ct_b + c * lo.Lr + i)
    for i, j in enumerate(lo.owned_ct):                         # d0 and d2, element-wise
        out += _setq(lo.moduli[j])
        out += [f"vle v0, abs+{a(0, i):#x}", f"vle v1, abs+{b(0, i):#x}", "vfqmul.vv v2, v0, v1", f"vse v2, abs+{lo.off(lo.rot + i):#x}"]
        out += [f"vle v0, abs+{a(1, i):#x}", f"vle v1, abs+{b(1, i):#x}", "vfqmul.vv v2, v0, v1", f"vse v2, abs+{lo.off(lo.s1 + i):#x}"]
    for i, j in enumerate(lo.owned_ct):                         # d1 = a0*b1 + a1*b0: a two-term sum in one pass
        out += _setq(lo.moduli[j])
        out += [f"vle v0, abs+{a(0, i):#x}", f"vle v1, abs+{b(1, i):#x}", "vfqmul.vv v6, v0, v1",
                f"vle v0, abs+{a(1, i):#x}", f"vle v1, abs+{b(0, i):#x}", "vfqmul.vv v3, v0, v1", "vfqadd.vv v6, v6, v3",
                f"vse v6, abs+{lo.off(lo.base1 + i):#x}"]
    for i, j in enumerate(lo.owned_ct):                         # digits: d2 in coefficient form
        out += _setq(lo.moduli[j])
        out += [f"vle v0, abs+{lo.off(lo.s1 + i):#x}", "vintt v2, v0", f"vse v2, abs+{lo.off(lo.dig_slot(j)):#x}"]
    return out + ["break"]


def program_digit_transforms(lo: Layout, jls: Sequence[int]) -> List[str]:
    """Phase 2a for the owned moduli ``jls``: every digit under each of them, NTT form, into SD[jl*L + d]."""
    out = [f"vsetvl {64 * lo.n:#x}"]
    for jl in jls:
        j = lo.owned_all[jl]
        out += _setq(lo.modulus(j))
        for d in range(lo.L):
            out += [f"vle v0, abs+{lo.off(lo.dig_slot(d)):#x}", "vntt v2, v0", f"vse v2, abs+{lo.off(lo.sd + jl * lo.L + d):#x}"]
    return out + ["break"]


def inner_words_per_modulus(lo: Layout) -> int:
    return 2 + 2 * (4 * lo.L)                                   # vsetq, vsetiq, per component: L x (vle, vle, mul[, add]) + vse


def program_inner(lo: Layout, jls: Sequence[int]) -> List[str]:
    """Phase 2b for the owned moduli with local indices ``jls``: the digit-by-key inner products."""
    out = [f"vsetvl {64 * lo.n:#x}"]
    for jl in jls:
        out += _setq(lo.modulus(lo.owned_all[jl]))
        for c in range(2):                                      # inner product with the key, per component
            acc = 6 + 2 * c                                     # v6, v8 (even bank)
            for d in range(lo.L):
                out += [f"vle v0, abs+{lo.off(lo.sd + jl * lo.L + d):#x}", f"vle v1, ksk+{((jl * lo.L + d) * 2 + c) * 8 * lo.n:#x}"]
                if d == 0:
                    out.append(f"vfqmul.vv v{acc}, v0, v1")
                else:
                    out += ["vfqmul.vv v3, v0, v1", f"vfqadd.vv v{acc}, v{acc}, v3"]
            out.append(f"vse v{acc}, abs+{lo.off(lo.acc + 2 * jl + c):#x}")
    return out + ["break"]


def program_moddown_terms(lo: Layout) -> List[str]:
    """Phase 3 (only on the rank that owns P)."""
    jl = lo.owned_all.index(lo.L)
    half = lo.special >> 1
    out = [f"vsetvl {64 * lo.n:#x}"] + _setq(lo.special)
    for c in range(2):
        out += [f"vle v0, abs+{lo.off(lo.acc + 2 * jl + c):#x}", "vintt v2, v0", f"vfqadd.vs v2, v2, {half:#x}",
                f"vse v2, abs+{lo.off(lo.t + c):#x}"]
    return out + ["break"]


def program_finish(lo: Layout, add_base1: bool = False) -> List[str]:
    """Phase 4.  ``add_base1``: the product form, out_1 = BASE1 + r_1 (the rotation has out_1 = r_1)."""
    half = lo.special >> 1
    out = [f"vsetvl {64 * lo.n:#x}"]
    own = list(enumerate(lo.owned_ct))
    for i, j in own:                                            # t_c - floor(P/2) under q_j
        out += _setq(lo.moduli[j])
        for c in range(2):
            out += [f"vle v0, abs+{lo.off(lo.t + c):#x}", f"vfqsub.vs v2, v0, {half:#x}",
                    f"vse v2, abs+{lo.off(lo.s1 + c * lo.Lr + i):#x}"]
    for i, j in own:                                            # ... to NTT form (in place)
        out += _setq(lo.moduli[j])
        for c in range(2):
            s = lo.off(lo.s1 + c * lo.Lr + i)
            out += [f"vle v0, abs+{s:#x}", "vntt v2, v0", f"vse v2, abs+{s:#x}"]
    for i, j in own:                                            # acc - that
        out += _setq(lo.moduli[j])
        jl = lo.owned_all.index(j)
        for c in range(2):
            out += [f"vle v0, abs+{lo.off(lo.acc + 2 * jl + c):#x}", f"vle v1, abs+{lo.off(lo.s1 + c * lo.Lr + i):#x}",
                    "vfqsub.vv v2, v0, v1", f"vse v2, abs+{lo.off(lo.s2 + c * lo.Lr + i):#x}"]
    for i, j in own:                                            # times P^-1; c1 is final
        q = lo.moduli[j]
        pinv = pow(lo.special % q, q - 2, q)
        out += _setq(q)
        out += [f"vle v0, abs+{lo.off(lo.s2 + i):#x}", f"vfqmul.vs v2, v0, {pinv:#x}", f"vse v2, abs+{lo.off(lo.s1 + i):#x}"]
        out += [f"vle v0, abs+{lo.off(lo.s2 + lo.Lr + i):#x}", f"vfqmul.vs v2, v0, {pinv:#x}",
                f"vse v2, abs+{lo.off((lo.s1 if add_base1 else lo.out) + lo.Lr + i):#x}"]
    for i, j in own:                                            # c0 += rotated c0
        out += _setq(lo.moduli[j])
        out += [f"vle v0, abs+{lo.off(lo.s1 + i):#x}", f"vle v1, abs+{lo.off(lo.rot + i):#x}", "vfqadd.vv v2, v0, v1",
                f"vse v2, abs+{lo.off(lo.out + i):#x}"]
        if add_base1:
            out += [f"vle v0, abs+{lo.off(lo.s1 + lo.Lr + i):#x}", f"vle v1, abs+{lo.off(lo.base1 + i):#x}",
                    "vfqadd.vv v2, v0, v1", f"vse v2, abs+{lo.off(lo.out + lo.Lr + i):#x}"]
    return out + ["break"]


class Exchange:
    """The two exchanges of the key-switch over ``torch.distributed`` (or a no-op for world 1).

    ``tensor_of(vp, row, nwords)`` must return an int64 torch tensor VIEW of engine memory
    (zero copy on the GPU, see ``device_tensor``); ``None`` selects the copy-in/copy-out
    path through ``vp.read`` / ``vp.write`` used by the CPU tests.

    ``stream``: the ``torch.cuda.Stream`` the engine was created on (``VP(stream=s.cuda_stream)``).  The
    collectives are then enqueued stream-ordered -- RCCL waits for the phase's kernels through an event,
    the next phase's kernels wait for the collective -- and the host never blocks inside a key-switch;
    without it every exchange is bracketed by host synchronisation.  ``stub`` skips the communication
    altogether (per-rank timing of a sharded layout on one GPU; results are then meaningless).

    ``p2p``: MI355X nodes connect every GPU pair by its own xGMI link; a ring all-gather moves each chunk over
    world-1 hops of ONE link each, while world-1 direct sends per rank (``batch_isend_irecv``: one RCCL group) put
    1/(world-1) of the traffic on every link at once.  Off by default -- RCCL picks its own algorithm --
    and identical in result (tests)."""

    def __init__(self, group=None, tensor_of: Optional[Callable] = None, always: bool = False, stream=None,
                 stub: bool = False, p2p: bool = False):
        self.p2p = p2p                # all-gather as world-1 concurrent sends + receives (one per xGMI link) instead of the collective
        self.group = group
        self.tensor_of = tensor_of
        self.always = always          # run the collectives even for world == 1 (single-rank NCCL smoke test)
        self.stream = stream
        self.stub = stub
        self.bytes_sent = 0           # payload this rank has put on / taken off the links so far (accounting only)
        self.bytes_received = 0

    def all_gather(self, vp, row0: int, chunk: int, world: int, rank: int) -> None:
        """All-gather of ``world`` chunks of ``chunk`` words laid out back to back from SPM row ``row0``
        (rank r owns chunk r)."""
        if (world == 1 and not self.always) or self.stub:
            return
        self.bytes_sent += (world - 1) * chunk * 8
        self.bytes_received += (world - 1) * chunk * 8
        import torch
        import torch.distributed as dist
        rows = chunk // LANES

        def gather(buf):
            mine = buf[rank * chunk:(rank + 1) * chunk]
            if not self.p2p or world == 1:
                dist.all_gather_into_tensor(buf, mine.clone(), group=self.group)
                return
            peer = (lambda r: dist.get_global_rank(self.group, r)) if self.group is not None else (lambda r: r)
            ops = []
            for d in range(1, world):                              # staggered so that no two ranks start on the same peer
                to, frm = (rank + d) % world, (rank - d) % world
                ops.append(dist.P2POp(dist.isend, mine, peer(to), group=self.group))
                ops.append(dist.P2POp(dist.irecv, buf[frm * chunk:(frm + 1) * chunk], peer(frm), group=self.group))
            for w in dist.batch_isend_irecv(ops):
                w.wait()

        if self.tensor_of is not None and self.stream is not None:
            with torch.cuda.stream(self.stream):
                gather(self.tensor_of(vp, row0, world * chunk))
            return
        vp.sync()
        if self.tensor_of is not None:
            gather(self.tensor_of(vp, row0, world * chunk))
            torch.cuda.current_stream().synchronize()
        elif self.p2p:
            buf = torch.zeros(world * chunk, dtype=torch.int64)
            buf[rank * chunk:(rank + 1) * chunk] = torch.from_numpy(vp.read(SPM, row0 + rank * rows, chunk).view(np.int64).copy())
            gather(buf)
            for r in range(world):
                if r != rank:
                    vp.write(SPM, row0 + r * rows, buf[r * chunk:(r + 1) * chunk].numpy().view(np.uint64))
        else:
            mine = torch.from_numpy(vp.read(SPM, row0 + rank * rows, chunk).view(np.int64).copy())
            parts = [torch.empty_like(mine) for _ in range(world)]
            dist.all_gather(parts, mine, group=self.group)
            for r, p in enumerate(parts):
                if r != rank:
                    vp.write(SPM, row0 + r * rows, p.numpy().view(np.uint64))

    def all_gather_digits(self, vp, lo: Layout) -> None:
        self.all_gather(vp, lo.row(lo.dig), lo.Dr * lo.n, lo.world, lo.rank)

    def broadcast_terms(self, vp, lo: Layout) -> None:
        if (lo.world == 1 and not self.always) or self.stub:
            return
        import torch
        import torch.distributed as dist
        src = owner(lo.L, lo.world)
        row = lo.row(lo.t)
        if lo.rank == src:
            self.bytes_sent += (lo.world - 1) * 2 * lo.n * 8
        else:
            self.bytes_received += 2 * lo.n * 8
        if self.tensor_of is not None and self.stream is not None:
            with torch.cuda.stream(self.stream):
                dist.broadcast(self.tensor_of(vp, row, 2 * lo.n),
                               src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
            return
        vp.sync()
        if self.tensor_of is not None:
            buf = self.tensor_of(vp, row, 2 * lo.n)
            dist.broadcast(buf, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
            torch.cuda.current_stream().synchronize()
        else:
            t = torch.from_numpy(vp.read(SPM, row, 2 * lo.n).view(np.int64).copy())
            dist.broadcast(t, src=src, group=self.group)
            if lo.rank != src:
                vp.write(SPM, row, t.numpy().view(np.uint64))


def device_tensor(vp, row: int, nwords: int):
    """int64 torch view of ``nwords`` words of the engine's SPM memory at ``row`` (zero copy)."""
    import torch

    class _Cai:
        pass

    c = _Cai()
    c.__cuda_array_interface__ = {"shape": (nwords,), "typestr": "<i8", "data": (vp.device_ptr(SPM, row), False),
                                  "version": 2}
    return torch.as_tensor(c, device="cuda")


class ShardedKeySwitch:
    """Rotate + key-switch of one ciphertext whose limbs are sharded over ``world`` ranks.

    The phase programs live side by side in the instruction ROM (4096 words,
    ``vp_defines.vh:31``); when they do not fit (large L on few ranks) the per-modulus programs
    of phase 2 are loaded one after the other at the same address."""

    def __init__(self, vp, n: int, moduli: Sequence[int], special: int, world: int = 1, rank: int = 0,
                 exchange: Optional[Exchange] = None, base_row: int = 0, reference_shape: bool = False):
        """``reference_shape=True``: phase 1 exactly as the reference writes it (``program_digits``): three times the
        transforms and two coefficient-domain permutations per limb for the same words."""
        self.vp = vp
        self.lo = Layout(n, list(moduli), special, world, rank, base_row)
        self.ex = exchange or Exchange()
        lo = self.lo
        asm = lambda lines: isa.assemble("\n".join(lines))
        self._p_digits = asm(program_digits(lo, reference_shape))
        def groups(words_per_modulus):                          # owned moduli per program, by ROM capacity
            per = max(1, (isa.IRAM_DEPTH - 2) // words_per_modulus)
            return [list(range(g, min(g + per, lo.Mr))) for g in range(0, lo.Mr, per)]
        self._p_inner = [asm(program_digit_transforms(lo, g)) for g in groups(2 + 3 * lo.L)] + \
                        [asm(program_inner(lo, g)) for g in groups(inner_words_per_modulus(lo))]
        self._p_terms = asm(program_moddown_terms(lo)) if lo.L in lo.owned_all else None
        self._p_finish = asm(program_finish(lo))
        self._p_tensor = asm(program_tensor(lo))
        self._p_finish_mul = asm(program_finish(lo, add_base1=True))
        progs = [self._p_digits, self._p_finish, self._p_tensor, self._p_finish_mul] + \
                ([self._p_terms] if self._p_terms is not None else [])
        fixed = sum(len(p) for p in progs)
        inner = sum(len(p) for p in self._p_inner)
        if max([len(p) for p in progs + self._p_inner]) > isa.IRAM_DEPTH:
            raise ValueError("a phase program exceeds the 4096-word instruction ROM")
        self._resident = fixed + inner <= isa.IRAM_DEPTH
        self._pc = {}
        if self._resident:                                      # everything loaded once
            pc = 0
            for name, p in [("digits", self._p_digits), ("finish", self._p_finish), ("terms", self._p_terms),
                            ("tensor", self._p_tensor), ("finish_mul", self._p_finish_mul)] + \
                           [(("inner", jl), p) for jl, p in enumerate(self._p_inner)]:
                if p is None:
                    continue
                vp.load_isram(pc, p)
                self._pc[name] = pc
                pc += len(p)

    # -- data placement -------------------------------------------------------------------------
    def load_key(self, ksk_full: np.ndarray) -> None:
        """``ksk_full``: the whole key, (L+1)*L*2 limbs ordered [modulus][digit][component]
        (keyswitch.mem:46-77); only the owned moduli's limbs are kept on this rank."""
        lo = self.lo
        k = np.ascontiguousarray(ksk_full, dtype=np.uint64).reshape(lo.L + 1, lo.L * 2 * lo.n)
        for jl, j in enumerate(lo.owned_all):
            self.vp.write(KSK, jl * lo.L * 2 * lo.rows, k[j])

    def load_ct(self, ct_full: np.ndarray) -> None:
        """``ct_full``: 2*L limbs, c0 limbs then c1 limbs; the owned ones go to CT_IN."""
        lo = self.lo
        ct = np.ascontiguousarray(ct_full, dtype=np.uint64).reshape(2, lo.L, lo.n)
        for c in range(2):
            for i, j in enumerate(lo.owned_ct):
                self.vp.write(SPM, lo.row(lo.ct_in + c * lo.Lr + i), ct[c, j])

    def load_ct_b(self, ct_full: np.ndarray) -> None:
        """Second operand of ``multiply`` (same format as ``load_ct``)."""
        lo = self.lo
        ct = np.ascontiguousarray(ct_full, dtype=np.uint64).reshape(2, lo.L, lo.n)
        for c in range(2):
            for i, j in enumerate(lo.owned_ct):
                self.vp.write(SPM, lo.row(lo.ct_b + c * lo.Lr + i), ct[c, j])

    def read_out(self) -> np.ndarray:
        """Owned result limbs, shape (2, Lr, n)."""
        lo = self.lo
        self.vp.sync()
        return self.vp.read(SPM, lo.row(lo.out), 2 * lo.Lr * lo.n).reshape(2, lo.Lr, lo.n)

    # -- the op -----------------------------------------------------------------------------------
    def _run(self, name, words, step: int = 0) -> None:
        if self._resident:
            pc = self._pc[name]
        else:
            pc = 0
            self.vp.load_isram(0, words)
        rc = self.vp.run(pc, src0=0, src1=0, rslt=0, ksk=0, step=step)
        if rc:
            raise RuntimeError(f"key-switch phase {name} failed with status {rc}: {getattr(self.vp, 'last_error', '')}")

    def rotate(self, k: int) -> None:
        """Apply X -> X^k (k odd, reduced mod N by the engine like the reference) and switch
        the key, reading CT_IN and writing OUT."""
        lo = self.lo
        if lo.Lr:
            self._run("digits", self._p_digits, step=k)
        self._switch("finish", self._p_finish)

    def multiply(self) -> None:
        """CT_IN * CT_B with relinearisation (the loaded key must be the relinearisation key, i.e.
        switch from s^2): OUT = (d0 + r_0, d1 + r_1).  No rescale."""
        lo = self.lo
        if lo.Lr:
            self._run("tensor", self._p_tensor)
        self._switch("finish_mul", self._p_finish_mul)

    def _switch(self, finish_name, finish_words) -> None:
        """Phases 2-4 on the digits in DIG."""
        lo = self.lo
        self.ex.all_gather_digits(self.vp, lo)
        for jl, words in enumerate(self._p_inner):
            self._run(("inner", jl), words)
        if self._p_terms is not None:
            self._run("terms", self._p_terms)
        self.ex.broadcast_terms(self.vp, lo)
        if lo.Lr:
            self._run(finish_name, finish_words)
