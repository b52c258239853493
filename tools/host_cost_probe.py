#!/usr/bin/env python3
"""Where the host time of a key-switch goes (GPU box): per phase, time in load_isram and in run (enqueue only)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from aloha_amd import hybrid, vp as avp

n, L, K, alpha = 1 << 16, 24, 4, 4
mods = [avp.nth_prime(16, i) for i in range(L)]; sp = [avp.nth_prime(16, L + k) for k in range(K)]
lo = hybrid.HybridLayout(n, mods, sp, alpha)
vp = avp.VP(spm_rows=lo.spm_rows_needed, ksk_rows=2 * lo.ksk_limbs * lo.rows, max_log_n=16)
hs = hybrid.HybridKeySwitch(vp, n, mods, sp, alpha)
print("resident:", hs._resident, {k: [len(w) for w in v] for k, v in hs._p.items()})
acc = collections.defaultdict(float)
orig_load, orig_run = vp.load_isram, vp.run
def load(pc, words):
    t = time.perf_counter(); r = orig_load(pc, words); acc["load_isram"] += time.perf_counter() - t; return r
def run(pc, **kw):
    t = time.perf_counter(); r = orig_run(pc, **kw); acc["run"] += time.perf_counter() - t; acc["runs"] += 1; return r
vp.load_isram, vp.run = load, run
k = pow(3, 5, 2 * n)
hs.hoist(); hs.rotate_hoisted(k, 0); vp.sync(); acc.clear()
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    hs.rotate_hoisted(k, 0)
enq = time.perf_counter() - t0
vp.sync()
tot = time.perf_counter() - t0
print({k2: round(v / reps * 1e6, 1) for k2, v in acc.items()}, "enqueue us", round(enq / reps * 1e6, 1), "total us", round(tot / reps * 1e6, 1))
