#!/usr/bin/env python3
"""Instruction histogram per kernel of a gfx950 .s file (hipcc -S --cuda-device-only)."""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_Z\S+):[^\n]*\n(.*?)s_endpgm", s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ops = [l.split()[0] for l in body.splitlines() if l.startswith("\t") and l.strip() and not l.strip().startswith((";", "."))]
    c = collections.Counter(ops)
    mult = sum(v for k, v in c.items() if k.startswith(("v_mad_u64", "v_mul_lo_u32", "v_mul_hi_u32")))
    valu = sum(v for k, v in c.items() if k.startswith("v_"))
    print(f"== {name[:110]}\n   total {len(ops)}  valu {valu}  mult {mult}  s_nop {c.get('s_nop',0)}  s_waitcnt {c.get('s_waitcnt',0)}")
    print("   " + ", ".join(f"{k}:{v}" for k, v in c.most_common(40)))
