#!/usr/bin/env python
"""Basic blocks of one kernel in a hipcc -S listing: instruction mix per block (MFMA, other VALU, SALU, LDS, loads,
branches), in program order — where a loop's non-MFMA work sits.  Usage: python tools/isa_blocks.py file.s <substring
of the mangled kernel name> [min instructions]"""
import collections
import sys

src, key = sys.argv[1], sys.argv[2]
minimum = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if key in l and l.rstrip().split(";")[0].rstrip().endswith(":") and not l.startswith(("\t", " ")))
blocks, cur = [], ["entry", collections.Counter(), []]
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith(".Lfunc_end"):
        break
    if not t or t.startswith(";"):
        continue
    if t.startswith(".LBB") and ":" in t:
        blocks.append(cur)
        cur = [t.split(":")[0], collections.Counter(), []]
        continue
    if t.startswith("."):
        continue
    op = t.split()[0]
    cur[1][op] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        cur[2].append(t.split()[-1])
blocks.append(cur)
for name, c, br in blocks:
    tot = sum(c.values())
    if tot < minimum:
        continue
    mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    valu = sum(v for k, v in c.items() if k.startswith("v_")) - mfma
    salu = sum(v for k, v in c.items() if k.startswith("s_"))
    dsr = sum(v for k, v in c.items() if k.startswith("ds_read"))
    dsw = sum(v for k, v in c.items() if k.startswith("ds_write"))
    vm = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "flat_", "scratch_")))
    print(f"{name:12s} {tot:5d}  mfma {mfma:3d}  valu {valu:4d}  salu {salu:4d}  ds_read {dsr:3d}  ds_write {dsw:3d}  vmem {vm:3d}"
          f"  waitcnt {c['s_waitcnt']:2d}  nop {c['s_nop']:2d}  barrier {c['s_barrier']}  -> {' '.join(br)}")
