#!/usr/bin/env python
"""Generates TFC_LENC_BLOCK (compression_amd/csrc/range_lanes.h): the 8-step encoder block with the table
lookups of a step (part A) issued TWO steps ahead of its interval update (part B), three rotating register
sets for (lo, hi), and the s_waitcnt lgkmcnt values that follow from the order of the LDS operations
(LDS operations complete in order, so "at most n outstanding" = everything older than the n latest is done).
Prints the macro; paste it over the old one when the schedule changes."""
import sys

SETS = [(152, 153, 154, 155), (156, 157, 158, 159), (176, 177, 178, 179)]
ROWS = [(148, 149, "A"), (150, 151, "B")]
AHEAD = 2
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16          # steps per block (kEncCadence)
VAL0 = 140 if N <= 8 else 124                              # the block's values live in v[VAL0 : VAL0 + N)

lines = []
lds = []          # tags of LDS operations in issue order


def emit(text):
    lines.append(text)


def wait_for(tag):
    idx = max(i for i, t in enumerate(lds) if t == tag)
    n = len(lds) - 1 - idx
    emit(f'"s_waitcnt lgkmcnt({n})\\n\\t"')


def part_a(k):
    r0, r1, _ = ROWS[k % 2]
    lo, _, hi, _ = SETS[k % 3]
    if k < N - 1:
        nxt = f"TFC_LENC_ROW_{ROWS[(k + 1) % 2][2]}({16 * (k + 1)})"
        lds.append(f"row{k + 1}")
    else:
        nxt = '""'
    emit(f"TFC_LENC_A({VAL0 + k}, {r0}, {r1}, {lo}, {hi}, {nxt}, NP{k % 3})")
    lds.append(f"lo{k}")
    lds.append(f"hi{k}")


def part_b(k):
    lo, loh, hi, hih = SETS[k % 3]
    emit(f"TFC_LENC_B({lo}, {loh}, {hi}, {hih}, PRE{k % 3})")
    lds.append(f"dig{k}")


emit('"s_mov_b64 s[56:57], exec\\n\\t"')
for q in range(N // 2):
    emit(f'"ds_read2_b32 v[{VAL0 + 2 * q}:{VAL0 + 2 * q + 1}], %[VP] offset0:{2 * q} offset1:{2 * q + 1}\\n\\t"')
    lds.append("val")
emit("TFC_LENC_ROW_A(0)")
lds.append("row0")
emit('"v_mov_b32 v153, 0\\n\\tv_mov_b32 v155, 0\\n\\tv_mov_b32 v157, 0\\n\\tv_mov_b32 v159, 0\\n\\t"')
emit('"v_mov_b32 v177, 0\\n\\tv_mov_b32 v179, 0\\n\\t"')
next_a = 0
for k in range(N):
    while next_a <= min(N - 1, k + AHEAD):
        wait_for(f"row{next_a}")
        part_a(next_a)
        next_a += 1
    wait_for(f"hi{k}")
    part_b(k)
emit('"s_mov_b64 exec, s[56:57]\\n\\t"')
emit('"s_waitcnt lgkmcnt(0)\\n\\t"')
print("#define TFC_LENC_BLOCK(NP0, NP1, NP2, PRE0, PRE1, PRE2)" + " " * 40 + "\\")
for i, l in enumerate(lines):
    print("  " + l + (" \\" if i + 1 < len(lines) else ""))
