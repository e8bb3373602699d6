#!/usr/bin/env python3
"""Per-basic-block instruction mix of one kernel in a device assembly file: isa_blocks.py file.s <mangled-name-prefix> [min_instructions]
(VALU / SALU / packed / AGPR copies / v_mov / s_nop / LDS per block: where a tick's instructions are)."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
least = int(sys.argv[3]) if len(sys.argv) > 3 else 100
lines = open(path, errors="replace").read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(name) and ":" in l and not l.startswith("\t")][0]
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start:end]
blk, cnt = "entry", collections.OrderedDict({"entry": collections.Counter()})
for l in body[1:]:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blk = m.group(1)
        cnt[blk] = collections.Counter()
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    cnt[blk][t.split()[0]] += 1
tot = collections.Counter()
for b, c in cnt.items():
    tot.update(c)
    n = sum(c.values())
    if n < least:
        continue
    f = lambda p: sum(v for k, v in c.items() if k.startswith(p))  # noqa: E731
    print(f"{b:12s} {n:5d}  valu {f('v_'):5d} salu {f('s_'):4d} pk {f('v_pk'):4d} agpr {sum(v for k, v in c.items() if 'accvgpr' in k):4d} "
          f"v_mov {c.get('v_mov_b32', 0):4d} nop {c.get('s_nop', 0):3d} lds {f('ds_'):3d} cndmask {c.get('v_cndmask_b32_e32', 0) + c.get('v_cndmask_b32_e64', 0):3d} "
          f"trans {sum(v for k, v in c.items() if re.match(r'v_(rcp|rsq|sqrt|sin|cos|exp|log)', k)):3d} scratch {f('scratch_'):3d}")
n = sum(tot.values())
print(f"whole kernel {n} instructions, agpr copies {sum(v for k, v in tot.items() if 'accvgpr' in k)}, scratch {sum(v for k, v in tot.items() if k.startswith('scratch_'))}")
