"""profiling tool: per-basic-block instruction statistics of one kernel in a --save-temps .s file.
usage: python profiles/tools/isa_blocks.py file.s <mangled-name substring> [min_block_size]"""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
minsz = int(sys.argv[3]) if len(sys.argv) > 3 else 15
L = open(path).read().split('\n')
start = next(i for i, l in enumerate(L) if l.startswith('_ZN') and pat in l and l.split(';')[0].rstrip().endswith(':'))
end = next(j for j in range(start, len(L)) if L[j].startswith('.Lfunc_end'))
blk = ['entry', 0, 0, 0, 0, []]
stats = [blk]
for l in L[start + 1:end]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blk = [m.group(1), 0, 0, 0, 0, []]
        stats.append(blk)
        continue
    t = l.strip()
    if not l.startswith('\t') or t.startswith(('.', ';')):
        continue
    op = t.split()[0]
    blk[1] += 1
    if 'readlane' in op or 'writelane' in op:
        blk[3] += 1
    elif op.startswith('v_'):
        blk[2] += 1
        if op.startswith(('v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos', 'v_log', 'v_exp')):
            blk[4] += 1
    if op.startswith('s_cbranch') or op == 's_branch':
        blk[5].append(t.replace('s_cbranch_', '').replace('s_branch', 'jmp'))
tot = [sum(b[k] for b in stats) for k in (1, 2, 3, 4)]
print(L[start][:90], 'total', tot)
for b in stats:
    if b[1] >= minsz:
        print(f"{b[0]:12s} n={b[1]:4d} valu={b[2]:4d} lane={b[3]:3d} trans={b[4]:2d} {b[5]}")
