"""profiling tool: per-kernel static instruction statistics from a --save-temps .s file."""
import sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
L = open(path).read().split('\n')
for start, l in enumerate(L):
    if l.startswith('_ZN') and pat in l.split(':')[0] and ': ' in l or (l.startswith('_ZN') and pat in l and l.rstrip().endswith(':')):
        end = next(i for i in range(start, len(L)) if L[i].startswith('.Lfunc_end'))
        lines = [x.strip() for x in L[start + 1:end] if x.startswith('\t') and not x.strip().startswith(('.', ';'))]
        c = Counter(x.split()[0] for x in lines)
        f = lambda p: sum(v for k, v in c.items() if p in k)
        print(l.split(':')[0][:70], len(lines), 'scratch', f('scratch'), 's_load', f('s_load'), 'ds', f('ds_'), 'v_mov', c['v_mov_b32_e32'],
              'readlane', c['v_readlane_b32'], 'writelane', c['v_writelane_b32'], 'waitcnt', c['s_waitcnt'],
              'cndmask', f('v_cndmask'), 'accvgpr', f('accvgpr'), 'rcp', f('v_rcp'), 'branch', f('s_cbranch'))
