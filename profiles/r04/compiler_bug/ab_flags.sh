#!/bin/bash
# A/B of compiler flags against the placement bug on repro.hip (runs here: hipcc cross-compiles without a GPU). Prints, per flag
# set, the number of misplaced instructions, the kernels' VGPR counts and scratch bytes, and the instruction count.
cd "$(dirname "$0")/../../.."
for f in "" "-O2" "-mllvm -amdgpu-spill-sgpr-to-vgpr=0" "-mllvm -greedy-regclass-priority-trumps-globalness=1" "-mllvm -split-spill-mode=size" \
         "-mllvm -enable-local-reassign=1" "-mllvm -sgpr-regalloc=fast" "-mllvm -vgpr-regalloc=basic"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -fno-slp-vectorize -mllvm -enable-misched=false -ffp-contract=off \
      --cuda-device-only -S profiles/r04/compiler_bug/repro.hip -o /tmp/pf_repro.s $f 2>/dev/null
  python - "$f" <<'PY'
import re, sys
sys.path.insert(0, ".")
from tools import isa_exec_check as chk
t = open("/tmp/pf_repro.s", errors="replace").read()
lines = t.split("\n")
n = sum(len(idx) for _, _, idx, _ in chk.sites(lines))
vg = re.findall(r"[.]vgpr_count:\s+(\d+)", t)
sc = re.findall(r"[.]private_segment_fixed_size:\s+(\d+)", t)
ni = sum(1 for l in lines if re.match(r"^\s+(v_|s_|ds_|global_|scratch_|buffer_)", l))
print("%-58s misplaced %3d  vgpr %s  scratch %s  instructions %d" % (sys.argv[1] or "(the build's flags)", n, vg, sc, ni))
PY
done
