#!/bin/bash
# scratch: A/B compile flags on the GPU box
cd $GRAFT_REPO_ROOT
for flags in "" "-fno-slp-vectorize" "-fno-slp-vectorize -ffast-math" ; do
  PF_HIPCC_FLAGS="$flags" python -c "import __graft_entry__ as g; g.build(force=True)" 2>&1 | grep -v warning | grep -v "^\s" | head -3
  echo "== flags: '$flags'"
  (cd scratch; PF_LPW=64 PF_WPS=2 python exp2.py "lpw64" 65536; PF_LPW=64 PF_WPS=2 python exp2.py "lpw64" 262144; PF_LPW=32 PF_WPS=2 python exp2.py "lpw32" 65536) 2>&1 | grep -v amdgpu.ids
done
