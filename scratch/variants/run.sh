#!/bin/bash
# usage (on the GPU box): bash scratch/variants/run.sh  -- benches every scratch/variants/lib_*.so
R=$GRAFT_REPO_ROOT; cp $R/pyflyt_amd/libpyflyt_amd.so /tmp/orig.so
for f in $R/scratch/variants/lib_*.so; do
  cp $f $R/pyflyt_amd/libpyflyt_amd.so
  for e in hover fixedwing_waypoints; do
    python $R/bench.py --env $e --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $f)', '$e', round(d['roofline']['launch_us'],2))"
  done
done
cp /tmp/orig.so $R/pyflyt_amd/libpyflyt_amd.so
