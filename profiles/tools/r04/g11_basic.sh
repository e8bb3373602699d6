R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for lib in "" "$R/build/variants/libpf_basic.so"; do
  echo "== lib: ${lib:-product}"
  for e in hover quadx_waypoints fixedwing_waypoints; do
    for b in 65536 524288; do
      PF_LIB_PATH=$lib timeout 100 python $R/bench.py --env $e --batch $b --steps 600 --warmup 100 --no-cpu-baseline --no-configs --rollout-steps 50 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$e', $b, 'launch_us', round(d['roofline']['launch_us'],2), 'rollout_us', round(d.get('rollout',{}).get('ms_per_step',0)*1e3,2))"
    done
  done
done
