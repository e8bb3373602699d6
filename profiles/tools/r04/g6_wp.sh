R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04f; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for w in "" "--world contact_iters=0" "--no-contact-response"; do
  echo "== $w"; for rep in 1 2; do timeout 100 python $R/bench.py --env quadx_waypoints --steps 2000 --warmup 100 --no-cpu-baseline --rollout-steps 0 $w 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_us', round(d['roofline']['launch_us'],2))"; done
done
echo "== PF_NO_CALM_PATH"; PF_NO_CALM_PATH=1 timeout 100 python $R/bench.py --env quadx_waypoints --steps 2000 --warmup 100 --no-cpu-baseline --rollout-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('launch_us', round(d['roofline']['launch_us'],2))"
