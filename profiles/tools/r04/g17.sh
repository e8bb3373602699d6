R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
timeout 100 python $R/profiles/tools/solver_bench.py 2>/dev/null | grep "us per tick"
timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" | cut -c1-100 | sed -n 4,14p
timeout 100 python $R/profiles/tools/bench_ma_shared2.py 2>/dev/null | grep "us/step" | head -6
timeout 100 python $R/bench.py --env quadx_waypoints --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline --no-configs --rollout-steps 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('waypoints 524288 launch_us', round(d['roofline']['launch_us'],2))"
cd $R; timeout 600 python -m pytest tests -m gpu -q --tb=short -k "aviary or kat or landing or dogfight or pz or golden" 2>&1 | tail -4
