R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" | cut -c1-120 > $O/dogfight.txt; cat $O/dogfight.txt
timeout 100 python $R/profiles/tools/solver_bench.py 2>/dev/null | grep "us per tick"
timeout 100 python $R/profiles/tools/bench_ma_shared2.py 2>/dev/null | grep "us/step"
timeout 100 python $R/bench.py --env dogfight --steps 150 --warmup 20 --no-cpu-baseline --min-timed-ms 0 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dogfight gentle launch_us', round(d['roofline']['launch_us'],2))"
