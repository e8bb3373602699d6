R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
rm -rf $O/kt_dogfight
timeout 170 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dogfight -- python $R/bench.py --env dogfight --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
timeout 100 python $R/bench.py --env dogfight --steps 150 --warmup 20 2>/dev/null | tail -1 > $O/bench_dogfight.json
timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" > $O/dogfight_step_time_vs_population.txt
FREEZE=1 timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" > $O/dogfight_step_time_vs_population_freeze_wrecks.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
cat $O/bench_dogfight.json | cut -c1-300; head -3 $O/dogfight_step_time_vs_population_freeze_wrecks.txt
