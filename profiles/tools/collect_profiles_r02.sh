#!/bin/bash
# Collects the round-2 judged profile artefacts into gpurun_out/r02/ (profiles/tools/summarize_profiles_r02.py copies them to profiles/r02).
# Every rocprofv3 run sits under `timeout`: a counter set the device does not like can hang the collection.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r02; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 170"
# headline line (per-step launches, hipGraph) + the rollout figure + the CPU baseline leg
timeout 200 python $R/bench.py --steps 2000 --warmup 200 2>/dev/null | tail -1 > $O/bench_n1.json
# the same command under the kernel tracer (CPU baseline skipped: not GPU work)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/kt_bench.json 2>/dev/null
# HBM traffic, each TCC counter in its own pass: per-step launches (prof_cfg.py) and rollout launches (prof_roll.py)
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --pmc $c --output-format csv -d $O/pmc_step_$c -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
  $T rocprofv3 --pmc $c --output-format csv -d $O/pmc_roll_$c -- python $R/profiles/tools/prof_roll.py > /dev/null 2>&1
done
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
$T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
$T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_roll_sq -- python $R/profiles/tools/prof_roll.py > /dev/null 2>&1
CR=1 $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq_response -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
for e in fixedwing:waypoints quadx:waypoints; do
  VEH=${e%%:*} TASK=${e##*:} $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq_${e%%:*}_${e##*:} -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
done
# the other BASELINE configs / sizes, and the contact-response and strong-scaling variants of the headline
for e in quadx_waypoints fixedwing_waypoints; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$e -- python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
  timeout 100 python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$e.json
done
timeout 100 python $R/bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b4096.json
timeout 100 python $R/bench.py --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline --rollout-steps 50 2>/dev/null | tail -1 > $O/bench_b524288.json
timeout 100 python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --world contact_response=1 2>/dev/null | tail -1 > $O/bench_contact_response_on.json
timeout 100 python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --scaling strong 2>/dev/null | tail -1 > $O/bench_strong_n1.json
# the dogfight task (auxiliary figure): bench line + kernel trace, and step time against the population's state
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dogfight -- python $R/bench.py --env dogfight --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
timeout 100 python $R/bench.py --env dogfight --steps 150 --warmup 20 2>/dev/null | tail -1 > $O/bench_dogfight.json
timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" > $O/dogfight_step_time_vs_population.txt
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
