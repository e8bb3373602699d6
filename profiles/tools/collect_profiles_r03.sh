#!/bin/bash
# Collects the round-3 judged profile artefacts into gpurun_out/r03/ (profiles/tools/summarize_profiles_r03.py copies them to
# profiles/r03). Every rocprofv3 run sits under `timeout`. The env tasks run stepSimulation's contact solve by default now
# (contact_response on): every figure below is WITH it unless the file name says otherwise.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r03; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 170"
# headline line (per-step launches, hipGraph) + the rollout figure + the CPU baseline leg
timeout 200 python $R/bench.py --steps 2000 --warmup 200 2>/dev/null | tail -1 > $O/bench_n1.json
# the same command under the kernel tracer (CPU baseline skipped: not GPU work)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > /dev/null 2>&1
# the driver's own invocation (20 timed steps: the wall clock then carries the graph launch and the final synchronisation)
timeout 100 python $R/bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_driver_shape.json
# HBM traffic, each TCC counter in its own pass: per-step launches (prof_cfg.py) and rollout launches (prof_roll.py)
for c in FETCH_SIZE WRITE_SIZE; do
  $T rocprofv3 --pmc $c --output-format csv -d $O/pmc_step_$c -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
  $T rocprofv3 --pmc $c --output-format csv -d $O/pmc_roll_$c -- python $R/profiles/tools/prof_roll.py > /dev/null 2>&1
done
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
$T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
$T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_roll_sq -- python $R/profiles/tools/prof_roll.py > /dev/null 2>&1
CR=0 $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq_detect_only -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
MODE=7 $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq_mode7 -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
for e in fixedwing:waypoints quadx:waypoints; do
  VEH=${e%%:*} TASK=${e##*:} $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_step_sq_${e%%:*}_${e##*:} -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
done
# the other BASELINE configs / sizes, the flight modes, the shared-world PettingZoo task, the detection-only opt-out
for e in quadx_waypoints fixedwing_waypoints; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$e -- python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
  timeout 100 python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$e.json
done
# (PettingZoo task: no auto-reset, so a short window while every drone of every world is still airborne)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_ma_hover -- python $R/bench.py --env ma_hover --steps 120 --warmup 20 --graph-steps 20 --no-cpu-baseline > /dev/null 2>&1
timeout 100 python $R/bench.py --env ma_hover --steps 120 --warmup 20 --graph-steps 20 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_ma_hover.json
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_mode7 -- python $R/bench.py --flight-mode 7 --steps 1000 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
for m in 7 6 4 1 -1; do timeout 100 python $R/bench.py --flight-mode=$m --steps 1000 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_mode$m.json; done
timeout 100 python $R/bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b4096.json
timeout 100 python $R/bench.py --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline --rollout-steps 50 2>/dev/null | tail -1 > $O/bench_b524288.json
timeout 100 python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-contact-response 2>/dev/null | tail -1 > $O/bench_detect_only.json
timeout 100 python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline --scaling strong 2>/dev/null | tail -1 > $O/bench_strong_n1.json
# the dogfight task (auxiliary): gentle and uniform actions; step time against the population's state; the contact solve's statistics
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_dogfight -- python $R/bench.py --env dogfight --steps 150 --warmup 20 --no-cpu-baseline > /dev/null 2>&1
timeout 100 python $R/bench.py --env dogfight --steps 150 --warmup 20 2>/dev/null | tail -1 > $O/bench_dogfight.json
timeout 100 python $R/bench.py --env dogfight --dogfight-actions uniform --steps 300 --warmup 200 2>/dev/null | tail -1 > $O/bench_dogfight_uniform.json
timeout 200 python $R/profiles/tools/dog_diag.py 2>/dev/null | grep "^steps" > $O/dogfight_step_time_vs_population.txt
timeout 100 python $R/profiles/tools/solver_bench.py 2>/dev/null | grep "us per tick" > $O/solver_bench_landed.txt
timeout 100 python $R/profiles/tools/bench_ma_shared2.py 2>/dev/null | grep "us/step" > $O/ma_hover_shared_step_time.txt
# per-wave phase timeline and the solver's call statistics (the -DPF_PHASE_TRACE variant library)
if [ -f $R/build/variants/libpf_trace.so ]; then
  for cr in 1 0; do PF_LIB_PATH=$R/build/variants/libpf_trace.so CR=$cr timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null > $O/phase_trace_hover65536_cr$cr.txt; done
  for w in landed hover rates calm; do WHAT=$w PF_LIB_PATH=$R/build/variants/libpf_trace.so timeout 150 python $R/profiles/tools/solver_trace.py 2>/dev/null; done > $O/solver_trace.txt
fi
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
