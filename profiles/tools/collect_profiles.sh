#!/bin/bash
# Collects the round's judged profile artefacts into gpurun_out/r01/ (copied to profiles/ afterwards).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r01; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
python $R/bench.py --steps 2000 --warmup 200 2>/dev/null | tail -1 > $O/bench_n1.json
# same command under the kernel tracer (cpu baseline skipped: it is not GPU work)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --steps 2000 --warmup 200 --no-cpu-baseline > $O/kt_bench.json 2>/dev/null
# counters, each TCC counter in its own pass
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_tcc -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
# instruction mix of the other two specialised kernels
VEH=fixedwing TASK=waypoints rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq_fixedwing_waypoints -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
VEH=quadx TASK=waypoints rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS --output-format csv -d $O/pmc_sq_quadx_waypoints -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
# kernel traces of the other two BASELINE configs
for e in quadx_waypoints fixedwing_waypoints; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$e -- python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline > /dev/null 2>&1
done
# other envs / sizes, short
for e in quadx_waypoints fixedwing_waypoints; do python $R/bench.py --env $e --steps 500 --warmup 100 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_$e.json; done
python $R/bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b4096.json
python $R/bench.py --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_b524288.json
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
ls -R $O | head -40
