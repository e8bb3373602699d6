#!/bin/bash
# Collects the round-6 judged profile artefacts into gpurun_out/r06/ (profiles/tools/summarize_profiles_r06.py copies them to
# profiles/r06 and writes profiles/pmc_latest.json). Every rocprofv3 run sits under `timeout`; PMC passes are separate runs without
# any tracing flag (kernel-trace / stats in their own runs). Contact model: the defaults (pyflyt_amd/params.py: WORLD).
# Trace variants (built on the CPU box, profiles/tools/r06/*_only_build.py): build/variants/t_hover.so, t_wp.so (-DPF_PHASE_TRACE),
# t_fw.so (-DPF_PHASE_TRACE), t_fwtick.so (-DPF_PHASE_TRACE -DPF_FW_TICK_TRACE); build/variants/r05.so = round 5's library (commit bcac199) for the same-box A/B.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
T="timeout 170"
# headline line (hipGraph; includes the secondary configs, the facade block and the CPU baseline leg)
timeout 400 python $R/bench.py --steps 2000 --warmup 200 2>/dev/null | tail -1 > $O/bench_n1.json
# the driver's own invocation
timeout 300 python $R/bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_shape.json
# kernel traces of the three 65 536-lane configs (the same bench command, CPU baseline and secondary configs off)
for e in hover quadx_waypoints fixedwing_waypoints; do
  $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$e -- python $R/bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --rollout-steps 0 > /dev/null 2>&1
  timeout 100 python $R/bench.py --env $e --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_$e.json
done
# the facade in a closed loop under the kernel trace: which kernels does make_vec(...).step() launch? (policy: addmm + clamp; env: one kernel)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_facade -- python $R/profiles/tools/r06/facade_loop.py > $O/facade_loop.txt 2>&1
# HBM traffic and instruction counters per config, each TCC counter in its own pass (prof_cfg.py: eager env steps)
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
for e in quadx:hover quadx:waypoints fixedwing:waypoints; do
  v=${e%%:*}; t=${e##*:}
  for c in FETCH_SIZE WRITE_SIZE; do
    VEH=$v TASK=$t $T rocprofv3 --pmc $c --output-format csv -d $O/pmc_${v}_${t}_$c -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
  done
  VEH=$v TASK=$t $T rocprofv3 --pmc $SQ --output-format csv -d $O/pmc_${v}_${t}_sq -- python $R/profiles/tools/prof_cfg.py > /dev/null 2>&1
done
# other sizes / variants
timeout 100 python $R/bench.py --batch 4096 --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_b4096.json
timeout 100 python $R/bench.py --batch 524288 --steps 300 --warmup 50 --no-cpu-baseline --no-configs --rollout-steps 50 2>/dev/null | tail -1 > $O/bench_b524288.json
timeout 100 python $R/bench.py --batch 8192 --steps 2000 --warmup 200 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_b8192.json
timeout 100 python $R/bench.py --env quadx_waypoints --steps 2000 --warmup 200 --no-cpu-baseline --no-configs --world contact_iters=10 --rollout-steps 0 2>/dev/null | tail -1 > $O/bench_quadx_waypoints_iters10.json
for m in 7 6; do timeout 100 python $R/bench.py --flight-mode=$m --steps 1000 --warmup 100 --no-cpu-baseline --no-configs 2>/dev/null | tail -1 > $O/bench_mode$m.json; done
timeout 100 python $R/bench.py --env ma_hover --steps 120 --warmup 20 --graph-steps 20 --no-cpu-baseline --min-timed-ms 0 2>/dev/null | tail -1 > $O/bench_ma_hover.json
timeout 100 python $R/bench.py --env dogfight --steps 150 --warmup 20 --no-cpu-baseline --min-timed-ms 0 2>/dev/null | tail -1 > $O/bench_dogfight.json
# the 2-rank launcher path on this one GPU (bench.py starts the ranks itself: no torchrun environment)
PF_BENCH_SINGLE_DEVICE=1 PF_BENCH_BACKEND=gloo timeout 300 python $R/bench.py --gpus 2 --steps 200 --warmup 20 --no-cpu-baseline --rollout-steps 0 2>/dev/null | tail -1 > $O/bench_self_launched_2ranks_one_gpu.json
# same-box A/B against round 5's library over the four BASELINE configurations
[ -f $R/build/variants/r05.so ] && (cd $R && timeout 500 bash profiles/tools/r06/g_all.sh build/variants/r05.so pyflyt_amd/libpyflyt_amd.so > /dev/null 2>&1; cp gpurun_out/g_all.txt $O/ab_r05_vs_r06_same_box.txt)
# ... and against this round's own first half (commit ef16607: the repaired build, before the plain build / the cascaded modes' fp64 state out of scratch / the packed sweeps)
[ -f $R/build/variants/r06_head.so ] && (cd $R && timeout 500 bash profiles/tools/r06/g_all.sh build/variants/r06_head.so pyflyt_amd/libpyflyt_amd.so > /dev/null 2>&1; cp gpurun_out/g_all.txt $O/ab_r06_first_half_vs_final_same_box.txt)
# the cascaded flight modes
(cd $R && timeout 300 bash profiles/tools/r06/g_modes.sh > /dev/null 2>&1; cp gpurun_out/g_modes.txt $O/cascaded_modes.txt)
# per-wave phase timelines and the in-register floor solve's statistics (the -DPF_PHASE_TRACE variant libraries)
V=$R/build/variants
[ -f $V/t_hover.so ] && TASK=hover PF_LIB_PATH=$V/t_hover.so timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null > $O/phase_trace_hover65536.txt
[ -f $V/t_wp.so ] && TASK=waypoints PF_LIB_PATH=$V/t_wp.so timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null > $O/phase_trace_waypoints65536.txt
[ -f $V/t_hover_modes.so ] && MODE=7 TASK=hover PF_LIB_PATH=$V/t_hover_modes.so timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null > $O/phase_trace_mode7.txt
[ -f $V/t_fw.so ] && VEH=fixedwing TASK=waypoints PF_LIB_PATH=$V/t_fw.so timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null | grep -v "per tick and wave" > $O/phase_trace_fixedwing_waypoints65536.txt
# (the tick's own split from the -DPF_FW_TICK_TRACE variant: its per-tick atomics inflate the wave's life tenfold, so only the RATIO of the two shares is kept)
[ -f $V/t_fwtick.so ] && (echo "tick split (t_fwtick.so: -DPF_PHASE_TRACE -DPF_FW_TICK_TRACE; clocks inflated by the counters' own atomics -- the shares are what counts):"; VEH=fixedwing TASK=waypoints PF_LIB_PATH=$V/t_fwtick.so timeout 100 python $R/profiles/tools/phase_trace.py 2>/dev/null | grep "per tick and wave") >> $O/phase_trace_fixedwing_waypoints65536.txt
: > $O/solver_trace.txt
[ -f $V/t_hover.so ] && WHAT=rates TASK=hover RINGS=0,100 PF_LIB_PATH=$V/t_hover.so timeout 200 python $R/profiles/tools/solver_trace.py 2>/dev/null >> $O/solver_trace.txt
[ -f $V/t_wp.so ] && WHAT=rates TASK=waypoints RINGS=0,100 PF_LIB_PATH=$V/t_wp.so timeout 200 python $R/profiles/tools/solver_trace.py 2>/dev/null >> $O/solver_trace.txt
[ -f $V/t_hover.so ] && WHAT=calm TASK=hover PF_LIB_PATH=$V/t_hover.so timeout 100 python $R/profiles/tools/solver_trace.py 2>/dev/null >> $O/solver_trace.txt
[ -f $V/t_wp.so ] && WHAT=calm TASK=waypoints PF_LIB_PATH=$V/t_wp.so timeout 100 python $R/profiles/tools/solver_trace.py 2>/dev/null >> $O/solver_trace.txt
[ -f $V/t_wp.so ] && WHAT=perlaunch TASK=waypoints LAUNCHES=1500 PF_LIB_PATH=$V/t_wp.so timeout 300 python $R/profiles/tools/solver_trace.py 2>/dev/null >> $O/solver_trace.txt
# the distribution of the launch durations behind the averages (QuadX-Waypoints: the tail is the launches in which one lane solves a floor contact)
python - <<PY > $O/launch_duration_distribution.txt
import csv,glob,os,numpy as np
for e,k in (("hover","quadx_m0_env_kernel"),("quadx_waypoints","quadx_m0_env_kernel"),("fixedwing_waypoints","fixedwing_wp_env_kernel")):
    fs=sorted(glob.glob("$O/kt_%s/*/*kernel_trace.csv"%e), key=os.path.getmtime)
    if not fs: continue
    d=np.array([(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3 for r in csv.DictReader(open(fs[-1])) if k in r["Kernel_Name"]][-2000:])
    print(e, "last %d launches of the per-step kernel (us): min %.2f p10 %.2f median %.2f mean %.2f p90 %.2f p99 %.2f max %.2f"%(len(d),d.min(),np.percentile(d,10),np.median(d),d.mean(),np.percentile(d,90),np.percentile(d,99),d.max()))
    print("   histogram, edges", [0,9,10,11,12,13,14,15,16,17,18,20,24,28,32,40,60,100], ":", np.histogram(d, bins=[0,9,10,11,12,13,14,15,16,17,18,20,24,28,32,40,60,100])[0].tolist())
PY
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
du -sh $O; ls $O
