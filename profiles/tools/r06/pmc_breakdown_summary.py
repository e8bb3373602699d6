"""profiling tool: gpurun_out/r06_breakdown/ (profiles/tools/r06/pmc_breakdown.sh) -> where a wave's issue slots go, per env step."""
import collections, csv, glob, os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
print("issue slots per wave per env step at 65 536 lanes (rocprofv3 --pmc, three SQ passes per config, prof_cfg.py's eager env steps; SQ_ACTIVE_INST_* are quad-cycles\n"
      "= slots of 4 clocks for a lone wave, SQ_INSTS_* are instruction counts; VALU 'arithmetic' = FMA + MUL + ADD (f32) + INT32 + TRANS + CVT as the counters class them,\n"
      "'other VALU' = the rest: v_mov, v_cndmask, v_cmp, min / max / med3, bit operations, v_readlane, v_accvgpr_*)")
for cfg, kname in (("quadx_hover", "quadx_m0_env_kernel<1"), ("quadx_waypoints", "quadx_m0_env_kernel<2"), ("fixedwing_waypoints", "fixedwing_wp_env_kernel")):
    tot = {}
    for p in "abc":
        fs = glob.glob(os.path.join(R, f"gpurun_out/r06_breakdown/{cfg}_{p}/*/*counter_collection.csv"))
        if not fs:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if kname in r["Kernel_Name"]:
                per[r["Counter_Name"]].append(float(r["Counter_Value"]))
        w = sum(per["SQ_WAVES"]) / max(1, len(per["SQ_WAVES"]))
        for k, v in per.items():
            if k != "SQ_WAVES":
                tot[k[3:]] = sum(v) / len(v) / w
    g = lambda k: tot.get(k, 0.0)
    arith = sum(g("INSTS_VALU_" + k) for k in ("FMA_F32", "MUL_F32", "ADD_F32", "INT32", "TRANS_F32", "CVT"))
    print(f"\n{cfg}: {g('ACTIVE_INST_ANY'):.0f} slots = VALU {g('ACTIVE_INST_VALU'):.0f} + scalar {g('ACTIVE_INST_SCA'):.0f} + branch / wait / nop {g('ACTIVE_INST_MISC'):.0f} + LDS {g('ACTIVE_INST_LDS'):.0f} (+ memory {g('INSTS_VMEM_RD') + g('INSTS_VMEM_WR'):.0f})")
    print(f"   instructions: VALU {g('INSTS_VALU'):.0f} (FMA {g('INSTS_VALU_FMA_F32'):.0f}, MUL {g('INSTS_VALU_MUL_F32'):.0f}, ADD {g('INSTS_VALU_ADD_F32'):.0f}, INT32 {g('INSTS_VALU_INT32'):.0f}, "
          f"transcendental {g('INSTS_VALU_TRANS_F32'):.0f}, CVT {g('INSTS_VALU_CVT'):.0f}; other VALU {g('INSTS_VALU') - arith:.0f}), SALU {g('INSTS_SALU'):.0f}, SMEM {g('INSTS_SMEM'):.0f}, "
          f"branches {g('INSTS_BRANCH'):.0f}, LDS {g('INSTS_LDS'):.0f}, loads {g('INSTS_VMEM_RD'):.0f}, stores {g('INSTS_VMEM_WR'):.0f}")
    print(f"   arithmetic VALU = {100 * arith / g('ACTIVE_INST_ANY'):.0f} % of the slots; other VALU {100 * (g('INSTS_VALU') - arith) / g('ACTIVE_INST_ANY'):.0f} %; scalar + branch + wait {100 * (g('ACTIVE_INST_SCA') + g('ACTIVE_INST_MISC')) / g('ACTIVE_INST_ANY'):.0f} %")
