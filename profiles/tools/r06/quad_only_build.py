#!/usr/bin/env python3
"""Experiment build: the library with ONLY the specialised QuadX kernels of ONE task instantiated, Philox noise, flight mode 0,
contact response on (a temporary copy of pyflyt_amd.hip with the other launchers stubbed), straight through hipcc -- a minute
instead of four; for A/B work on quadx_fast.hpp (PF_LIB_PATH=<out> python bench.py --env quadx_waypoints ...).
Usage: quad_only_build.py out.so HOVER|WAYPOINTS [--modes] [--save-asm] [hipcc flags]. Never the product library: no lint, no repair."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

out, task, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
modes = "--modes" in extra  # keep the cascaded-flight-mode instantiations (MODES = true) instead of the mode-0 ones
extra = [e for e in extra if e != "--modes"]
src = open(G.HIP_SRC).read()
keep = f"PF_TASK_{task}"
for t in ("PF_TASK_HOVER", "PF_TASK_MA_HOVER", "PF_TASK_WAYPOINTS"):
    if t != keep:
        src = src.replace(f"launch_fast<{t}>(ctx, b, op, mask, s)", f"launch_fast<{keep}>(ctx, b, op, mask, s)")
        src = src.replace(f"launch_rollout<{t}>(ctx, b, k_steps, step_index0, s)", f"launch_rollout<{keep}>(ctx, b, k_steps, step_index0, s)")
# Philox only, mode 0 only, contact response only
src = src.replace("  else if (ctx->P.noise_mode == PF_NOISE_INJECT) PF_FAST(PF_NOISE_INJECT);\n  else PF_FAST(PF_NOISE_OFF);\n#undef PF_FAST\n#undef PF_FAST3", "#undef PF_FAST\n#undef PF_FAST3")
src = src.replace("#define PF_FAST(NZ) do { if (ctx->K.mode != 0) PF_FAST3(NZ, true, true); else if (ctx->P.contact_response) PF_FAST3(NZ, true, false); else PF_FAST3(NZ, false, false); } while (0)",
                  "#define PF_FAST(NZ) do { PF_FAST3(NZ, true, %s); } while (0)" % ("true" if modes else "false"))
src = src.replace("#define PF_ROLL(NZ, R) do { if (ctx->K.mode != 0) PF_ROLL3(NZ, R, true, true); else if (ctx->P.contact_response) PF_ROLL3(NZ, R, true, false); else PF_ROLL3(NZ, R, false, false); } while (0)",
                  "#define PF_ROLL(NZ, R) do { PF_ROLL3(NZ, R, true, %s); } while (0)" % ("true" if modes else "false"))
src = src.replace("    else PF_ROLL(PF_NOISE_OFF, 1);", "").replace("    else PF_ROLL(PF_NOISE_OFF, 2);", "")
src = re.sub(r"(static void launch_env_t\([^{]*\{)(.*?)(\n\}\nextern \"C\")", r"\1\n  (void)ctx; (void)b; (void)op; (void)mask; (void)s; (void)roll_steps; (void)step0;\3", src, flags=re.S)
src = src.replace("#define PF_DF(AA, VV) hipLaunchKernelGGL(", "#define PF_DF(AA, VV) if (false) hipLaunchKernelGGL(")
for head in ("static void launch_fast_fw(pf_ctx* ctx", "static void launch_rollout_fw(pf_ctx* ctx"):
    i = src.index(head)
    j = src.index("{", src.index(")", i)) + 1
    k = src.index("\n}\n", j)
    src = src[:j] + "\n  (void)ctx; (void)b; (void)s;\n" + src[k:]
tmp = os.path.join(os.path.dirname(G.HIP_SRC), f"_quad_only_tmp_{os.getpid()}.hip")  # (one per process: variants build side by side)
open(tmp, "w").write(src)
try:
    save = "--save-asm" in extra
    extra = [e for e in extra if e != "--save-asm"]
    base = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *G.HIPCC_FLAGS, *extra]
    if save:
        subprocess.check_call(base + ["--cuda-device-only", "-S", tmp, "-o", out + ".s"])
    subprocess.check_call(base + ["-shared", "-fPIC", tmp, "-o", out])
finally:
    os.remove(tmp)
