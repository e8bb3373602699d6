#!/usr/bin/env python3
"""Experiment build: the library with ONLY the specialised Fixedwing-Waypoints kernels instantiated (every other launcher's body
stubbed out in a temporary copy of pyflyt_amd.hip), straight through hipcc -- a minute instead of four; for A/B work on
fixedwing_fast.hpp only (PF_LIB_PATH=<out> python bench.py --env fixedwing_waypoints ...). Never the product library: no lint,
no repair (the Fixedwing kernels have no site; tests/test_isa_lint.py checks the product)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

out, extra = sys.argv[1], sys.argv[2:]
src = open(G.HIP_SRC).read()


def stub(src, head, tail):
    """replace everything between the line that starts with `head` (kept, up to its opening brace) and the first later line equal to `tail`"""
    i = src.index(head)
    j = src.index("{", src.index(")", i)) + 1
    k = src.index(tail, j)
    return src[:j] + "\n  (void)ctx; (void)b; (void)s;\n" + src[k:]


for head in ("static void launch_fast(pf_ctx* ctx", "static void launch_rollout(pf_ctx* ctx"):
    src = stub(src, head, "\n}\n")
src = re.sub(r"(static void launch_env_t\([^{]*\{)(.*?)(\n\}\nextern \"C\")", r"\1\n  (void)ctx; (void)b; (void)op; (void)mask; (void)s; (void)roll_steps; (void)step0;\3", src, flags=re.S)
src = src.replace("#define PF_DF(AA, VV) hipLaunchKernelGGL(", "#define PF_DF(AA, VV) if (false) hipLaunchKernelGGL(")
tmp = os.path.join(os.path.dirname(G.HIP_SRC), f"_fw_only_tmp_{os.getpid()}.hip")  # (one per process: variants build side by side)
open(tmp, "w").write(src)
try:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", *G.HIPCC_FLAGS, *extra, "-shared", "-fPIC", tmp, "-o", out]
    if "--save-asm" in extra:
        cmd.remove("--save-asm")
        subprocess.check_call(cmd[:-4] + ["--cuda-device-only", "-S", tmp, "-o", out + ".s"])
    subprocess.check_call(cmd)
finally:
    os.remove(tmp)
