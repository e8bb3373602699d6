"""The shipped code object is free of the compiler's exec-restore placement bug (tools/isa_exec_check.py): vector copies that a
join block runs before it re-enables its lanes. The build repairs the device assembly; this lints what was actually linked."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("PF_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def test_the_lint_sees_the_pattern_and_the_repair_removes_it():
    from tools import isa_exec_check as chk

    asm = """
_Z1kv:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\tv_add_f32_e32 v1, v2, v3
.LBB0_2:
\tv_writelane_b32 v252, s60, 62
\tv_accvgpr_write_b32 a41, v7
\ts_mov_b64 s[36:37], s[76:77]
\ts_or_b64 exec, exec, s[0:1]
\tv_mul_f32_e32 v1, v1, v1
.LBB0_3:
\tv_mov_b32_e32 v5, v6
\ts_or_b64 exec, exec, s[2:3]
\ts_endpgm
""".split("\n")
    found = [i for _, _, idx, _ in chk.sites(asm) for i in idx]
    assert [asm[i].strip() for i in found] == ["v_accvgpr_write_b32 a41, v7"]  # (.LBB0_3 is no divergent branch's target)
    fixed, report = chk.fix(asm)
    assert len(report) == 1 and not list(chk.sites(fixed))
    assert fixed.index("\tv_accvgpr_write_b32 a41, v7") == fixed.index("\ts_or_b64 exec, exec, s[0:1]") + 1
    # a copy whose registers the instructions in between touch is refused, and so is anything that is not a plain copy
    for bad in ("\tv_mov_b32_e32 v9, s36", "\tds_read_b32 v9, v10"):
        broken = [bad if l.strip().startswith("v_accvgpr_write") else l for l in asm]
        with pytest.raises(RuntimeError):
            chk.fix(broken)


def test_the_repair_moves_stack_spills_and_keeps_the_wait_counts_honest():
    from tools import isa_exec_check as chk

    asm = """
_Z1kv:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\tv_add_f32_e32 v1, v2, v3
.LBB0_2:
\tscratch_store_dwordx2 off, v[80:81], off offset:28
\ts_waitcnt vmcnt(3)
\ts_or_b64 exec, exec, s[0:1]
\ts_endpgm
""".split("\n")
    fixed, report = chk.fix(asm)
    assert len(report) == 1 and not list(chk.sites(fixed))
    i = fixed.index("\ts_or_b64 exec, exec, s[0:1]")
    assert fixed[i + 1].strip() == "scratch_store_dwordx2 off, v[80:81], off offset:28"
    assert "\ts_waitcnt vmcnt(2)" in fixed and "\ts_waitcnt vmcnt(3)" not in fixed  # one operation fewer in flight at that wait
    # a spill addressed through a register (not a constant stack slot) is not something the repair understands
    with pytest.raises(RuntimeError):
        chk.fix([l.replace("off, v[80:81], off offset:28", "v5, v[80:81], off") for l in asm])


def test_a_moved_reload_is_waited_for_before_anything_reads_it():
    """A flagged stack reload followed by the compiler's wait and a flagged copy that reads the reloaded register: both move behind
    the exec restore, the wait does not -- the repair must put a wait of its own right behind the moved reload."""
    from tools import isa_exec_check as chk

    asm = """
_Z1kv:
\ts_and_saveexec_b64 s[0:1], vcc
\ts_cbranch_execz .LBB0_2
\tv_add_f32_e32 v1, v2, v3
.LBB0_2:
\tscratch_load_dword v9, off, off offset:12
\ts_waitcnt vmcnt(0)
\tv_mov_b32_e32 v10, v9
\ts_or_b64 exec, exec, s[0:1]
\tv_mul_f32_e32 v1, v10, v9
\ts_endpgm
""".split("\n")
    fixed, report = chk.fix(asm)
    assert len(report) == 2 and not list(chk.sites(fixed))
    ins = [chk.instruction(l) for l in fixed if chk.instruction(l)]
    i = ins.index("s_or_b64 exec, exec, s[0:1]")
    assert ins[i + 1] == "scratch_load_dword v9, off, off offset:12"
    assert ins[i + 2] == "s_waitcnt vmcnt(0)"            # the reload completes before ...
    assert ins[i + 3] == "v_mov_b32_e32 v10, v9"         # ... the copy moved along with it reads it
    assert ins[i + 4] == "v_mul_f32_e32 v1, v10, v9"


def test_shipped_code_object_is_clean(tmp_path):
    from tools import isa_exec_check as chk

    import __graft_entry__ as G

    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        pytest.skip("no llvm-objdump in this image")
    G.build()
    lib = shutil.copy(G.HIP_LIB, tmp_path / "lib.so")
    subprocess.check_call([objdump, "--offloading", os.path.basename(lib)], cwd=tmp_path, stdout=subprocess.DEVNULL)
    objs = [f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f]
    assert len(objs) == 1, objs
    dis = subprocess.run([objdump, "-d", "--symbolize-operands", objs[0]], cwd=tmp_path, check=True, capture_output=True, text=True).stdout.split("\n")
    assert sum(1 for l in dis if "s_cbranch_execz" in l) > 1000  # (the disassembly is the real thing)
    assert chk.lint("libpyflyt_amd.so", dis) == 0
