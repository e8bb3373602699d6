"""Lint (and repair) for a register-allocator placement bug of the ROCm 7.2 compiler that bit these kernels on gfx950.

At the top of the join block of a divergent branch -- the block `s_cbranch_execz` jumps to, or the fall-through of a loop's
`s_cbranch_execnz` -- the lanes that sat the branch out are re-enabled by `s_or_b64 exec, exec, s[..]` (or, in an if/else flow block,
`s_or_saveexec_b64`). Only scalar instructions and SGPR spill traffic (v_writelane / v_readlane) belong in front of that
instruction. The VGPR allocation runs after the SGPR allocation, and when the latter has left a scalar copy or a spill in front of
the exec restore, the former no longer recognises the block prologue: a live-range-split copy of a value that is live in ALL lanes
(seen: `v_accvgpr_write_b32 a41, v7`, the last action component of the generic QuadX-Waypoints env kernel; `v_mov_b32 v121, v55` in
the specialised one) lands BEFORE the restore, executes under the branch's narrower -- on the skip edge: empty -- exec mask, and
the other lanes read garbage later. Which kernels are hit changes with every edit that moves register pressure.

  python tools/isa_exec_check.py dev.s [...]            lint: prints every such site, exit status 1 if there is any
  python tools/isa_exec_check.py --fix dev.s -o out.s   repair: moves the copies to just behind the exec restore (where the
                                                        allocator meant them to run: for every lane that enters the block),
                                                        refusing anything that is not a register copy or a stack spill / reload, and any move it cannot prove safe

The build (__graft_entry__.build) compiles the device code to assembly, repairs it with this, lints the result and only then
assembles it; tests/test_isa_lint.py lints the disassembly of the shipped code object."""
import re
import sys

LABEL = re.compile(r"^(\.LBB\S+|[A-Za-z_][\w$.]*):")
OBJ_LABEL = re.compile(r"^[0-9a-f]+ <([^>]+)>:")
WIDEN = re.compile(r"^(s_or_b64\s+exec,\s*exec,|s_or_saveexec_b64\s)")
# The flow block of an if / else: `s_andn2_saveexec_b64 s[a:b], s[a:b]` switches exec from the then-lanes to the else-lanes. A copy in
# front of it runs for the then-lanes (none, when the block was entered through s_cbranch_execz), behind it for the else-lanes: a
# value that is live in all lanes needs BOTH, so the repair REPEATS such an instruction behind the flip instead of moving it
# (round 5: `v_accvgpr_write_b32 a23, v177` in front of the flip inside the in-register floor solve of the one-wave-per-SIMD
# QuadX-Waypoints kernel -- the else-lanes read a stale a23 afterwards and two crashing lanes of a wave came out of the solve with
# or without their impulses from one run to the next; the lint had classed the flip as "exec rewritten some other way").
ELSE_FLIP = re.compile(r"^s_andn2_saveexec_b64\s")
SCALAR_OK = ("v_writelane_b32", "v_readlane_b32", "v_readfirstlane_b32", "s_")
MOVABLE = re.compile(r"^(v_mov_b32_e32|v_mov_b64_e32|v_accvgpr_write_b32|v_accvgpr_read_b32|v_accvgpr_mov_b32|v_pk_mov_b32)\s")
SPILL = re.compile(r"^scratch_(store|load)_(dword|dwordx2|dwordx3|dwordx4|short|byte|ubyte)\s+(off,\s*[va]\S+,\s*off|[va]\S+,\s*off,\s*off)")  # (stack slot at a constant offset)
REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|scc|m0)(?:_lo|_hi)?\b")


def instruction(raw):
    """the instruction text of an assembly / objdump line, '' for labels, directives, comments and blanks"""
    line = raw.split("//")[0].split(";")[0].strip()
    if not line or line.startswith(".") or LABEL.match(line) or OBJ_LABEL.match(line):
        return ""
    return line


def regs(ins):
    out = set()
    for m in REG.finditer(ins.split(None, 1)[1] if " " in ins else ""):
        if m.group(1):
            out.add(m.group(1) + m.group(2))
        elif m.group(3):
            out.update(m.group(3) + str(k) for k in range(int(m.group(4)), int(m.group(5)) + 1))
        else:
            out.add(m.group(6))
    return out


def label_of(raw):
    m = LABEL.match(raw.strip()) or OBJ_LABEL.match(raw.strip())
    return m.group(1) if m else None


def sites(lines):
    """yields (function, label, [line indices of vector instructions in front of the exec restore], index of the restore)"""
    joins, func = set(), None
    for raw in lines:
        lab = label_of(raw)
        if lab and not lab.startswith((".LBB", "L")):
            func = lab
        ins = instruction(raw)
        if ins.startswith("s_cbranch_execz"):
            joins.add((func, ins.split()[1].strip("<>").split("+")[0]))
    func, label, pending = None, None, None
    for i, raw in enumerate(lines):
        lab = label_of(raw)
        if lab:
            label = lab
            if not lab.startswith((".LBB", "L")):
                func = lab
            pending = [] if (func, lab) in joins else None
            continue
        ins = instruction(raw)
        if not ins:
            continue
        if ins.startswith("s_cbranch_execnz"):  # the fall-through of a loop's back edge runs with an empty exec mask
            label, pending = f"(after line {i + 1})", []
            continue
        if pending is None:
            continue
        if WIDEN.match(ins):
            if pending:
                yield func, label, pending, i
            pending = None
            continue
        if ELSE_FLIP.match(ins):
            # (already repaired: the same instructions, in the same order, right behind the flip)
            behind = [instruction(x) for x in lines[i + 1:i + 1 + 3 * len(pending)]] if pending else []
            behind = [x for x in behind if x and not x.startswith("s_waitcnt")]
            todo = [k for n, k in enumerate(pending) if not (n < len(behind) and behind[n] == instruction(lines[k]))] if pending else []
            if todo:
                yield func, label, todo, i
            pending = None
            continue
        op = ins.split()[0]
        if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")) or re.match(r"^s_\w+\s+exec", ins) or "saveexec" in op:
            pending = None  # the block ends, or exec is rewritten some other way: not this pattern
            continue
        if not op.startswith(SCALAR_OK):
            pending.append(i)


def lint(path, lines):
    n = 0
    for func, label, idx, _ in sites(lines):
        for i in idx:
            print(f"{path}:{i + 1}: {func} {label}: `{instruction(lines[i])}` runs before the block's exec restore")
            n += 1
    return n


def fix(lines):
    """returns (new lines, report); raises if a flagged instruction is not a plain register copy or cannot be moved safely"""
    moves, report, waits, repeats = {}, [], {}, set()
    for func, label, idx, at in sites(lines):
        moved = set()
        for i in idx:
            ins = instruction(lines[i])
            spill = bool(SPILL.match(ins))  # (a spill to / reload from the stack: the same misplacement, the lanes outside the mask lose the value)
            if not (MOVABLE.match(ins) or spill):
                raise RuntimeError(f"line {i + 1} ({func} {label}): `{ins}` sits before the exec restore and is neither a register copy nor a spill")
            mine = regs(ins)
            for j in range(i + 1, at + 1):
                other = instruction(lines[j])
                if other and j not in idx and (regs(other) & mine):
                    raise RuntimeError(f"line {i + 1} ({func} {label}): cannot move `{ins}` past `{other}`")
                if other and spill and not ELSE_FLIP.match(instruction(lines[at])) and other.startswith("s_waitcnt") and "vmcnt" in other:
                    # one memory operation fewer is in flight at this wait than the compiler counted: "at most N outstanding" must
                    # become "at most N - 1" to keep guaranteeing the same older operations complete
                    n = int(re.search(r"vmcnt\((\d+)\)", other).group(1))
                    waits[j] = waits.get(j, n) - 1
                    if waits[j] < 0:
                        waits[j] = 0
            moved.add(i)
            if ELSE_FLIP.match(instruction(lines[at])):
                repeats.add(i)
            report.append(f"{func} {label}: `{ins}` {'repeated' if i in repeats else 'moved'} behind `{instruction(lines[at])}`")
        moves[at] = sorted(moved)
    skip = {i for v in moves.values() for i in v if i not in repeats}
    out = []
    for i, raw in enumerate(lines):
        if i in skip:
            continue
        if i in waits:
            raw = re.sub(r"vmcnt\(\d+\)", f"vmcnt({waits[i]})", raw)
        out.append(raw)
        for j in moves.get(i, ()):
            out.append(lines[j])
            if instruction(lines[j]).startswith("scratch_load"):
                # The wait the compiler placed for this reload stays where it was -- in front of the exec restore, now in front of the
                # load itself -- so nothing waits for it any more: neither a copy moved along with it that reads the reloaded
                # register, nor the block's own code. A moved reload therefore completes on the spot.
                out.append("\ts_waitcnt vmcnt(0)")
    return out, report


def main(argv):
    if argv and argv[0] == "--fix":
        src, dst = argv[1], argv[argv.index("-o") + 1]
        lines = open(src, errors="replace").read().split("\n")
        out, report = fix(lines)
        open(dst, "w").write("\n".join(out))
        for r in report:
            print("isa_exec_check:", r)
        return 1 if lint(dst, out) else 0
    bad = 0
    for p in argv:
        bad += lint(p, open(p, errors="replace").read().split("\n"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
