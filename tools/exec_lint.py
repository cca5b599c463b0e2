#!/usr/bin/env python3
"""Lint gfx950 assembly for vector instructions placed ahead of a block's EXEC restore.

Background (open issues): at 256 VGPR + ~250 AGPR the register allocator splits live ranges through AGPRs
(`v_accvgpr_write` before a region, `v_accvgpr_read` after it).  In some builds of the hot kernels such copies landed at
the top of a block that is entered with EXEC = 0 — the exit block of a divergent loop (e.g. the ziggurat loop of
rng_std_normal), reached by `s_cbranch_execz` — AHEAD of the `s_or_b64 exec, exec, s[..]` that re-enables the lanes.
The copies then save nothing and the later reads return stale register contents: a chain's `draw` / window index came back
as garbage (memory fault on p.closures[cidx]) or the MH proposal became a polynomial constant.  The source is correct;
the placement is a code-generation fault that comes and goes with unrelated edits.

The check: in every block entered through `s_cbranch_execz <label>` (EXEC = 0 when the branch is taken, the surviving
subset of the lanes when the block is reached by falling through) or by falling out of `s_cbranch_execnz`, walk the
fall-through path — across `; %bb.N` markers, up to the first branch or label — to the first instruction that writes EXEC.
If that instruction can ADD lanes (any restore form: `s_or_b64 exec, exec, ..`, `s_mov_b64 exec, ..`, `s_xor_b64 exec, ..`,
`s_or_saveexec_b64 ..`, ...) no vector / memory / AGPR instruction may precede it: such an instruction ran for a subset of the
lanes it was meant for.  Lane moves of SGPR spills (`v_writelane` / `v_readlane`) ignore EXEC and are allowed — but they are
what moves the allocator's insertion point ahead of the restore (tools/ubench/execz_copy.hip reproduces that in 40 lines).

History of the rule ("Code-generation hazard"): round 1 checked only `s_or_b64 exec, exec` directly after the
label; the two historical builds that can be rebuilt were re-checked in round 2 — the faulting `<glmm-poisson, SUBKLHRSINH,
over-relaxed>` unit of commit e4cb1ae is flagged (36 AGPR copies ahead of the restore in klhr_iter_kernel<2,true,2>), the
`-DKLHR_STAMPS` octet build of commit e2cbbed is NOT, by either form of the rule: its wrong MH proposal has another cause.

usage: exec_lint.py file.s [file.s ...]     exit status 1 if anything is flagged
"""
import re
import sys

VEC = re.compile(r"^\s*(v_(?!readlane|writelane|readfirstlane)|ds_|global_|scratch_|buffer_|flat_)")      # lane moves ignore EXEC
LABEL = re.compile(r"^(\.LBB[0-9_]+):")
FALL = re.compile(r"^; %bb\.\d+:")
FUNC = re.compile(r"^(_Z\w+):")
EXECZ = re.compile(r"^\s*s_cbranch_execz\s+(\.LBB[0-9_]+)")
EXECNZ = re.compile(r"^\s*s_cbranch_execnz\s+")
ANYBR = re.compile(r"^\s*s_(c?branch\w*|setpc_b64|swappc_b64|endpgm)\b")
WRITES_EXEC = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_\w*saveexec")
# EXEC writes that can switch lanes ON (restores); s_and / s_andn2 (and their saveexec forms) only remove lanes
COPIES_EXEC = re.compile(r"^\s*s_mov_b64\s+s\[\d+:\d+\],\s*exec\b")      # `s_mov sX, exec; s_and sX, sX, ..; s_mov exec, sX` narrows: not a restore
MOV_TO_EXEC = re.compile(r"^\s*s_mov_b64\s+exec\b")
ADDS_LANES = re.compile(r"^\s*s_(or|mov|xor|xnor|orn2|cselect|nor|nand)_b64\s+exec\b|^\s*s_(or|xor|orn2|nor|nand|xnor|andn1|orn1)\w*_saveexec")


def lint(path):
    lines = open(path).read().splitlines()
    execz_targets = set()
    for ln in lines:
        m = EXECZ.match(ln)
        if m:
            execz_targets.add(m.group(1))
    found = []
    func = "?"
    i, n = 0, len(lines)
    prev_instr = ""
    while i < n:
        ln = lines[i]
        m = FUNC.match(ln)
        if m:
            func = m.group(1)
        entry = None
        ml = LABEL.match(ln)
        if ml and ml.group(1) in execz_targets:
            entry = ml.group(1)
        elif FALL.match(ln) and EXECNZ.match(prev_instr):
            entry = ln.split(":")[0].strip("; ")
        if entry:
            vec = []
            copied_exec = False
            j = i + 1
            while j < n:
                s = lines[j]
                if FUNC.match(s) or ANYBR.match(s) or LABEL.match(s):
                    break          # (a label starts another block with a restore of its own; `; %bb.N` markers are walked through)
                if COPIES_EXEC.match(s):
                    copied_exec = True
                if WRITES_EXEC.match(s):
                    restore = ADDS_LANES.match(s) and not (MOV_TO_EXEC.match(s) and copied_exec)
                    if restore and vec:
                        found.append((func, entry, j + 1, len(vec), vec[0].strip()))
                    break
                if VEC.match(s):
                    vec.append(s)
                j += 1
        st = ln.strip()
        if st and not st.startswith(";") and not st.startswith(".") and not st.endswith(":"):
            prev_instr = ln
        i += 1
    return found


def main(argv):
    bad = 0
    for path in argv:
        for func, entry, line, nvec, first in lint(path):
            bad += 1
            print(f"{path}:{line}: {func}: block {entry}: {nvec} vector instruction(s) ahead of the EXEC restore (first: {first})")
    print(f"exec_lint: {bad} block(s) flagged in {len(argv)} file(s)")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
