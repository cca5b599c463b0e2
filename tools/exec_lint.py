#!/usr/bin/env python3
"""Lint gfx950 assembly for vector instructions placed ahead of a block's EXEC restore.

Background (open issues): at 256 VGPR + ~250 AGPR the register allocator splits live ranges through AGPRs
(`v_accvgpr_write` before a region, `v_accvgpr_read` after it).  In some builds of the hot kernels such copies landed at
the top of a block that is entered with EXEC = 0 — the exit block of a divergent loop (e.g. the ziggurat loop of
rng_std_normal), reached by `s_cbranch_execz` — AHEAD of the `s_or_b64 exec, exec, s[..]` that re-enables the lanes.
The copies then save nothing and the later reads return stale register contents: a chain's `draw` / window index came back
as garbage (memory fault on p.closures[cidx]) or the MH proposal became a polynomial constant.  The source is correct;
the placement is a code-generation fault that comes and goes with unrelated edits.

The check: in every block entered through `s_cbranch_execz <label>` or by falling out of `s_cbranch_execnz`, the first
instruction that writes EXEC must not be preceded by a vector / memory instruction if it is an `s_or_b64 exec, exec, ..`.

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
WRITES_EXEC = re.compile(r"^\s*s_\w+\s+exec\b|^\s*s_\w*saveexec")
RESTORE = re.compile(r"^\s*s_or_b64\s+exec,\s*exec,")


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
            j = i + 1
            while j < n:
                s = lines[j]
                if LABEL.match(s) or FALL.match(s) or FUNC.match(s) or s.strip().startswith("s_cbranch") or s.strip().startswith("s_branch"):
                    break
                if WRITES_EXEC.match(s):
                    if RESTORE.match(s) and vec:
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
