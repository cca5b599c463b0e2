"""Build libklhr_hip.so for gfx950 in-tree (klhr_amd/lib/).  `python -m klhr_amd.build [--force] [-v]`

Eleven translation units are compiled in parallel and linked: klhr_api.hip (C ABI + small kernels), klhr_hot.hip once per
(sampler variant, over-relaxation) pair (six) plus once for the slice sampler — see the comment at the top of klhr_hot.hip —
and klhr_split.hip once per variant (three).  Only the units whose sources or flags changed are recompiled.

Every unit's device assembly is kept (-save-temps) and checked with tools/exec_lint.py: at 256 VGPR + ~250 AGPR the
register allocator has been seen to place AGPR save copies ahead of a block's EXEC restore (they then save nothing and
a chain's counters come back as garbage).  A flagged unit is rebuilt once with another allocation order
(RETRY_FLAGS); if it is still flagged the build fails rather than ship the kernel."""
import glob
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.environ.get("KLHR_BUILD_OUT") or os.path.join(HERE, "lib", "libklhr_hip.so")      # KLHR_BUILD_OUT: diagnostic builds
OBJ = os.path.join(os.path.dirname(OUT), "obj" if not os.environ.get("KLHR_BUILD_OUT") else "obj_diag")
COMMON = ["klhr_kernels.hpp", "device_opt.hpp", "device_models.hpp", "device_rng.hpp", "zig_tables.h", os.path.join("..", "..", "include", "klhr.h")]
UNIT_DEPS = {"api": ["klhr_api.hip"], "hot": ["klhr_hot.hip", "klhr_lane_kernel.hpp"], "split": ["klhr_split.hip", "klhr_split.hpp", "klhr_lane_kernel.hpp"]}
DEPS = sorted(set(COMMON + [d for v in UNIT_DEPS.values() for d in v]))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"]
# diagnostic builds: KLHR_EXTRA_FLAGS (e.g. -DKLHR_STAMPS) for the units whose name starts with one of KLHR_EXTRA_UNITS (comma
# separated prefixes such as "free" or "hot_v0"; default: every unit)
EXTRA_FLAGS = os.environ.get("KLHR_EXTRA_FLAGS", "").split()
EXTRA_UNITS = [u for u in os.environ.get("KLHR_EXTRA_UNITS", "").split(",") if u]


def _unit_flags(name):
    return EXTRA_FLAGS if (not EXTRA_UNITS or any(name.startswith(u) for u in EXTRA_UNITS)) else []

RETRY_FLAGS = ["-mllvm", "-greedy-regclass-priority-trumps-globalness=1"]
LINT = os.path.join(HERE, "..", "tools", "exec_lint.py")


def source_hash():
    """sha256 (hex) of the library's sources — every file of DEPS, by name and content.  Compiled into the library as
    klhr_build_id() and stored with every profiles/traffic.json entry: bench.py reports measured HBM traffic only when the
    entry was taken on the build that is loaded (evidence of a kernel that no longer exists must not be quoted)."""
    import hashlib
    h = hashlib.sha256()
    for d in sorted(DEPS):
        with open(os.path.join(SRC, d), "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _lint(objdir):
    """Flagged blocks in the device assembly that -save-temps left in objdir (empty list = clean).  Fails closed: a unit
    whose device assembly cannot be found (toolchain naming change, flags that drop -save-temps) is an error, not "clean"."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(LINT)))
    import exec_lint
    out = []
    files = glob.glob(os.path.join(objdir, "*amdgcn*.s"))
    if not files:
        raise RuntimeError(f"klhr_amd.build: no device assembly (*amdgcn*.s) found in {objdir}: the code-generation check "
                           "cannot run (did -save-temps change its file naming?)")
    for asm in files:
        out += exec_lint.lint(asm)
    if os.environ.get("KLHR_BUILD_VERBOSE"):
        sys.stderr.write(f"klhr_amd.build: {os.path.basename(objdir)}: linted {len(files)} device assembly file(s), {len(out)} block(s) flagged\n")
    return out


def needs_build():
    """the library is missing, or was built from other sources (its id file beside it names the source hash it was built from;
    file times are not trusted: a snapshot copy to another machine may change them)"""
    idf = OUT + ".id"
    if not os.path.exists(OUT) or not os.path.exists(idf):
        return True
    return open(idf).read().strip() != source_hash() + " " + " ".join(FLAGS + EXTRA_FLAGS + EXTRA_UNITS)


def _unit_key(kind, args, flags):
    """what an object file depends on: its sources, the flags, the compiler"""
    import hashlib
    h = hashlib.sha256()
    h.update(repr((sorted(flags), [a for a in args if a.startswith("-D")], os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"))).encode())
    for d in sorted(COMMON + UNIT_DEPS[kind]):
        with open(os.path.join(SRC, d), "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def build(force=False, verbose=False, extra=()):
    """Compile the translation units whose sources changed (per-unit key = hash of the unit's sources and flags, kept beside the
    object file), lint their device assembly, link.  api.o carries the build id (source_hash) and is rebuilt with any change."""
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = FLAGS + list(extra) + (["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
    jobs = [("api.o", "api", ["-c", os.path.join(SRC, "klhr_api.hip"), f'-DKLHR_BUILD_ID="{source_hash()}"'])]
    for v in (0, 1, 2):
        for o in (0, 1):
            jobs.append((f"hot_v{v}_o{o}.o", "hot", ["-c", os.path.join(SRC, "klhr_hot.hip"), f"-DKLHR_VARIANT={v}", f"-DKLHR_OVERRELAX={o}"]))
    jobs.append(("hot_v3_o0.o", "hot", ["-c", os.path.join(SRC, "klhr_hot.hip"), "-DKLHR_VARIANT=3", "-DKLHR_OVERRELAX=0"]))     # slice.py
    for v in (0, 1, 2):           # split-phase pipeline (klhr_split.hpp), one unit per variant
        jobs.append((f"split_v{v}.o", "split", ["-c", os.path.join(SRC, "klhr_split.hip"), f"-DKLHR_VARIANT={v}"]))

    def objpath(name):
        return os.path.join(OBJ, name[:-2], name)

    def launch(name, args, extra_flags=()):
        d = os.path.join(OBJ, name[:-2])
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d)
        cmd = [hipcc] + flags + _unit_flags(name) + list(extra_flags) + args + ["-save-temps=obj", "-o", os.path.join(d, name)]
        return subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)

    def finish(name, pr):
        out, _ = pr.communicate()
        if verbose or pr.returncode != 0:
            sys.stderr.write(out)
        if pr.returncode != 0:
            raise RuntimeError(f"hipcc failed on {name}")

    keys = {name: _unit_key(kind, args, flags + _unit_flags(name)) for name, kind, args in jobs}
    todo = []
    for name, kind, args in jobs:
        kf = os.path.join(OBJ, name[:-2], "key")
        fresh = not force and os.path.exists(objpath(name)) and os.path.exists(kf) and open(kf).read() == keys[name]
        if not fresh:
            todo.append((name, args))
    ncpu = int(os.environ.get("KLHR_BUILD_JOBS", str(os.cpu_count() or 4)))
    flagged = []
    for i0 in range(0, len(todo), ncpu):          # at most ncpu compilers at a time
        procs = [(name, args, launch(name, args)) for name, args in todo[i0:i0 + ncpu]]
        for name, args, pr in procs:
            finish(name, pr)
            bad = _lint(os.path.join(OBJ, name[:-2]))
            if bad:
                flagged.append((name, args, bad))
    retry = [(name, args, bad, launch(name, args, RETRY_FLAGS)) for name, args, bad in flagged]
    for name, args, bad, pr in retry:
        sys.stderr.write(f"klhr_amd.build: {name}: {len(bad)} block(s) with vector code ahead of the EXEC restore "
                         f"({bad[0][0]} {bad[0][1]}); rebuilding with {' '.join(RETRY_FLAGS)}\n")
        finish(name, pr)
        still = _lint(os.path.join(OBJ, name[:-2]))
        if still:
            raise RuntimeError(f"{name}: device code still fails tools/exec_lint.py after the retry: {still[:3]}")
    keep_asm = bool(os.environ.get("KLHR_KEEP_ASM"))          # 10 MB per unit; kept only on request
    for name, _ in todo:          # drop the bulky -save-temps files, record the unit's key
        d = os.path.join(OBJ, name[:-2])
        for f in os.listdir(d):
            if not (f == name or (keep_asm and f.endswith("gfx950.s"))):
                os.remove(os.path.join(d, f))
        with open(os.path.join(d, "key"), "w") as f:
            f.write(keys[name])
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [objpath(n) for n, _, _ in jobs])
    with open(OUT + ".id", "w") as f:
        f.write(source_hash() + " " + " ".join(FLAGS + EXTRA_FLAGS + EXTRA_UNITS))
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
