"""Build libklhr_hip.so for gfx950 in-tree (klhr_amd/lib/).  `python -m klhr_amd.build [--force] [-v]`

Eight translation units are compiled in parallel and linked: klhr_api.hip (C ABI + small kernels) and klhr_hot.hip once
per (sampler variant, over-relaxation) pair (six) plus once for the slice sampler — see the comment at the top of klhr_hot.hip."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libklhr_hip.so")
OBJ = os.path.join(HERE, "lib", "obj")
DEPS = ["klhr_api.hip", "klhr_hot.hip", "klhr_kernels.hpp", "klhr_lane_kernel.hpp", "device_opt.hpp", "device_models.hpp",
        "device_rng.hpp", "zig_tables.h", os.path.join("..", "..", "include", "klhr.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden"] + os.environ.get("KLHR_EXTRA_FLAGS", "").split()


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(SRC, d)) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not needs_build():
        return OUT
    os.makedirs(OBJ, exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = FLAGS + list(extra) + (["-Rpass-analysis=kernel-resource-usage"] if verbose else [])
    jobs = [("api.o", ["-c", os.path.join(SRC, "klhr_api.hip")])]
    for v in (0, 1, 2):
        for o in (0, 1):
            jobs.append((f"hot_v{v}_o{o}.o", ["-c", os.path.join(SRC, "klhr_hot.hip"), f"-DKLHR_VARIANT={v}", f"-DKLHR_OVERRELAX={o}"]))
    jobs.append(("hot_v3_o0.o", ["-c", os.path.join(SRC, "klhr_hot.hip"), "-DKLHR_VARIANT=3", "-DKLHR_OVERRELAX=0"]))     # slice.py
    procs = []
    for name, args in jobs:
        cmd = [hipcc] + flags + args + ["-o", os.path.join(OBJ, name)]
        procs.append((name, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for name, pr in procs:
        out, _ = pr.communicate()
        if verbose or pr.returncode != 0:
            sys.stderr.write(out)
        failed |= pr.returncode != 0
    if failed:
        raise RuntimeError("hipcc failed")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + [os.path.join(OBJ, n) for n, _ in jobs])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
