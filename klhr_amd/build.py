"""Build libklhr_hip.so for gfx950 in-tree (klhr_amd/lib/).  `python -m klhr_amd.build [--force]`"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "lib", "libklhr_hip.so")
SOURCES = ["klhr_api.hip"]
DEPS = ["klhr_api.hip", "klhr_kernels.hpp", "klhr_lane_kernel.hpp", "device_opt.hpp", "device_models.hpp", "device_rng.hpp", "zig_tables.h",
        os.path.join("..", "..", "include", "klhr.h")]


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(SRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-o", OUT] + [os.path.join(SRC, s) for s in SOURCES]
    if verbose:
        cmd.append("-Rpass-analysis=kernel-resource-usage")
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
