#!/bin/bash
# usage (GPU box): bash tools/stamps_tile.sh   — cycle split of the tile kernel / the 16-chain streaming kernel (diagnostic library
# klhr_amd/lib/diag/libklhr_stamps.so = tools/build_unit_variant.sh stamps split_v0 -DKLHR_STAMPS), after checking that the
# diagnostic build returns the production library's bits (tools/libcmp_iter.py)
R=$GRAFT_REPO_ROOT
L=$R/klhr_amd/lib/diag/libklhr_stamps.so
python3 $R/tools/libcmp_iter.py prod ill-normal 100 4096 && KLHR_LIB=$L python3 $R/tools/libcmp_iter.py stamps ill-normal 100 4096 || exit 1
python3 - <<PY
import numpy as np
a=np.load("$R/gpurun_out/libcmp2_prod.npz"); b=np.load("$R/gpurun_out/libcmp2_stamps.npz")
for f in ("theta","rho","eta","rng"): print("stamps build vs production,", f, "equal:", np.array_equal(a[f],b[f]))
PY
export KLHR_LIB=$L
python3 $R/tools/stamps_stream.py ill-normal 100 16384
python3 $R/tools/stamps_stream.py ill-normal 100 8192
python3 $R/tools/stamps_stream.py funnel 100 16384
KLHR_FUSED=0 KLHR_GROUPS=1 KLHR_STREAM_PAIR=0 python3 $R/tools/stamps_stream.py ill-normal 100 16384
