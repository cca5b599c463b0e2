"""Copy the judged summaries of a tools/profile_cfg.sh run from gpurun_out/cfg_<tag>_* into profiles/<tag>/ and refresh
profiles/traffic.json[<key>] with the HBM bytes per launch of the dominant kernel.

    python tools/profile_cfg_collect.py <tag> <key> [calibration: tiled|row]

Corrections (MI355X_MICROARCH.md, HBM: calibrate on a known byte count in your own access pattern; tools/ubench/rowcal.hip,
profiles/r02_calibration/rowcal.csv): FETCH_SIZE reports 0.500x the bytes of the split pipeline's tiled reads (8 B per lane,
512-B or 128-B contiguous segments: double it) and 1.40x those of lane-private 8-byte row reads at D = 10 (monolithic
kernels: divide; three runs of the same microbenchmark gave 1.3832 (profiles/r01b), 1.4547 (r02_calibration) and 1.3565
(r03_calibration): the factor moves by +-3.5 % between runs, their mean is used); WRITE_SIZE 1.000x (tiled) / 1.003x (rows).

Every entry carries `build_id` = klhr_build_id() of the library the bench line was produced with (bench.py prints it): bench.py
quotes an entry only while the loaded library is that build.  The entry also carries the counter-derived fp64 rate of the
dominant kernel: (2 FMA + ADD + MUL + TRANS) fp64 instructions x 64 lanes x the lanes-active share, per launch."""
import csv, glob, json, os, shutil, sys, collections
tag, key = sys.argv[1], sys.argv[2]
cal = sys.argv[3] if len(sys.argv) > 3 else "tiled"
FETCH_FACTOR = {"tiled": 0.5, "row": 1.40}[cal]
out = f"profiles/{tag}"
os.makedirs(out, exist_ok=True)
base = f"gpurun_out/cfg_{tag}"
ks = max(glob.glob(base + "_kt/*/*_kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(ks, f"{out}/kernel_stats.csv")
stats = [r for r in csv.DictReader(open(ks)) if "klhr" in r["Name"] and "init" not in r["Name"]]
# the roofline's kernel (bench.py): the one that moves the iteration's algorithmic bytes — the streaming / whole-iterations-per-tile
# kernel of the split pipeline, the hot kernel of the monolithic mappings
mover = [r for r in stats if "klhr_stream" in r["Name"] or "klhr_tile" in r["Name"]]
dom = max(mover or stats, key=lambda r: float(r["TotalDurationNs"]))["Name"].split("(")[0]
COLS = ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size",
        "Counter_Name", "Counter_Value")
def rows_of(part):
    f = max(glob.glob(f"{base}_{part}/*/*_counter_collection.csv"), key=os.path.getmtime)
    return [{k: r.get(k, "") for k in COLS} for r in csv.DictReader(open(f)) if "klhr" in r["Kernel_Name"] and "init" not in r["Kernel_Name"]]
# SQ counters: mean per dispatch and kernel over the last third of the dispatches (the timed regions)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for part in ("sq1", "sq2"):
    for r in rows_of(part):
        k = r["Kernel_Name"].split("(")[0]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta[k] = (r["Grid_Size"], r["Workgroup_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"])
with open(f"{out}/pmc_sq.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "grid", "workgroup", "lds_bytes", "vgpr", "agpr", "scratch", "counter", "mean_per_dispatch", "dispatches_averaged"])
    for k, d in agg.items():
        for c, v in sorted(d.items()):
            tail = v[len(v) * 2 // 3:]
            w.writerow([k, *meta[k], c, sum(tail) / len(tail), len(tail)])
# FETCH / WRITE per dispatch
fw = collections.defaultdict(lambda: collections.defaultdict(list))
allrows = []
for part in ("fetch", "write"):
    for r in rows_of(part):
        allrows.append(r)
        fw[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_fetch_write.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "counter", "mean_KB_per_dispatch", "dispatches_averaged", "min_KB", "max_KB"])
    for k, d in fw.items():
        for c, v in sorted(d.items()):
            tail = v[len(v) * 2 // 3:]
            w.writerow([k, c, sum(tail) / len(tail), len(tail), min(tail), max(tail)])
d = fw[dom]
tail = lambda v: v[len(v) * 2 // 3:]
fetch_raw = sum(tail(d["FETCH_SIZE"])) / len(tail(d["FETCH_SIZE"])) * 1024
write = sum(tail(d["WRITE_SIZE"])) / len(tail(d["WRITE_SIZE"])) * 1024
fetch = fetch_raw / FETCH_FACTOR
bench = json.loads([l for l in open(base + "_kt.log") if l.startswith("{")][-1])
if not bench.get("build_id"):
    raise SystemExit("the bench line carries no build_id: profile with a bench.py / library that reports klhr_build_id()")
ipl = bench["roofline"].get("iters_per_launch") or bench["steps"] / max(1, bench["roofline"].get("launches", 1))
sq = {c: sum(tail(v)) / len(tail(v)) for c, v in agg[dom].items()}
lanes_active = sq["SQ_THREAD_CYCLES_VALU"] / (sq["SQ_ACTIVE_INST_VALU"] * 64)
fp64_insts = 2 * sq["SQ_INSTS_VALU_FMA_F64"] + sq["SQ_INSTS_VALU_ADD_F64"] + sq["SQ_INSTS_VALU_MUL_F64"] + sq["SQ_INSTS_VALU_TRANS_F64"]
tj_path = "profiles/traffic.json"
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[key] = {"kernel": dom, "build_id": bench["build_id"], "profile_dir": f"profiles/{tag}",
           "hbm_bytes_per_launch": fetch + write, "fetch_bytes_calibrated": fetch, "fetch_bytes_raw": fetch_raw, "write_bytes": write,
           "iters_per_launch": ipl, "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_chain_iter"] * bench["config"]["chains_per_gpu"] * ipl,
           "source": f"profiles/{tag}/pmc_fetch_write.csv (mean of the last third of the dispatches); FETCH_SIZE / {FETCH_FACTOR}: calibration of the "
                     f"{'tiled 8-byte coalesced' if cal == 'tiled' else 'lane-private row'} read pattern, profiles/r01b, r02_calibration, r03_calibration (rowcal.csv)",
           "fp64_flops_per_launch": fp64_insts * 64 * lanes_active, "lanes_active": lanes_active,
           "valu_active": sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"], "waiting": sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"],
           "valu_per_wavefront_iter": sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] / ipl,
           "fp64_source": f"profiles/{tag}/pmc_sq.csv: (2 FMA + ADD + MUL + TRANS)_F64 x 64 lanes x SQ_THREAD_CYCLES_VALU / (64 SQ_ACTIVE_INST_VALU), "
                          "mean of the last third of the dispatches",
           "bench_value_when_profiled": bench["value"], "bench_ms_per_step_when_profiled": bench["ms_per_step"],
           "rocprof_avg_launch_ms": float(next(r for r in stats if r["Name"].split("(")[0] == dom)["AverageNs"]) / 1e6,
           # all launches of the run, per draw: setup draws + --warmup + the timed regions (bench.py: SETUP_DRAWS = 150, REPEATS = 5)
           "rocprof_ms_per_iteration_all_kernels": sum(float(r["TotalDurationNs"]) for r in stats) / 1e6 / (150 + bench["warmup"] + 5 * bench["steps"] + (0 if ipl > 1 else min(bench["steps"], 20)))}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(f"{out}/kernel_stats.csv").read()[:1500])
print(json.dumps(tj[key], indent=1))
