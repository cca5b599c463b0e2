"""Copy the judged summaries of a tools/profile_cfg.sh run from gpurun_out/cfg_<tag>_* into profiles/<tag>/ and refresh
profiles/traffic.json[<key>] with the HBM bytes per launch of the dominant kernel.

    python tools/profile_cfg_collect.py <tag> <key> [calibration: tiled|row]

Corrections (MI355X_MICROARCH.md, HBM: calibrate on a known byte count in your own access pattern; tools/ubench/rowcal.hip,
profiles/r02_calibration/rowcal.csv): FETCH_SIZE reports 0.500x the bytes of the split pipeline's tiled reads (8 B per lane,
512-B or 128-B contiguous segments: double it) and 1.383-1.455x those of lane-private 8-byte row reads (monolithic kernels:
divide); WRITE_SIZE 1.000x / 1.002x."""
import csv, glob, json, os, shutil, sys, collections
tag, key = sys.argv[1], sys.argv[2]
cal = sys.argv[3] if len(sys.argv) > 3 else "tiled"
FETCH_FACTOR = {"tiled": 0.5, "row": 1.383}[cal]
out = f"profiles/{tag}"
os.makedirs(out, exist_ok=True)
base = f"gpurun_out/cfg_{tag}"
ks = max(glob.glob(base + "_kt/*/*_kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(ks, f"{out}/kernel_stats.csv")
stats = [r for r in csv.DictReader(open(ks)) if "klhr" in r["Name"] and "init" not in r["Name"]]
dom = max(stats, key=lambda r: float(r["TotalDurationNs"]))["Name"].split("(")[0]
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
ipl = bench["roofline"].get("iters_per_launch") or bench["steps"] / max(1, bench["roofline"].get("launches", 1))
tj_path = "profiles/traffic.json"
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[key] = {"kernel": dom, "hbm_bytes_per_launch": fetch + write, "fetch_bytes_calibrated": fetch, "fetch_bytes_raw": fetch_raw, "write_bytes": write,
           "iters_per_launch": ipl, "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_chain_iter"] * bench["config"]["chains_per_gpu"] * ipl,
           "source": f"profiles/{tag}/pmc_fetch_write.csv (mean of the last third of the dispatches); FETCH_SIZE / {FETCH_FACTOR}: calibration of the "
                     f"{'tiled 8-byte coalesced' if cal == 'tiled' else 'lane-private row'} read pattern, profiles/r02_calibration/rowcal.csv"}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(f"{out}/kernel_stats.csv").read()[:1500])
print(json.dumps(tj[key], indent=1))
