"""Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/ into profiles/<tag>/ and refresh
profiles/traffic.json (HBM bytes per launch of the hot kernel: FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KB units)."""
import csv, glob, json, os, shutil, sys
tag = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "funnel-10-65536"
iters_per_launch = int(sys.argv[3]) if len(sys.argv) > 3 else 200
n_timed = int(sys.argv[4]) if len(sys.argv) > 4 else 1          # dispatches of the timed region (bench default: 200 steps in one launch)
out = f"profiles/{tag}"
os.makedirs(out, exist_ok=True)
ks = max(glob.glob(f"gpurun_out/prof_{tag}_kt/*/*_kernel_stats.csv"), key=os.path.getmtime)
shutil.copy(ks, f"{out}/kernel_stats.csv")
rows = []
vals = {}
for name in ("fetch", "write"):
    f = max(glob.glob(f"gpurun_out/prof_{tag}_{name}/*/*_counter_collection.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if "lane_kernel" in r["Kernel_Name"] or "iter_kernel" in r["Kernel_Name"]:
            rows.append({k: r[k] for k in ("Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "VGPR_Count",
                                           "Accum_VGPR_Count", "SGPR_Count", "Scratch_Size", "Counter_Name", "Counter_Value")})
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
w = csv.DictWriter(open(f"{out}/pmc_fetch_write.csv", "w"), fieldnames=list(rows[0].keys()))
w.writeheader(); w.writerows(rows)
# timed launches = the last 4 dispatches (200 steps / 50 per launch); draws are recorded only there
# Corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE under-reports wide coalesced reads by 2x and is "uncalibrated" for
# other widths.  This kernel's reads are lane-private 8-byte row reads; calibrated on a known byte count
# (tools/ubench/rowcal.hip, profiles/r01b/rowcal.csv) FETCH_SIZE reports 1.383x the bytes of that pattern, WRITE_SIZE 1.002x.
ROW_FETCH_FACTOR = 1.383
fetch_raw = sum(vals["FETCH_SIZE"][-n_timed:]) / n_timed * 1024
fetch = fetch_raw / ROW_FETCH_FACTOR
write = sum(vals["WRITE_SIZE"][-n_timed:]) / n_timed * 1024
tj_path = "profiles/traffic.json"
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
tj[key] = {"hbm_bytes_per_launch": fetch + write, "fetch_bytes_calibrated": fetch, "fetch_bytes_raw": fetch_raw, "fetch_bytes_if_doubled": 2 * fetch_raw,
           "write_bytes": write, "iters_per_launch": iters_per_launch,
           "source": f"profiles/{tag}/pmc_fetch_write.csv (mean of the timed dispatches); FETCH_SIZE / 1.383: calibration of the lane-private row-read pattern, profiles/r01b/rowcal.csv (the 2x rule of MI355X_MICROARCH.md §HBM applies to wide coalesced reads only)"}
json.dump(tj, open(tj_path, "w"), indent=1)
print(open(f"{out}/kernel_stats.csv").read())
print(json.dumps(tj[key], indent=1))
