"""profiles/<tag>/rowcal.csv from a tools/ubench/rowcal.sh run (gpurun_out/rowcal_{fetch,write}/): FETCH_SIZE / WRITE_SIZE reported
for every kernel of tools/ubench/rowcal.hip against the bytes it really touches.   python tools/rowcal_collect.py <tag> [D]"""
import csv, glob, os, sys
tag = sys.argv[1]
D = int(sys.argv[2]) if len(sys.argv) > 2 else 10
true_bytes = (1 << 25) * D * 8
rows = []
for part, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    f = max(glob.glob(f"gpurun_out/rowcal_{part}/*/*_counter_collection.csv"), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != ctr:
            continue
        name = r["Kernel_Name"].split("(")[0]
        kb = float(r["Counter_Value"])
        rows.append((name, r["Grid_Size"], ctr, kb, kb * 1024 / true_bytes))
os.makedirs(f"profiles/{tag}", exist_ok=True)
with open(f"profiles/{tag}/rowcal.csv", "w", newline="") as f:
    f.write(f"# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/rowcal.hip, {true_bytes} B per kernel, D = {D}, rocprofv3 --pmc, MI355X)\n"
            "# tile_* = the split pipeline's pattern: a wavefront walks rows of W lanes x 8 B (grid 33554432 threads: W = 64; 134217728: W = 16)\n")
    w = csv.writer(f)
    w.writerow(["kernel", "grid_threads", "counter", "reported_KB", "reported_bytes/true_bytes"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{r[4]:.4f}"])
print(open(f"profiles/{tag}/rowcal.csv").read())
