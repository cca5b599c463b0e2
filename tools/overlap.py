"""Kernel timeline of a rocprofv3 --kernel-trace run: how much of the wall time two or more kernels were in flight.
usage: python tools/overlap.py <kernel_trace.csv> [skip_first_n_kernels]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = []
ks = [r for r in rows if "klhr" in r["Kernel_Name"] and "init" not in r["Kernel_Name"]][skip:]
for r in ks:
    ev.append((int(r["Start_Timestamp"]), 1)); ev.append((int(r["End_Timestamp"]), -1))
ev.sort()
t0, t1 = ev[0][0], ev[-1][0]
depth, last, hist = 0, t0, {}
for t, d in ev:
    hist[depth] = hist.get(depth, 0) + (t - last)
    depth += d; last = t
tot = t1 - t0
print(f"{len(ks)} kernels over {tot/1e3:.1f} us")
for k in sorted(hist): print(f"  {k} kernels in flight: {100*hist[k]/tot:5.1f} %")
for r in ks[:12]:
    print(f"  {(int(r['Start_Timestamp'])-t0)/1e3:9.1f} -> {(int(r['End_Timestamp'])-t0)/1e3:9.1f} us  queue {r.get('Queue_Id','?')}  {r['Kernel_Name'][12:60]}")
