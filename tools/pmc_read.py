import csv,glob,sys,collections
tag=sys.argv[1]
agg=collections.OrderedDict()
for f in sorted(glob.glob(f"gpurun_out/pmc_{tag}_*/*/*_counter_collection.csv")):
    rows=[r for r in csv.DictReader(open(f)) if 'iter_kernel' in r['Kernel_Name'] or 'lane_kernel' in r['Kernel_Name']]
    last=max(int(r['Dispatch_Id']) for r in rows)
    for r in rows:
        if int(r['Dispatch_Id'])==last: agg[r['Counter_Name']]=float(r['Counter_Value'])
for k,v in agg.items(): print(f"{k:32s} {v:16.0f}")
w=agg.get("SQ_WAVES",1024)
if 'SQ_INSTS_VALU' in agg: print("VALU insts per wave per iteration (50 it):", agg['SQ_INSTS_VALU']/w/50)
