"""Mean SQ counters per dispatch and per wavefront for every hot kernel of a tools/sqstats.sh run (last third of the dispatches)."""
import csv, glob, sys, collections
base = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(base + "_*/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "klhr" not in k or "init" in k:
            continue
        k = k.split("(")[0].replace("void klhr::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_vgpr"] = [r["VGPR_Count"], r["Accum_VGPR_Count"], r["Scratch_Size"], r["LDS_Block_Size"] if "LDS_Block_Size" in r else "?"]
for k, d in agg.items():
    m = {c: sum(v[len(v) * 2 // 3:]) / max(1, len(v[len(v) * 2 // 3:])) for c, v in d.items() if not c.startswith("_")}
    w = m.get("SQ_WAVES", 1)
    print(f"== {k}  (vgpr/agpr/scratch/lds {d['_vgpr']})  waves {w:.0f}")
    if "SQ_INSTS_VALU" in m:
        wc = m["SQ_WAVE_CYCLES"] * 4
        print(f"   per wave: VALU {m['SQ_INSTS_VALU']/w:9.0f}  SALU {m['SQ_INSTS_SALU']/w:8.0f}  cycles {wc/w:10.0f}  cyc/VALU {wc/m['SQ_INSTS_VALU']:.2f}"
              f"  VALU-active {m['SQ_ACTIVE_INST_VALU']*4/wc*100:.0f}%  wait {m['SQ_WAIT_ANY']*4/wc*100:.0f}%  issue-stall {m['SQ_WAIT_INST_ANY']*4/wc*100:.0f}%"
              f"  lanes-on {m['SQ_THREAD_CYCLES_VALU']/(m['SQ_ACTIVE_INST_VALU']*64)*100:.0f}%")
    if "SQ_INSTS_VALU_FMA_F64" in m:
        print(f"   per wave: fma {m['SQ_INSTS_VALU_FMA_F64']/w:8.0f} add {m['SQ_INSTS_VALU_ADD_F64']/w:8.0f} mul {m['SQ_INSTS_VALU_MUL_F64']/w:8.0f} trans {m['SQ_INSTS_VALU_TRANS_F64']/w:7.0f}"
              f"  LDS {m['SQ_INSTS_LDS']/w:7.0f} VMEM {m['SQ_INSTS_VMEM']/w:7.0f}  busy-cycles {m['SQ_BUSY_CYCLES']*4:.0f}")
