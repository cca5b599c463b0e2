import csv,glob,collections
import os; f=max(glob.glob("gpurun_out/pmc_ph/*/*_counter_collection.csv"),key=os.path.getmtime)
rows=[r for r in csv.DictReader(open(f)) if 'lane_kernel' in r['Kernel_Name'] or 'iter_kernel' in r['Kernel_Name']]
by=collections.OrderedDict()
for r in rows: by.setdefault(int(r['Dispatch_Id']),{})[r['Counter_Name']]=float(r['Counter_Value'])
names=["step50","step50","step20","direction","fit","KL eval","line eval","mh","adapt","step1"]
import sys
waves=int(sys.argv[1]) if len(sys.argv)>1 else 1024
for (d,c),n in zip(by.items(),names):
    print(f"{n:10s} VALU/wave {c['SQ_INSTS_VALU']/waves:10.0f}  SALU/wave {c['SQ_INSTS_SALU']/waves:9.0f}  LDS {c['SQ_INSTS_LDS']/waves:7.0f} VMEM {c['SQ_INSTS_VMEM']/waves:7.0f}  wavecyc/wave {4*c['SQ_WAVE_CYCLES']/waves:10.0f}  valu-active {c['SQ_ACTIVE_INST_VALU']/c['SQ_WAVE_CYCLES']:.2f} wait {c['SQ_WAIT_ANY']/c['SQ_WAVE_CYCLES']:.2f}")
