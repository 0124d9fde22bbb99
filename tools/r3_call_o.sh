#!/bin/bash
export TMPDIR=/tmp PYTHONPATH=$PWD
ROOT=$PWD
O=$PWD/gpurun_out/r3o; mkdir -p $O
cd /tmp
export MVD_WGRAD16Z_NG=${NG:-2}
for d in ${DBGS:-0 2 3 8}; do
  MVD_WG16Z_DBG=$d timeout -k 10 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/p$d -- python3 $ROOT/tools/bench_conv.py --dtype bf16 --layers enc0.conv1 --what wgrad --iters 10 > $O/log$d.txt 2>&1
  python3 - <<PY
import csv,glob
cc=glob.glob('$O/p$d/*/*counter_collection.csv')[0]; kt=glob.glob('$O/p$d/*/*kernel_trace.csv')[0]
dur={}
for r in csv.DictReader(open(kt)):
    if 'k_wgrad16z' in r['Kernel_Name']: dur[r['Dispatch_Id']]=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
cyc={}
for r in csv.DictReader(open(cc)):
    if 'k_wgrad16z' in r['Kernel_Name'] and r['Counter_Name']=='GRBM_GUI_ACTIVE': cyc[r['Dispatch_Id']]=cyc.get(r['Dispatch_Id'],0)+float(r['Counter_Value'])
ks=sorted(set(dur)&set(cyc))[3:]
import statistics
d=statistics.median(dur[k] for k in ks); c=statistics.median(cyc[k] for k in ks)
print('DBG=$d dur_us=%.1f cycles/XCD=%.0f GHz=%.2f'%(d/1e3,c/8,c/8/d))
PY
  rm -rf $O/p$d
done
