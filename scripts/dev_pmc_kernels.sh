#!/bin/bash
# Dev: issue / stall counters (one rocprofv3 --pmc pass) summed per kernel over three train steps; run ON the GPU box from the repo root:
#   bash scripts/dev_pmc_kernels.sh      (writes gpurun_out/r02am_pmc*, prints the table profiles/r02am_sq_counters_by_kernel.txt was made from)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE --output-format csv -d /root/repo/gpurun_out/r02am_pmc -o sq -- python3 /root/repo/bench.py --steps 2 --warmup 1 --no-roofline --no-infer --no-cpu-baseline > /root/repo/gpurun_out/r02am_pmc_bench.json 2>/root/repo/gpurun_out/r02am_pmc.err
python3 - <<'PY'
import csv,glob,collections
f=glob.glob('/root/repo/gpurun_out/r02am_pmc/**/*counter_collection.csv',recursive=True)[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
seen=set()
for r in csv.DictReader(open(f)):
    k=r['Kernel_Name'].split('(')[0][:60]
    agg[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(r['Dispatch_Id'])
    if (k,key) not in seen: seen.add((k,key)); n[k]+=1
rows=[]
for k,c in agg.items():
    wc=c.get('SQ_WAVE_CYCLES',0)
    if wc<=0: continue
    rows.append((c.get('GRBM_GUI_ACTIVE',0),k,n[k],c))
rows.sort(reverse=True)
print('kernel | launches | GUI_ACTIVE share | wait_any/wave | wait_inst/wave | active_any/wave | active_valu/wave | valu insts per wave-cycle')
tot=sum(r[0] for r in rows)
for g,k,nn,c in rows[:16]:
    wc=c['SQ_WAVE_CYCLES']
    print(f"{k:58s} {nn:5d} {g/tot:6.3f}  {c['SQ_WAIT_ANY']/wc:5.2f} {c['SQ_WAIT_INST_ANY']/wc:5.2f} {c['SQ_ACTIVE_INST_ANY']/wc:5.2f} {c['SQ_ACTIVE_INST_VALU']/wc:5.2f}  busy/gui {c['SQ_BUSY_CYCLES']/max(c['GRBM_GUI_ACTIVE'],1):5.2f} valu/busy {c['SQ_ACTIVE_INST_VALU']/max(c['SQ_BUSY_CYCLES'],1):6.2f}")
PY
