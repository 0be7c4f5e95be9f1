cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export CFRK_MSP_CHUNKS=4
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tr -- python3 bench.py --cpu-reads 0 --steps 1 --warmup 1 > gpurun_out/tr.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/tr/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'msp_p' in r['Kernel_Name']]
rows=rows[-9:]
t0=int(rows[0]['Start_Timestamp'])
for r in rows:
    n=r['Kernel_Name']; n=n[n.find('msp_p'):n.find('msp_p')+6]
    print(n, r.get('Queue_Id'), (int(r['Start_Timestamp'])-t0)/1e6, (int(r['End_Timestamp'])-t0)/1e6)
PY
