#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c5; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o ring -- python tools/ring_probe.py > $D/ring.json 2> $D/ring.err
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c5/ring_kernel_stats.csv')):
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8))
PY
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c5/ring_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
t0=int(rows[0]['Start_Timestamp'])
fin=[i for i,r in enumerate(rows) if 'k_bus_finalize' in r['Kernel_Name']]
print('finalize at rows', fin[:6], '...', fin[-6:], 'of', len(rows))
def show(i0,i1):
    for r in rows[i0:i1]:
        print('%10.1f %8.1f q%-3s %s'%((int(r['Start_Timestamp'])-t0)/1e3,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,r.get('Queue_Id','?'),r['Kernel_Name'][:60]))
i=fin[5]
show(i-14,i+14)
print('....')
i=fin[-3]
show(i-8,i+40)
PY
rm -f $D/*kernel_trace.csv
