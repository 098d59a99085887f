#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c15; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --output-format csv -d $D -o tr -- python tools/stagger_probe.py > $D/tr.json 2> $D/tr.err
tail -1 $D/tr.json
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/c15/tr_kernel_trace.csv')))
print(rows[0].keys())
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'],r.get('Queue_Id'),r.get('Stream_Id')) for r in rows]
ev.sort()
# take a window late in the run where tiles kernels dominate
idx=[i for i,e in enumerate(ev) if '4, 8, 4, 9' in e[2]]
mid=idx[len(idx)//2]
t0=ev[mid][0]
out=open('gpurun_out/c15/window.txt','w')
for e in ev[mid:mid+40]:
    name='L' if '4, 8, 4, 9' in e[2] else ('G' if '4, 4, 4, 10' in e[2] else e[2][:30])
    line='%-6s q=%s s=%s start %8.1f end %8.1f dur %6.1f'%(name,e[3],e[4],(e[0]-t0)/1e3,(e[1]-t0)/1e3,(e[1]-e[0])/1e3)
    print(line); out.write(line+'\n')
PY
rm -f $D/tr_kernel_trace.csv
