#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c8; rm -rf $D; mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_onsets.py -m gpu -q -x 2>&1 | tail -5
python tools/stagger_probe.py 2>&1 | tail -1
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o st -- python tools/stagger_probe.py > $D/st.json 2> $D/st.err
tail -1 $D/st.json
python - <<'PY'
import csv
for r in csv.DictReader(open('gpurun_out/c8/st_kernel_stats.csv')):
    print(r['Name'][:110].ljust(110), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8))
PY
rm -f $D/*kernel_trace.csv
