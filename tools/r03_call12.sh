#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c12; rm -rf $D; mkdir -p $D
timeout 600 python -m pytest tests/test_gpu_onsets.py -m gpu -q -x 2>&1 | tail -3
python tools/stagger_probe.py 2>&1 | tail -1
run() {
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $name -- python tools/stagger_probe.py > $D/$name.json 2> $D/$name.err
  tail -1 $D/$name.json
  python - $name <<'PY'
import csv,sys
for r in csv.DictReader(open('gpurun_out/c12/%s_kernel_stats.csv'%sys.argv[1])):
    if 'render' in r['Name'] or 'tiles' in r['Name']:
        print(r['Name'][:100].ljust(100), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8))
PY
}
run serial SYNTHHIP_NO_OVERLAP=1
run default X=1
rm -f $D/*kernel_trace.csv
