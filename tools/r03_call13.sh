#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c13; rm -rf $D; mkdir -p $D
run() {
  name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $D -o $name -- python tools/stagger_probe.py > $D/$name.json 2> $D/$name.err
  echo "== $name"; tail -1 $D/$name.json | cut -c1-200
  python - $name <<'PY'
import csv,sys
for r in csv.DictReader(open('gpurun_out/c13/%s_kernel_stats.csv'%sys.argv[1])):
    if 'render' in r['Name'] or 'tiles' in r['Name']:
        print(r['Name'][:90].ljust(90), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MinNs'])/1e3)).rjust(8), ('%.1f'%(float(r['MaxNs'])/1e3)).rjust(8))
PY
}
for v in tpw2 tpw12 nostore; do
run ${v}_nospec SYNTHHIP_LIB=tools/ab/$v.so SYNTHHIP_NO_SPECULATION=1 SYNTHHIP_NO_OVERLAP=1
done
for v in tpw2 tpw12; do
echo "== $v default"; SYNTHHIP_LIB=tools/ab/$v.so python tools/stagger_probe.py 2>&1 | tail -1 | cut -c1-200
done
rm -f $D/*kernel_trace.csv
