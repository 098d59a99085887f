#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c16; rm -rf $D; mkdir -p $D
( time python bench.py ) > $D/bench.json 2> $D/bench.err
tail -3 $D/bench.err
python - <<'PY'
import json
lines=[l for l in open('gpurun_out/c16/bench.json') if l.strip()]
print('stdout lines', len(lines))
d=json.loads(lines[-1])
print({k:d[k] for k in ('value','ms_per_step','steps','warmup')})
print('verified', d.get('verified',{}).get('ok'))
print('roofline frac', d['roofline']['frac'], 'stale', d['roofline'].get('profile_stale'))
print('staggered', {k:v for k,v in d['staggered_notes'].items() if k!='note'})
print('cpu', {k:(v if not isinstance(v,dict) else {kk:vv for kk,vv in v.items() if kk!='note'}) for k,v in d['cpu_baseline'].items() if k!='sample'})
for k,v in d['configs'].items():
    print(k, {kk:vv for kk,vv in v.items() if kk not in ('note','roofline')} if isinstance(v,dict) else v)
PY
