#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c3; mkdir -p $D
( time timeout 1200 python -m pytest tests -m gpu -q ) > $D/pytest.log 2>&1
tail -15 $D/pytest.log
python bench.py --steps 20 --warmup 5 --no-pcm-rows > $D/bench.json 2> $D/bench.err
tail -3 $D/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c3/bench.json'))
print('headline', d['ms_per_step'], d['value'], 'verified', d.get('verified'))
print('configs', json.dumps(d.get('configs'), indent=1)[:3500])
PY
