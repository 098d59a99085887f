#!/bin/bash
# round 3, call 2: the refactored library -- full GPU tests, then config 2 under the new schedule and its knobs
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c2; mkdir -p $D
( time timeout 900 python -m pytest tests -m gpu -q ) > $D/pytest.log 2>&1
tail -15 $D/pytest.log
python tools/config2_probe.py default > $D/probe_default.json 2> $D/probe_default.err
SYNTHHIP_PREPARE_IN_TILE=1 python tools/config2_probe.py in_tile > $D/probe_in_tile.json 2>/dev/null
SYNTHHIP_NO_SMALL_PIPELINE=1 python tools/config2_probe.py no_small > $D/probe_no_small.json 2>/dev/null
SYNTHHIP_NO_SMALL_PIPELINE=1 SYNTHHIP_PREPARE_IN_TILE=1 python tools/config2_probe.py round2_schedule > $D/probe_r2.json 2>/dev/null
for v in 421 821 1621 1611 441; do
  SYNTHHIP_VARIANT=$v python tools/config2_probe.py var$v > $D/probe_var$v.json 2>/dev/null
  SYNTHHIP_NO_SMALL_PIPELINE=1 SYNTHHIP_VARIANT=$v python tools/config2_probe.py var${v}_serial > $D/probe_var${v}_serial.json 2>/dev/null
done
cat $D/probe_*.json
python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-pcm-rows > $D/bench.json 2> $D/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c2/bench.json'))
print('headline', d['ms_per_step'], d['value'], 'int16', d.get('int16_stream',{}).get('ms_per_step'), 'job0', d.get('job_from_frame_0',{}).get('ms'))
print('two_step', {k:(v if not isinstance(v,dict) else v.get('avg_launch_ms', v.get('generate_ms'))) for k,v in d.get('two_step',{}).items()})
print('configs', {k:v.get('ms_per_1s_block', v.get('ms')) for k,v in d.get('configs',{}).items()})
PY
