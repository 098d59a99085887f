#!/bin/bash
# BASELINE config 3 (1024 FM Sine voices) under each of the libraries named: us per block.
for L in "$@"; do
  SYNTHHIP_ALLOW_STALE=1 SYNTHHIP_LIB=$L timeout 100 python bench.py --only-config config3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); c=d.get('configs', d); r=c['config3_fm_1024v_48k_stereo']
print('$L', round(r['ms_per_1s_block']*1e3,2))"
done
