#!/bin/bash
# BASELINE config 3 under the shapes named (SYNTHHIP_VARIANT; 0 = the rule's choice): us per block, same box.
for V in "$@"; do
  SYNTHHIP_VARIANT=$V timeout 100 python bench.py --only-config config3 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); c=d.get('configs', d); r=c['config3_fm_1024v_48k_stereo']
print('variant $V', round(r['ms_per_1s_block']*1e3,2))"
done
