#!/bin/bash
# The staggered_notes row (us per block, from a standing start, per 4096-frame chunk) under each of the libraries named: A/B of build variants.
for L in "$@"; do
  SYNTHHIP_ALLOW_STALE=1 SYNTHHIP_LIB=$L timeout 100 python bench.py --only-config staggered 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().splitlines()[-1]); s=d['configs']['staggered_notes'] if 'configs' in d and 'staggered_notes' in d['configs'] else d.get('staggered_notes', d)
print('$L', round(s['ms_per_step']*1e3,2), round(s['from_a_standing_start_ms_per_step']*1e3,2), round(s['realtime_chunks_4096']['ms_per_chunk']*1e3,2))"
done
