#!/bin/bash
# Headline block time against the number of voice groups of the 4163 shape (SYNTHHIP_VARIANT=4163 SYNTHHIP_GROUPS=g), two streams and one.
for g in 0 12 13 14 15 16 17 18 20 22; do
  for ov in 0 1; do
    r=$(SYNTHHIP_VARIANT=$([ $g = 0 ] && echo 0 || echo 4163) SYNTHHIP_GROUPS=$g SYNTHHIP_NO_OVERLAP=$ov timeout 60 python bench.py --no-pcm-rows --no-two-step --no-configs --cpu-frames 0 --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('%.2f %.2f %s' % (d['ms_per_step']*1e3, d['passes']['min_ms_per_step']*1e3, d['roofline']['kernel']))")
    echo "groups $g no_overlap $ov: $r"
  done
done
