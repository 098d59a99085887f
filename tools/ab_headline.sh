#!/bin/bash
# The headline (us per block: median pass, best pass) under each of the libraries named, two streams and one: A/B of build variants on one box.
for L in "$@"; do
  for ov in 0 1; do
    r=$(SYNTHHIP_ALLOW_STALE=1 SYNTHHIP_LIB=$L SYNTHHIP_NO_OVERLAP=$ov timeout 60 python bench.py --no-pcm-rows --no-two-step --no-configs --cpu-frames 0 --min-seconds 0.6 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().splitlines()[-1]); print('%.2f %.2f' % (d['ms_per_step']*1e3, d['passes']['min_ms_per_step']*1e3))")
    echo "$L no_overlap $ov: $r"
  done
done
