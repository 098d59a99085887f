#!/bin/bash
# A/B of the materialisation kernel under variant libraries: tools/gen_ab.sh LIB [LIB ...]   (GPU box, repo root)
for LIB in "$@"; do
  for rep in 1 2; do
    SYNTHHIP_ALLOW_STALE=1 SYNTHHIP_LIB=$LIB python bench.py --no-configs --no-pcm-rows --cpu-frames 0 --min-seconds 0.3 --no-runs --detail /tmp/gen_ab_detail.json > /dev/null 2>/tmp/gen_ab.err || tail -3 /tmp/gen_ab.err
    python - "$LIB" <<'P'
import json, sys
d = json.load(open("/tmp/gen_ab_detail.json"))
t = d["two_step"]; g = t["roofline_generate"]; i = d["two_step_i16"]
print("%-50s generate f32 %.4f ms (min %.4f) %.3f of HBM | i16 rows %.4f ms | fused mixdown %.4f ms | mix %.4f ms | headline %.2f us" % (
    sys.argv[1].split("/")[-1], g["avg_launch_ms"], g["min_ms"], g["frac"], i["roofline_generate"]["avg_launch_ms"], i["fused_mono_mixdown"]["ms"],
    t["roofline_mix"]["avg_launch_ms"], d["ms_per_step"] * 1e3), flush=True)
P
  done
done
