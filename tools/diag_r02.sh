#!/bin/bash
# Round-2 diagnostics on the GPU box: GPU tests, smoke, and where the headline kernel's time goes (debug knobs).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/diag1; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
B="python bench.py --no-pcm-rows --no-two-step --cpu-frames 0"
run() { name=$1; shift; env "$@" $B --steps 100 --warmup ${WARM:-5} 2>$O/$name.err | tail -1 > $O/$name.json
  python - "$O/$name.json" "$name" <<'PY'
import json,sys
j=json.load(open(sys.argv[1])); r=j["roofline"]
print("%-28s value %.0f Ms/s  ms/step %.4f  ev %.4f  int16 %.4f" % (sys.argv[2], j["value"], j["ms_per_step"], r["avg_launch_ms"], j.get("int16_stream",{}).get("ms_per_step",0)))
PY
}
run base A=0
WARM=205 run late A=0
WARM=205 run late_noprep SYNTHHIP_DEBUG=1
WARM=205 run late_nofold SYNTHHIP_DEBUG=2
WARM=205 run late_neither SYNTHHIP_DEBUG=3
WARM=205 run late_serial SYNTHHIP_NO_OVERLAP=1
WARM=205 run late_serial_neither SYNTHHIP_NO_OVERLAP=1 SYNTHHIP_DEBUG=3
WARM=205 run late_g16 SYNTHHIP_GROUPS=16
WARM=205 run late_g4 SYNTHHIP_GROUPS=4
WARM=205 run late_844 SYNTHHIP_VARIANT=844
[ -x tools/ubench_fastloop.bin ] && timeout 120 tools/ubench_fastloop.bin > $O/ubench_fastloop.txt 2>&1; grep -i "production\|FPL8 4w min4 groups 16" $O/ubench_fastloop.txt
