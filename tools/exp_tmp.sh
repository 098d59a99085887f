cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_bank.py tests/test_gpu_split.py tests/test_gpu_dist.py tests/test_gpu_realtime_mixer.py -x -q 2>&1 | tail -3
B="python bench.py --no-pcm-rows --no-two-step --cpu-frames 0 --no-configs --steps 100 --warmup 5 --min-seconds 0.3"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('$name: median ms/step %.4f min %.4f value %.0f' % (j['ms_per_step'], j['passes']['min_ms_per_step'], j['value']))"; }
run base SYNTHHIP_LIB=tools/ab/libsynthhip_base.so
run new A=1
run new_always_general SYNTHHIP_ALWAYS_GENERAL=1
run new_serial SYNTHHIP_NO_OVERLAP=1
run base_serial SYNTHHIP_LIB=tools/ab/libsynthhip_base.so SYNTHHIP_NO_OVERLAP=1
run new A=1
