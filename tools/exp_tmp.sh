cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
B="python bench.py --no-pcm-rows --no-two-step --cpu-frames 0 --steps 100 --warmup 5 --min-seconds 0.3"
run() { name=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); c=j.get('configs',{}); print('$name: median ms/step %.4f min %.4f value %.0f | int16 %.4f cfg2 %.4f cfg3 %.4f' % (j['ms_per_step'], j['passes']['min_ms_per_step'], j['value'], j['int16_stream']['ms_per_step'], c['config2_additive_64v_adsr_48k_stereo']['ms_per_1s_block'], c['config3_fm_1024v_48k_stereo']['ms_per_1s_block']))"; }
run base SYNTHHIP_LIB=tools/ab/libsynthhip_base.so
run new A=1
run new_nosplit SYNTHHIP_NO_SPLIT=1
run new_444split SYNTHHIP_VARIANT=444
run base SYNTHHIP_LIB=tools/ab/libsynthhip_base.so
run new A=1
python bench.py --no-pcm-rows --no-two-step --cpu-frames 0 --no-configs --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('K=20: ms/step %.4f value %.0f frac %.3f' % (j['ms_per_step'], j['value'], j['roofline']['frac']))"
