python -m pytest tests/test_gpu_pcm.py -x -q 2>&1 | tail -3
for m in 0 1; do echo "== SYNTHHIP_NO_PERIOD=$m"; SYNTHHIP_NO_PERIOD=$m python tools/resample_ab.py 2>&1; done
