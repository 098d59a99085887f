#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c4; mkdir -p $D
( time timeout 900 python -m pytest tests/test_gpu_dist.py tests/test_gpu_multi.py tests/test_gpu_realtime_mixer.py tests/test_gpu_pipeline.py -m gpu -q ) > $D/pytest.log 2>&1
tail -8 $D/pytest.log
python bench.py --only-config config4 > $D/cfg4.json 2> $D/cfg4.err
cat $D/cfg4.json | cut -c1-1500; tail -5 $D/cfg4.err
