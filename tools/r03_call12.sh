#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c12; rm -rf $D; mkdir -p $D
timeout 900 python -m pytest tests/test_gpu_onsets.py tests/test_gpu_pipeline.py tests/test_gpu_split.py tests/test_gpu_bank.py -m gpu -q -x 2>&1 | tail -3
python tools/stagger_probe.py 2>&1 | tail -1
python tools/stagger_probe.py 2>&1 | tail -1
python tools/config2_probe.py 2>&1 | tail -3
