#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/resample_pk_probe.py 2>&1 | tail -1
SYNTHHIP_RESAMPLE_PK=0 python tools/resample_pk_probe.py 2>&1 | tail -1
python -m pytest tests/test_gpu_pcm.py tests/test_gpu_pcm24.py tests/test_gpu_huge.py -m gpu -q 2>&1 | tail -3
