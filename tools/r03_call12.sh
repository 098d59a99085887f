#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/stagger_short_probe.py 2>&1 | tail -8
python tools/stagger_probe.py 2>&1 | tail -1
timeout 900 python -m pytest tests/test_gpu_onsets.py tests/test_gpu_pipeline.py tests/test_gpu_bank.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head -12
