#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_bank.py tests/test_gpu_split.py tests/test_gpu_pipeline.py tests/test_gpu_onsets.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head -12
python tools/stagger_short_probe.py 2>&1 | tail -5 | cut -c1-100
python tools/stagger_probe.py 2>&1 | tail -1
timeout 600 python tools/fuzz_tiles.py 9 30 2>&1 | tail -2
