#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_onsets.py tests/test_gpu_pipeline.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head -12
python tools/stagger_probe.py 2>&1 | tail -1
python tools/stagger_probe.py 2>&1 | tail -1
timeout 800 python tools/fuzz_tiles.py 3 20 2>&1 | tail -4
