#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_onsets.py -m gpu -q -x 2>&1 | grep -E "^E  |passed|failed" | cut -c1-200 | head -12
timeout 800 python tools/fuzz_tiles.py 1 25 2>&1 | tail -8
timeout 800 python tools/fuzz_tiles.py 2 25 2>&1 | tail -8
