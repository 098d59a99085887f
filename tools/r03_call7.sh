#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c7; mkdir -p $D
( time timeout 1200 python -m pytest tests -m gpu -q -x ) > $D/pytest.log 2>&1
tail -30 $D/pytest.log
