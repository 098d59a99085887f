#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/short_blocks_probe.py 2>&1 | tail -8
