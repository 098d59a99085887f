#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
( time timeout 900 python tools/fuzz_tiles.py 1 30 ) 2>&1 | tail -8
( time timeout 900 python tools/fuzz_tiles.py 2 30 ) 2>&1 | tail -8
