#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for t in 0 1; do
echo "== TILES_FOR_ALL=$t"
SYNTHHIP_TILES_FOR_ALL=$t python tools/release_probe.py 2>&1 | tail -1
SYNTHHIP_TILES_FOR_ALL=$t python tools/block0_probe.py 2>&1 | tail -3
SYNTHHIP_TILES_FOR_ALL=$t python tools/job_probe.py 2>&1 | tail -2
done
