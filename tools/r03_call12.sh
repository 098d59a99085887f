#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for g in 8 16 64; do echo "== groups $g"; SYNTHHIP_GROUPS=$g python tools/stagger_short_probe.py 2>&1 | tail -5 | cut -c1-70; done
