#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/stagger_probe.py 2>&1 | tail -1
python tools/stagger_probe.py 2>&1 | tail -1
