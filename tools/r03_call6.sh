#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/ring_probe.py 2>&1 | tail -3
