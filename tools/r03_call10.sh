#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python tools/gen_rows_probe.py 2>&1 | tail -1
for p in 1 2 4; do SYNTHHIP_GEN_ROWS=$p python tools/gen_rows_probe.py 2>&1 | tail -1; done
