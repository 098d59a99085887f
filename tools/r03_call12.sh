#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
./tools/ubench_hbm 2>&1 | grep -i "fill\|memset"
