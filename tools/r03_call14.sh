#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c14; rm -rf $D; mkdir -p $D
SH_X_TIME=$D/wg_serial.txt SYNTHHIP_NO_OVERLAP=1 python tools/stagger_probe.py 2>&1 | tail -1
SH_X_TIME=$D/wg_default.txt python tools/stagger_probe.py 2>&1 | tail -1
ls -la $D
