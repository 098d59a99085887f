#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for v in default tpw9 tpw12 gs1 tpw12gs1 default; do
  if [ $v = default ]; then L=synthesizer_amd/libsynthhip.so; else L=tools/ab/$v.so; fi
  echo "== $v"; SYNTHHIP_LIB=$L python tools/stagger_probe.py 2>&1 | tail -1 | cut -c1-160
done
