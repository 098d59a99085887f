#!/bin/bash
# round 3, call 1: baseline tests + where config 2's block goes
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/c1
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/c1/pytest.log 2>&1
python tools/config2_probe.py default > gpurun_out/c1/probe_default.json 2> gpurun_out/c1/probe_default.err
SYNTHHIP_DEBUG=1 python tools/config2_probe.py noprepare > gpurun_out/c1/probe_noprepare.json 2>/dev/null
SYNTHHIP_NO_SPECULATION=1 python tools/config2_probe.py nospec > gpurun_out/c1/probe_nospec.json 2>/dev/null
for v in 421 441 444 844 826 1621; do
  SYNTHHIP_VARIANT=$v python tools/config2_probe.py var$v > gpurun_out/c1/probe_var$v.json 2>/dev/null
done
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1/prof -o c2 -- python tools/config2_probe.py prof > gpurun_out/c1/probe_prof.json 2> gpurun_out/c1/prof.err
rm -f gpurun_out/c1/prof/*/*kernel_trace.csv gpurun_out/c1/prof/*kernel_trace.csv
tail -3 gpurun_out/c1/pytest.log
cat gpurun_out/c1/probe_*.json
