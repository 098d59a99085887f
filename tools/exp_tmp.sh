#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_split.py tests/test_gpu_bank.py -q -x 2>&1 | tail -5
python tools/block0_probe.py
SYNTHHIP_NO_SEG=1 python tools/block0_probe.py
D=gpurun_out/b0s; rm -rf $D; mkdir -p $D
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o seg -- python tools/block0_counters.py env > /dev/null 2>$D/err.txt
rm -f $D/*kernel_trace.csv $D/*domain_stats.csv
