#!/bin/bash
# What clock and power does the chip hold under the headline's stream of launches?  (round 4: the stream is not bound by its instruction
# count -- shapes with 7-10 % fewer instructions run no faster.)  A sampler (rocm-smi every 0.2 s) beside bench.py --min-seconds 6.
O=gpurun_out/r04_clock
mkdir -p $O
( for i in $(seq 1 60); do date +%s.%N; rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "sclk|Power|GPU use|Socket"; sleep 0.2; done ) > $O/smi_log.txt 2>&1 &
SMI=$!
sleep 1
python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-configs --no-pcm-rows --no-two-step --min-seconds 6 > $O/bench.json 2> $O/bench.err
sleep 1
kill $SMI 2>/dev/null
wait $SMI 2>/dev/null
python - <<'P'
import json,re
d=json.load(open("gpurun_out/r04_clock/bench.json")); print("bench", d["ms_per_step"], d["passes"]["count"])
txt=open("gpurun_out/r04_clock/smi_log.txt").read()
sclk=[int(x) for x in re.findall(r"sclk clock level: \d+: \((\d+)Mhz\)", txt)]
pw=[float(x) for x in re.findall(r"Power \(W\): ([0-9.]+)", txt)]
use=[int(x) for x in re.findall(r"GPU use \(%\): (\d+)", txt)]
print("sclk MHz samples", sclk)
print("power W samples", pw)
print("use % samples", use)
P
tail -20 $O/smi_log.txt
