#!/bin/bash
# Collect the rocprofv3 evidence for one round on the GPU box:  tools/profile_round.sh r01
# (run through gpurun from the repo root; afterwards, here: python tools/summarize_profiles.py gpurun_out/prof_r01 r01)
# Kernel durations and counters are separate runs: --pmc is never combined with a runtime / sys trace.
set -u
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/prof_$TAG
rm -rf "$D"; mkdir -p "$D"
python -c "from synthesizer_amd import build as B; print(B.source_hash())" > "$D/source_hash.txt"
rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o stats -- \
    python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-configs --no-runs --min-seconds 0.25 > gpurun_out/prof_${TAG}_bench.json 2> gpurun_out/prof_${TAG}_stats.err
# the same with the render launches on ONE stream: the default run overlaps consecutive render launches pairwise
# on two streams, so a kernel's own start-to-end duration there is about twice the time per launch; serialised, the
# kernel-trace average is the per-launch time and can be set against bench.py's HIP-event figure of the same run
SYNTHHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o serial -- \
    python bench.py --steps 100 --warmup 5 --cpu-frames 0 --no-pcm-rows --no-two-step --no-configs --no-runs --min-seconds 0.25 > gpurun_out/prof_${TAG}_bench_serial.json 2>> gpurun_out/prof_${TAG}_stats.err
# counters: blocks 100 .. 190 of the stream (the steady state the timed passes of a default run sit in)
SHORT="python bench.py --steps 30 --warmup 100 --cpu-frames 0 --min-seconds 0 --no-configs --no-runs"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$D" -o fetch -- $SHORT > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$D" -o write -- $SHORT > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVES SQ_WAVE_CYCLES \
    --output-format csv -d "$D" -o sq1 -- $SHORT > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
    --output-format csv -d "$D" -o sq2 -- $SHORT > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
# the other BASELINE configs that fit one GPU, each in passes of its own (bench.py --only-config: that config's steady-state loop
# and nothing else), so that its kernels' durations and counters are not averaged with the headline's launches of the same kernel
for CFG in config2 config3 staggered mixed job; do
  CMD="python bench.py --only-config $CFG"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o ${CFG}_stats -- $CMD > gpurun_out/prof_${TAG}_${CFG}.json 2>> gpurun_out/prof_${TAG}_stats.err
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$D" -o ${CFG}_fetch -- $CMD > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$D" -o ${CFG}_write -- $CMD > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVES SQ_WAVE_CYCLES \
      --output-format csv -d "$D" -o ${CFG}_sq1 -- $CMD > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES \
      --output-format csv -d "$D" -o ${CFG}_sq2 -- $CMD > /dev/null 2>> gpurun_out/prof_${TAG}_stats.err
done
rm -f "$D"/*_kernel_trace.csv "$D"/*_domain_stats.csv          # large; the stats and counter tables are what is summarised
python tools/reduce_counters.py "$D"                            # a row per (kernel, counter): the raw tables exceed what travels back
ls -la "$D"
