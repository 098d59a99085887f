cd /tmp && export TMPDIR=/tmp; cd "${GRAFT_REPO_ROOT:-/root/repo}"
for L in synthesizer_amd/libsynthhip.so synthesizer_amd/build/libsynthhip_nocheck.so; do
  tag=$(basename $L .so)
  for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_ANY SQ_IFETCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY"; do
    SYNTHHIP_ALLOW_STALE=1 SYNTHHIP_LIB=$L rocprofv3 --kernel-trace --pmc $C --output-format csv -d gpurun_out/pmc_guard -o ${tag}_$(echo $C | cut -c1-12 | tr ' ' _) -- python tools/gen_i16_once.py > /dev/null 2>> gpurun_out/pmc_guard.err
  done
done
python - <<'P'
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/pmc_guard/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "lean_harm" not in k: continue
        acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f.split("/")[-1])
    for k, d in acc.items():
        print("  ", k, {c: "%.4g" % (sum(v) / len(v)) for c, v in d.items()})
P
