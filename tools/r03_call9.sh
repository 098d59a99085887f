#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
D=gpurun_out/c9; rm -rf $D; mkdir -p $D
SYNTHHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d $D -o sq1 -- python tools/stagger_probe.py > /dev/null 2> $D/err1
SYNTHHIP_NO_OVERLAP=1 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_FLAT --output-format csv -d $D -o sq2 -- python tools/stagger_probe.py > /dev/null 2> $D/err2
python - <<'PY'
import csv, collections, re
for f in ('sq1','sq2'):
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open('gpurun_out/c9/%s_counter_collection.csv'%f)):
        m=re.search(r'(k_[a-z_]+(<[^>]*>)?)', r['Kernel_Name'])
        k=m.group(1) if m else r['Kernel_Name'][:30]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in agg:
        if 'k_bank_render' in k or 'prepare_tiles' in k:
            print(k, {c: round(sum(v)/len(v)) for c,v in agg[k].items()}, 'n=',len(list(agg[k].values())[0]))
PY
rm -f $D/*kernel_trace.csv $D/*counter_collection.csv
