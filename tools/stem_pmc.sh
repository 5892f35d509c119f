#!/bin/bash
# PMC look at rf_stem_kernel (tools only)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/stem_pmc
mkdir -p $O
unset TA_PROFILE_OPS
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  T=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --kernel-trace -d $O/$T -o c -- python $R/tools/detector_profile.py > $O/$T.log 2>&1
  python - <<PY
import csv, glob, collections
for f in glob.glob('$O/$T/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        if 'rf_stem' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
    for k, v in acc.items():
        print(k, len(v), sum(v) / len(v))
PY
done
