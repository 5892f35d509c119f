R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/collect
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $C --kernel-trace -d $O/pmc_det_$C -o c -- python $R/tools/detector_profile.py 32 640 640 f16x3 > $O/pmc_det_$C.log 2>&1
done
cd $R; python - <<'EOF'
import sys
sys.path.insert(0,'profiles')
import summarize_round as S
S.detector_table('gpurun_out/collect','gpurun_out/collect/summary')
import json
d=json.load(open('gpurun_out/collect/summary_pmc_detector.json'))
for o in d['ops']: print(o)
EOF
