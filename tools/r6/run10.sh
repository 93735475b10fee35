#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6k}; mkdir -p $out; cd $R
timeout 600 python -m pytest tests/test_accept_compact_gpu.py tests/test_rejection_posterior_gpu.py tests/test_npe_gpu.py -x -q -m gpu 2>&1 | tail -5 | tee $out/pytest.log
cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/st; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/st -- python $R/tools/diag/sample_timing.py > /tmp/st.log 2>&1; cat /tmp/st.log | grep -v "WARN\|amdgpu" | tee $out/sample_timing.txt
f=$(ls /tmp/st/*/*kernel_stats.csv | head -1)
python - $f <<'PY' | tee -a $out/sample_timing.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'compact' in r['Name'] or 'nsf_flow' in r['Name']: print('   ', r['Name'][:60], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
