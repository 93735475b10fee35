#!/bin/bash
# cooperative kernels at 8192 rows: one 16-row tile per workgroup (512 workgroups, two per CU) against two tiles (256)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6l}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_debug.so
for nt in 0 1 2 0 1; do
  rm -rf /tmp/nt_$nt
  SBI_AMD_COOP_NT=$nt SB_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nt_$nt -- python $R/tools/diag/small_batch.py 8192 6144 > /tmp/nt_$nt.log 2>&1
  echo "== COOP_NT=$nt: $(grep batch /tmp/nt_$nt.log | awk '{print $2 $NF}' | tr '\n' ' ')" | tee -a $out/nt.txt
  f=$(ls /tmp/nt_$nt/*/*kernel_stats.csv | head -1)
  python - $f <<'PY' | tee -a $out/nt.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'coop' in r['Name']: print('   ', r['Name'][:44], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
done
