#!/bin/bash
# A/B of library variants: fused step at 65536 / 8192 / 200 rows + reduce kernel time
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6i}; mkdir -p $out; shift; cd /tmp; export TMPDIR=/tmp
for v in default "$@" default "$@"; do
  if [ $v = default ]; then unset SBI_AMD_LIB; else export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_$v.so; fi
  rm -rf /tmp/tr_$v
  SB_NO_GRAPH=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr_$v -- python $R/tools/diag/small_batch.py 65536 8192 200 > /tmp/tr_$v.log 2>&1
  f=$(ls /tmp/tr_$v/*/*kernel_stats.csv | head -1)
  echo "== $v: $(grep batch /tmp/tr_$v.log | awk '{print $2 $NF}' | tr '\n' ' ')" | tee -a $out/ab.txt
  python - $f <<'PY' | tee -a $out/ab.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'reduce' in r['Name']: print('   ', r['Name'][:30], r['Calls'], round(float(r['AverageNs'])/1e3,1), 'us')
PY
done
