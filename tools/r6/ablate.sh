#!/bin/bash
# ablation sweep of the backward kernel on the -DNSF_DEBUG library (timing only: results are invalid)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6abl}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_debug.so
for a in ${2:-0 32 4 36 1 2 64 103}; do
  rm -rf /tmp/abl_$a
  SBI_AMD_ABLATE=$a SB_NO_GRAPH=1 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_$a -- python $R/tools/diag/small_batch.py 65536 > /tmp/abl_$a.log 2>&1
  f=$(ls /tmp/abl_$a/*/*kernel_stats.csv | head -1)
  python - $f $a >> $out/abl.txt <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pick = lambda s: next((float(r["AverageNs"]) / 1e3 for r in rows if s in r["Name"]), float("nan"))
print(f"ablate {sys.argv[2]:>4}: bwd {pick('nsf_bwd_layer'):8.1f} us  fwd {pick('nsf_flow_kernel'):7.1f} us  reduce {pick('nsf_grad_reduce'):5.1f}")
PY
done
cat $out/abl.txt
