#!/bin/bash
# A/B of library variants on log_prob / sample / train step
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6g}; mkdir -p $out; shift; cd $R
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset SBI_AMD_LIB; else export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_$v.so; fi
  lp=$(timeout 200 python bench.py --mode log_prob --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step']*1000,1))")
  lpb=$(timeout 200 python bench.py --mode log_prob_broadcast --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step']*1000,1))")
  sm=$(timeout 200 python tools/diag/sample_timing.py 2>/dev/null | head -1 | awk '{print $2}')
  tr=$(SB_NO_GRAPH=1 timeout 200 python tools/diag/small_batch.py 65536 2>&1 | grep batch | awk '{print $NF}')
  echo "$v: log_prob $lp us, one-x_o $lpb us, 1e6 draws $sm ms, train step $tr ms" | tee -a $out/ab.txt
done; done
