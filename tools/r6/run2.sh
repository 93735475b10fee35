#!/bin/bash
# A/B of kernel-library variants: parity subset on the default library, fused-step time per variant, timeline
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6b}; mkdir -p $out; shift
cd $R
run() { timeout -k 5 "$@" < /dev/null; }
run 900 python -m pytest tests/test_nsf_train_gpu.py tests/test_parity_full_size_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -3 $out/pytest.log
for rep in 1 2; do
for v in default "$@"; do
  if [ $v = default ]; then unset SBI_AMD_LIB; else export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_$v.so; fi
  echo "== $v" >> $out/ab.txt
  SB_NO_GRAPH=1 run 200 python tools/diag/small_batch.py 65536 2>&1 | grep batch >> $out/ab.txt
done; done
cat $out/ab.txt
unset SBI_AMD_LIB
run 300 python tools/timeline.py > $out/timeline.txt 2>&1; head -62 $out/timeline.txt
