#!/bin/bash
# spline rewrite A/B: HEAD library (libsbi_amd_nsf_old.so) against the working tree -- timing twice, interleaved; accuracy; targeted tests
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6h}; mkdir -p $out; cd $R
for v in old default old default; do
  if [ $v = default ]; then unset SBI_AMD_LIB; else export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_$v.so; fi
  echo "== $v" | tee -a $out/ab.txt; timeout 300 python tools/diag/fwd_ab.py 2>&1 | grep -v "WARNING\|amdgpu.ids" | tail -3 | tee -a $out/ab.txt
done
for v in old default; do
  if [ $v = default ]; then unset SBI_AMD_LIB; else export SBI_AMD_LIB=$R/sbi_amd/libsbi_amd_nsf_$v.so; fi
  echo "== $v" >> $out/gates.txt
  timeout 300 python tools/diag/measure_gates.py 2>&1 | grep -v "WARNING\|amdgpu.ids\|UserWarning\|Consider\|print(" >> $out/gates.txt
done
cat $out/gates.txt
unset SBI_AMD_LIB
timeout -k 5 900 python -m pytest tests/test_rq_spline_abi_gpu.py tests/test_spline_adversarial_gpu.py tests/test_nsf_parity_gpu.py tests/test_nsf_train_gpu.py tests/test_maf_gpu.py tests/test_zuko_gpu.py tests/test_coop_gpu.py tests/test_parity_full_size_gpu.py tests/test_trained_parity_gpu.py tests/test_wide_gpu.py tests/test_broadcast_x_gpu.py tests/test_mcmc_gpu.py -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -12 $out/pytest.log
