#!/bin/bash
# clip + Adam inside the reduction kernels (fence-free rendezvous): targeted tests three times, then step times with / without
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6g}; mkdir -p $out; cd $R
for i in 1 2 3; do
timeout -k 5 600 python -m pytest tests/test_step_tail_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $out/pytest.log
done
timeout -k 5 600 python -m pytest tests/test_nsf_train_gpu.py tests/test_npe_gpu.py tests/test_abi_guards_gpu.py tests/test_reference_trainer_replay_gpu.py tests/test_rccl_one_rank_gpu.py -x -q -m gpu 2>&1 | tail -2 | tee -a $out/pytest.log
for f in 1 0 1 0; do
  echo "SBI_AMD_FUSED_UPDATE=$f $(SBI_AMD_FUSED_UPDATE=$f SB_NO_GRAPH=1 timeout -k 5 300 python tools/diag/small_batch.py 200 8192 65536 2>&1 | grep batch | awk '{print $2 $NF}' | tr '\n' ' ')" | tee -a $out/small_batch.txt
done
