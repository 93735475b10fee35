#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6p}; mkdir -p $out; cd $R
timeout 1200 python -m pytest tests/test_shuffle_prefetch_gpu.py tests/test_shuffle_gpu.py tests/test_npe_gpu.py tests/test_reference_trainer_replay_gpu.py tests/test_npe_multiround_gpu.py tests/test_dp_two_rank_gpu.py tests/test_bench_two_ranks_gpu.py tests/test_coop_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.log
for i in 1 2; do timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline --no-rccl-leg --no-small-batch 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("train", d["value"], d["ms_per_step"], d["roofline"]["device_ms_per_step"])' | tee -a $out/bench.txt; done
timeout 400 python bench.py --mode npe_train --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-600 | tee -a $out/bench.txt
