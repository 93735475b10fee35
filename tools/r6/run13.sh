#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6n}; mkdir -p $out; cd $R
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "rc $?"; cut -c1-400 $out/bench.json
timeout 600 python -m pytest tests/test_bench_two_ranks_gpu.py tests/test_bench_wall_vs_device_gpu.py tests/test_rccl_one_rank_gpu.py -x -q -m gpu 2>&1 | tail -3
