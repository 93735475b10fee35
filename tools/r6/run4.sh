#!/bin/bash
# round-6 run: new tests + sampling / training bench legs
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6e}; mkdir -p $out; cd $R
run() { timeout -k 5 "$@" < /dev/null; }
run 900 python -m pytest tests/test_accept_compact_gpu.py tests/test_state_dict_interchange.py tests/test_abi_guards_gpu.py tests/test_golden_rejection.py tests/test_rejection_posterior_gpu.py tests/test_npe_gpu.py tests/test_embedding_gpu.py tests/test_mcmc_gpu.py -x -q > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -15 $out/pytest.log
run 600 python bench.py --mode sample --steps 10 --warmup 3 --no-cpu-baseline > $out/sample.json 2> $out/sample.err; cut -c1-1500 $out/sample.json
