cd $GRAFT_REPO_ROOT
echo "== lean fwd"; SB_NO_GRAPH=1 python tools/diag/small_batch.py 200 4096 8192 12288 2>&1 | grep batch
echo "== two-tile fwd"; SBI_AMD_COOP_LEAN=0 SB_NO_GRAPH=1 python tools/diag/small_batch.py 8192 12288 2>&1 | grep batch
python tools/diag/coop_crossover.py 2>&1 | tail -12
python -m pytest tests/test_coop_gpu.py tests/test_nsf_train_gpu.py tests/test_nsf_parity_gpu.py -x -q 2>&1 | tail -3
