#!/bin/bash
# round-6 iteration run: parity of the training path, fused-step timing, kernel trace, backward timeline
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6a}; mkdir -p $out
cd $R
run() { timeout -k 5 "$@" < /dev/null; }
run 900 python -m pytest tests/test_nsf_train_gpu.py tests/test_parity_full_size_gpu.py tests/test_step_tail_gpu.py -x -q -m gpu > $out/pytest.log 2>&1; echo "pytest rc $?" >> $out/pytest.log
tail -5 $out/pytest.log
SB_NO_GRAPH=1 run 200 python tools/diag/small_batch.py 65536 8192 > $out/small_batch.txt 2>&1; cat $out/small_batch.txt
cd /tmp && export TMPDIR=/tmp
SB_NO_GRAPH=1 run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/tools/diag/small_batch.py 65536 > $out/trace.log 2>&1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv; rm -rf $out/trace
head -12 $out/kernel_stats.csv | cut -c1-200
cd $R
run 300 python tools/timeline.py > $out/timeline.txt 2>&1; head -70 $out/timeline.txt
