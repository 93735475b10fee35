#!/bin/bash
# full GPU suite + smoke at the working tree
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6i}; mkdir -p $out; cd $R
timeout -k 5 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -8 $out/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
