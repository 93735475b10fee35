#!/bin/bash
# full GPU suite + the round's profile set
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6full}; mkdir -p $out; cd $R
timeout -k 5 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; tail -2 $out/smoke.log
bash tools/profile_round.sh ${1:-r6full} ${2:-unknown} > $out/profile_round.log 2>&1; tail -3 $out/profile_round.log | cut -c1-600
