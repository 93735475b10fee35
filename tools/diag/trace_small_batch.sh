#!/bin/bash
# usage (GPU box): bash tools/diag/trace_small_batch.sh <tag> [batch ...]  -> gpurun_out/<tag>_kernel_stats_<batch>.csv
# per-kernel durations of the small-batch training loop (tools/diag/small_batch.py) from rocprofv3 --kernel-trace
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for B in "$@"; do
  rm -rf /tmp/sbprof_$B
  SB_NO_GRAPH=1 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sbprof_$B -- python $R/tools/diag/small_batch.py $B > /tmp/sbprof_$B.log 2>&1 < /dev/null
  f=$(ls /tmp/sbprof_$B/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $R/gpurun_out/${tag}_kernel_stats_$B.csv
  grep "^batch" /tmp/sbprof_$B.log
done
