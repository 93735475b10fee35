#!/bin/bash
# kernel trace of one bench mode: bash tools/r6/prof_mode.sh <tag> <mode> [extra bench args]
R=$GRAFT_REPO_ROOT; tag=$1; mode=$2; shift; shift; out=$R/gpurun_out/$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pm_$mode
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pm_$mode -- python $R/bench.py --mode $mode --steps 20 --warmup 3 --no-cpu-baseline "$@" > $out/${mode}_trace.log 2>&1
cp $(ls /tmp/pm_$mode/*/*kernel_stats.csv | head -1) $out/${mode}_kernel_stats.csv
python - $out/${mode}_kernel_stats.csv <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:12]:
    print(r['Name'][:70], r['Calls'], round(float(r['AverageNs'])/1e3,1), r['Percentage'])
PY
