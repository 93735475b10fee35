#!/bin/bash
# instruction-cache behaviour of the training kernels (separate PMC pass, no trace domains)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/icache
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/icache/$tag -- python $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/icache/$tag.log 2>&1
done
python - <<PY
import glob, csv, collections
for f in sorted(glob.glob("$R/gpurun_out/icache/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
