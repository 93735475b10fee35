#!/bin/bash
# usage (GPU box): bash tools/diag/pmc_bwd.sh <tag>  -> gpurun_out/<tag>_pmc.txt : wait / issue counters of the training kernels
tag=${1:-pmc}
R=$GRAFT_REPO_ROOT
out=/tmp/pmc_$tag
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  timeout -k 5 300 rocprofv3 --pmc $grp --output-format csv -d ${out}_$i -- python $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-rccl-leg > ${out}_$i.log 2>&1 < /dev/null
done
python - <<PY > $R/gpurun_out/${tag}_pmc.txt
import glob, csv, collections
for f in sorted(glob.glob("${out}_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if k.startswith("void at::") or k.startswith("__amd"): continue
        print(k, {c: round(v) for c, v in d.items()}, "launches", max(calls[(k, c)] for c in d))
PY
cat $R/gpurun_out/${tag}_pmc.txt | cut -c1-900
