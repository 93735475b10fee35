#!/bin/bash
# usage: tools_pmc.sh <outdir-name> <bench args...>   -- run separate PMC passes (no trace domains mixed in)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/$name/$tag -- python $R/bench.py "$@" --no-cpu-baseline > $R/gpurun_out/$name/$tag.log 2>&1
done
python - <<PY
import glob, csv, collections
for f in sorted(glob.glob("$R/gpurun_out/$name/*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in agg.items():
        print(k, {c: round(v) for c, v in d.items()})
PY
