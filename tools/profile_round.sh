#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [commit]
#   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc.txt,traffic.json,{maf,zuko}_bench.json,
#      {maf,generic}_kernel_stats.csv}; copy what should be judged into profiles/
tag=${1:-r2}
commit=${2:-unknown}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $out/bench.json 2> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline > $out/trace.log 2>&1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"; do
  t=$(echo $grp | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$t -- python $R/bench.py --mode both --steps 3 --warmup 1 --no-cpu-baseline --npe-epochs 4 > $out/pmc_$t.log 2>&1
done
# HBM traffic: one pair of passes per leg, so that bytes per step are attributable (3 timed + 1 warm-up step = 4)
specs=""
for mode in log_prob train fmpe; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $out/hbm_${mode}_$c -- python $R/bench.py --mode $mode --steps 3 --warmup 1 --no-cpu-baseline --skip-sampling > $out/hbm_${mode}_$c.log 2>&1
  done
  specs="$specs $mode=$out/hbm_${mode}_FETCH_SIZE,$out/hbm_${mode}_WRITE_SIZE"
done
python $R/tools/pmc_traffic.py $out/traffic.json $commit 4 $specs
python - <<PY > $out/pmc.txt
import glob, csv, collections
print("# rocprofv3 --pmc passes (one counter group per pass, no trace domains), bench.py --mode both --steps 3 --warmup 1 --npe-epochs 4")
print("# sums over ALL launches of the run")
for f in sorted(glob.glob("$out/pmc_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:44]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if k.startswith("void at::") or k.startswith("__amd"): continue
        print(k, {c: round(v) for c, v in d.items()}, "launches", max(calls[(k, c)] for c in d))
PY
# sibling flows and the generic training pass: bench lines + kernel traces (bounded: a hung profiler must not eat the box)
for m in maf zuko; do
  timeout -k 5 200 python $R/bench.py --mode $m --no-cpu-baseline > $out/${m}_bench.json 2> $out/${m}_bench.err < /dev/null
done
timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_maf -- python $R/bench.py --mode maf --steps 20 --warmup 5 --no-cpu-baseline > $out/trace_maf.log 2>&1 < /dev/null
f=$(ls $out/trace_maf/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/maf_kernel_stats.csv
SBI_AMD_ABLATE=2048 timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_gen -- python $R/bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $out/trace_gen.log 2>&1 < /dev/null
f=$(ls $out/trace_gen/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/generic_kernel_stats.csv
rm -rf $out/trace $out/trace_maf $out/trace_gen $out/pmc_*/ $out/hbm_*/
cat $out/bench.json | cut -c1-600
