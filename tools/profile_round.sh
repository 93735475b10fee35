#!/bin/bash
# usage (on the GPU box): bash tools/profile_round.sh <tag> [commit]
#   -> gpurun_out/<tag>/{bench.json,kernel_stats.csv,pmc.txt,traffic.json,fmpe_bench.json,{maf,zuko}_bench.json,
#      small_batch_kernel_stats_{200,8192}.csv,small_batch.txt}; copy what should be judged into profiles/
# Every profiler run is bounded (timeout): a hung profiler must not eat the box.
tag=${1:-r3}
commit=${2:-unknown}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
run() { timeout -k 5 "$@" < /dev/null; }
# 1. the driver's command (CPU baselines included) and a per-kernel trace of the same legs
run 400 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err
run 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-rccl-leg > $out/trace.log 2>&1
cp $(ls $out/trace/*/*kernel_stats.csv | head -1) $out/kernel_stats.csv
# 2. counters of the training kernels: one group per pass, no trace domains.  Counter passes COUNT, they do not time:
#    bench.py's untimed preheat steps are switched off so that launches per run = warm-up + timed steps
export SBI_AMD_BENCH_PREHEAT_MS=0
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  run 200 rocprofv3 --pmc $grp --output-format csv -d $out/pmc_$i -- python $R/bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-rccl-leg > $out/pmc_$i.log 2>&1
done
python - <<PY > $out/pmc.txt
import glob, csv, collections
print("# rocprofv3 --pmc passes (one counter group per pass, no trace domains), bench.py --mode train --steps 3 --warmup 1")
print("# sums over ALL launches of the run (the small_batch object of the train leg launches the cooperative kernels)")
for f in sorted(glob.glob("$out/pmc_*/*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); calls[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if k.startswith("void at::") or k.startswith("__amd") or "rocprim" in k: continue
        print(k, {c: round(v) for c, v in d.items()}, "launches", max(calls[(k, c)] for c in d))
PY
# 3. HBM traffic: one pair of passes per leg, so that bytes per step are attributable (3 timed + 1 warm-up step = 4)
specs=""
for mode in log_prob train fmpe; do
  for c in FETCH_SIZE WRITE_SIZE; do
    run 200 rocprofv3 --pmc $c --output-format csv -d $out/hbm_${mode}_$c -- python $R/bench.py --mode $mode --steps 3 --warmup 1 --no-cpu-baseline --skip-sampling --no-rccl-leg --no-small-batch > $out/hbm_${mode}_$c.log 2>&1
  done
  specs="$specs $mode=$out/hbm_${mode}_FETCH_SIZE,$out/hbm_${mode}_WRITE_SIZE"
done
python $R/tools/pmc_traffic.py $out/traffic.json $commit 4 $specs
unset SBI_AMD_BENCH_PREHEAT_MS
# 4. small batches: per-step times of both kernel families and per-kernel traces of the cooperative path
run 200 python $R/tools/diag/coop_crossover.py > $out/small_batch.txt 2>&1
for B in 200 8192; do
  rm -rf /tmp/sbprof_$B
  SB_NO_GRAPH=1 run 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sbprof_$B -- python $R/tools/diag/small_batch.py $B > /tmp/sbprof_$B.log 2>&1
  f=$(ls /tmp/sbprof_$B/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp $f $out/small_batch_kernel_stats_$B.csv
done
# 5. FMPE at BASELINE configs[4] (step, FMPE.train(), ODE sampling, log_prob) and the sibling flows
run 400 python $R/bench.py --mode fmpe --no-cpu-baseline > $out/fmpe_bench.json 2> $out/fmpe_bench.err
for m in maf zuko; do
  run 200 python $R/bench.py --mode $m --no-cpu-baseline > $out/${m}_bench.json 2> $out/${m}_bench.err
done
rm -rf $out/trace $out/pmc_*/ $out/hbm_*/
cut -c1-700 $out/bench.json
