#!/bin/bash
# kernel sequence of one NPE.train() epoch at batch 65536 (M2 leg)
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6r}; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/nt; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/nt -- python $R/bench.py --mode npe_train --npe-epochs 40 --no-cpu-baseline > /tmp/nt.log 2>&1
f=$(ls /tmp/nt/*/*kernel_trace.csv | head -1)
python - $f <<'PY' | tee $out/epoch_trace.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "nsf_bwd_layer_kernel" in r["Kernel_Name"]]
# the 20th and 21st backward launches bracket one epoch
a, b = idx[20], idx[21]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b + 1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  {r['Kernel_Name'][:100]}")
PY
