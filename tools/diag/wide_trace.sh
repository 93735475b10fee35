#!/bin/bash
# usage (GPU box): bash tools/diag/wide_trace.sh  -> per-kernel durations of the hidden-100 training step at batch 200 / 8192
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
cat > /tmp/wide_step.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from sbi_amd.inference.trainers.fused import FusedTrainStep
from sbi_amd.neural_nets.net_builders.flow import build_nsf
B = int(sys.argv[1])
torch.manual_seed(0)
theta = torch.randn(20000, 10); x = theta + 0.3 * torch.randn(20000, 10)
est = build_nsf(theta, x, hidden_features=100).cuda()
st = FusedTrainStep(est)
tb, xb = theta[:B].cuda().contiguous(), x[:B].cuda().contiguous()
for _ in range(60): st.step(tb, xb)
torch.cuda.synchronize()
PY
for B in 200 8192; do
  rm -rf /tmp/wtr_$B
  timeout -k 5 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/wtr_$B -- python /tmp/wide_step.py $B > /tmp/wtr_$B.log 2>&1 < /dev/null
  python - $B <<'PY'
import csv, glob, sys
f = glob.glob(f"/tmp/wtr_{sys.argv[1]}/*/*kernel_stats.csv")
print("batch", sys.argv[1])
for i, r in enumerate(csv.DictReader(open(f[0]))):
    if i < 8: print("   ", r["Name"][:50], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
done
