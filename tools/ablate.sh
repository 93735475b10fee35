#!/bin/bash
# timing breakdown by phase ablation (results of ablated runs are numerically meaningless)
cd $GRAFT_REPO_ROOT
for m in 0 1 2 4 8 16 7 31; do
  echo -n "log_prob ablate=$m: "; SBI_AMD_ABLATE=$m python bench.py --mode log_prob --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step']*1000,1),'us')"
done
for m in 0 1 2 4 8 16 32 3 63; do
  echo -n "train ablate=$m: "; SBI_AMD_ABLATE=$m python bench.py --mode train --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step']*1000,1),'us')"
done
