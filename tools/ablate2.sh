#!/bin/bash
cd $GRAFT_REPO_ROOT
for m in 0 64 4 2 1; do
  echo -n "train ablate=$m: "; SBI_AMD_ABLATE=$m python bench.py --mode train --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['roofline']['device_ms_per_step']*1000,1),'us')"
done
