#!/bin/bash
# lookahead experiment: rebuild the library on the GPU box with TR_LA=3 / 4 and time the train leg
cd $GRAFT_REPO_ROOT
python bench.py --mode train --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LA=2', d['ms_per_step'])"
for la in 3 4; do
  SBI_AMD_EXTRA_HIPCC_FLAGS="-DTR_LA=$la" python -c "from sbi_amd import _build; _build.build(force=True)" > /dev/null 2>&1
  SBI_AMD_EXTRA_HIPCC_FLAGS="-DTR_LA=$la" python bench.py --mode train --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('LA=$la', d['ms_per_step'])"
done
