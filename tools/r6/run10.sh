#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6k}; mkdir -p $out; cd $R
timeout 900 python -m pytest tests/test_accept_compact_gpu.py tests/test_rejection_posterior_gpu.py tests/test_npe_gpu.py tests/test_density_estimator_contract_gpu.py tests/test_npe_multiround_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee $out/pytest.log
timeout 300 python bench.py --mode sample --no-cpu-baseline 2>/dev/null | tail -1 > $out/sample_bench.json
python - $out/sample_bench.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); p=d.get("posterior_sample", d)
print("10^6 draws", p["ms_per_step"], "box1", p["acceptance_below_one"]["box_uniform_1"]["ms_per_step"], "box0.5", p["acceptance_below_one"]["box_uniform_0.5"]["ms_per_step"])
PY
