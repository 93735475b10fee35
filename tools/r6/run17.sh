#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6s}; mkdir -p $out; cd $R
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; echo "rc $? lines $(wc -l < $out/bench.json)"
python -c "
import json; d=json.load(open('$out/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['posterior_sample']['ms_per_step'], d['posterior_sample']['acceptance_below_one']['box_uniform_0.5']['ms_per_step'], d['log_prob']['roofline']['frac'], d['npe_train']['ms_per_epoch'])"
