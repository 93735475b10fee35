#!/bin/bash
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${1:-r6m}; mkdir -p $out; cd $R
timeout 300 python tools/diag/clock_ramp.py 2>&1 | grep -v "WARN\|amdgpu" | tee $out/ramp.txt
rocm-smi --showclocks 2>/dev/null | head -20 | tee -a $out/ramp.txt
