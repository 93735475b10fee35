#!/bin/bash
R=$GRAFT_REPO_ROOT; cd $R; timeout 600 python tools/diag/compact_stress.py 400 2>&1 | tail -3
