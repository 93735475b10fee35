"""Helper launched by tests/test_bench_launch_cpu.py through bench.launch_ranks: every rank joins a gloo group,
all-reduces its rank + 1 and rank 0 writes what it saw."""
import json
import os
import sys

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
t = torch.tensor([float(dist.get_rank() + 1)])
dist.all_reduce(t)
if dist.get_rank() == 0:
    with open(sys.argv[1], "w") as f:
        json.dump({"world": dist.get_world_size(), "sum": t.item(), "env_world": int(os.environ["WORLD_SIZE"])}, f)
dist.destroy_process_group()
