"""The RCCL path on real hardware: a child process trains through every fused path as rank 0 of a
ONE-rank `nccl` process group on the test box's single GPU and must reproduce the group-less run bit for bit
(tests/_rccl_one_rank.py).  The CPU gloo tests (tests/test_distributed_cpu.py) cover world size 2 with the oracle
estimator; this covers what they cannot: communicator init on the device, broadcasts and all-reduces on DEVICE
buffers between the fused kernels, `global_batch` weighting and optimizer snapshots under DP."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(900)
def test_one_rank_rccl_group_reproduces_the_single_process_run_bit_for_bit(tmp_path):
    out = tmp_path / "rccl.json"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    r = subprocess.run([sys.executable, os.path.join(HERE, "_rccl_one_rank.py"), str(out)], env=env,
                       capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, f"child failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    rep = json.loads(out.read_text())
    assert rep["backend"] == "nccl" and rep["world"] == 1 and rep["probe_sum"] == 1.0
    # the fused steps, both trainers' epoch loops (loss sums) and NPE.train()'s setup all went through RCCL
    assert rep["collective_calls"]["all_reduce"] >= 4 + 3 + 6, rep["collective_calls"]
    assert rep["collective_calls"]["broadcast"] >= 3, rep["collective_calls"]
    bad = {leg: {k: v for k, v in d.items() if not v["bit_identical"]} for leg, d in rep["legs"].items()}
    bad = {k: v for k, v in bad.items() if v}
    assert not bad, f"the 1-rank RCCL run differs from the run without a process group: {json.dumps(bad)}"
    for leg, d in rep["legs"].items():
        for k, v in d.items():
            assert v.get("finite", True), f"{leg}.{k} not finite"
    from tests.parity_log import record

    record("rccl_one_rank", "all_legs", **{f"{leg}.{k}": float(v["bit_identical"]) for leg, d in rep["legs"].items()
                                           for k, v in d.items()}, all_reduce_calls=rep["collective_calls"]["all_reduce"],
           broadcast_calls=rep["collective_calls"]["broadcast"])
