"""`sbi_amd_allreduce_flat` (include/sbi_amd_nsf.h): the gradient all-reduce of the fused training step
through the kernel library's own RCCL entry point, on real hardware with the one rank a test box has.  With one rank a
SUM all-reduce is the identity, so the run must reproduce the group-less run bit for bit (tests/_native_allreduce_rank.py);
world size 2 is covered on the host by tests/test_distributed_cpu.py (gloo) -- no multi-GPU box exists for this repo."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.timeout(600)
def test_native_allreduce_one_rank(tmp_path):
    out = tmp_path / "native.json"
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(HERE, "_native_allreduce_rank.py"), str(out)], env=env,
                       capture_output=True, text=True, timeout=550)
    assert r.returncode == 0, f"child failed:\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    rep = json.loads(out.read_text())
    assert rep["one_rank_identity"] and all(rep["bit_identical"].values()), rep
    assert rep["allreduce_us_98025_floats"] < 200.0, rep
    from tests.parity_log import record

    record("native_allreduce_one_rank", "default_net", allreduce_us=rep["allreduce_us_98025_floats"],
           **{f"bit_identical.{k}": float(v) for k, v in rep["bit_identical"].items()})
