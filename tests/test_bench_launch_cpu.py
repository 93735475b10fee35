"""bench.py --gpus N must start N ranks or fail loudly (the flag was parsed and ignored).
CPU-side checks of the launcher; the RCCL legs themselves need GPUs."""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_launch_ranks_starts_world_size_2(tmp_path):
    out = tmp_path / "probe.json"
    rc = bench.launch_ranks(2, os.path.join(ROOT, "tests", "_rank_probe.py"), [str(out)], require_gpus=False)
    assert rc == 0
    got = json.loads(out.read_text())
    assert got == {"world": 2, "sum": 3.0, "env_world": 2}


def test_more_ranks_than_gpus_is_refused():
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1 if have else 2),
                          "--steps", "1", "--warmup", "0"], capture_output=True, text=True,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert res.returncode != 0
    assert "refusing to run fewer ranks" in (res.stderr + res.stdout)
    assert '"n_gpus"' not in res.stdout          # no benchmark line was printed


def test_world_size_mismatch_under_a_launcher_is_refused():
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                         capture_output=True, text=True, env=env)
    assert res.returncode != 0
    assert "WORLD_SIZE=1" in (res.stderr + res.stdout)


def test_traffic_comes_from_a_profile_file_not_a_constant():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "PROFILED_TRAFFIC" not in src
    t = bench.load_traffic()
    if t is not None:          # once a round's profile is committed it must say where it came from
        assert "commit" in t and "bytes_per_step" in t and t["_file"].startswith("profiles/")
