"""The driver's command (`python bench.py --gpus 1 --steps 20 --warmup 5`) must report wall times that ARE device
times: VERDICT r2 item 3 found `fmpe_train.ms_per_step` = 3.97 ms wall against 0.72 ms of device time (host time from
earlier legs landing inside the timed region).  Every leg of the default line is held to wall <= 1.2 x device."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(900)
def test_every_leg_of_the_default_line_is_device_bound():
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                        "--no-cpu-baseline", "--npe-epochs", "40"], env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads(next(l for l in reversed(r.stdout.splitlines()) if l.startswith("{")))
    legs = {"train": (j["ms_per_step"], j["roofline"]["whole_step_device_ms"]),
            "log_prob": (j["log_prob"]["ms_per_step"], j["log_prob"]["roofline"]["device_ms_per_step"]),
            "posterior_sample": (j["posterior_sample"]["ms_per_step"],
                                 j["posterior_sample"]["roofline"]["device_ms_per_step"]),
            "fmpe_train": (j["fmpe_train"]["ms_per_step"], j["fmpe_train"]["roofline"]["device_ms_per_step"])}
    print({k: (round(w, 4), round(d, 4)) for k, (w, d) in legs.items()}, j.get("host"), j["fmpe_train"].get("host"))
    for name, (wall, dev) in legs.items():
        assert wall <= 1.2 * dev + 0.02, f"{name}: {wall:.3f} ms wall per step against {dev:.3f} ms on the device"
    assert j["rccl_1rank"].get("rccl_ranks") == 1, j["rccl_1rank"]
    assert j["rccl_1rank"]["ms_per_step_vs_no_process_group"] < 1.25
