"""Parity artifact: the GPU parity tests record their measured distances here; the file is merged back from the
GPU box (gpurun_out/parity.json) and committed per round as profiles/parity_rN.json."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.environ.get("SBI_AMD_PARITY_OUT", os.path.join(ROOT, "gpurun_out", "parity.json"))


def record(test: str, config: str, **numbers) -> None:
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        data.setdefault(test, {})[config] = {k: (float(v) if isinstance(v, (int, float)) else v)
                                             for k, v in numbers.items()}
        with open(PATH, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass   # a read-only tree must not fail a parity test
