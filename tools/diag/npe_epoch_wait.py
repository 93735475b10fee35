"""NPE.train() per-epoch time (bench.py's npe_train leg: 100 000 simulations, batch 65 536, one training + one validation
step per epoch) with the epoch event POLLED (default) and with the blocking wait (SBI_AMD_EVENT_SPIN=0), alternating on
one box.  usage: python tools/diag/npe_epoch_wait.py [epochs]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
for spin in ("1", "0", "1", "0"):
    os.environ["SBI_AMD_EVENT_SPIN"] = spin
    out = bench.npe_train_leg(dev, 0, 1, epochs)
    print(f"SBI_AMD_EVENT_SPIN={spin}: ms_per_epoch {out['ms_per_epoch']:.3f}  dense {out['dense_epochs']['ms_per_epoch']:.3f}",
          flush=True)
