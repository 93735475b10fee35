#!/usr/bin/env python
"""Generate tests/golden/fmpe_trainer_reference.pt: the REAL early-stopping rule of sbi's vector-field trainers
(`VectorFieldTrainer._converged`, sbi/inference/trainers/vfpe/base_vf_inference.py:352-407) driven in the order of
the reference training loop (trainers/base.py:1100-1119: `_converged(epoch)` first, then the epoch's validation
loss and its EMA-smoothed summary entry, base_vf_inference.py:598-636) on synthetic validation-loss sequences.
Recorded per epoch: converged flag, fruitless-epoch counter, best validation loss.  Build container only."""

import os
import sys
import types

import torch


def main():
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import make_golden

    for mod in ["matplotlib", "matplotlib.pyplot", "matplotlib.axes", "matplotlib.figure", "joblib"]:
        try:
            __import__(mod)
        except Exception:
            make_golden.stub(mod)
    from sbi.inference.trainers.vfpe.base_vf_inference import VectorFieldTrainer

    torch.manual_seed(0)
    seqs = {
        "plateau_then_worse": torch.cat([torch.linspace(2.0, 1.0, 15), 1.0 + 0.05 * torch.randn(40),
                                         1.6 + 0.05 * torch.randn(40)]).tolist(),
        "noisy_plateau": (1.0 + 0.2 * torch.randn(120)).tolist(),
        "steady_improvement": torch.linspace(3.0, 0.5, 60).tolist(),
        "late_spike": torch.cat([torch.linspace(2.0, 1.0, 30), torch.full((25,), 5.0)]).tolist(),
    }
    out = {}
    for name, seq in seqs.items():
        for stop, decay in ((5, 0.1), (20, 0.1), (3, 0.5)):
            net = torch.nn.Linear(2, 2)
            self = types.SimpleNamespace(_neural_net=net, _val_loss=float("inf"), _best_val_loss=float("inf"),
                                         _summary={"validation_loss": []}, _epochs_since_last_improvement=0,
                                         _best_model_state_dict=None)
            trace = []
            for ep, v in enumerate(seq):
                c = VectorFieldTrainer._converged(self, ep, stop)
                trace.append((bool(c), int(self._epochs_since_last_improvement), float(self._best_val_loss)))
                if c:
                    break
                self._val_loss = v
                hist = self._summary["validation_loss"]
                hist.append(v if not hist else (1 - decay) * hist[-1] + decay * v)
            out[(name, stop, decay)] = dict(seq=seq, trace=trace)
            print(name, stop, decay, "epochs", len(trace), "converged", trace[-1][0])
    torch.save(out, os.path.join(make_golden.OUT, "fmpe_trainer_reference.pt"))


if __name__ == "__main__":
    main()
