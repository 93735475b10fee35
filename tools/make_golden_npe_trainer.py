#!/usr/bin/env python
"""Generate tests/golden/npe_trainer_reference.pt: sbi's real early-stopping / best-weights rule
(`NeuralInference._converged`, sbi/inference/trainers/base.py:1254-1284) driven in the reference loop order
(base.py:1100-1119) on synthetic validation-loss sequences.  Recorded per epoch: converged flag, fruitless-epoch
counter, best validation loss and a checksum of the weights the rule holds as best.  Build container only."""

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
    from sbi.inference.trainers.base import NeuralInference

    torch.manual_seed(1)
    seqs = {
        "improve_then_flat": torch.cat([torch.linspace(5.0, 1.0, 12), 1.0 + 0.1 * torch.rand(40)]).tolist(),
        "noisy": (2.0 + torch.randn(80)).tolist(),
        "monotone": torch.linspace(4.0, 1.0, 30).tolist(),
    }
    out = {}
    for name, seq in seqs.items():
        for stop in (1, 5, 20):
            net = torch.nn.Linear(1, 1, bias=False)
            with torch.no_grad():
                net.weight.fill_(-1.0)
            self = types.SimpleNamespace(_neural_net=net, _val_loss=float("inf"), _best_val_loss=float("inf"),
                                         _epochs_since_last_improvement=0, _best_model_state_dict=None)
            trace = []
            for ep, v in enumerate(seq):
                c = NeuralInference._converged(self, ep, stop)
                trace.append((bool(c), int(self._epochs_since_last_improvement), float(self._best_val_loss),
                              float(self._best_model_state_dict["weight"].item()), float(net.weight.item())))
                if c:
                    break
                with torch.no_grad():
                    net.weight.fill_(float(ep + 1))      # "training" changes the weights every epoch
                self._val_loss = v
            out[(name, stop)] = dict(seq=seq, trace=trace)
            print(name, stop, "epochs", len(trace), "converged", trace[-1][0])
    torch.save(out, os.path.join(make_golden.OUT, "npe_trainer_reference.pt"))


if __name__ == "__main__":
    main()
