#!/usr/bin/env python
"""Pin the parity that this repository could not pin in its build container: compare the oracle (and, when a
ROCm device is present, the HIP estimator) with a REAL sbi + nflows install.

The build container has neither nflows nor network access, so `oracle/nsf_oracle.py` restates nflows 0.14 from
its published algorithm ("PARITY UNPINNED at the nflows boundary").  Wherever
`pip install sbi` works, run

    python tools/compare_with_nflows.py            # CPU: real NFlowsFlow vs oracle
    python tools/compare_with_nflows.py --device cuda   # + the HIP kernels

It builds sbi's own `build_nsf` estimator, copies its `state_dict` into the oracle / `NSFFlow`
(`load_nflows_state_dict`: same key names), perturbs the weights so that no identity-at-init
structure hides a discrepancy, and asserts |d log_prob| <= 1e-5 (norm-wise) and |d sample| <= 1e-5 on fixed seeds
for several shapes, incl. theta-dim 1 (ContextSplineMap).  Exit code 0 = pinned.
"""

import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--device", default="cpu")
    args = ap.parse_args()
    try:
        from sbi.neural_nets.net_builders.flow import build_nsf as ref_build_nsf
    except Exception as e:   # noqa: BLE001
        print(f"real sbi/nflows not importable here ({type(e).__name__}: {e}); nothing compared")
        return 2
    from oracle.nsf_oracle import NSFOracle

    worst = 0.0
    for cfg in [dict(D=10, C=10), dict(D=2, C=2), dict(D=4, C=7, num_bins=8), dict(D=1, C=3),
                dict(D=5, C=3, hidden_features=32, num_transforms=3)]:
        D, C = cfg.pop("D"), cfg.pop("C")
        g = torch.Generator().manual_seed(0)
        theta = torch.randn(2000, D, generator=g) * 0.5
        x = theta[:, :1].expand(-1, C) * 0.3 + torch.randn(2000, C, generator=g)
        torch.manual_seed(1)
        ref = ref_build_nsf(theta, x, **cfg)
        with torch.no_grad():
            for p in ref.parameters():
                p.add_(0.05 * torch.randn(p.shape, generator=g))
        sd = ref.state_dict()
        oracle = NSFOracle(theta, x, **cfg)
        missing = oracle.load_state_dict(sd, strict=True)
        th, xx = theta[:512], x[:512]
        with torch.no_grad():
            lp_ref = ref.log_prob(th.unsqueeze(0), xx)[0]
            lp_or = oracle.log_prob(th, xx)[0]
        d_lp = (lp_ref - lp_or).abs().max().item() / (1 + lp_ref.abs().max().item())
        torch.manual_seed(7)
        s_ref = ref.sample((64,), xx[:4])
        torch.manual_seed(7)
        s_or = oracle.sample((64,), xx[:4])
        d_s = (s_ref - s_or).abs().max().item()
        line = f"D={D} C={C} {cfg}: oracle vs nflows  d_logp(norm)={d_lp:.2e}  d_sample={d_s:.2e}"
        if args.device != "cpu":
            from sbi_amd.neural_nets.net_builders.flow import build_nsf

            est = build_nsf(theta, x, **cfg)
            est.net.load_nflows_state_dict(sd)
            est = est.to(args.device)
            with torch.no_grad():
                lp_hip = est.log_prob(th.to(args.device).unsqueeze(0), xx.to(args.device))[0].cpu()
            d_hip = (lp_ref - lp_hip).abs().max().item() / (1 + lp_ref.abs().max().item())
            torch.manual_seed(7)
            s_hip = est.sample((64,), xx[:4].to(args.device)).cpu()
            line += f" | HIP vs nflows  d_logp(norm)={d_hip:.2e}  d_sample={(s_ref - s_hip).abs().max().item():.2e}"
            d_lp, d_s = max(d_lp, d_hip), max(d_s, (s_ref - s_hip).abs().max().item())
        print(line)
        worst = max(worst, d_lp, d_s)
    ok = worst <= 1e-5
    print("PINNED" if ok else f"MISMATCH (worst {worst:.2e})")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
