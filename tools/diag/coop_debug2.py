import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from tests.helpers import matched_pair
from sbi_amd.neural_nets.estimators.nsf_flow import train_forward, train_workspace
oracle, est, theta, x = matched_pair(D=10, C=10, n=9000)
for n in [int(a) for a in sys.argv[1:]]:
    th, xx = theta[:n].cuda(), x[:n].cuda()
    print("train_forward n", n, flush=True)
    ws = train_workspace(est.net, n, "cuda")
    lp = train_forward(est.net, th, xx, ws)
    torch.cuda.synchronize()
    with torch.no_grad():
        ref = oracle.log_prob(theta[:n], x[:n])[0]
    print("   max err", (lp.cpu() - ref).abs().max().item(), flush=True)
print("done")
