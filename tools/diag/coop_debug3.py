import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib
from tests.helpers import matched_pair
import sbi_amd.neural_nets.estimators.nsf_flow as nf
oracle, est, theta, x = matched_pair(D=10, C=10)
real_ptr = _lib.ptr
def ptr(t):
    p = real_ptr(t)
    if t is not None:
        print(f"   ptr {p:#x} .. {p + t.numel()*t.element_size():#x} shape {tuple(t.shape)}", flush=True)
    return p
_lib.ptr = ptr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 777
mode = sys.argv[2] if len(sys.argv) > 2 else "autograd"
th, xx = theta[:n].cuda(), x[:n].cuda()
if mode == "autograd":
    lp = est.log_prob(th, xx)
else:
    ws = nf.train_workspace(est.net, n, "cuda")
    lp = nf.train_forward(est.net, th, xx, ws)
torch.cuda.synchronize()
print("ok", lp.shape)
