import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from sbi_amd import _lib
real_ptr = _lib.ptr
def ptr(t):
    p = real_ptr(t)
    if t is not None:
        print(f"   ptr {p:#x} .. {p + t.numel()*t.element_size():#x} shape {tuple(t.shape)}", flush=True)
    return p
_lib.ptr = ptr
import tests.test_coop_gpu as T
torch.manual_seed(1)
T.test_image_kind_and_log_prob_match_oracle_and_throughput_kernels(T.CONFIGS[0])
print("test body ok")
