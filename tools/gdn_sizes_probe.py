import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from compression_amd.layers import functional, gdn_forward
C = 192
beta = (1 + 0.1 * torch.rand(C)).cuda(); gamma = (0.1 * torch.eye(C) + 0.01 * torch.rand(C, C)).cuda()
prep = functional.GDNPrepared(beta, gamma, torch.bfloat16)
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps
for M in (262144, 524288, 1048576, 2097152, 4194304):
    xs = [torch.randn(M, C, device="cuda").bfloat16() for _ in range(3)]
    k = [0]
    def call():
        k[0] += 1
        return gdn_forward(xs[k[0] % 3], beta, gamma, prepared=prep)
    ms = t(call)
    print(f"[{M}, {C}]: {ms*1e3:.1f} us, {2*M*C*2/ms/1e6:.0f} GB/s")
