import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops, _lib
M, N, K = 5000, 3072, 768
A = torch.rand(M, K, device="cuda") * 2 - 1
W = ((torch.rand(N, K, device="cuda") * 2 - 1) * 0.05).half().float()
b = torch.randn(N, device="cuda"); x = torch.empty(M, N, device="cuda")
for mode, slot in (("f16", "gemm_f16"), ("f32", "gemm_s16")):
    ops.set_precision(mode)
    for act in (0, 1, 2, 3):
        fn = lambda: ops.gemm_nt(A, W, b, act=act, out=x)
        for _ in range(3): fn()
        torch.cuda.synchronize(); _lib.prof_enable(True)
        for _ in range(20): fn()
        torch.cuda.synchronize(); ms, n = _lib.prof_read(slot); _lib.prof_enable(False)
        print(mode, "act", act, "%.1f us" % (ms * 1e3))
