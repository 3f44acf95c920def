import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops
for (M, N, K) in ((5000, 2304, 768), (5000, 768, 768), (5000, 3072, 768), (5000, 768, 3072), (4096, 4096, 4096), (20000, 2304, 768), (20000, 768, 768), (20000, 768, 3072)):
    A = torch.randn(M, K, device="cuda") * 3.0
    W = (torch.randn(N, K, device="cuda") * 0.05).half().float()
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda")
    for _ in range(3): ops.gemm_nt(A, W, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_nt(A, W, b, out=out)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rows = slice(0, 512)
    ref = A[rows].double() @ W.double().t() + b.double()
    err = float((out[rows].double() - ref).abs().max() / ref.abs().max())
    print("%s M=%5d N=%5d K=%5d %8.3f ms %7.1f TF err %.1e" % (os.environ.get("XMH_GEMM_DIRECT_A", "default"), M, N, K, ms, 2.0 * M * N * K / ms / 1e9, err))
