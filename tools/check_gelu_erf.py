import os, sys
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "clip-based-cross-modal-hash_amd")]
import torch
from xmh import ops
# act(A @ I + 0): feed x through an identity GEMM in exact mode (fp32 products) to read the activation alone
x = torch.cat([torch.linspace(-12, 12, 200001), torch.randn(100000) * 3, torch.tensor([0.0, -0.0, 1e-6, -1e-6, 30.0, -30.0])]).cuda()
n = (x.numel() + 31) // 32 * 32
xp = torch.zeros(n, device="cuda"); xp[:x.numel()] = x
A = xp.reshape(-1, 32).contiguous()
I = torch.eye(32, device="cuda")
ops.set_precision("f32x")
y = ops.gemm_nt(A, I, act=ops.ACT_GELU_ERF).reshape(-1)[:x.numel()]
ref = torch.nn.functional.gelu(x.double())
err = (y.double() - ref).abs()
print("max abs err %.3e at x=%.4f; max err / max(|x|,1e-3) %.3e" % (err.max(), x[err.argmax()], (err / x.abs().clamp_min(1e-3).double()).max()))
