import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/clip-based-cross-modal-hash_amd"]
import numpy as np, torch
from xmh import dense, ops
g = np.load("/root/repo/tests/golden/calc_utils_ternary_float.npz")
a, b = torch.from_numpy(g["fa"]).cuda(), torch.from_numpy(g["fb"]).cuda()
d = dense.pairwise_l2(a, b).cpu().numpy()
print("euc max diff", np.abs(d - g["euc"]).max())
print("gram diff", (ops.gemm_nt(a, b).cpu() - (a.cpu() @ b.cpu().t())).abs().max())
print("sqn diff", (dense._sqnorm(a).cpu() - (a.cpu() ** 2).sum(1)).abs().max())
