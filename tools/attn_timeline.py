"""Phase timing of the global attention kernel (library built with -DGLB_TIMING; results in `out` are stamps, not data)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
n_img, heads, hd, grid = 8, 16, 80, 64
D = heads * hd
g = torch.Generator().manual_seed(0)
qkv = torch.randn(n_img * 4096, 3 * D, generator=g).to(dev).to(torch.float16)
rh = (0.02 * torch.randn(127, hd, generator=g)).to(dev); rw = (0.02 * torch.randn(127, hd, generator=g)).to(dev)
out = torch.zeros(n_img * 4096, D, dtype=torch.float16, device=dev)
for _ in range(2):
    assert lib.samrs_k_global_attention(1, qkv.data_ptr(), rh.data_ptr(), rw.data_ptr(), out.data_ptr(), n_img, grid, heads, hd, s) == 0
    torch.cuda.synchronize()
nb = 32 * heads * n_img
t = out.view(torch.int64).flatten()[: nb * 8].cpu().numpy().reshape(nb, 8)
names = ["issue DMA + rh", "QK^T (+scale, bias, max)", "softmax", "PV", "wait for own DMA", "barrier"]
tot = t[:, 6].astype(np.float64)
print(f"blocks {nb}; main loop per block: median {np.median(tot):.0f} cycles = {np.median(tot) / 64:.0f} per tile")
for i, n in enumerate(names):
    print(f"  {n:28s} {np.median(t[:, i]) / 64:7.0f} cycles per tile ({100 * np.median(t[:, i]) / np.median(tot):4.1f} %)")
print(f"setup before the first tile (Q fragments, rel-pos tables, first DMA): median {np.median(t[:, 7]):.0f} cycles, p90 {np.percentile(t[:, 7], 90):.0f}")
