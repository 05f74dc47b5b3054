"""Phase timing of the windowed attention kernel (library built with -DWIN_TIMING; `out` holds stamps, not data)."""
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
bias = torch.randn(3 * D, generator=g).to(dev)
rh = (0.02 * torch.randn(27, hd, generator=g)).to(dev); rw = (0.02 * torch.randn(27, hd, generator=g)).to(dev)
out = torch.zeros(n_img * 4096, D, dtype=torch.float16, device=dev)
for _ in range(2):
    assert lib.samrs_k_window_attention(1, qkv.data_ptr(), bias.data_ptr(), rh.data_ptr(), rw.data_ptr(), out.data_ptr(), n_img, grid, 14, heads, hd, s) == 0
    torch.cuda.synchronize()
nb = 256
t = out.view(torch.int64).flatten()[: nb * 64].cpu().numpy().reshape(nb, 8, 8)       # [block][wave][stamp]
names = ["barrier (previous item's readers done)", "LDS stores of K / V^T (+ wait for the prefetched loads)", "barrier",
         "issue of the next item's loads", "rel-pos setup (2 table products + LDS transposes)", "7 key tiles: QK^T, softmax, PV"]
items = t[:, 0, 7].astype(np.float64)
tot = t[:, 0, 6] / items
print(f"blocks {nb}; items per block {np.median(items):.1f}; cycles per item (wave 0): median {np.median(tot):.0f}")
print("per wave (median over blocks, cycles per item); waves w and w + 4 share a SIMD, wave 7 only stages")
print("  " + " " * 58 + "".join(f"   w{w:<5d}" for w in range(8)))
for i, n in enumerate(names):
    v = [np.median(t[:, w, i] / np.maximum(t[:, w, 7], 1)) for w in range(8)]
    print(f"  {n:58s}" + "".join(f"{x:9.0f}" for x in v))
v = [np.median(t[:, w, 6] / np.maximum(t[:, w, 7], 1)) for w in range(8)]
print(f"  {'total':58s}" + "".join(f"{x:9.0f}" for x in v))
