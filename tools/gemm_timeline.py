"""Per-block timeline of the pair-stage GEMM (variant 164 = timing build of variant 27): where a tile's time goes.
Run on the GPU box: python tools/gemm_timeline.py [M N K]."""
import sys, os, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
lib.samrs_debug_set_gemm_variant.argtypes = [ctypes.c_int]
M, N, K = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (32768, 3840, 1280)
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
A = torch.randn(M, K, generator=g).to(dev).half()
W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).half()
bias = torch.randn(N, generator=g).to(dev)
C = torch.zeros(M, N, dtype=torch.int16, device=dev)
ntiles = (M // 256) * (N // 320)
tl = torch.zeros(max(ntiles * 32, N), dtype=torch.int64, device=dev)
lib.samrs_debug_set_gemm_variant(164)
for it in range(3):
    tl.zero_()
    rc = lib.samrs_k_gemm(engine.PREC_F16, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), tl.data_ptr(), 1, M, N, K, 0, 0, 0, s)
    assert rc == 0
    torch.cuda.synchronize()
t = tl.cpu().numpy()[: ntiles * 32].reshape(ntiles, 32)
nst = K // 64
hw = t[:, 28]
cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xF)      # xcc, se, cu (gfx9 HW_ID layout)
t0 = t[:, 0].min()
rel = lambda x: (x - t0)
print(f"{ntiles} tiles, {len(np.unique(cu))} distinct CU ids; kernel span {(t[:, 27].max() - t0)} cycles")
fill = t[:, 1] - t[:, 0]
stages = np.diff(t[:, 1:2 + nst], axis=1)             # stage t: from previous stamp to end of stage t
loop = t[:, 26] - t[:, 1]
epi = t[:, 27] - t[:, 26]
tot = t[:, 27] - t[:, 0]
q = lambda a: f"min {np.min(a):7.0f} med {np.median(a):7.0f} p90 {np.percentile(a, 90):7.0f} max {np.max(a):7.0f}"
print("entry -> stage 0 landed :", q(fill))
print("main loop (all stages)  :", q(loop))
print("per stage               :", q(stages))
print("  first stage / last    :", q(stages[:, 0]), "|", q(stages[:, -1]))
print("epilogue + store drain  :", q(epi))
print("block total             :", q(tot))
gaps = []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    idx = idx[np.argsort(t[idx, 0])]
    for a, b in zip(idx[:-1], idx[1:]):
        gaps.append(t[b, 0] - t[a, 27])
if gaps:
    print("gap block end -> next block entry on the same CU:", q(np.array(gaps)))
rounds = np.argsort(t[:, 0])
print("first 3 blocks by start: ", [(int(rel(t[i, 0])), int(rel(t[i, 27]))) for i in rounds[:3]])
