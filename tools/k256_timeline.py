"""Phase timing of the K = 256 streaming GEMM (library built with -DK2_TIMING; the first rows of C hold stamps)."""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
M, K, period = 131072, 256, 4096
for N in (256, 384):
    g = torch.Generator().manual_seed(0)
    A = torch.randn(M, K, generator=g).to(dev).to(torch.float16)
    W = (torch.randn(N, K, generator=g) / 16).to(dev).to(torch.float16)
    bias = torch.randn(N, generator=g).to(dev); add = torch.randn(period, N, generator=g).to(dev)
    C = torch.zeros(M, N, dtype=torch.int16, device=dev)
    lib.samrs_debug_set_gemm_variant(40)
    for with_add in (True, False):
        for _ in range(3):
            assert lib.samrs_k_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), add.data_ptr() if with_add else None,
                                    period if with_add else 0, M, N, K, 0, 0, 0, s) == 0
            torch.cuda.synchronize()
        t = C.view(torch.int64).flatten()[: 256 * 8].cpu().numpy().reshape(256, 8)
        n = t[:, 5].astype(np.float64)
        print(f"N={N} addend={with_add}: tiles per block {np.median(n):.0f}; block total {np.median(t[:, 4]):.0f} cycles")
        for i, name in enumerate(["wait own DMA pieces + older stores (vmcnt 0)", "barrier", "issue next tile's DMA", "fragment reads + MFMA + addend + stores"]):
            v = t[:, i] / n
            print(f"  {name:48s} {np.median(v):8.0f} cycles per tile")
    lib.samrs_debug_set_gemm_variant(8)
