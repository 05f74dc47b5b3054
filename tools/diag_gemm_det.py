import sys, os, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
lib.samrs_debug_set_gemm_variant.argtypes = [ctypes.c_int]; lib.samrs_debug_set_gemm_variant.restype = None
dev = torch.device("cuda"); s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(0)
shapes = [("up2", 65536, 128, 64, 0, 1, 0, 0), ("up1", 16384, 256, 256, 1, 0, 0, 0), ("kvq", 16384, 384, 256, 0, 0, 0, 1),
          ("i2t_out", 16384, 256, 128, 1, 0, 1, 0), ("lin1", 8192, 5120, 1280, 0, 1, 0, 0), ("qkv", 8192, 3840, 1280, 0, 0, 0, 0),
          ("lin2", 8192, 1280, 5120, 1, 0, 1, 0)]
for var in [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["8"])]:
    lib.samrs_debug_set_gemm_variant(var)
    for name, M, N, K, of32, gelu, acc, add in shapes:
        A = torch.randn(M, K, generator=g).to(dev).half()
        W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).half()
        bias = torch.randn(N, generator=g).to(dev)
        add2d = torch.randn(4096, N, generator=g).to(dev) if add else None
        C0 = torch.randn(M, N, generator=g).to(dev) if of32 else torch.zeros(M, N, dtype=torch.int16, device=dev)
        outs = []
        for rep in range(6):
            C = C0.clone()
            rc = lib.samrs_k_gemm(1, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), add2d.data_ptr() if add else None, 4096, M, N, K, of32, gelu, acc, s)
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(C.clone())
        ref = A.float() @ W.float().t() + bias + (add2d.repeat(M // 4096, 1) if add else 0)
        if gelu: ref = torch.nn.functional.gelu(ref)
        if acc: ref = ref + C0
        got = outs[0] if of32 else outs[0].view(torch.float16).float()
        nd = sum(int((o != outs[0]).sum().item()) for o in outs[1:])
        bad_rows = (outs[1] != outs[0]).any(1).nonzero().flatten()[:8].tolist() if nd else []
        print(f"v{var} {name:8s} M={M} N={N} K={K}: rel err {((got - ref).norm() / ref.norm()).item():.2e}  elements differing across 5 repeats: {nd}  rows {bad_rows}", flush=True)
