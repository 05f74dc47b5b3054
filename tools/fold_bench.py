"""GPU micro-benchmark of the folded-LayerNorm GEMMs against the stand-alone path on the ViT-H shapes of an 8-tile batch
(run via gpurun): consumer (LayerNorm kernel + GEMM vs folded GEMM) for qkv and lin1 + GELU, producer (plain fp32-residual GEMM
vs the GEMM that also writes ET(x) and the row statistics) for proj and lin2.  Interleaved rounds in one process."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine

lib = engine.load_library()
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
prec, dt = engine.PRECISIONS["f16"], torch.float16
M, D = 32768, 1280
g = torch.Generator().manual_seed(0)


def timeit(fns, rounds=5, reps=10):
    best = {k: [] for k in fns}
    for _ in range(rounds):
        for k, f in fns.items():
            f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                f()
            e1.record(); torch.cuda.synchronize()
            best[k].append(e0.elapsed_time(e1) / reps * 1e3)
    return {k: min(v) for k, v in best.items()}


X = (torch.randn(M, D, generator=g) * 2 + 0.3).to(dev)
gamma, beta = (1 + 0.2 * torch.randn(D, generator=g)).to(dev), (0.1 * torch.randn(D, generator=g)).to(dev)
Y = torch.empty(M, D, dtype=torch.int16, device=dev)
Xh = torch.empty(M, D, dtype=torch.int16, device=dev)
stats = torch.empty(M, 8, 2, device=dev)
lib.samrs_k_rowstats_convert(prec, X.data_ptr(), Xh.data_ptr(), stats.data_ptr(), M, D, s)
rowstat = torch.empty(M, 2, device=dev)
lib.samrs_k_ln_rowstat(stats.data_ptr(), rowstat.data_ptr(), M, 1e-6, s)
for name, N, gelu in (("qkv", 3840, 0), ("lin1+gelu", 5120, 1)):
    W = (torch.randn(N, D, generator=g) / D ** 0.5).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    We = W.to(dt).view(torch.int16)
    Wf = torch.empty(N, D, dtype=torch.int16, device=dev)
    cvec, bf = torch.empty(N, device=dev), torch.empty(N, device=dev)
    lib.samrs_k_ln_fold_weight(prec, W.data_ptr(), gamma.data_ptr(), beta.data_ptr(), bias.data_ptr(), Wf.data_ptr(), cvec.data_ptr(), bf.data_ptr(), N, D, s)
    out = torch.empty(M, N, dtype=torch.int16, device=dev)
    fns = {
        "ln": lambda: lib.samrs_k_layernorm(prec, X.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-6, Y.data_ptr(), None, M, D, 0, 1, 64, 0, s),
        "gemm": lambda: lib.samrs_k_gemm(prec, Y.data_ptr(), We.data_ptr(), out.data_ptr(), bias.data_ptr(), None, 0, M, N, D, 0, gelu, 0, s),
        "ln+gemm": lambda: (lib.samrs_k_layernorm(prec, X.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-6, Y.data_ptr(), None, M, D, 0, 1, 64, 0, s),
                            lib.samrs_k_gemm(prec, Y.data_ptr(), We.data_ptr(), out.data_ptr(), bias.data_ptr(), None, 0, M, N, D, 0, gelu, 0, s)),
        "fold": lambda: lib.samrs_k_gemm_fold(prec, Xh.data_ptr(), Wf.data_ptr(), out.data_ptr(), bf.data_ptr(), cvec.data_ptr(), rowstat.data_ptr(), M, N, D, gelu, s),
        "rowstat": lambda: lib.samrs_k_ln_rowstat(stats.data_ptr(), rowstat.data_ptr(), M, 1e-6, s),
    }
    r = timeit(fns)
    print(f"{name:10s} N={N}: " + " | ".join(f"{k} {v:7.1f} us" for k, v in r.items()), flush=True)
for name, K in (("proj", 1280), ("lin2", 5120)):
    A = torch.randn(M, K, generator=g).to(dev).to(dt).view(torch.int16)
    W = (torch.randn(D, K, generator=g) / K ** 0.5).to(dev).to(dt).view(torch.int16)
    bias = torch.randn(D, generator=g).to(dev)
    C = torch.zeros(M, D, device=dev)
    fns = {
        "plain": lambda: lib.samrs_k_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, 0, M, D, K, 1, 0, 1, s),
        "stats": lambda: lib.samrs_k_gemm_stats(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), Xh.data_ptr(), stats.data_ptr(), M, D, K, s),
    }
    r = timeit(fns)
    print(f"{name:10s} K={K}: " + " | ".join(f"{k} {v:7.1f} us" for k, v in r.items()), flush=True)
