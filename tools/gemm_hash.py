"""GPU: sha256 of the outputs of the encoder's four block GEMMs (ViT-H shapes, seeded operands, fp32 outputs accumulate into a non-zero C)
under the library named by SAMRS_LIB_PATH -- two builds that print the same digests compute the same bits.  usage: gemm_hash.py [f16|bf16]"""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine

lib = engine.load_library()
prec_name = sys.argv[1] if len(sys.argv) > 1 else "f16"
prec = engine.PRECISIONS[prec_name]
dt = torch.float16 if prec_name == "f16" else torch.bfloat16
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
g = torch.Generator().manual_seed(5)
for name, M, N, K, of32, gelu, acc in [("qkv", 32768, 3840, 1280, 0, 0, 0), ("proj+res", 32768, 1280, 1280, 1, 0, 1), ("lin1+gelu", 32768, 5120, 1280, 0, 1, 0),
                                       ("lin2+res", 32768, 1280, 5120, 1, 0, 1), ("proj b=3", 3 * 4096, 1280, 1280, 1, 0, 1), ("neck-like", 4096, 256, 1280, 1, 0, 0)]:
    A = torch.randn(M, K, generator=g).to(dev).to(dt)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dt)
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.randn(M, N, generator=g).to(dev) if of32 else torch.zeros(M, N, dtype=torch.int16, device=dev)
    rc = lib.samrs_k_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), None, 0, M, N, K, of32, gelu, acc, s)
    torch.cuda.synchronize()
    print(name, rc, hashlib.sha256(C.cpu().numpy().tobytes()).hexdigest()[:16], flush=True)
