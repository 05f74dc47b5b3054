"""GPU: the split block GEMMs in their three forms on the ViT-H shapes (M = 32768): plain f16 (samrs_k_gemm), f16 lo terms in one
launch (samrs_k_gemm_split3), MXFP4 lo terms (samrs_k_gemm_mx).  us per launch and the cost relative to the plain product."""
import math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
s = torch.cuda.current_stream().cuda_stream
M, D = 32768, 1280
reps = int(os.environ.get("REPS", "10"))
g = torch.Generator().manual_seed(0)

def pack(x, G, GP, is_b):
    rows, K = x.shape
    Kp = K // G * GP
    hi = torch.zeros(rows, K, dtype=torch.int16, device="cuda")
    q = [torch.zeros(rows, Kp // 2, dtype=torch.uint8, device="cuda") for _ in range(2)]
    sc = [torch.zeros(int(lib.samrs_k_mx_scale_bytes(rows, Kp, int(is_b))), dtype=torch.uint8, device="cuda") for _ in range(2)]
    xd = x.cuda()
    assert lib.samrs_k_mx4_pack(1, xd.data_ptr(), None, None, hi.data_ptr(), q[0].data_ptr(), q[1].data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(),
                                rows, K, G, GP, int(is_b), s) == 0
    lo = torch.zeros(rows, K, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_convert_split(1, xd.data_ptr(), hi.data_ptr(), lo.data_ptr(), rows * K, s) == 0
    return hi, lo, q, sc

def timeit(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for name, N, K, G, GP, out_f32, from_n in (("qkv (v third split)", 3 * D, D, D, D, 0, 2 * D), ("qkv (all split)", 3 * D, D, D, D, 0, 0),
                                           ("proj (+residual, K' 1536)", D, D, 80, 96, 1, 0), ("lin1 (ET out)", 4 * D, D, D, D, 0, 0),
                                           ("lin2 (+residual)", D, 4 * D, 4 * D, 4 * D, 1, 0)):
    A = torch.randn(M, K, generator=g)
    B = (torch.rand(N, K, generator=g) * 2 - 1) / math.sqrt(K)
    Ah, Al, qa, sa = pack(A, G, GP, False)
    Bh, Bl, qb, sb = pack(B, G, GP, True)
    Kp = K // G * GP
    bias = torch.randn(N, generator=g).cuda()
    C = torch.zeros(M, N, dtype=torch.float32 if out_f32 else torch.int16, device="cuda")
    t_plain = timeit(lambda: lib.samrs_k_gemm(1, Ah.data_ptr(), Bh.data_ptr(), C.data_ptr(), bias.data_ptr(), None, 0, M, N, K, out_f32, 0, out_f32, s))
    t_s3 = timeit(lambda: lib.samrs_k_gemm_split3(1, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), C.data_ptr(), bias.data_ptr(), M, N, K, out_f32, out_f32, from_n, s))
    t_mx = timeit(lambda: lib.samrs_k_gemm_mx(1, Ah.data_ptr(), Bh.data_ptr(), C.data_ptr(), bias.data_ptr(), M, N, K, Kp, qa[1].data_ptr(), qa[0].data_ptr(),
                                              sa[1].data_ptr(), sa[0].data_ptr(), qb[0].data_ptr(), qb[1].data_ptr(), sb[0].data_ptr(), sb[1].data_ptr(), out_f32, out_f32, from_n, s))
    t_pack = timeit(lambda: lib.samrs_k_mx4_pack(1, None, Ah.data_ptr(), Al.data_ptr(), None, qa[0].data_ptr(), qa[1].data_ptr(), sa[0].data_ptr(), sa[1].data_ptr(), M, K, G, GP, 0, s))
    fl = 2.0 * M * N * K
    print(f"{name:28s} plain {t_plain:7.1f} us ({fl / t_plain / 1e6:6.0f} TF) | f16 lo terms {t_s3:7.1f} us ({t_s3 / t_plain:.2f}x) | MXFP4 lo terms {t_mx:7.1f} us ({t_mx / t_plain:.2f}x)"
          f" | A pack {t_pack:6.1f} us", flush=True)
