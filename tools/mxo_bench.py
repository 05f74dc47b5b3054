"""GPU: lin1 of the all-split mode (gemm_et_mx_kernel with the exact-erf GELU epilogue that also emits its output as MXFP4 hi / lo rows).

    python tools/mxo_bench.py            # sha256 of every output on seeded inputs (ET, q_hi, q_lo, s_hi, s_lo) + us per launch at M = 32768

The hashes pin the epilogue bit for bit across refactorings of its store path (tests/test_kernels_gpu.py pins hi against the pack kernel and
lo only through its error); the timing table separates the lo terms, the GELU and the MX rows: plain f16 + GELU, MX + GELU, MX + GELU + rows."""
import hashlib, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine
lib = engine.load_library()
s = torch.cuda.current_stream().cuda_stream
reps = int(os.environ.get("REPS", "10"))


def pack(x, is_b):
    rows, K = x.shape
    hi = torch.zeros(rows, K, dtype=torch.int16, device="cuda")
    q = [torch.zeros(rows, K // 2, dtype=torch.uint8, device="cuda") for _ in range(2)]
    sc = [torch.zeros(int(lib.samrs_k_mx_scale_bytes(rows, K, int(is_b))), dtype=torch.uint8, device="cuda") for _ in range(2)]
    assert lib.samrs_k_mx4_pack(1, x.cuda().data_ptr(), None, None, hi.data_ptr(), q[0].data_ptr(), q[1].data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(),
                                rows, K, K, K, int(is_b), s) == 0
    return hi, q, sc


def timeit(fn):
    assert fn() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def sha(t):
    return hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:16]


def run(M, N, K, time_it):
    g = torch.Generator().manual_seed(77)
    A = torch.randn(M, K, generator=g)
    B = (torch.rand(N, K, generator=g) * 2 - 1) * 2.0 / math.sqrt(K)
    bias = (torch.randn(N, generator=g) * 0.5).cuda()
    Ah, qa, sa = pack(A, False)
    Bh, qb, sb = pack(B, True)
    Kp = N // 80 * 96
    out = torch.zeros(M, N, dtype=torch.int16, device="cuda")
    q = [torch.zeros(M, Kp // 2, dtype=torch.uint8, device="cuda") for _ in range(2)]
    sc = [torch.zeros(int(lib.samrs_k_mx_scale_bytes(M, Kp, 0)), dtype=torch.uint8, device="cuda") for _ in range(2)]
    mxo = lambda flags, rows: lib.samrs_k_gemm_mx_gelu_mxout(
        1, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), bias.data_ptr(), M, N, K, K, qa[1].data_ptr(), qa[0].data_ptr(), sa[1].data_ptr(), sa[0].data_ptr(),
        qb[0].data_ptr(), qb[1].data_ptr(), sb[0].data_ptr(), sb[1].data_ptr(), flags, *(t.data_ptr() if rows else None for t in (q[0], q[1], sc[0], sc[1])), s)
    assert mxo(1, True) == 0
    torch.cuda.synchronize()
    print(f"M {M} N {N} K {K}: ET {sha(out)}  q_hi {sha(q[0])}  q_lo {sha(q[1])}  s_hi {sha(sc[0])}  s_lo {sha(sc[1])}", flush=True)
    if time_it:
        t_plain = timeit(lambda: lib.samrs_k_gemm(1, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 1, 0, s))
        t_nog = timeit(lambda: mxo(0, False))
        t_g = timeit(lambda: mxo(1, False))
        t_gr = timeit(lambda: mxo(1, True))
        print(f"   us per launch: plain f16 + GELU {t_plain:7.1f} | MXFP4 lo terms {t_nog:7.1f} | + GELU {t_g:7.1f} | + GELU + MX rows {t_gr:7.1f}", flush=True)


run(512, 640, 1280, False)
run(32768, 5120, 1280, True)
