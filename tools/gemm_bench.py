"""GPU micro-benchmark of gemm_et variants on the encoder's hot shapes (run via gpurun).
Interleaved rounds in ONE process (cdna_hip_programming.md 5.4 rule 24); random operands."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samrs_amd import engine

lib = engine.load_library()
lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
lib.samrs_debug_set_gemm_variant.restype = None
dev = torch.device("cuda")
s = torch.cuda.current_stream().cuda_stream
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "1"])]
# SKEWS="a,b;c,d": every variant is additionally run with these first-round start skews (variant id printed as v<id>s<a>.<b>)
skews = [tuple(int(x) for x in sk.split(",")) for sk in os.environ.get("SKEWS", "").split(";") if sk]
lib.samrs_debug_set_gemm_skew.argtypes = [__import__("ctypes").c_int] * 2
lib.samrs_debug_set_gemm_skew.restype = None
if skews:
    variants = [(v, sk) for v in variants for sk in [(0, 0)] + skews]
else:
    variants = [(v, (0, 0)) for v in variants]
_set_variant = lib.samrs_debug_set_gemm_variant


def set_variant(vs):
    _set_variant(vs[0])
    lib.samrs_debug_set_gemm_skew(vs[1][0], vs[1][1])
prec_name = sys.argv[2] if len(sys.argv) > 2 else "f16"
prec = engine.PRECISIONS[prec_name]
dt = torch.float16 if prec_name == "f16" else torch.bfloat16
shapes = [("qkv-like", 32768, 3840, 1280, 0, 0, 0), ("lin2-like ET", 32768, 1280, 5120, 0, 0, 0)] if os.environ.get("ABL") else [("lin1+gelu", 32768, 5120, 1280, 0, 1, 0), ("lin2+res", 32768, 1280, 5120, 1, 0, 1),
          ("qkv     ", 32768, 3840, 1280, 0, 0, 0), ("proj+res", 32768, 1280, 1280, 1, 0, 1),
          ("lin1 b=1", 4096, 5120, 1280, 0, 1, 0)]
if os.environ.get("DEC_SHAPES"):
    # the decoder's image-side projections for 32 prompts (rows = 32 x 4096 keys, K = 256, 2-D addend = folded PE with period 4096)
    shapes = [("dec kvq  ", 131072, 384, 256, 0, 0, 0), ("dec finkv", 131072, 256, 256, 0, 0, 0)]
if os.environ.get("STRIDE_PROBE"):
    # L2-channel probe: the same GEMM with a row stride of 10 x 256 B (K = 1280), 11 x 256 B (K = 1408), 40 x 256 B (K = 5120), 41 x 256 B
    shapes = [("K=1280   ", 32768, 5120, 1280, 0, 1, 0), ("K=1408   ", 32768, 5120, 1408, 0, 1, 0), ("K=1344   ", 32768, 5120, 1344, 0, 1, 0),
              ("K=5120   ", 32768, 1280, 5120, 1, 0, 1), ("K=5248   ", 32768, 1280, 5248, 1, 0, 1), ("K=5184   ", 32768, 1280, 5184, 1, 0, 1)]
if os.environ.get("ONLY"):        # ONLY=lin1,qkv: shapes whose name starts with one of these (per-shape PMC passes)
    shapes = [sh for sh in shapes if any(sh[0].strip().startswith(o) for o in os.environ["ONLY"].split(","))]
g = torch.Generator().manual_seed(0)
for name, M, N, K, of32, gelu, acc in shapes:
    A = torch.randn(M, K, generator=g).to(dev).to(dt)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).to(dt)
    if os.environ.get("ZERO"):
        # DVFS probe (MI355X_MICROARCH.md, "DVFS give-back"): the same kernels on all-zero operands draw less power and clock
        # higher; the ratio to the random-operand rate is the share of the gap to the roof that is power, not schedule
        A.zero_(); W.zero_()
    bias = torch.randn(N, generator=g).to(dev)
    C = torch.zeros(M, N, dtype=torch.float32 if of32 else torch.int16, device=dev)
    ref = A.float() @ W.float().t() + bias
    add2d, period = None, 0
    if os.environ.get("DEC_SHAPES"):
        period = 4096
        add2d = torch.randn(period, N, generator=g).to(dev)
        ref = (ref.view(-1, period, N) + add2d).view(M, N)
    add_ptr = add2d.data_ptr() if add2d is not None else None
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    res = {}
    first = None
    for v in variants:
        set_variant(v)
        C.zero_()
        lib.samrs_k_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), add_ptr, period, M, N, K, of32, gelu, acc, s)
        got = C.float() if of32 else C.view(dt).float()
        err = ((got - ref).norm() / ref.norm()).item()
        if first is None:
            first = C.clone()
        same = bool(torch.equal(C, first))          # every tile shape accumulates over k in the same order
        res[v] = {"err": err, "ms": [], "same": same}
        del got
    for rnd in range(5):
        for v in variants:
            set_variant(v)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                lib.samrs_k_gemm(prec, A.data_ptr(), W.data_ptr(), C.data_ptr(), bias.data_ptr(), add_ptr, period, M, N, K, of32, gelu, acc, s)
            e1.record(); torch.cuda.synchronize()
            res[v]["ms"].append(e0.elapsed_time(e1) / 10)
    lib.samrs_debug_set_gemm_skew(0, 0)
    # vendor reference point (hipBLASLt through torch), same operands, no epilogue
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    out = torch.empty(M, N, dtype=dt, device=dev)
    for _ in range(3): torch.matmul(A, W.t(), out=out)
    e0.record()
    for _ in range(10): torch.matmul(A, W.t(), out=out)
    e1.record(); torch.cuda.synchronize()
    lib_ms = e0.elapsed_time(e1) / 10
    fl = 2.0 * M * N * K
    line = f"{name:10s} M={M} N={N} K={K}: " + " | ".join(
        f"v{v[0]}{'' if v[1] == (0, 0) else 's%d.%d' % v[1]}: {min(r['ms'])*1e3:7.1f}us {fl/min(r['ms'])/1e9:7.1f}TF (med {sorted(r['ms'])[2]*1e3:.1f}us) err {r['err']:.1e}{'' if r['same'] else ' BITS-DIFFER'}" for v, r in res.items())
    print(line + f" | torch.matmul {lib_ms*1e3:.1f}us {fl/lib_ms/1e9:.1f}TF", flush=True)
