"""GPU: every HIP kernel alone, through the C ABI's kernel-level entry points, against a plain
torch fp32/fp64 CPU statement of the same op on the same (already rounded) operands.

Tolerances: MFMA operands are bf16 / f16, accumulation is fp32.  With operands pre-rounded on the
host the only differences left are fp32 summation order and the rounding of ET outputs
(one ulp: 2^-8 relative for bf16, 2^-11 for f16).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

PRECS = [("f16", 1, torch.float16, 2.0 ** -10), ("bf16", 0, torch.bfloat16, 2.0 ** -7)]


@pytest.fixture(scope="module")
def lib():
    from samrs_amd import engine
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return engine.load_library()


def dev(t):
    return t.cuda().contiguous()


def need_experiments(lib):
    """Kernels that were measured and not adopted (32x32x16 GEMMs, the LayerNorm fold) are left out of the product build
    (samrs_amd/csrc/Makefile: make EXPERIMENTS=1 builds them); their tests run against such a build only."""
    lib.samrs_debug_has_experiments.restype = __import__("ctypes").c_int
    if not lib.samrs_debug_has_experiments():
        pytest.skip("built without EXPERIMENTS=1: the retired kernels are not in this library")


def stream():
    return torch.cuda.current_stream().cuda_stream


def et_bits(x, dt):
    """fp32 tensor -> (rounded fp32 values, int16 bit-pattern tensor)"""
    r = x.to(dt)
    return r.to(torch.float32), r.view(torch.int16)


def rel_err(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp(min=1e-30)).item(), (a - b).abs().max().item()


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_convert_bit_exact(lib, name, prec, dt, ulp):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(4096 * 8, generator=g) * torch.logspace(-6, 4, 4096 * 8)
    x[:8] = torch.tensor([0.0, -0.0, 1.0, -1.0, 65504.0, 70000.0, -70000.0, 1e-8])
    out = torch.empty(x.numel(), dtype=torch.int16, device="cuda")
    xd = dev(x)
    assert lib.samrs_k_convert(prec, xd.data_ptr(), out.data_ptr(), x.numel(), stream()) == 0
    ref = x.clamp(-65504, 65504).to(dt).view(torch.int16) if dt == torch.float16 else x.to(dt).view(torch.int16)
    assert torch.equal(out.cpu(), ref), f"{name}: {(out.cpu() != ref).sum().item()} mismatching bit patterns"


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (384, 128, 640), (128, 256, 2304),
                                   (256, 128, 64), (256, 256, 192), (512, 256, 640), (768, 128, 1280)])
def test_gemm_et_variants(lib, name, prec, dt, ulp, M, N, K):
    g = torch.Generator().manual_seed(M + N + K)
    A, Ab = et_bits(torch.randn(M, K, generator=g), dt)
    B, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
    bias = torch.randn(N, generator=g)
    add2d = torch.randn(64, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ B.double().t()
    Ad, Bd, biasd, add2dd = dev(Ab), dev(Bb), dev(bias), dev(add2d)
    # (a) plain fp32 out, no bias
    out = torch.empty(M, N, device="cuda")
    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), None, None, 0, M, N, K, 1, 0, 0, stream()) == 0
    r, mx = rel_err(out.cpu(), ref)
    print(f"gemm {name} {M}x{N}x{K} plain: rel {r:.2e} max {mx:.2e}")
    assert r < 2e-6, "transposed / permuted output?" if r > 0.1 else "accumulation error too large"
    # (b) bias + add2d(period 64) + accumulate into existing fp32 C
    out = dev(C0.clone())
    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), biasd.data_ptr(), add2dd.data_ptr(), 64,
                            M, N, K, 1, 0, 1, stream()) == 0
    ref_b = ref + bias.double() + add2d.double().repeat(M // 64, 1) + C0.double()
    r, mx = rel_err(out.cpu(), ref_b)
    print(f"gemm {name} {M}x{N}x{K} bias+add2d+acc: rel {r:.2e}")
    assert r < 2e-6
    # (c) ET out with bias + GELU
    out = torch.empty(M, N, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), biasd.data_ptr(), None, 0, M, N, K, 0, 1, 0, stream()) == 0
    ref_c = F.gelu((ref + bias.double()).float())
    got = out.cpu().view(dt).float()
    err = (got - ref_c).abs() / ref_c.abs().clamp(min=1e-2)
    print(f"gemm {name} {M}x{N}x{K} gelu->ET: max rel {err.max().item():.2e}")
    assert err.max().item() < 1.5 * ulp


@pytest.mark.parametrize("variant,M,N,K", [(5, 512, 256, 192), (6, 512, 512, 192), (7, 512, 256, 192), (9, 256, 128, 128),
                                           (10, 512, 640, 192), (10, 256, 640, 64), (6, 256, 256, 64),
                                           # pair-stage (64-deep, whole-cache-line DMA) kernels: one stage, odd / even stage counts, many tiles
                                           (20, 256, 256, 64), (20, 512, 512, 192), (20, 1024, 768, 320), (21, 256, 640, 64),
                                           (21, 512, 640, 192), (21, 2048, 1280, 256), (22, 512, 512, 192), (22, 256, 256, 128),
                                           (23, 512, 640, 192), (23, 768, 640, 320),
                                           # one block-wide barrier per pair stage
                                           (24, 512, 512, 192), (24, 256, 256, 64), (25, 512, 640, 192), (25, 2048, 1280, 256),
                                           (26, 512, 512, 320), (27, 768, 640, 320), (27, 256, 640, 64),
                                           # persistent pair-stage kernel: 1 tile per block, blocks that walk 2 / 3 tiles, odd stage count
                                           (28, 512, 640, 192), (28, 8192, 3200, 192), (28, 256, 640, 64), (28, 16384, 3840, 320)])
def test_gemm_every_tile_variant(lib, variant, M, N, K):
    """(variant 28 = the persistent 256x320 kernel; 8192 x 3200 gives 320 tiles, so 64 blocks walk two tiles.)
    Each tile shape of the pipelined GEMM forced explicitly (the auto rule only picks the 256x256 / 256x320
    kernels at sizes the unit tests do not reach): fp32 accumulate output, f16 output with 2-D addend, GELU output."""
    name, prec, dt, ulp = PRECS[0]
    g = torch.Generator().manual_seed(variant * 1000 + M + N + K)
    A, Ab = et_bits(torch.randn(M, K, generator=g), dt)
    B, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
    bias = torch.randn(N, generator=g)
    add2d = torch.randn(64, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    ref = A.double() @ B.double().t()
    Ad, Bd, biasd, add2dd = dev(Ab), dev(Bb), dev(bias), dev(add2d)
    lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
    lib.samrs_debug_set_gemm_variant(variant)
    try:
        out = dev(C0.clone())
        assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), biasd.data_ptr(), add2dd.data_ptr(), 64,
                                M, N, K, 1, 0, 1, stream()) == 0
        r, _ = rel_err(out.cpu(), ref + bias.double() + add2d.double().repeat(M // 64, 1) + C0.double())
        assert r < 2e-6, f"variant {variant} fp32 out: rel {r:.2e}"
        out = torch.empty(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), biasd.data_ptr(), add2dd.data_ptr(), 64,
                                M, N, K, 0, 0, 0, stream()) == 0
        ref_e = (ref + bias.double() + add2d.double().repeat(M // 64, 1)).float()
        err = (out.cpu().view(dt).float() - ref_e).abs() / ref_e.abs().clamp(min=1e-2)
        assert err.max().item() < 1.5 * ulp, f"variant {variant} ET out + add2d"
        assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), out.data_ptr(), biasd.data_ptr(), None, 0, M, N, K, 0, 1, 0, stream()) == 0
        ref_c = F.gelu((ref + bias.double()).float())
        err = (out.cpu().view(dt).float() - ref_c).abs() / ref_c.abs().clamp(min=1e-2)
        print(f"gemm variant {variant} {M}x{N}x{K}: gelu->ET max rel {err.max().item():.2e}")
        assert err.max().item() < 1.5 * ulp, f"variant {variant} GELU out"
    finally:
        lib.samrs_debug_set_gemm_variant(8)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_tile_variants_are_bit_identical(lib, name, prec, dt, ulp):
    """Every tile shape accumulates each output element over k in the same ascending 32-wide MFMA steps, so the pair-stage
    kernels (20-23) must reproduce the staggered 256x256 / 256x320 kernels (6 / 10) BIT FOR BIT -- also the strongest
    race screen for a new LDS pipeline: repeated launches, fp32 and ET outputs, several K depths."""
    lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
    try:
        for (M, N, K) in [(2048, 1280, 1280), (1024, 1280, 320), (512, 1280, 64), (1536, 2560, 704), (32768, 1280, 192), (24576, 3840, 128)]:
            g = torch.Generator().manual_seed(M + N + K)
            _, Ab = et_bits(torch.randn(M, K, generator=g), dt)
            _, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
            bias = dev(torch.randn(N, generator=g))
            Ad, Bd = dev(Ab), dev(Bb)
            outs = {}
            for variant in (10, 6, 20, 21, 22, 23, 24, 25, 26, 27, 28):
                lib.samrs_debug_set_gemm_variant(variant)
                for rep in range(3):
                    of = torch.zeros(M, N, device="cuda")
                    oe = torch.zeros(M, N, dtype=torch.int16, device="cuda")
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), of.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 1, 0, 0, stream()) == 0
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), oe.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 1, 0, stream()) == 0
                    if not outs:
                        outs = {"f": of, "e": oe}
                    else:
                        assert torch.equal(of, outs["f"]), f"{name} {M}x{N}x{K}: variant {variant} rep {rep} fp32 output differs from variant 10"
                        assert torch.equal(oe, outs["e"]), f"{name} {M}x{N}x{K}: variant {variant} rep {rep} GELU/ET output differs from variant 10"
    finally:
        lib.samrs_debug_set_gemm_variant(8)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_w4x_kernel_is_bit_identical(lib, name, prec, dt, ulp):
    """Round 5: the four-wave 256 x 256 x 64 kernel on v_mfma_f32_16x16x32 with 128 x 128 wave tiles and all 256
    accumulator registers in AGPRs (variant 38; ET outputs).  It walks k in the same ascending 32-wide steps over the same LDS image
    as the eight-wave pair-stage kernels, so it must reproduce them BIT FOR BIT -- ET output with and without the GELU -- on 4 to 80
    stages, one to ten tiles per block (the persistent stage stream runs across tiles), and from launch to launch (the race screen
    for its one-barrier-per-stage hand-over).  fp32 outputs fall back to the eight-wave kernel (checked: same bits, trivially)."""
    lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
    shapes = [(512, 256, 256), (2048, 1280, 1280), (16384, 2560, 256), (24576, 3840, 384), (32768, 5120, 1280), (32768, 1280, 5120),
              # round 6: ODD stage counts (5, 7, 21 = lin1 with the 64-column outlier extension): the buffer parity carries across tiles
              (16384, 2560, 320), (24576, 3840, 448), (32768, 5120, 1344)]
    try:
        for (M, N, K) in shapes:
            g = torch.Generator().manual_seed(M + N + K + 38)
            _, Ab = et_bits(torch.randn(M, K, generator=g), dt)
            _, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
            bias = dev(torch.randn(N, generator=g))
            res = dev(torch.randn(M, N, generator=g))
            Ad, Bd = dev(Ab), dev(Bb)
            ref = None
            for variant, reps in ((27, 1), (38, 3)):
                lib.samrs_debug_set_gemm_variant(variant)
                for rep in range(reps):
                    of = torch.zeros(M, N, device="cuda")
                    oa = res.clone()
                    oe = torch.zeros(M, N, dtype=torch.int16, device="cuda")
                    og = torch.zeros(M, N, dtype=torch.int16, device="cuda")
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), of.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 1, 0, 0, stream()) == 0
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), oa.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 1, 0, 1, stream()) == 0
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), oe.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 0, 0, stream()) == 0
                    assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), og.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 1, 0, stream()) == 0
                    if ref is None:
                        ref = (of, oa, oe, og)
                        assert of.abs().max() > 0 and not torch.equal(oe, og)
                    else:
                        for tag, x, y in zip(("fp32", "fp32 accumulate", "ET", "ET + GELU"), (of, oa, oe, og), ref):
                            assert torch.equal(x, y), f"{name} {M}x{N}x{K}: variant 38 rep {rep} {tag} output differs from variant 27"
    finally:
        lib.samrs_debug_set_gemm_variant(8)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("variant", [30, 31, 32, 33, 34, 36])
def test_gemm_m32_kernel(lib, name, prec, dt, ulp, variant):
    """The 32x32x16 symmetric-schedule kernel (30 persistent / 31 one tile per block; 32 / 33 with the DMA pieces spread over
    two steps; 34 / 36 = its four-wave, 512-register flavour with hand-allocated AGPR accumulators).  It sums k in 16-wide MFMA steps, so it is not bit-identical with the 16x16x32 tile family; it must agree with
    it to fp32-summation-order accuracy (fp32 output: rel. L2 < 2e-6; ET output: <= 1 ulp apart, and only where the fp32 value
    sits on a rounding boundary), with fp64 on the small shapes, and with ITSELF bit for bit across repeated launches (the race
    screen: 2, 4, 6 and 80 stages, one to three tiles per block, the real proj / lin2 / qkv shapes of an 8-tile batch)."""
    need_experiments(lib)
    lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
    shapes = [(256, 640, 128), (512, 640, 256), (8192, 3200, 256), (16384, 3840, 384), (32768, 1280, 1280)]
    if variant in (30, 34):
        shapes += [(32768, 1280, 5120), (32768, 3840, 1280)]
    try:
        for (M, N, K) in shapes:
            g = torch.Generator().manual_seed(M + N + K + variant)
            A, Ab = et_bits(torch.randn(M, K, generator=g), dt)
            B, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
            bias = dev(torch.randn(N, generator=g))
            C0 = dev(torch.randn(M, N, generator=g))
            Ad, Bd = dev(Ab), dev(Bb)

            def run(v):
                lib.samrs_debug_set_gemm_variant(v)
                of = C0.clone()                                                       # fp32 out: bias + residual (proj / lin2)
                oe = torch.full((M, N), 0x7E00, dtype=torch.int16, device="cuda")     # NaN canaries
                og = torch.full((M, N), 0x7E00, dtype=torch.int16, device="cuda")
                assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), of.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 1, 0, 1, stream()) == 0
                assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), oe.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 0, 0, stream()) == 0
                assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), og.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 1, 0, stream()) == 0
                torch.cuda.synchronize()
                return of, oe, og

            rf, re_, rg = run(27)                     # the 16x16x32 pair-stage kernel
            first = None
            for rep in range(3):
                of, oe, og = run(variant)
                if first is None:
                    first = (of, oe, og)
                else:
                    assert torch.equal(of, first[0]) and torch.equal(oe, first[1]) and torch.equal(og, first[2]), \
                        f"{name} variant {variant} {M}x{N}x{K}: rep {rep} differs from rep 0 (race)"
            of, oe, og = first
            r = ((of - rf).double().norm() / rf.double().norm()).item()
            mx = (of - rf).abs().max().item()
            assert r < 2e-6 and mx < 1e-4 * math.sqrt(K / 128), f"{name} variant {variant} {M}x{N}x{K} fp32 out: rel {r:.2e} max {mx:.2e}"
            for tag, o, ref in (("ET", oe, re_), ("GELU", og, rg)):
                a, b = o.view(dt).float(), ref.view(dt).float()
                assert torch.isfinite(a).all(), f"{name} variant {variant} {M}x{N}x{K} {tag}: untouched / non-finite outputs"
                err = ((a - b).abs() / b.abs().clamp(min=1e-2)).max().item()
                frac = (o != ref).float().mean().item()
                assert err < 1.1 * ulp and frac < 2e-2, f"{name} variant {variant} {M}x{N}x{K} {tag}: max rel {err:.2e}, {frac:.2e} of the outputs differ"
            if M * N * K <= 512 * 640 * 256:
                ref = A.double() @ B.double().t() + bias.cpu().double()
                r, _ = rel_err(of.cpu(), ref + C0.cpu().double())
                assert r < 2e-6, f"{name} variant {variant} {M}x{N}x{K} vs fp64: {r:.2e}"
                got = oe.cpu().view(dt).float()
                err = ((got - ref.float()).abs() / ref.float().abs().clamp(min=1e-2)).max().item()
                assert err < 1.5 * ulp, f"{name} variant {variant} {M}x{N}x{K} ET vs fp64: {err:.2e}"
            print(f"m32 variant {variant} {name} {M}x{N}x{K}: fp32 rel {r:.2e}")
    finally:
        lib.samrs_debug_set_gemm_variant(8)


def _merge_stats(stats):
    """(mean, M2) partials [M, 8, 2] of equal-size (160) groups -> (mean, biased variance) per row, in float64."""
    st = stats.double()
    mean = st[:, :, 0].mean(1)
    m2 = st[:, :, 1].sum(1) + 160.0 * ((st[:, :, 0] - mean[:, None]) ** 2).sum(1)
    return mean, m2 / 1280.0


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_ln_fold_weight_and_rowstats(lib, name, prec, dt, ulp):
    """The two small kernels of the folded LayerNorm: weight preparation (ET(W diag(gamma)) bit-exact, its row sums, b + W beta)
    and the entry kernel (Xh = ET(X) bit-exact; merged row statistics vs float64, with row means up to 50 sigma)."""
    import ctypes
    g = torch.Generator().manual_seed(5)
    N, K = 640, 1280
    W = torch.randn(N, K, generator=g) / math.sqrt(K)
    gamma, beta, bias = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g), torch.randn(N, generator=g)
    Wf = torch.empty(N, K, dtype=torch.int16, device="cuda")
    cvec, bf = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    Wd, gd, bd, biasd = dev(W), dev(gamma), dev(beta), dev(bias)
    assert lib.samrs_k_ln_fold_weight(prec, Wd.data_ptr(), gd.data_ptr(), bd.data_ptr(), biasd.data_ptr(), Wf.data_ptr(),
                                      cvec.data_ptr(), bf.data_ptr(), N, K, stream()) == 0
    ref_wf = (W * gamma).to(dt)
    assert torch.equal(Wf.cpu(), ref_wf.view(torch.int16))
    assert (cvec.cpu().double() - ref_wf.double().sum(1)).abs().max().item() < 1e-6
    assert (bf.cpu().double() - (bias.double() + W.double() @ beta.double())).abs().max().item() < 1e-6
    rows = 1000
    X = torch.randn(rows, 1280, generator=g) * torch.logspace(-2, 2, rows)[:, None] + \
        torch.linspace(-50, 50, rows)[:, None] * torch.logspace(-2, 2, rows)[:, None]
    X[:, 7] *= 30                                  # a massive-activation channel
    Xd = dev(X)
    Xh = torch.empty(rows, 1280, dtype=torch.int16, device="cuda")
    stats = torch.empty(rows, 8, 2, device="cuda")
    assert lib.samrs_k_rowstats_convert(prec, Xd.data_ptr(), Xh.data_ptr(), stats.data_ptr(), rows, 1280, stream()) == 0
    refh = X.clamp(-65504, 65504).to(dt) if dt == torch.float16 else X.to(dt)
    assert torch.equal(Xh.cpu(), refh.view(torch.int16))
    mean, var = _merge_stats(stats.cpu())
    rm, rv = X.double().mean(1), X.double().var(1, unbiased=False)
    assert ((mean - rm).abs() / rv.sqrt()).max().item() < 1e-5
    assert ((var - rv).abs() / rv).max().item() < 1e-5


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_stats_epilogue(lib, name, prec, dt, ulp):
    """Producer of the folded LayerNorm (proj / lin2): C += A B^T + bias on the 32x32x16 kernel, plus Xh = ET(C) and the
    per-row partial statistics of C.  C must equal the plain fp32-residual GEMM, Xh must be ET(C) bit for bit, the merged
    statistics must match float64 statistics of C; repeated launches are bit-identical."""
    need_experiments(lib)
    for (M, K) in [(512, 128), (8192, 1280), (32768, 1280), (32768, 5120)]:
        N = 1280
        g = torch.Generator().manual_seed(M + K)
        _, Ab = et_bits(torch.randn(M, K, generator=g), dt)
        _, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
        bias = dev(torch.randn(N, generator=g))
        C0 = dev(torch.randn(M, N, generator=g) * 3 + torch.randn(M, 1, generator=g) * 4)      # residual with a row mean
        Ad, Bd = dev(Ab), dev(Bb)
        ref = C0.clone()
        assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), ref.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 1, 0, 1, stream()) == 0
        outs = []
        for rep in range(2):
            C = C0.clone()
            Xh = torch.full((M, N), 0x7E00, dtype=torch.int16, device="cuda")
            stats = torch.full((M, 8, 2), float("nan"), device="cuda")
            assert lib.samrs_k_gemm_stats(prec, Ad.data_ptr(), Bd.data_ptr(), C.data_ptr(), bias.data_ptr(), Xh.data_ptr(),
                                          stats.data_ptr(), M, N, K, stream()) == 0
            torch.cuda.synchronize()
            outs.append((C, Xh, stats))
        assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1])), f"{name} {M}x{N}x{K}: launches differ (race)"
        C, Xh, stats = outs[0]
        r = ((C - ref).double().norm() / ref.double().norm()).item()
        print(f"gemm_stats {name} {M}x{N}x{K}: C vs plain kernel rel {r:.2e}, bit-identical {torch.equal(C, ref)}")
        assert r < 2e-6
        want = C.clamp(-65504, 65504).to(dt) if dt == torch.float16 else C.to(dt)
        assert torch.equal(Xh, want.view(torch.int16)), f"{name} {M}x{N}x{K}: Xh != ET(C)"
        assert torch.isfinite(stats).all()
        mean, var = _merge_stats(stats)
        rm, rv = C.double().mean(1), C.double().var(1, unbiased=False)
        assert ((mean - rm).abs() / rv.sqrt()).max().item() < 1e-5
        assert ((var - rv).abs() / rv).max().item() < 1e-5


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("gelu", [0, 1])
def test_gemm_folded_layernorm(lib, name, prec, dt, ulp, gelu):
    """Consumer of the folded LayerNorm (qkv / lin1): rstd (Xh Wf^T - mean cvec) + bias_f [GELU] against float64
    LayerNorm + Linear [+ GELU], next to the engine's other path (stand-alone LayerNorm kernel -> ET -> GEMM) on the same data:
    same error class.  Rows carry means of several sigma and a massive-activation channel."""
    need_experiments(lib)
    for (M, N) in [(512, 640), (8192, 3840)]:
        K = 1280
        g = torch.Generator().manual_seed(M + N + gelu)
        X = torch.randn(M, K, generator=g) * (0.5 + torch.rand(M, 1, generator=g) * 4) + torch.randn(M, 1, generator=g) * 3
        X[:, 11] += 40
        W = torch.randn(N, K, generator=g) / math.sqrt(K)
        gamma, beta, bias = 1 + 0.3 * torch.randn(K, generator=g), 0.2 * torch.randn(K, generator=g), torch.randn(N, generator=g)
        Xd, Wd, gd, bd, biasd = dev(X), dev(W), dev(gamma), dev(beta), dev(bias)
        ref = F.layer_norm(X.double(), (K,), gamma.double(), beta.double(), 1e-6) @ W.double().t() + bias.double()
        if gelu:
            ref = F.gelu(ref)
        # folded path
        Xh = torch.empty(M, K, dtype=torch.int16, device="cuda")
        stats = torch.empty(M, 8, 2, device="cuda")
        Wf = torch.empty(N, K, dtype=torch.int16, device="cuda")
        cvec, bf = torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
        out = torch.full((M, N), 0x7E00, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_rowstats_convert(prec, Xd.data_ptr(), Xh.data_ptr(), stats.data_ptr(), M, K, stream()) == 0
        assert lib.samrs_k_ln_fold_weight(prec, Wd.data_ptr(), gd.data_ptr(), bd.data_ptr(), biasd.data_ptr(), Wf.data_ptr(),
                                          cvec.data_ptr(), bf.data_ptr(), N, K, stream()) == 0
        rowstat = torch.empty(M, 2, device="cuda")
        assert lib.samrs_k_ln_rowstat(stats.data_ptr(), rowstat.data_ptr(), M, 1e-6, stream()) == 0
        assert lib.samrs_k_gemm_fold(prec, Xh.data_ptr(), Wf.data_ptr(), out.data_ptr(), bf.data_ptr(), cvec.data_ptr(),
                                     rowstat.data_ptr(), M, N, K, gelu, stream()) == 0
        # stand-alone LayerNorm path
        Y = torch.empty(M, K, dtype=torch.int16, device="cuda")
        We = dev(W.to(dt).view(torch.int16))
        out0 = torch.empty(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_layernorm(prec, Xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-6, Y.data_ptr(), None, M, K, 0, 1, 64, 0, stream()) == 0
        assert lib.samrs_k_gemm(prec, Y.data_ptr(), We.data_ptr(), out0.data_ptr(), biasd.data_ptr(), None, 0, M, N, K, 0, gelu, 0, stream()) == 0
        torch.cuda.synchronize()
        got, got0 = out.cpu().view(dt).double(), out0.cpu().view(dt).double()
        assert torch.isfinite(got).all()
        e1 = ((got - ref).norm() / ref.norm()).item()
        e0 = ((got0 - ref).norm() / ref.norm()).item()
        print(f"folded LN {name} {M}x{N} gelu={gelu}: folded {e1:.3e}, stand-alone {e0:.3e} (rel. L2 vs float64)")
        assert e1 < 1.5 * e0 + 1e-5, (e1, e0)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("N", [256, 384])
def test_gemm_k256_streaming_kernel_is_bit_identical(lib, name, prec, dt, ulp, N):
    """The decoder's image-side projections (rows = prompts x 4096 keys, K = 256, 2-D addend with period 4096): the persistent
    weights-in-registers kernel (variant 40; automatic from 512 tiles of 128 rows) must reproduce the tiled 2-blocks-per-CU
    kernel (7) bit for bit -- more tiles than blocks (each block walks 3 tiles: both LDS buffers are reused), a ragged last
    round, repeated launches, with and without the addend -- and agree with fp64."""
    lib.samrs_debug_set_gemm_variant.argtypes = [__import__("ctypes").c_int]
    M, K, period = 128 * 700, 256, 4096
    g = torch.Generator().manual_seed(N)
    A, Ab = et_bits(torch.randn(M, K, generator=g), dt)
    B, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
    bias = torch.randn(N, generator=g)
    add = torch.randn(period, N, generator=g)
    Ad, Bd, biasd, addd = dev(Ab), dev(Bb), dev(bias), dev(add)
    try:
        for with_add in (True, False):
            outs = {}
            for variant in (7, 40, 40, 8):
                lib.samrs_debug_set_gemm_variant(variant)
                o = torch.full((M, N), 0x7E00, dtype=torch.int16, device="cuda")      # NaN canary
                assert lib.samrs_k_gemm(prec, Ad.data_ptr(), Bd.data_ptr(), o.data_ptr(), biasd.data_ptr(),
                                        addd.data_ptr() if with_add else None, period if with_add else 0, M, N, K, 0, 0, 0, stream()) == 0
                torch.cuda.synchronize()
                if not outs:
                    outs["ref"] = o
                else:
                    assert torch.equal(o, outs["ref"]), f"{name} N={N} add2d={with_add}: variant {variant} differs from the tiled kernel"
            ref = A.double() @ B.double().t() + bias.double()
            if with_add:
                ref = ref + add.double()[torch.arange(M) % period]
            got = outs["ref"].cpu().view(dt).double()
            r, _ = rel_err(got, ref)
            assert r < (1e-3 if name == "f16" else 8e-3), r
    finally:
        lib.samrs_debug_set_gemm_variant(8)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_group_layernorm_gelu_epilogue(lib, name, prec, dt, ulp):
    """ConvT #1 of the mask upscaler: GEMM + LayerNorm2d over each 64-channel group + GELU (mask_decoder.py:53-56)."""
    g = torch.Generator().manual_seed(31)
    M, N, K = 512, 256, 256
    A, Ab = et_bits(torch.randn(M, K, generator=g), dt)
    B, Bb = et_bits(torch.randn(N, K, generator=g) / math.sqrt(K), dt)
    bias = torch.randn(N, generator=g) * 0.5
    gamma = 1 + 0.2 * torch.randn(64, generator=g)
    beta = 0.2 * torch.randn(64, generator=g)
    y = (A.double() @ B.double().t() + bias.double()).view(M, N // 64, 64)
    mu = y.mean(-1, keepdim=True)
    var = ((y - mu) ** 2).mean(-1, keepdim=True)
    ref = F.gelu((((y - mu) / torch.sqrt(var + 1e-6)) * gamma.double() + beta.double()).float()).view(M, N)
    out = torch.empty(M, N, dtype=torch.int16, device="cuda")
    gb = dev(torch.cat([gamma, beta]))
    assert lib.samrs_k_gemm_gln(prec, dev(Ab).data_ptr(), dev(Bb).data_ptr(), out.data_ptr(), dev(bias).data_ptr(), gb.data_ptr(),
                                M, N, K, None, None, stream()) == 0
    err = (out.cpu().view(dt).float() - ref).abs() / ref.abs().clamp(min=1e-2)
    print(f"gemm+groupLN+gelu {name}: max rel {err.max().item():.2e}")
    assert err.max().item() < 1.5 * ulp


def split_bits(lib, prec, x):
    """fp32 CPU tensor -> (hi bits, lo bits) device int16 tensors through samrs_k_convert_split, and the fp64 value hi + lo."""
    xd = dev(x)
    hi = torch.empty(x.shape, dtype=torch.int16, device="cuda")
    lo = torch.empty(x.shape, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_convert_split(prec, xd.data_ptr(), hi.data_ptr(), lo.data_ptr(), x.numel(), stream()) == 0
    return hi, lo


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_convert_split_is_the_two_term_split(lib, name, prec, dt, ulp):
    """hi = ET(x) (bit-identical with samrs_k_convert), lo = ET(x - hi); hi + lo reproduces x to ~ulp^2."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1 << 16, generator=g) * torch.logspace(-2, 2, 1 << 16)
    hi, lo = split_bits(lib, prec, x)
    h = x.to(dt)
    assert torch.equal(hi.cpu(), h.view(torch.int16))
    l = (x - h.float()).to(dt)
    assert torch.equal(lo.cpu(), l.view(torch.int16))
    rec = h.double() + l.double()
    # |x - hi| <= ulp/2 |x| and lo rounds that remainder to ulp/2 of ITSELF -- unless it falls into the operand type's
    # subnormal range (f16: steps of 2^-24), where the error is absolute
    excess = ((rec - x.double()).abs() - 2.0 ** -24).clamp(min=0) / x.double().abs()
    print(f"split {name}: max rel reconstruction error beyond one subnormal step {excess.max().item():.2e}")
    assert excess.max().item() < ulp * ulp


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_gln_split_precision(lib, name, prec, dt, ulp):
    """ConvT #1 with hi + lo operands (A_lo B + A B_lo + A B, fp32 output after LayerNorm2d + GELU) against the fp64 product of the
    UN-rounded fp32 operands: the error must be of the fp32-accumulation class, far below one operand ulp."""
    g = torch.Generator().manual_seed(32)
    M, N, K = 512, 256, 256
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.5
    gamma = 1 + 0.2 * torch.randn(64, generator=g)
    beta = 0.2 * torch.randn(64, generator=g)
    y = (A.double() @ B.double().t() + bias.double()).view(M, N // 64, 64)
    mu = y.mean(-1, keepdim=True)
    var = ((y - mu) ** 2).mean(-1, keepdim=True)
    ref = F.gelu(((y - mu) / torch.sqrt(var + 1e-6)) * gamma.double() + beta.double()).view(M, N)
    Ah, Al = split_bits(lib, prec, A)
    Bh, Bl = split_bits(lib, prec, B)
    out = torch.full((M, N), float("nan"), device="cuda")
    gb = dev(torch.cat([gamma, beta]))
    assert lib.samrs_k_gemm_gln(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), gb.data_ptr(),
                                M, N, K, Al.data_ptr(), Bl.data_ptr(), stream()) == 0
    err = (out.cpu().double() - ref).abs() / ref.abs().clamp(min=1e-2)
    print(f"split gemm+groupLN+gelu {name}: max rel {err.max().item():.2e} (one operand ulp = {ulp:.1e})")
    assert not torch.isnan(out).any()
    assert err.max().item() < (1e-4 if name == "f16" else 2e-3)        # an order of magnitude under one operand ulp


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("M,N,K,out_f32", [(256, 320, 64, 1), (512, 640, 192, 0), (512, 1280, 1280, 1), (768, 3840, 1280, 0),
                                           (512, 256, 2304, 1), (256, 768, 768, 1)])      # N % 256 only: the neck / ViT-B patch embed (fp32 out)
def test_gemm_split3_one_launch(lib, name, prec, dt, ulp, M, N, K, out_f32):
    """The block GEMMs of the reference-grade mode as ONE launch over a three-segment K axis (A_lo B + A B_lo + A B in the
    register accumulators): against the fp64 product of the un-rounded fp32 operands, far below one operand ulp; ET output =
    that result rounded once; fp32 output accumulates into the residual stream (image_encoder.py:175,182)."""
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.5
    ref = A.double() @ B.double().t() + bias.double()
    Ah, Al = split_bits(lib, prec, A)
    Bh, Bl = split_bits(lib, prec, B)
    if out_f32:
        res = torch.randn(M, N, generator=g)
        out = dev(res.clone())
        assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), out.data_ptr(),
                                       dev(bias).data_ptr(), M, N, K, 1, 1, 0, stream()) == 0
        got = out.cpu().double() - res.double()
        tol = 2e-5 if name == "f16" else 1e-3
    else:
        out = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), out.data_ptr(),
                                       dev(bias).data_ptr(), M, N, K, 0, 0, 0, stream()) == 0
        got = out.cpu().view(dt).double()
        tol = 0.51 * ulp                                                # one rounding of the fp32 result to the operand type
    err = (got - ref).abs() / ref.abs().clamp(min=1.0)
    print(f"split3 {name} {M}x{N}x{K} out_f32={out_f32}: max err {err.max().item():.2e} (operand ulp {ulp:.1e})")
    assert err.max().item() < tol
    # against the plain product of the rounded operands the difference must be visible: the lo terms are really there
    plain = A.to(dt).double() @ B.to(dt).double().t() + bias.double()
    assert ((plain - ref).abs().max() > 4 * (got - ref).abs().max()) or not out_f32
    assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), out.data_ptr(),
                                   dev(bias).data_ptr(), M, N + 32, K, out_f32, 0, 0, stream()) != 0      # shape outside the tiles: refused


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_split3_column_mask(lib, name, prec, dt, ulp):
    """split_from_n: the tiles in front of it are the plain hi x hi product (bit-identical with samrs_k_gemm on the hi operands),
    the tiles from it on carry the lo terms (bit-identical with the unmasked launch) -- qkv with only its v third split."""
    g = torch.Generator().manual_seed(11)
    M, N, K = 8192, 3840, 128                      # 32 x 12 = 384 tiles: the persistent blocks walk light and heavy tiles in turn
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) / math.sqrt(K)
    bias = dev(torch.randn(N, generator=g) * 0.5)
    Ah, Al = split_bits(lib, prec, A)
    Bh, Bl = split_bits(lib, prec, B)
    outs = {}
    for from_n in (0, 2560, 3840 - 320):
        out = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), out.data_ptr(),
                                       bias.data_ptr(), M, N, K, 0, 0, from_n, stream()) == 0
        outs[from_n] = out.cpu()
    plain = torch.zeros(M, N, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_gemm(prec, Ah.data_ptr(), Bh.data_ptr(), plain.data_ptr(), bias.data_ptr(), None, 0, M, N, K, 0, 0, 0, stream()) == 0
    plain = plain.cpu()
    for from_n in (2560, 3840 - 320):
        assert torch.equal(outs[from_n][:, :from_n], plain[:, :from_n])
        assert torch.equal(outs[from_n][:, from_n:], outs[0][:, from_n:])
    assert not torch.equal(outs[0][:, :2560], plain[:, :2560])
    # race screen: the persistent blocks hand ring buffers from tile to tile (stage 0 of the next tile lands under the epilogue);
    # a missed wait shows up as run-to-run differences
    for _ in range(12):
        again = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), again.data_ptr(),
                                       bias.data_ptr(), M, N, K, 0, 0, 2560, stream()) == 0
        assert torch.equal(again.cpu(), outs[2560])
    assert lib.samrs_k_gemm_split3(prec, Ah.data_ptr(), Al.data_ptr(), Bh.data_ptr(), Bl.data_ptr(), plain.data_ptr(),
                                   bias.data_ptr(), M, N, K, 0, 0, 100, stream()) != 0          # not a whole tile


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("n_sel,sel0", [(1, 0), (3, 1)])
def test_upscale2_masks_fused(lib, name, prec, dt, ulp, n_sel, sel0):
    """ConvT #2 (as a K = 64 GEMM) + GELU + hypernetwork dot, fused (mask_decoder.py:57-59,154-167)."""
    g = torch.Generator().manual_seed(77 + n_sel)
    n, grid = 3, 16
    rows = n * grid * grid * 4
    U, Ub = et_bits(torch.randn(rows, 64, generator=g), dt)
    Wt, Wb = et_bits(torch.randn(128, 64, generator=g) / 8, dt)
    bias = 0.3 * torch.randn(128, generator=g)
    hyper = torch.randn(n, 4, 32, generator=g)
    up2 = F.gelu((U.double() @ Wt.double().t() + bias.double()).float()).double()        # [rows][sub2*32 + c]
    up2 = up2.view(n, grid, grid, 2, 2, 2, 2, 32)                                          # b, y, x, dy, dx, dy2, dx2, c
    S = 4 * grid
    ref = torch.einsum("byxijklc,bsc->bsyikxjl", up2, hyper[:, sel0:sel0 + n_sel].double()).reshape(n, n_sel, S, S)
    low = torch.full((n, n_sel, S, S), float("nan"), device="cuda")
    assert lib.samrs_k_upscale2_masks(prec, dev(Ub).data_ptr(), dev(Wb).data_ptr(), None, dev(bias).data_ptr(), dev(hyper).data_ptr(),
                                      low.data_ptr(), n, grid, 4, sel0, n_sel, stream()) == 0
    r, mx = rel_err(low.cpu(), ref)
    print(f"upscale2+mask {name} n_sel={n_sel}: rel {r:.2e} max {mx:.2e}")
    assert not torch.isnan(low).any(), "some low-res pixels were never written"
    assert r < 2e-5
    # split precision: fp32 activations (split in registers) x hi + lo weights, against the fp64 product of the UN-rounded operands
    Uf = torch.randn(rows, 64, generator=g)
    Wf = torch.randn(128, 64, generator=g) / 8
    up2 = F.gelu(Uf.double() @ Wf.double().t() + bias.double()).view(n, grid, grid, 2, 2, 2, 2, 32)
    ref = torch.einsum("byxijklc,bsc->bsyikxjl", up2, hyper[:, sel0:sel0 + n_sel].double()).reshape(n, n_sel, S, S)
    Wh, Wl = split_bits(lib, prec, Wf)
    low = torch.full((n, n_sel, S, S), float("nan"), device="cuda")
    assert lib.samrs_k_upscale2_masks(prec, dev(Uf).data_ptr(), Wh.data_ptr(), Wl.data_ptr(), dev(bias).data_ptr(), dev(hyper).data_ptr(),
                                      low.data_ptr(), n, grid, 4, sel0, n_sel, stream()) == 0
    r, mx = rel_err(low.cpu(), ref)
    print(f"upscale2+mask split {name} n_sel={n_sel}: rel {r:.2e} max {mx:.2e}")
    assert not torch.isnan(low).any()
    assert r < (3e-6 if name == "f16" else 1e-4)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("n_sel,sel0", [(1, 0), (3, 1)])
@pytest.mark.parametrize("split", [False, True])
def test_upscaler_fused_one_kernel(lib, name, prec, dt, ulp, n_sel, sel0, split):
    """mask_decoder.py:53-59,154-167 as ONE kernel: ConvT #1 -> LayerNorm2d(64) -> GELU -> ConvT #2 -> GELU -> hypernetwork dot,
    against the same chain in fp64 (torch conv_transpose2d semantics restated as the two per-token GEMMs).  Un-split: operands
    pre-rounded on the host, what remains is the rounding of the ConvT #1 activations to the operand type; split: the UN-rounded
    operands, error an order of magnitude below one operand ulp."""
    g = torch.Generator().manual_seed(200 + n_sel + 10 * split)
    n, grid = 3, 16
    tokens = grid * grid
    keys = torch.randn(n * tokens, 256, generator=g)
    w1 = torch.randn(256, 256, generator=g) / 16
    b1 = 0.3 * torch.randn(256, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(64, generator=g), 0.2 * torch.randn(64, generator=g)
    w2 = torch.randn(128, 64, generator=g) / 8
    b2 = 0.3 * torch.randn(128, generator=g)
    hyper = torch.randn(n, 4, 32, generator=g)
    if not split:
        keys, w1, w2 = keys.to(dt).float(), w1.to(dt).float(), w2.to(dt).float()
    y = (keys.double() @ w1.double().t() + b1.double()).view(-1, 4, 64)                        # (token, s1, c)
    mu = y.mean(-1, keepdim=True)
    u1 = F.gelu((y - mu) / torch.sqrt(((y - mu) ** 2).mean(-1, keepdim=True) + 1e-6) * gamma.double() + beta.double())
    u1r = u1 if split else u1.float().to(dt).double()                                             # the un-split kernel rounds here
    up2 = F.gelu(u1r.reshape(-1, 64) @ w2.double().t() + b2.double())                            # rows (token, s1), cols (s2, c2)
    up2 = up2.view(n, grid, grid, 2, 2, 2, 2, 32)                                                # b, y, x, dy, dx, dy2, dx2, c
    S = 4 * grid
    ref = torch.einsum("byxijklc,bsc->bsyikxjl", up2, hyper[:, sel0:sel0 + n_sel].double()).reshape(n, n_sel, S, S)
    low = torch.full((n, n_sel, S, S), float("nan"), device="cuda")
    if split:
        kh, kl = split_bits(lib, prec, keys)
        w1h, w1l = split_bits(lib, prec, w1)
        w2h, w2l = split_bits(lib, prec, w2)
        ptrs = (kh.data_ptr(), kl.data_ptr(), w1h.data_ptr(), w1l.data_ptr(), w2h.data_ptr(), w2l.data_ptr())
    else:
        kh, w1h, w2h = dev(keys.to(dt).view(torch.int16)), dev(w1.to(dt).view(torch.int16)), dev(w2.to(dt).view(torch.int16))
        ptrs = (kh.data_ptr(), None, w1h.data_ptr(), None, w2h.data_ptr(), None)
    gb = dev(torch.cat([gamma, beta]))
    assert lib.samrs_k_upscaler_fused(prec, ptrs[0], ptrs[1], ptrs[2], ptrs[3], dev(b1).data_ptr(), gb.data_ptr(), ptrs[4], ptrs[5],
                                      dev(b2).data_ptr(), dev(hyper).data_ptr(), low.data_ptr(), n, grid, 4, sel0, n_sel, stream()) == 0
    torch.cuda.synchronize()
    assert not torch.isnan(low).any(), "some low-res pixels were never written"
    r, mx = rel_err(low.cpu(), ref)
    print(f"fused upscaler {name} split={split} n_sel={n_sel}: rel {r:.2e} max {mx:.2e}")
    tol = {("f16", False): 4e-4, ("f16", True): 1e-5, ("bf16", False): 3e-3, ("bf16", True): 2e-4}[(name, split)]
    assert r < tol


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_neck_kernel_level(lib, name, prec, dt, ulp):
    """The neck alone (image_encoder.py:88-104, LayerNorm2d common.py:31-43): 1x1 conv as a GEMM -> LayerNorm over C (eps 1e-6)
    -> im2col + GEMM for the bias-free 3x3 / pad 1 conv -> LayerNorm, against torch conv2d in fp64 on the same
    ET-rounded operands.  The im2col itself is pure data movement: bit-exact against F.unfold."""
    g = torch.Generator().manual_seed(91)
    n, grid, D, C = 2, 16, 256, 256
    X, Xb = et_bits(torch.randn(n * grid * grid, D, generator=g), dt)
    W0, W0b = et_bits(torch.randn(C, D, generator=g) / math.sqrt(D), dt)
    W2 = torch.randn(C, C, 3, 3, generator=g) / math.sqrt(9 * C)
    W2r = W2.permute(0, 2, 3, 1).reshape(C, 9 * C).contiguous()              # [co][(ky*3+kx)*C + ci], the engine's repack
    W2e, W2b = et_bits(W2r, dt)
    g1, b1 = 1 + 0.1 * torch.randn(C, generator=g), 0.05 * torch.randn(C, generator=g)
    g2, b2 = 1 + 0.1 * torch.randn(C, generator=g), 0.05 * torch.randn(C, generator=g)
    M = n * grid * grid
    s = stream()
    n1 = torch.empty(M, C, device="cuda")
    assert lib.samrs_k_gemm(prec, dev(Xb).data_ptr(), dev(W0b).data_ptr(), n1.data_ptr(), None, None, 0, M, C, D, 1, 0, 0, s) == 0
    n1e = torch.empty(M, C, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_layernorm(prec, n1.data_ptr(), dev(g1).data_ptr(), dev(b1).data_ptr(), 1e-6, n1e.data_ptr(), None, M, C, 0, n, grid, 0, s) == 0
    col = torch.empty(M, 9 * C, dtype=torch.int16, device="cuda")
    assert lib.samrs_k_neck_im2col(n1e.data_ptr(), col.data_ptr(), n, grid, C, s) == 0
    # im2col: exact
    img = n1e.cpu().view(n, grid, grid, C)
    ref_col = torch.zeros(n, grid, grid, 9, C, dtype=torch.int16)
    for ky in range(3):
        for kx in range(3):
            ys, xs = slice(max(0, 1 - ky), grid - max(0, ky - 1)), slice(max(0, 1 - kx), grid - max(0, kx - 1))
            yd, xd = slice(max(0, ky - 1), grid - max(0, 1 - ky)), slice(max(0, kx - 1), grid - max(0, 1 - kx))
            ref_col[:, ys, xs, ky * 3 + kx] = img[:, yd, xd]
    assert torch.equal(col.cpu().view(n, grid, grid, 9, C), ref_col), "neck im2col is not the 3x3 / pad 1 gather"
    n2 = torch.empty(M, C, device="cuda")
    assert lib.samrs_k_gemm(prec, col.data_ptr(), dev(W2b).data_ptr(), n2.data_ptr(), None, None, 0, M, C, 9 * C, 1, 0, 0, s) == 0
    out = torch.empty(M, C, device="cuda")
    assert lib.samrs_k_layernorm(prec, n2.data_ptr(), dev(g2).data_ptr(), dev(b2).data_ptr(), 1e-6, None, out.data_ptr(), M, C, 0, n, grid, 0, s) == 0
    # reference: same rounding points (ET operands into each conv), everything else fp64
    x = X.double().view(n, grid, grid, D).permute(0, 3, 1, 2)
    y = F.conv2d(x, W0.double().view(C, D, 1, 1))
    def ln2d(t, w, b):
        u = t.mean(1, keepdim=True)
        v = ((t - u) ** 2).mean(1, keepdim=True)
        return (t - u) / torch.sqrt(v + 1e-6) * w.double().view(1, -1, 1, 1) + b.double().view(1, -1, 1, 1)
    y = ln2d(y, g1, b1)
    y_et = n1e.cpu().view(dt).double().view(n, grid, grid, C).permute(0, 3, 1, 2)        # the engine's rounded LN output
    r1, _ = rel_err(y_et.float(), y)
    assert r1 < 1.5 * ulp, f"neck stage 1 (1x1 conv + LayerNorm2d): rel {r1:.2e}"
    z = F.conv2d(y_et, W2e.double().view(C, 3, 3, C).permute(0, 3, 1, 2), padding=1)
    z = ln2d(z, g2, b2).permute(0, 2, 3, 1).reshape(M, C)
    r2, mx = rel_err(out.cpu(), z)
    print(f"neck {name}: stage 1 rel {r1:.2e}, output rel {r2:.2e} max {mx:.2e}")
    assert r2 < 5e-6


def test_gemm_f32_exact_class(lib):
    g = torch.Generator().manual_seed(5)
    for (M, N, K, lda_pad, relu, acc) in [(7, 32, 256, 0, 0, 0), (224, 2048, 256, 0, 1, 0), (224, 256, 2048, 0, 0, 1),
                                          (32, 4, 256, 1792 - 256, 0, 0), (4096, 128, 256, 0, 0, 0), (100, 70, 128, 0, 1, 1),
                                          (33, 40, 48, 4, 0, 0), (224, 128, 256, 0, 0, 0)]:
        lda = K + lda_pad
        A = torch.randn(M, lda, generator=g)
        Wt = torch.randn(N, K, generator=g) / math.sqrt(K)
        b = torch.randn(N, generator=g)
        C0 = torch.randn(M, N + 8, generator=g)
        ref = A[:, :K].double() @ Wt.double().t() + b.double()
        if relu:
            ref = ref.clamp(min=0)
        if acc:
            ref = ref + C0[:, :N].double()
        Ad, Wd, bd, Cd = dev(A), dev(Wt), dev(b), dev(C0.clone())
        assert lib.samrs_k_gemm_f32(Ad.data_ptr(), lda, Wd.data_ptr(), bd.data_ptr(), Cd.data_ptr(), N + 8, M, N, K, relu, acc, stream()) == 0
        out = Cd.cpu()
        r, mx = rel_err(out[:, :N], ref)
        print(f"gemm_f32 {M}x{N}x{K} lda={lda} relu={relu} acc={acc}: rel {r:.2e} max {mx:.2e}")
        assert r < 3e-6
        assert torch.equal(out[:, N:], C0[:, N:]), "wrote outside the [M, N] window"
        # K is split over the waves of a block and combined in a fixed order: bit-reproducible
        Cd2 = dev(C0.clone())
        assert lib.samrs_k_gemm_f32(Ad.data_ptr(), lda, Wd.data_ptr(), bd.data_ptr(), Cd2.data_ptr(), N + 8, M, N, K, relu, acc, stream()) == 0
        assert torch.equal(Cd2.cpu(), out)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_layernorm_plain_and_window(lib, name, prec, dt, ulp):
    g = torch.Generator().manual_seed(9)
    for D in (128, 256, 768, 1280):
        n_img, grid, win = 2, 64, 14
        X = torch.randn(n_img * grid * grid, D, generator=g) * 3 + 0.5
        gam = 1 + 0.1 * torch.randn(D, generator=g)
        bet = 0.1 * torch.randn(D, generator=g)
        ref = F.layer_norm(X, (D,), gam, bet, eps=1e-6)
        Xd, gd, bd = dev(X), dev(gam), dev(bet)
        # plain: fp32 + ET outputs
        oe = torch.empty(X.shape, dtype=torch.int16, device="cuda")
        of = torch.empty(X.shape, device="cuda")
        assert lib.samrs_k_layernorm(prec, Xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-6, oe.data_ptr(), of.data_ptr(),
                                     X.shape[0], D, 0, n_img, grid, win, stream()) == 0
        r, mx = rel_err(of.cpu(), ref)
        assert mx < 2e-5, (D, mx)
        assert (oe.cpu().view(dt).float() - ref).abs().max().item() < 8 * ulp
        # window gather (image_encoder.py:243-264): 25 windows of 14x14, zero rows in the padding
        nw = 5
        rows = n_img * nw * nw * win * win
        oe = torch.empty(rows, D, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_layernorm(prec, Xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-6, oe.data_ptr(), None,
                                     rows, D, 1, n_img, grid, win, stream()) == 0
        y = F.pad(ref.view(n_img, grid, grid, D), (0, 0, 0, 6, 0, 6))
        y = y.view(n_img, nw, win, nw, win, D).permute(0, 1, 3, 2, 4, 5).reshape(rows, D)
        got = oe.cpu().view(dt).float()
        assert (got - y).abs().max().item() < 8 * ulp
        assert (got[y.abs().sum(1) == 0] == 0).all()


def _attention_ref(qkv, rel_h, rel_w, heads, S):
    """fp32 attention on [B, S*S, 3D] (already rounded) with decomposed rel-pos
    (image_encoder.py:224-240,325-361)."""
    B, N, D3 = qkv.shape
    D = D3 // 3
    d = D // heads
    t = qkv.reshape(B, N, 3, heads, d).permute(2, 0, 3, 1, 4).reshape(3, B * heads, N, d).double()
    q, k, v = t[0], t[1], t[2]
    attn = (q * d ** -0.5) @ k.transpose(1, 2)
    idx = torch.arange(S)[:, None] - torch.arange(S)[None, :] + (S - 1)
    Rh, Rw = rel_h.double()[idx], rel_w.double()[idx]
    rq = q.reshape(B * heads, S, S, d)
    rh = torch.einsum("bhwc,hkc->bhwk", rq, Rh)
    rw = torch.einsum("bhwc,wkc->bhwk", rq, Rw)
    attn = (attn.view(-1, S, S, S, S) + rh[..., :, None] + rw[..., None, :]).view(-1, N, N)
    o = attn.softmax(-1) @ v
    return o.view(B, heads, N, d).permute(0, 2, 1, 3).reshape(B, N, D).float()


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("hd,heads", [(64, 2), (80, 2)])
def test_window_attention(lib, name, prec, dt, ulp, hd, heads):
    """Token-order qkv in, un-partitioned output; padding positions of the 14x14 windows are zero
    tokens in the reference (pad AFTER norm1), i.e. their q/k/v equal the qkv bias."""
    g = torch.Generator().manual_seed(hd)
    n_img, grid, win, nw = 2, 64, 14, 5
    D = hd * heads
    qkv, qkvb = et_bits(torch.randn(n_img, grid, grid, 3 * D, generator=g), dt)
    bias, _ = et_bits(0.5 * torch.randn(3 * D, generator=g), dt)       # pre-rounded: the GEMM epilogue rounds too
    rel_h, _ = et_bits(0.3 * torch.randn(2 * win - 1, hd, generator=g), dt)
    rel_w, _ = et_bits(0.3 * torch.randn(2 * win - 1, hd, generator=g), dt)
    # reference: pad with the bias rows, partition (image_encoder.py:243-264), attend, un-partition + crop
    padded = bias.view(1, 1, 1, -1).expand(n_img, nw * win, nw * win, 3 * D).clone()
    padded[:, :grid, :grid] = qkv
    xw = padded.view(n_img, nw, win, nw, win, 3 * D).permute(0, 1, 3, 2, 4, 5).reshape(n_img * nw * nw, win * win, 3 * D)
    ref_w = _attention_ref(xw, rel_h, rel_w, heads, win)
    ref = ref_w.view(n_img, nw, nw, win, win, D).permute(0, 1, 3, 2, 4, 5).reshape(n_img, nw * win, nw * win, D)[:, :grid, :grid]
    out = torch.full((n_img * grid * grid, D), 0x7E00, dtype=torch.int16, device="cuda")  # NaN canary
    qd, bd, rhd, rwd = dev(qkvb), dev(bias), dev(rel_h), dev(rel_w)
    assert lib.samrs_k_window_attention(prec, qd.data_ptr(), bd.data_ptr(), rhd.data_ptr(), rwd.data_ptr(), out.data_ptr(),
                                        n_img, grid, win, heads, hd, stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(dt).float().view(n_img, grid, grid, D)
    assert torch.isfinite(got).all(), "some output rows were never written"
    r, mx = rel_err(got, ref)
    print(f"window attention {name} hd={hd}: rel {r:.2e} max {mx:.2e}")
    assert r < (3e-3 if name == "f16" else 2e-2)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("hd,heads", [(64, 2), (80, 1)])
def test_global_attention(lib, name, prec, dt, ulp, hd, heads):
    g = torch.Generator().manual_seed(100 + hd)
    n_img, grid = 1, 64
    D = hd * heads
    qkv, qkvb = et_bits(torch.randn(n_img, grid * grid, 3 * D, generator=g), dt)
    rel_h, _ = et_bits(0.3 * torch.randn(2 * grid - 1, hd, generator=g), dt)
    rel_w, _ = et_bits(0.3 * torch.randn(2 * grid - 1, hd, generator=g), dt)
    ref = _attention_ref(qkv, rel_h, rel_w, heads, grid)
    out = torch.full((n_img * grid * grid, D), 0x7E00, dtype=torch.int16, device="cuda")
    qd, rhd, rwd = dev(qkvb), dev(rel_h), dev(rel_w)
    assert lib.samrs_k_global_attention(prec, qd.data_ptr(), rhd.data_ptr(), rwd.data_ptr(), out.data_ptr(), n_img, grid,
                                        heads, hd, stream()) == 0
    torch.cuda.synchronize()
    got = out.cpu().view(dt).float().view(n_img, grid * grid, D)
    assert torch.isfinite(got).all()
    r, mx = rel_err(got, ref)
    print(f"global attention {name} hd={hd}: rel {r:.2e} max {mx:.2e}")
    assert r < (3e-3 if name == "f16" else 2e-2)


def test_global_attention_online_softmax_rescale(lib):
    """Force the running-max rescale branch: one key late in the sequence dominates."""
    hd, heads, grid, dt, prec = 64, 1, 64, torch.float16, 1
    g = torch.Generator().manual_seed(77)
    x = torch.randn(1, grid * grid, 3 * hd, generator=g)
    x[0, 3000, hd:2 * hd] *= 12.0            # a spiky key far into the stream
    x[0, 100, hd:2 * hd] *= 8.0
    qkv, qkvb = et_bits(x, dt)
    rel_h = torch.zeros(2 * grid - 1, hd)
    rel_w = torch.zeros(2 * grid - 1, hd)
    ref = _attention_ref(qkv, rel_h, rel_w, heads, grid)
    out = torch.empty(grid * grid, hd, dtype=torch.int16, device="cuda")
    qd, rhd, rwd = dev(qkvb), dev(rel_h), dev(rel_w)
    assert lib.samrs_k_global_attention(prec, qd.data_ptr(), rhd.data_ptr(), rwd.data_ptr(), out.data_ptr(), 1, grid, heads, hd, stream()) == 0
    got = out.cpu().view(dt).float().view(1, grid * grid, hd)
    r, mx = rel_err(got, ref)
    print(f"global attention spiky keys: rel {r:.2e} max {mx:.2e}")
    assert r < 3e-3


@pytest.mark.parametrize("in_hw,orig_hw", [((1024, 1024), (1024, 1024)), ((768, 1024), (600, 800)), ((1024, 683), (1500, 1000))])
def test_postprocess_matches_interpolate(lib, in_hw, orig_hw):
    g = torch.Generator().manual_seed(3)
    low = torch.randn(3, 256, 256, generator=g)
    ref = F.interpolate(low[None], (1024, 1024), mode="bilinear", align_corners=False)
    ref = ref[..., : in_hw[0], : in_hw[1]]
    ref = F.interpolate(ref, orig_hw, mode="bilinear", align_corners=False)[0]
    lowd = dev(low)
    outf = torch.empty(3, *orig_hw, device="cuda")
    assert lib.samrs_k_postprocess(lowd.data_ptr(), 3, in_hw[0], in_hw[1], orig_hw[0], orig_hw[1], 1024, 1, outf.data_ptr(), stream()) == 0
    d = (outf.cpu() - ref).abs().max().item()
    print(f"postprocess {in_hw}->{orig_hw}: logits max abs diff {d:.2e}")
    assert d < 1e-5
    outm = torch.empty(3, *orig_hw, dtype=torch.uint8, device="cuda")
    assert lib.samrs_k_postprocess(lowd.data_ptr(), 3, in_hw[0], in_hw[1], orig_hw[0], orig_hw[1], 1024, 0, outm.data_ptr(), stream()) == 0
    mism = (outm.cpu().bool() != (ref > 0)).sum().item()
    assert mism <= 3, mism      # only pixels whose logit is within fp32 noise of zero


@pytest.mark.parametrize("hw", [(600, 800), (800, 800), (1500, 1000), (2048, 1365), (1024, 1024)])
def test_device_resize_is_bit_exact_with_pil(lib, hw):
    """N3: ResizeLongestSide.apply_image on the GPU == PIL BILINEAR (utils/transforms.py:26-31), integer work."""
    from PIL import Image
    from samrs_amd import synth
    from samrs_amd.transforms import ResizeLongestSide
    t = ResizeLongestSide(1024)
    img = synth.make_image(3, *hw)
    nh, nw = t.get_preprocess_shape(hw[0], hw[1], 1024)
    ref = np.array(Image.fromarray(img).resize((nw, nh), Image.BILINEAR)) if (nh, nw) != hw else img
    got = t.apply_image_device(torch.as_tensor(img).cuda()).cpu().numpy()
    assert got.shape == ref.shape and np.array_equal(got, ref)


# ------------------------------------------------------------------------------------------------------------------------------
# round 4: the operand split's correction terms on MXFP4 operands (gemm_et_mx_kernel, mx4_pack_kernel, LayerNorm's MX outputs)
# ------------------------------------------------------------------------------------------------------------------------------
def _mx_pad(x, G, GP):
    """[rows][K] -> [rows][K / G * GP]: every group of G elements followed by GP - G zeros (the per-head padded K axis)."""
    rows, K = x.shape
    out = torch.zeros(rows, K // G, GP, dtype=x.dtype)
    out[:, :, :G] = x.reshape(rows, K // G, G)
    return out.reshape(rows, K // G * GP)


def _mx_pack(lib, prec, x, G, GP, is_b):
    """samrs_k_mx4_pack on fp32 x: (hi ET bits [rows][K], q_hi, q_lo [rows][Kp / 2] u8, s_hi, s_lo scale tiles) on the device."""
    rows, K = x.shape
    Kp = K // G * GP
    hi = torch.zeros(rows, K, dtype=torch.int16, device="cuda")
    q = [torch.zeros(rows, Kp // 2, dtype=torch.uint8, device="cuda") for _ in range(2)]
    nb = int(lib.samrs_k_mx_scale_bytes(rows, Kp, int(is_b)))
    sc = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(2)]
    xd = dev(x)
    assert lib.samrs_k_mx4_pack(prec, xd.data_ptr(), None, None, hi.data_ptr(), q[0].data_ptr(), q[1].data_ptr(), sc[0].data_ptr(),
                                sc[1].data_ptr(), rows, K, G, GP, int(is_b), stream()) == 0
    return hi, q, sc


def _mx_decode(q, s, rows, Kp, is_b):
    """Host inverse of the packed format (codes + tiled scales -> fp64 values [rows][Kp]): pins the data AND the scale-tile layout."""
    E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], dtype=torch.float64)
    qb = q.cpu().to(torch.int64)
    nib = torch.stack([qb & 15, qb >> 4], dim=-1).reshape(rows, Kp)              # element 2 i = low nibble of byte i
    val = E2M1[nib & 7] * torch.where((nib & 8) != 0, -1.0, 1.0)
    sb = s.cpu().to(torch.int64)
    r = torch.arange(rows)[:, None]
    b = torch.arange(Kp // 32)[None, :]
    st, kh, fq = b >> 3, (b >> 2) & 1, b & 3
    nst4 = Kp // 256
    if not is_b:
        idx = ((r >> 8) * nst4 + st) * 2048 + ((kh * 2 + ((r >> 7) & 1)) * 64 + fq * 16 + (r & 15)) * 8 + ((r >> 4) & 7)
    else:
        rr = r % 320
        idx = ((r // 320) * nst4 + st) * 4096 + ((kh * 4 + rr // 80) * 64 + fq * 16 + (rr & 15)) * 8 + ((rr % 80) >> 4)
    scale = torch.pow(2.0, (sb[idx] - 127).double())                            # [rows][Kp / 32]
    return val * scale.repeat_interleave(32, dim=1)


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("G,GP", [(1280, 1280), (80, 96)])
def test_mx4_pack_matches_the_oracle_quantiser(lib, name, prec, dt, ulp, G, GP):
    """mx4_pack_kernel against oracle/sam_oracle.py split_fp8_lo(fmt="e2m1", block=32)._q8 -- the emulation the error budget was
    run with (plans9 / plans10) -- for hi = ET(x) and lo = x - hi, plain and per-head padded K axis, A- and B-operand scale tiles:
    every decoded value identical."""
    from oracle import sam_oracle as so
    g = torch.Generator().manual_seed(5 + G)
    K = 1280
    for rows, is_b in ((512, False), (640, True)):
        x = torch.randn(rows, K, generator=g) * torch.exp(torch.randn(rows, K, generator=g))       # a wide magnitude range per block
        x[3, 64:96] = 0.0                                                                           # an all-zero block
        hi_bits, q, sc = _mx_pack(lib, prec, x, G, GP, is_b)
        hi = x.to(dt).to(torch.float32)
        assert torch.equal(hi_bits.cpu().view(dt).to(torch.float32), hi)
        q4 = so.split_fp8_lo(dt, fmt="e2m1", block=32)
        Kp = K // G * GP
        for h, src in ((0, hi), (1, x - hi)):
            want = q4._q8(_mx_pad(src, G, GP).double())      # fp64: floor(log2(amax)) exact next to powers of two
            got = _mx_decode(q[h], sc[h], rows, Kp, is_b)
            bad = (got != want)
            assert not bad.any(), f"{name} G={G} is_b={is_b} {'lo' if h else 'hi'}: {int(bad.sum())} of {bad.numel()} values differ"


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("G,GP,out_f32,from_n", [(1280, 1280, 1, 0), (80, 96, 1, 0), (1280, 1280, 0, 0), (1280, 1280, 0, 320)])
def test_gemm_mx_lo_terms(lib, name, prec, dt, ulp, G, GP, out_f32, from_n):
    """gemm_et_mx_kernel: C = A_hi B_hi^T + q4(A_lo) q4(B_hi)^T + q4(A_hi) q4(B_lo)^T + bias, the two correction terms on gfx950's
    block-scaled fp4 MFMA.  (i) against the fp64 evaluation of exactly that expression (the oracle's emulation, decoded from the
    kernel's own packed operands): fp32 accumulation noise only; (ii) against the fp64 product of the UN-rounded operands: the
    error of a plain f16 GEMM cut by more than 3x (oracle/error_budget.py: 2.9e-4 -> 5.2e-5); (iii) fp32 output accumulates into
    C, ET output is rounded once; split_from_n: the tiles in front of it are the plain product, bit for bit."""
    g = torch.Generator().manual_seed(21 + GP + out_f32 + from_n)
    M, N, K = 512, 640, 1280
    A = torch.randn(M, K, generator=g)
    B = (torch.rand(N, K, generator=g) * 2 - 1) / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.5
    Ah, qa, sa = _mx_pack(lib, prec, A, G, GP, False)
    Bh, qb, sb = _mx_pack(lib, prec, B, G, GP, True)
    Kp = K // G * GP
    ah, bh = A.to(dt).double(), B.to(dt).double()
    emul = (ah @ bh.t() + _mx_decode(qa[1], sa[1], M, Kp, False) @ _mx_decode(qb[0], sb[0], N, Kp, True).t()
            + _mx_decode(qa[0], sa[0], M, Kp, False) @ _mx_decode(qb[1], sb[1], N, Kp, True).t() + bias.double())
    exact = A.double() @ B.double().t() + bias.double()
    plain = ah @ bh.t() + bias.double()
    args = (qa[1].data_ptr(), qa[0].data_ptr(), sa[1].data_ptr(), sa[0].data_ptr(), qb[0].data_ptr(), qb[1].data_ptr(), sb[0].data_ptr(), sb[1].data_ptr())
    if out_f32:
        res = torch.randn(M, N, generator=g)
        out = dev(res.clone())
        assert lib.samrs_k_gemm_mx(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), M, N, K, Kp, *args, 1, 1, 0, stream()) == 0
        got = out.cpu().double() - res.double()
        e_emul = ((got - emul).norm() / emul.norm()).item()
        e_exact, e_plain = ((got - exact).norm() / exact.norm()).item(), ((plain - exact).norm() / exact.norm()).item()
        print(f"gemm_mx {name} G={G}->{GP} fp32 out: vs emulation {e_emul:.2e}; vs exact {e_exact:.2e} (plain {name} GEMM: {e_plain:.2e})")
        assert e_emul < 2e-6
        assert e_exact < 0.3 * e_plain
    else:
        out = torch.zeros(M, N, dtype=torch.int16, device="cuda")
        assert lib.samrs_k_gemm_mx(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), M, N, K, Kp, *args, 0, 0, from_n, stream()) == 0
        got = out.cpu().view(dt).double()
        want = emul.clone()
        want[:, :from_n] = plain[:, :from_n]
        err = ((got - want).abs() / want.abs().clamp(min=1.0)).max().item()
        print(f"gemm_mx {name} ET out, split_from_n={from_n}: max err vs emulation {err:.2e} (operand ulp {ulp:.1e})")
        assert err < 0.51 * ulp
        if from_n:
            ref = torch.zeros(M, N, dtype=torch.int16, device="cuda")
            assert lib.samrs_k_gemm(prec, Ah.data_ptr(), Bh.data_ptr(), ref.data_ptr(), dev(bias).data_ptr(), None, 0, M, N, K, 0, 0, 0, stream()) == 0
            assert torch.equal(out.cpu()[:, :from_n], ref.cpu()[:, :from_n])
            assert not torch.equal(out.cpu()[:, from_n:], ref.cpu()[:, from_n:])
    # race screen (two wave groups half a stage apart, DMA pieces of two stage kinds in flight): run-to-run identical
    first = out.clone()
    for _ in range(8):
        if out_f32:
            out.copy_(dev(res))
        assert lib.samrs_k_gemm_mx(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), M, N, K, Kp, *args, out_f32, out_f32, from_n, stream()) == 0
        assert torch.equal(out, first)
    assert lib.samrs_k_gemm_mx(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), M, N + 64, K, Kp, *args, out_f32, 0, 0, stream()) != 0


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_layernorm_mx_outputs_equal_the_pack_kernel(lib, name, prec, dt, ulp):
    """The LayerNorm's own MXFP4 outputs (what the engine feeds the qkv GEMM in lo_format 4) == samrs_k_mx4_pack of the fp32
    LayerNorm output, byte for byte (codes and scale tiles), and its ET output is unchanged."""
    g = torch.Generator().manual_seed(9)
    rows, D = 768, 1280
    x = torch.randn(rows, D, generator=g) * 3 + 0.5
    gamma, beta = 1 + 0.1 * torch.randn(D, generator=g), 0.05 * torch.randn(D, generator=g)
    xd, gd, bd = dev(x), dev(gamma), dev(beta)
    et = torch.zeros(rows, D, dtype=torch.int16, device="cuda")
    f32 = torch.zeros(rows, D, dtype=torch.float32, device="cuda")
    assert lib.samrs_k_layernorm(prec, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-6, et.data_ptr(), f32.data_ptr(), rows, D, 0, 1, 64, 14, stream()) == 0
    hi, q, sc = _mx_pack(lib, prec, f32.cpu(), D, D, False)
    et2 = torch.zeros_like(et)
    q2 = [torch.zeros_like(t) for t in q]
    s2 = [torch.zeros_like(t) for t in sc]
    assert lib.samrs_k_layernorm_mx(prec, xd.data_ptr(), gd.data_ptr(), bd.data_ptr(), 1e-6, et2.data_ptr(), rows, D, q2[0].data_ptr(),
                                    q2[1].data_ptr(), s2[0].data_ptr(), s2[1].data_ptr(), stream()) == 0
    assert torch.equal(et2, et) and torch.equal(hi, et)
    for h in range(2):
        assert torch.equal(q2[h], q[h]), f"{'lo' if h else 'hi'} codes differ"
        assert torch.equal(s2[h], sc[h])


@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
@pytest.mark.parametrize("is_global", [0, 1])
def test_attention_mx_outputs_equal_the_pack_kernel(lib, name, prec, dt, ulp, is_global):
    """Both attention kernels' own MXFP4 outputs (what the engine feeds the proj GEMM in lo_format 4: per-head padded K axis
    80 -> 96, block-internal order of store_attention_row_mx) against the pack kernel run on the (out, out_lo) pair of the f16-lo
    variant of the same kernel: the ET output identical in all three variants, the hi codes / scales identical, the lo codes within
    one fp4 step (the MX variant quantises the fp32 remainder, the pack kernel its f16 rounding) and almost always equal."""
    g = torch.Generator().manual_seed(31 + is_global)
    n_img, heads, hd, grid = 1, 16, 80, 64
    D = heads * hd
    rows = n_img * grid * grid
    qkv = dev((torch.randn(rows, 3 * D, generator=g)).to(dt)).view(torch.int16)
    bias = dev(torch.randn(3 * D, generator=g))
    tab = 127 if is_global else 27
    rh, rw = dev(0.02 * torch.randn(tab, hd, generator=g)), dev(0.02 * torch.randn(tab, hd, generator=g))
    Kp = heads * 96
    out = [torch.zeros(rows, D, dtype=torch.int16, device="cuda") for _ in range(3)]
    lo = torch.zeros(rows, D, dtype=torch.int16, device="cuda")
    q = [torch.zeros(rows, Kp // 2, dtype=torch.uint8, device="cuda") for _ in range(4)]
    nb = int(lib.samrs_k_mx_scale_bytes(rows, Kp, 0))
    sc = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(4)]
    common = (qkv.data_ptr(), bias.data_ptr(), rh.data_ptr(), rw.data_ptr())
    assert lib.samrs_k_attention_mx(prec, is_global, *common, out[0].data_ptr(), None, n_img, grid, heads, hd, None, None, None, None, stream()) == 0
    assert lib.samrs_k_attention_mx(prec, is_global, *common, out[1].data_ptr(), lo.data_ptr(), n_img, grid, heads, hd, None, None, None, None, stream()) == 0
    assert lib.samrs_k_attention_mx(prec, is_global, *common, out[2].data_ptr(), None, n_img, grid, heads, hd, q[0].data_ptr(), q[1].data_ptr(),
                                    sc[0].data_ptr(), sc[1].data_ptr(), stream()) == 0
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], out[2])
    assert lib.samrs_k_mx4_pack(prec, None, out[1].data_ptr(), lo.data_ptr(), None, q[2].data_ptr(), q[3].data_ptr(), sc[2].data_ptr(), sc[3].data_ptr(),
                                rows, D, hd, 96, 2, stream()) == 0                     # is_b = 2: A scale tiles, attention block order
    assert torch.equal(q[0], q[2]) and torch.equal(sc[0], sc[2]), "hi codes / scales differ"
    same_scale = torch.equal(sc[1], sc[3])
    diff = (q[1] != q[3]).float().mean().item()
    print(f"attention MX {name} global={is_global}: lo bytes differing from the pack of the f16-rounded remainder: {diff:.2e}; lo scales equal: {same_scale}")
    assert diff < 0.15            # a block whose lo maximum crosses a power of two between fp32 and its f16 rounding changes all its codes
    # decoded values: hi exact; lo within one fp4 step of the block scale
    a = _mx_decode(q[1], sc[1], rows, Kp, False)
    b = _mx_decode(q[3], sc[3], rows, Kp, False)
    assert ((a - b).abs() <= 0.51 * b.abs().reshape(rows, -1, 32).amax(-1, keepdim=True).clamp(min=1e-30).expand(-1, -1, 32).reshape(rows, Kp)).all()


@pytest.mark.parametrize("plain", [0, 1])
@pytest.mark.parametrize("name,prec,dt,ulp", PRECS)
def test_gemm_mx_gelu_epilogue_emits_mx_rows(lib, name, prec, dt, ulp, plain):
    """lin1 of the all-split mode: gemm_et_mx_kernel with the exact-erf GELU in its epilogue, which also writes its output as MXFP4
    hi / lo rows for lin2 (K axis padded per 80-column wave tile to 96, block-internal order of the epilogue's lanes).  (i) the ET
    output = GELU of the fp64 evaluation of the kernel's own expression, rounded once; (ii) the hi codes and scales are EXACTLY what
    the pack kernel makes of that ET output (same layout, same order); (iii) the lo rows carry the remainder: adding the decoded lo
    to the ET value cuts the distance to the exact GELU by more than half in rms.
    plain = 1: lin1 of split 207 (lin2 alone takes lo terms): NO tile takes lo terms -- the persistent plain kernel
    (gemm_et_x64p_kernel<MXO>) with the same MX-row epilogue; the product is the plain one, everything else as above."""
    g = torch.Generator().manual_seed(77)
    M, N, K = 512, 640, 1280
    A = torch.randn(M, K, generator=g)
    B = (torch.rand(N, K, generator=g) * 2 - 1) * 2.0 / math.sqrt(K)
    bias = torch.randn(N, generator=g) * 0.5
    Ah, qa, sa = _mx_pack(lib, prec, A, K, K, False)
    Bh, qb, sb = _mx_pack(lib, prec, B, K, K, True)
    emul = A.to(dt).double() @ B.to(dt).double().t() + bias.double()
    if not plain:
        emul = (emul + _mx_decode(qa[1], sa[1], M, K, False) @ _mx_decode(qb[0], sb[0], N, K, True).t()
                + _mx_decode(qa[0], sa[0], M, K, False) @ _mx_decode(qb[1], sb[1], N, K, True).t())
    ref = F.gelu(emul)                                                            # exact erf GELU, fp64
    Kp = N // 80 * 96
    out = torch.zeros(M, N, dtype=torch.int16, device="cuda")
    q = [torch.zeros(M, Kp // 2, dtype=torch.uint8, device="cuda") for _ in range(4)]
    nb = int(lib.samrs_k_mx_scale_bytes(M, Kp, 0))
    sc = [torch.zeros(nb, dtype=torch.uint8, device="cuda") for _ in range(4)]
    assert lib.samrs_k_gemm_mx_gelu_mxout(prec, Ah.data_ptr(), Bh.data_ptr(), out.data_ptr(), dev(bias).data_ptr(), M, N, K, K, qa[1].data_ptr(),
                                          qa[0].data_ptr(), sa[1].data_ptr(), sa[0].data_ptr(), qb[0].data_ptr(), qb[1].data_ptr(), sb[0].data_ptr(),
                                          sb[1].data_ptr(), 1 | (2 if plain else 0), q[0].data_ptr(), q[1].data_ptr(), sc[0].data_ptr(), sc[1].data_ptr(),
                                          stream()) == 0
    got = out.cpu().view(dt).double()
    err = ((got - ref).abs() / ref.abs().clamp(min=1.0)).max().item()
    print(f"gemm_mx + GELU {name}: max err vs GELU(emulation) {err:.2e} (operand ulp {ulp:.1e})")
    assert err < 0.6 * ulp
    # (ii) hi rows == pack kernel on the ET output itself (fp32 copies of the ET values: hi = the value, lo = 0)
    x = out.cpu().view(dt).to(torch.float32)
    xd = dev(x)
    assert lib.samrs_k_mx4_pack(prec, xd.data_ptr(), None, None, None, q[2].data_ptr(), q[3].data_ptr(), sc[2].data_ptr(), sc[3].data_ptr(),
                                M, N, 80, 96, 4, stream()) == 0                   # is_b = 4: A scale tiles, epilogue block order
    assert torch.equal(q[0], q[2]) and torch.equal(sc[0], sc[2]), "hi codes / scales of the epilogue differ from the pack kernel's"
    # (iii) decode the lo rows back to columns: position 8 f + 4 i + e of block b3 of wave tile wt holds column 80 wt + 32 b3 + 16 i + 4 f + e
    lo_pad = _mx_decode(q[1], sc[1], M, Kp, False).reshape(M, N // 80, 3, 4, 2, 4)         # [row][wt][b3][f][i][e]
    lo_cols = lo_pad.permute(0, 1, 2, 4, 3, 5).reshape(M, N // 80, 96)[:, :, :80].reshape(M, N)
    pad = lo_pad.permute(0, 1, 2, 4, 3, 5).reshape(M, N // 80, 96)[:, :, 80:]
    assert (pad == 0).all()
    before = (ref - got).pow(2).mean().sqrt().item()
    after = (ref - (got + lo_cols)).pow(2).mean().sqrt().item()
    print(f"   rms distance to the exact GELU: ET output {before:.3e}, ET + decoded lo {after:.3e}")
    assert after < 0.5 * before
