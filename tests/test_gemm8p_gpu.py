"""gemm8p_pers_kernel (the ping-pong persistent GEMM, speechclip_amd/csrc/gemm8p.hip) against fp32 torch: every epilogue variant it serves --
bf16 / fp32 output, bias, GELU / QuickGELU, bf16 / fp32 residual -- ragged M (the last panel shifted back), overlapping rows (conv as GEMM), operands
beyond 32-bit offsets, and the dispatcher's choice.  Replaces in the reference: every nn.Linear / Conv1d of the encoders
(avssl/module/speech_encoder_plus.py:49-56,75,84-85 -> fairseq TransformerSentenceEncoderLayer [3P]; clip ResidualAttentionBlock [3P])."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w, bias, act, res):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = y * torch.sigmoid(1.702 * y)
    if res is not None:
        y = y + res.float()
    return y


@pytest.fixture()
def force_8p():
    from speechclip_amd._lib import lib
    lib().sc_debug_set_gemm_mode(16)
    yield
    lib().sc_debug_set_gemm_mode(-1)


@pytest.mark.parametrize("M,N,K,lda", [(256 * 37, 768, 768, None), (256 * 20 + 77, 1024, 128, None), (19000, 512, 1536, 1024), (9001, 2304, 192, None)])
@pytest.mark.parametrize("act,res,f32", [(0, False, False), (1, False, False), (0, True, False), (2, True, False), (0, True, True), (1, False, True), (2, True, True)])
def test_gemm8p_epilogue_variants_vs_fp32(force_8p, M, N, K, lda, act, res, f32):
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator(device="cpu").manual_seed(M + N + K + act)
    ld = lda or K
    flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else torch.bfloat16) if res else None
    y = ops.gemm(flat, w, bias, act, r, out_f32=f32, M=M, K=K, lda=ld)
    assert lib().sc_gemm_last_path() == 3                      # gemm8p_pers_kernel ran (ragged M, overlapping rows and fp32 outputs included)
    assert y.dtype == (torch.float32 if f32 else torch.bfloat16)
    a = torch.as_strided(flat, (M, K), (ld, 1))
    tol = 3e-3 if f32 else 2e-2
    torch.testing.assert_close(y.float(), _ref(a, w, bias, act, r), atol=tol, rtol=tol)


def test_gemm8p_is_transpose_detecting_and_bias_free(force_8p):
    """Identity-like A with a non-symmetric W: a swapped fragment layout or a transposed store shows up as W instead of W^T; bias = None reads zeros."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    M = N = K = 256 * 12
    a = torch.eye(M, device="cuda", dtype=torch.bfloat16)
    w = ((torch.arange(N * K, device="cuda").reshape(N, K) * 7) % 251).to(torch.bfloat16)
    y = ops.gemm(a, w)
    assert lib().sc_gemm_last_path() == 3
    torch.testing.assert_close(y.float(), w.float().t())


def test_gemm8p_is_the_default_for_the_step_shapes_and_old_kernels_keep_the_rest():
    """Dispatcher rule (gemm.hip): bf16 / fp32 output, N % 256 == 0, N <= 8192, >= 128 tiles -> gemm8p (path 3); everything else -> gemm256_kernel /
    gemm_bf16_kernel (path 0).  Both are hand-written; path 1 (vendor library) only with the comparator switched on."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    lib().sc_debug_set_gemm_mode(-1)
    g = torch.Generator().manual_seed(3)
    for (M, N, K, f32), want in (((12800, 768, 768, True), 3), ((32000, 2304, 768, False), 3), ((4096, 768, 768, False), 0), ((12800, 520, 768, False), 0),
                                 ((8192, 8192 + 256, 512, False), 0)):
        a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
        y = ops.gemm(a, w, out_f32=f32)
        assert lib().sc_gemm_last_path() == want, (M, N, K, f32, lib().sc_gemm_last_path())
        torch.testing.assert_close(y.float(), a.float() @ w.float().t(), atol=3e-2, rtol=2e-2)


def test_gemm8p_operands_beyond_32bit_offsets(force_8p):
    """A operand of 4.4e9 elements (conv layer 1 at B = 256 has 4.19e9): tile base pointers are 64-bit, per-lane offsets inside a half tile 32-bit."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    M, N, K, ld = 4300000, 256, 192, 1024
    flat = torch.empty(M * ld + K + 8, device="cuda", dtype=torch.bfloat16)
    flat.normal_(0, 0.5)
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).to(torch.bfloat16)
    y = ops.gemm(flat, w, None, 0, None, M=M, K=K, lda=ld)
    assert lib().sc_gemm_last_path() == 3
    for m0 in (0, 2200000, M - 4096):
        a = torch.as_strided(flat[m0 * ld:], (4096, K), (ld, 1))
        torch.testing.assert_close(y[m0:m0 + 4096].float(), a.float() @ w.float().t(), atol=2e-2, rtol=2e-2)


def test_gemm8p_dynamic_tile_order_equals_static_and_survives_ring_reuse():
    """The persistent kernel takes its tiles from per-XCD counters (one 64-byte slot of a 4096-slot ring per launch, words tagged with the launch's generation;
    gemm8p.hip, K >= 384).  Mode 26 = the static stride order.  Same tiles, same arithmetic per tile -> bit-identical outputs; more launches than ring
    slots, half of them racing on a second stream, must find every slot usable."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator().manual_seed(11)
    M, N, K = 256 * 123 + 100, 1280, 768                      # 620 tiles on 256 blocks, ragged last panel
    a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16)
    try:
        lib().sc_debug_set_gemm_mode(26)
        y_static = ops.gemm(a, w, bias, 1, r)
        assert lib().sc_gemm_last_path() == 3
        lib().sc_debug_set_gemm_mode(16)
        y_dyn = ops.gemm(a, w, bias, 1, r)
        assert torch.equal(y_static, y_dyn)
        # ring reuse: 2 x 2200 small launches (40 x 8 = 320 tiles each) on two streams at once, then the big shape again
        a2 = a[:256 * 40]
        w2 = (torch.randn(2048, 384, generator=g) * 384 ** -0.5).to("cuda", torch.bfloat16)
        M2 = a2.shape[0]
        want = ops.gemm(a2, w2, M=M2, K=384, lda=K)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        outs = [torch.empty_like(want), torch.empty_like(want)]
        for _ in range(2200):
            ops.gemm(a2, w2, out=outs[0], M=M2, K=384, lda=K)
            with torch.cuda.stream(side):
                ops.gemm(a2, w2, out=outs[1], M=M2, K=384, lda=K)
        torch.cuda.current_stream().wait_stream(side)
        assert torch.equal(outs[0], want) and torch.equal(outs[1], want)
        assert torch.equal(ops.gemm(a, w, bias, 1, r), y_static)
    finally:
        lib().sc_debug_set_gemm_mode(-1)
    torch.testing.assert_close(y_dyn.float(), _ref(a, w, bias, 1, r), atol=2e-2, rtol=2e-2)


def test_dynamic_tile_order_survives_a_poisoned_counter_ring():
    """a launch that faults or is torn down mid-flight leaves its slot of the tile-counter ring dirty; the launch that gets the slot next must
    not read it as "tiles already taken" (silently skipped tiles).  Counter words carry the launch generation and every block raises its word to its own
    generation before the first fetch (gemm8p.hip, g_sched).  Here: the WHOLE ring is overwritten with an old generation whose counts say every tile is gone
    (sc_debug_poison_gemm_sched), on an output buffer pre-filled with NaN: the result must be bit-identical to the static order's, every time."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator().manual_seed(13)
    M, N, K = 256 * 70 + 36, 1024, 512                        # 284 tiles on 256 blocks, ragged last panel
    a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    try:
        lib().sc_debug_set_gemm_mode(26)
        y_static = ops.gemm(a, w, bias, 0)
        lib().sc_debug_set_gemm_mode(16)
        for _ in range(3):
            torch.cuda.synchronize()
            assert lib().sc_debug_poison_gemm_sched() == 0
            out = torch.full((M, N), float("nan"), device="cuda", dtype=torch.bfloat16)
            y = ops.gemm(a, w, bias, 0, out=out)
            assert lib().sc_gemm_last_path() == 3
            assert torch.equal(y, y_static)
    finally:
        lib().sc_debug_set_gemm_mode(-1)


@pytest.mark.parametrize("sigma,mean_tol,binned_tol", [(0.55, 1.6e-4, 3.5e-4), (1.0, 1.6e-4, 5.9e-4)])
def test_gelu_epilogue_systematic_error(sigma, mean_tol, binned_tol):
    """What the every-input test below cannot see: the SYSTEMATIC part of the polynomial's error.  Rounding noise averages out over the K sum of the next GEMM, an
    approximation error (a smooth function of x) adds up coherently -- a minimax degree-4 fit passed every per-element tolerance and still moved the bench line's
    centred cosine from 0.9966 to 0.9943 (EXPERIMENTS.md R6-3).  1 M pre-activations x ~ N(0, sigma^2) (the conv stack's and fc1's range) through the kernel (identity W):
    the MEAN error and the density-weighted rms of the per-bin mean error (64 bins) must stay at the shipped fit's level.  float16 simulation of the instruction
    sequence, sigma 0.55 / 1.0 -- mean: shipped fit -8.7e-5 / -5.7e-5, degree 6 -4.9e-5 / -4e-6, the rejected minimax fit -3.8e-4 / -2.9e-4;
    binned rms: 2.2e-4 / 4.9e-4, 1.7e-4 / 4.3e-4, 5.5e-4 / 6.9e-4 (its floor is the half-precision rounding of Phi, deterministic per bf16 input value)."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator().manual_seed(17)
    a = (sigma * torch.randn(4096, 256, generator=g)).to(torch.bfloat16)
    w = torch.eye(256, dtype=torch.bfloat16)
    try:
        lib().sc_debug_set_gemm_mode(16)
        y = ops.gemm(a.cuda(), w.cuda(), None, 1).float().cpu()
        assert lib().sc_gemm_last_path() == 3
    finally:
        lib().sc_debug_set_gemm_mode(-1)
    x = a.double()
    err = (y.double() - torch.nn.functional.gelu(x)).flatten()
    assert abs(err.mean().item()) <= mean_tol, err.mean().item()
    edges = torch.linspace(-4 * sigma, 4 * sigma, 65, dtype=torch.float64)
    idx = torch.bucketize(x.flatten(), edges).clamp(1, 64) - 1
    cnt = torch.zeros(64, dtype=torch.float64).index_add_(0, idx, torch.ones_like(err))
    sm = torch.zeros(64, dtype=torch.float64).index_add_(0, idx, err)
    bias = sm / cnt.clamp(min=1)
    binned = torch.sqrt((cnt * bias * bias).sum() / cnt.sum()).item()
    assert binned <= binned_tol, binned


def test_gelu_epilogue_every_bf16_input():
    """The fused-GELU epilogue (packed-half polynomial -- degree 4 since round 6 --, common.h gelu_poly2_x8) on EVERY finite bf16 pre-activation with |x| <= 60000, each fed through the
    kernel exactly (one non-zero per A row against an identity W): against exact erf-GELU within bf16 rounding + the polynomial's 3.2e-3, and saturated
    exactly (y == x / y == 0) beyond |x| = 5.5 -- the region where only the clamp of Phi keeps the un-clamped polynomial argument in check.
    Reference arithmetic: torch.nn.functional.gelu (fairseq TransformerSentenceEncoderLayer activation_fn, speech_encoder_plus.py:49-56 [3P])."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    bits = torch.arange(0, 65536, dtype=torch.int32)
    vals = bits.to(torch.int16).view(torch.bfloat16)
    vals = vals[torch.isfinite(vals.float()) & (vals.float().abs() <= 60000)]
    n = vals.numel()
    M = (n + 255) // 256 * 256
    a = torch.zeros(M, 256, dtype=torch.bfloat16)
    rows = torch.arange(n)
    a[rows, rows % 256] = vals
    w = torch.eye(256, dtype=torch.bfloat16)
    try:
        lib().sc_debug_set_gemm_mode(16)
        y = ops.gemm(a.cuda(), w.cuda(), None, 1).float().cpu()
        assert lib().sc_gemm_last_path() == 3
    finally:
        lib().sc_debug_set_gemm_mode(-1)
    got = y[rows, rows % 256]
    x = vals.float()
    want = torch.nn.functional.gelu(x.double()).float()
    assert torch.isfinite(got).all()
    tol = 3.2e-3 + want.abs() * 2.0 ** -7            # polynomial + half input + bf16 output rounding
    bad = (got - want).abs() > tol
    assert not bad.any(), (x[bad][:8], got[bad][:8], want[bad][:8])
    big = x.abs() >= 5.5
    assert torch.equal(got[big & (x > 0)], x[big & (x > 0)])          # Phi == 1: y is x (bf16 -> half -> bf16 is exact up to 65504)
    assert (got[big & (x < 0)] == 0).all()                             # Phi == 0
    off = y.clone(); off[rows, rows % 256] = 0
    assert (off == 0).all()                                            # GELU(0) == 0 everywhere else
