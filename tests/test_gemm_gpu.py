"""GPU parity: sc_gemm_bf16 against a torch fp32 reference on the same bf16-rounded operands."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(a, w, bias, act, residual):
    y = a.float() @ w.float().t()
    if bias is not None:
        y = y + bias
    if act == 1:
        y = torch.nn.functional.gelu(y)
    elif act == 2:
        y = y * torch.sigmoid(1.702 * y)
    if residual is not None:
        y = y + residual.float()
    return y


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 768, 768), (499, 2304, 768), (1000, 48, 6144), (130, 512, 1536),
                                   (37, 132, 128)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_gemm_matches_fp32(M, N, K, act):
    from speechclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M * 7 + N + K + act)
    a = (torch.randn(M, K, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)  # asymmetric, non-identity
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16)
    for use_bias, use_res, f32 in ((True, True, False), (False, False, False), (True, True, True)):
        r = res.float() if f32 else res
        y = ops.gemm(a, w, bias if use_bias else None, act, r if use_res else None, out_f32=f32)
        ref = _ref(a, w, bias if use_bias else None, act, r if use_res else None)
        tol = 2e-3 if f32 else 2e-2
        torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)


def test_gemm_overlapping_rows_is_conv1d():
    """lda < K: channels-last conv1d (k=3, s=2) as a GEMM over overlapping rows."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(3)
    T, C, k, s = 257, 64, 3, 2
    x = (torch.randn(T, C, generator=g)).to("cuda", torch.bfloat16)
    w = (torch.randn(128, C, k, generator=g) * 0.1).to("cuda", torch.bfloat16)     # [out, in, k]
    w_g = w.permute(0, 2, 1).reshape(128, k * C).contiguous()                       # [out, k*in]
    t_out = (T - k) // s + 1
    y = ops.gemm(x, w_g, M=t_out, K=k * C, lda=s * C)
    ref = torch.nn.functional.conv1d(x.float().t()[None], w.float(), stride=s)[0].t()
    torch.testing.assert_close(y.float(), ref, atol=2e-2, rtol=2e-2)


def test_gemm_transpose_detecting():
    """A = I with an asymmetric W: catches a swapped C-write."""
    from speechclip_amd import ops
    a = torch.eye(128, device="cuda", dtype=torch.bfloat16)
    w = (torch.arange(128 * 128, device="cuda").reshape(128, 128) % 251).to(torch.bfloat16)
    y = ops.gemm(a, w)
    torch.testing.assert_close(y.float(), w.float().t())


@pytest.mark.parametrize("M,N,K,lda", [(20001, 768, 768, None), (16390, 1028, 128, None), (33000, 512, 1536, 1024), (25000, 2304, 64, None)])
@pytest.mark.parametrize("act,f32", [(0, False), (1, False), (2, True)])
def test_gemm_256_tile_kernel(M, N, K, lda, act, f32):
    """Shapes large enough (>= 192 256x256 tiles) to take the 4-slot-ring 256^2 kernel; M/N tails; overlapping rows."""
    from speechclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    ld = lda or K
    flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else torch.bfloat16)
    y = ops.gemm(flat, w, bias, act, res, out_f32=f32, M=M, K=K, lda=ld)
    a = torch.as_strided(flat, (M, K), (ld, 1))
    ref = _ref(a, w, bias, act, res)
    tol = 2e-3 if f32 else 2e-2
    torch.testing.assert_close(y.float(), ref, atol=tol, rtol=tol)


def test_plain_gemms_run_on_the_hand_written_kernel_by_default():
    """north_star: hand-written HIP kernels.  A plain (bias-only) GEMM of a library-eligible shape must hit gemm256_kernel unless the
    comparator was switched on explicitly (the suite used to validate hipBLASLt on these shapes)."""
    import os
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    assert os.environ.get("SC_GEMM_VENDOR", "0") != "1" and not ops.vendor_gemm_enabled()
    g = torch.Generator().manual_seed(1)
    for M, N, K in ((128000 // 8, 2304, 768), (8192, 768, 3072)):
        a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", torch.bfloat16)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
        bias = torch.randn(N, generator=g).cuda()
        y = ops.gemm(a, w, bias)
        assert lib().sc_gemm_last_path() in (0, 2, 3)      # 1 would be the vendor library
        torch.testing.assert_close(y.float(), a.float() @ w.float().t() + bias, atol=3e-2, rtol=2e-2)


def test_vendor_comparator_path_correct_on_three_streams():
    """The comparator (ops.set_vendor_gemm(True): hipBLASLt behind the same entry, used only by bench.py's `vendor_comparator` leg) hands one
    workspace half to each of two streams and sends a third stream to the hand-written kernels: paths == [1, 1, 0], the two library streams own
    DIFFERENT workspace halves, and every result is correct when checked IN STREAM ORDER on the stream that issued it (no device-wide
    synchronize before the comparison: a GEMM launched on the null stream instead of the caller's would race the check -- ADVICE r3, the loader's
    typedef once carried an extra int that did exactly that).  Switching the comparator off restores the hand-written path."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator().manual_seed(0)
    M, N, K = 8192, 768, 768
    a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    ref = a.float() @ w.float().t() + bias
    torch.cuda.synchronize()
    ops.set_vendor_gemm(True)
    try:
        streams = [torch.cuda.current_stream(), torch.cuda.Stream(), torch.cuda.Stream()]
        paths, slots, errs = [], [], []
        for rep in range(3):                                     # several rounds: an ordering bug needs a chance to show
            for st in streams:
                with torch.cuda.stream(st):
                    # a long kernel first, so that the GEMM is queued BEHIND work on its own stream: on the wrong stream it would start early and
                    # the stream-ordered comparison below would read a half-written / stale `y`
                    y = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
                    busy = torch.randn(4096, 4096, device="cuda") @ torch.randn(4096, 4096, device="cuda")
                    ops.gemm(a, w, bias, out=y)
                    if rep == 0:
                        paths.append(lib().sc_gemm_last_path())
                        slots.append(lib().sc_debug_vendor_stream_slot(st.cuda_stream))
                    errs.append((st, (y.float() - ref).abs().max()))       # same stream: ordered after the GEMM, no synchronize
                    del busy
        assert paths == [1, 1, 0], paths                         # two streams get the library, the third the hand-written kernel
        assert slots[:2] == [0, 1] and slots[2] == -2, slots     # ... on different workspace halves
        for st, e in errs:
            st.synchronize()
            assert e.item() < 0.15, e.item()
    finally:
        ops.set_vendor_gemm(False)
    assert lib().sc_debug_vendor_stream_slot(torch.cuda.current_stream().cuda_stream) in (-1, -3)
    out = ops.gemm(a, w, bias)
    assert lib().sc_gemm_last_path() in (0, 2, 3)      # 1 would be the vendor library
    torch.testing.assert_close(out.float(), ref, atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("M,N,K,res,f32", [(70000, 768, 128, False, False), (70000, 768, 128, True, False), (66000, 512, 192, False, True),
                                            (131072, 768, 768, True, False), (40000, 2304, 64, False, False)])
def test_gemm_256_tile_switch_counted_wait(M, N, K, res, f32):
    """Many tiles per persistent block with SHORT k-loops (nk = 1..3), where the tile-start wait is counted (`s_waitcnt vmcnt(N)`: the tail
    of the previous epilogue's stores and the next tile's stage 1 stay in flight): a wrong count would read a stage before it landed.
    Checked against fp32 and for run-to-run bitwise stability over several launches."""
    from speechclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K)
    a = (torch.randn(M, K, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else torch.bfloat16) if res else None
    outs = [ops.gemm(a, w, bias, 0, r, out_f32=f32) for _ in range(4)]
    torch.cuda.synchronize()
    ref = _ref(a, w, bias, 0, r)
    tol = 2e-3 if f32 else 2e-2
    torch.testing.assert_close(outs[0].float(), ref, atol=tol, rtol=tol)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


@pytest.mark.parametrize("S,M,N,K", [(4, 300, 260, 128), (30, 300, 520, 128), (14, 768, 768, 1024), (3, 3072, 768, 512), (40, 256, 256, 64)])
def test_batched_products_on_the_256_tile_kernel(S, M, N, K):
    """sc_gemm_bf16_batched with M, N >= 256 and >= 100 tiles runs all products as one persistent tile list on gemm256 (BATCH variant: the
    split-K partial products of the weight gradients); edge tiles (300 x 260) are shifted back inside their own product.  fp32 outputs vs torch."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(S * 1000 + M + N + K)
    a = (0.5 * torch.randn(S, M, K, generator=g)).to(torch.bfloat16)
    w = (0.5 * torch.randn(S, N, K, generator=g)).to(torch.bfloat16)
    out = torch.full((S, M, N), float("nan"), dtype=torch.float32, device="cuda")
    ops.gemm_batched(a.cuda(), K, M * K, w.cuda(), N * K, S, out, N, M * N, None, M, N, K, S)
    ref = a.float() @ w.float().transpose(1, 2)
    assert torch.isfinite(out).all()
    torch.testing.assert_close(out.cpu(), ref, atol=2e-2 * K ** 0.5 * 0.25, rtol=2e-3)
    # the same with a shared second operand (w_mod = 1) and bf16 outputs
    out16 = torch.empty(S, M, N, dtype=torch.bfloat16, device="cuda")
    ops.gemm_batched(a.cuda(), K, M * K, w[0].contiguous().cuda(), N * K, 1, out16, N, M * N, None, M, N, K, S)
    ref16 = a.float() @ w[0].float().t()
    torch.testing.assert_close(out16.float().cpu(), ref16, atol=6e-2 * K ** 0.5 * 0.25, rtol=2e-2)


def test_two_level_batched_products_vs_torch():
    """sc_gemm_bf16_batched2: (outer, inner) products with independent strides per level -- the (utterance, head) pairs of the attention backward."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, T, L = 3, 4, 70, 128
    q = (0.5 * torch.randn(B * T, 3 * H * 64, generator=g)).to(torch.bfloat16)
    kT = (0.5 * torch.randn(B * H, 64, L, generator=g)).to(torch.bfloat16)
    dS = (0.5 * torch.randn(B * H, L, L, generator=g)).to(torch.bfloat16)
    out = torch.zeros(B * T, 3 * H * 64, dtype=torch.bfloat16, device="cuda")
    # out[b, t, h*64:(h+1)*64] = dS[b*H + h][:T] @ kT[b*H + h]^T   (M = T rows, N = 64, contraction over L)
    ops.gemm_batched2(dS.cuda(), L, H * L * L, L * L, kT.cuda(), L, H * 64 * L, 64 * L, out, 3 * H * 64, T * 3 * H * 64, 64, T, 64, L, B, H)
    ref = torch.einsum("zik,zdk->zid", dS.float()[:, :T], kT.float()).view(B, H, T, 64).permute(0, 2, 1, 3).reshape(B * T, H * 64)
    torch.testing.assert_close(out.float().cpu()[:, :H * 64], ref, atol=0.15, rtol=2e-2)
    assert out[:, H * 64:].abs().max().item() == 0                       # nothing else was touched
    # strided first operand: S[b*H + h] = q_h k_h^T straight from packed rows
    S = torch.empty(B * H, T, L, dtype=torch.float32, device="cuda")
    qq = q.cuda()
    ops.gemm_batched2(qq, 3 * H * 64, T * 3 * H * 64, 64, qq[:, H * 64:], 3 * H * 64, T * 3 * H * 64, 64, S, L, H * T * L, T * L, T, 64, 64, B, H)
    x = q.float().view(B, T, 3, H, 64)
    refS = torch.einsum("bihd,bjhd->bhij", x[:, :, 0], x[:, :64, 1]).reshape(B * H, T, 64)
    torch.testing.assert_close(S.cpu()[:, :, :64], refS, atol=0.1, rtol=2e-2)


@pytest.mark.parametrize("M,N,K,lda", [(128000, 768, 768, None), (32768, 2304, 256, None), (40960, 512, 1536, 1024), (65536, 512, 1024, 1024),
                                       (23040, 3072, 768, None), (30720, 768, 3072, None), (196608, 256, 512, None), (15360, 1280, 384, None)])
@pytest.mark.parametrize("act,use_res", [(0, False), (1, False), (0, True), (2, False), (1, True)])
def test_gemm_large_full_tile_shapes(M, N, K, lda, act, use_res):
    """Large shapes made of interior 256 x 256 tiles only (M, N multiples of 256): every tile takes the branch-free fast epilogue (plain / GELU /
    QuickGELU) or the residual epilogue; with and without bias; run-to-run bitwise stable.  (The shape list is the one the round-3 staggered
    two-group kernel was validated on: profiles/r03_gemm_stagger_experiment.txt.)"""
    from speechclip_amd import ops
    g = torch.Generator(device="cpu").manual_seed(M + N + K + act)
    ld = lda or K
    flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", torch.bfloat16)
    bias = torch.randn(N, generator=g).cuda()
    res = torch.randn(M, N, generator=g).to("cuda", torch.bfloat16) if use_res else None
    y = ops.gemm(flat, w, bias, act, res, M=M, K=K, lda=ld)
    y2 = ops.gemm(flat, w, bias, act, res, M=M, K=K, lda=ld)
    assert torch.equal(y, y2), "not run-to-run deterministic"
    a = torch.as_strided(flat, (M, K), (ld, 1))
    # reference in row chunks (fp32 [M, N] of the largest shape is 1.5 GB: fine, but keep the peak low)
    worst = 0.0
    for r0 in range(0, M, 16384):
        sl = slice(r0, min(M, r0 + 16384))
        ref = _ref(a[sl], w, bias, act, res[sl] if use_res else None)
        torch.testing.assert_close(y[sl].float(), ref, atol=2e-2, rtol=2e-2)
        worst = max(worst, (y[sl].float() - ref).abs().max().item())
    # no bias, too
    y0 = ops.gemm(flat, w, None, act, res, M=M, K=K, lda=ld)
    ref0 = _ref(a[:4096], w, None, act, res[:4096] if use_res else None)
    torch.testing.assert_close(y0[:4096].float(), ref0, atol=2e-2, rtol=2e-2)


@pytest.mark.parametrize("M,N,K,lda,act", [(4096000, 512, 1536, 1024, 1),      # conv layer 1 of the B = 256 step: the A operand spans 4.19e9 bf16 elements
                                           (4352000, 512, 1536, 1024, 1),      # ... and one past 2^32 ELEMENTS (4.46e9), A = 8.9 GB
                                           (2250000, 2304, 768, None, 0)])     # plain GEMM whose OUTPUT passes 2^32 elements (5.18e9 bf16 = 10.4 GB)
def test_gemm_operands_beyond_32bit_offsets(M, N, K, lda, act):
    """at B = 256 the conv1 A operand is 4 096 000 rows x lda 1024 = 4.19e9 bf16 elements (8.4 GB: byte offsets pass 2^32 and
    2^33, element offsets pass 2^31), and the largest operand any earlier GEMM test compared with a reference was 1e8 elements -- a 32-bit offset
    anywhere in the tile addressing would pass every one of them and bench.py.  Overlapping rows (lda < K) exactly as the conv stack uses them
    (module/hubert.py), operands filled from a position-dependent pattern (so a row read from the WRONG place cannot match), checked against fp32
    torch on row blocks at the start, the end, and on both sides of the rows whose element / byte offsets cross 2^31, 2^32 (elements) and 2^32,
    2^33 (bytes) in A and in C."""
    from speechclip_amd import ops
    ld = lda or K
    n_el = M * ld + K + 8
    g = torch.Generator(device="cuda").manual_seed(M % 1000 + N + K)
    flat = torch.empty(n_el, device="cuda", dtype=torch.bfloat16)
    CH = 1 << 28
    for s0 in range(0, n_el, CH):                                   # fill in chunks (no 17 GB fp32 temporary)
        n = min(CH, n_el - s0)
        flat[s0:s0 + n] = (torch.randn(n, generator=g, device="cuda") * 0.5).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g, device="cuda") * K ** -0.5).to(torch.bfloat16)
    bias = torch.randn(N, generator=g, device="cuda")
    y = ops.gemm(flat, w, bias, act, None, M=M, K=K, lda=ld)
    assert y.shape == (M, N)
    a = torch.as_strided(flat, (M, K), (ld, 1))
    marks = {0, M - 512}
    for lim in (2 ** 31, 2 ** 32):
        for per_row in (ld, 2 * ld, N, 2 * N):                      # element and byte offsets of A rows and of C rows
            r = lim // per_row
            if 512 <= r < M - 512:
                marks |= {r - 256}
    marks |= {(M // 3) // 256 * 256 + 77, (2 * M // 3) // 256 * 256 + 131}
    worst = 0.0
    for r0 in sorted(marks):
        sl = slice(r0, r0 + 512)
        ref = _ref(a[sl], w, bias, act, None)
        torch.testing.assert_close(y[sl].float(), ref, atol=2e-2, rtol=2e-2)
        worst = max(worst, (y[sl].float() - ref).abs().max().item())
    # every row was written (a wrapped offset would leave part of C untouched and write another part twice): compare a cheap per-row checksum of the
    # WHOLE output against the fp32 reference of the same checksum, a[M,K] @ (w^T 1) + sum(bias), for the linear case; finite everywhere otherwise
    if act == 0:
        got = torch.empty(M, device="cuda")
        want = torch.empty(M, device="cuda")
        wsum = w.float().sum(0)
        for r0 in range(0, M, 1 << 18):
            sl = slice(r0, min(M, r0 + (1 << 18)))
            got[sl] = y[sl].float().sum(1)
            want[sl] = a[sl].float() @ wsum + bias.sum()
        torch.testing.assert_close(got, want, atol=1.5, rtol=2e-2)
    else:
        for r0 in range(0, M, 1 << 20):
            assert torch.isfinite(y[r0:r0 + (1 << 20)].float().sum()).item()
    print(f"M={M} N={N} K={K} lda={ld}: A spans {n_el:.3e} elements, C {M * N:.3e}; {len(marks)} row blocks, max abs err {worst:.4f}")


@pytest.mark.parametrize("M,N,K,act,res", [(256, 768, 3072, 0, True), (64, 768, 3072, 0, False), (256, 3072, 2048, 1, False), (40, 512, 4096, 0, True)])
def test_hp_linear_deterministic_split_k(M, N, K, act, res):
    """hp_linear on few-row, deep-K shapes (the CLS rows through linear2 of the pooling head): hi/lo-split operands, K chunks as a batched GEMM into fp32
    partials, sc_splitk_reduce_f32 in fixed order -- fp32-grade result (vs float64), bitwise run-to-run stable (no atomics), bias / GELU / residual
    applied in the finish kernel."""
    from speechclip_amd.module.kw_modules.TransformerModels import hp_linear
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g).cuda()
    lin = torch.nn.Linear(K, N).cuda()
    with torch.no_grad():
        lin.weight.copy_(torch.randn(N, K, generator=g) * K ** -0.5)
        lin.bias.copy_(torch.randn(N, generator=g))
    r = torch.randn(M, N, generator=g).cuda() if res else None
    y = hp_linear(a, lin.weight, lin.bias, act, r)
    y2 = hp_linear(a, lin.weight, lin.bias, act, r)
    assert y.dtype == torch.float32 and torch.equal(y, y2), "split-K finish is not run-to-run deterministic"
    ref = a.double() @ lin.weight.double().t() + lin.bias.double()
    if act == 1:
        ref = torch.nn.functional.gelu(ref)
    if res:
        ref = ref + r.double()
    rel = ((y.double() - ref).norm() / ref.norm()).item()
    assert rel < 2e-5, rel                      # bf16 operands would give ~2e-3
    # broadcast residual (ldr = 0: the CLS token added to every row, TransformerModels.forward_cls)
    if res:
        row = torch.randn(1, N, generator=g).cuda()
        yb = hp_linear(a, lin.weight, lin.bias, act, row.expand(M, N))
        refb = a.double() @ lin.weight.double().t() + lin.bias.double() + row.double()
        assert ((yb.double() - refb).norm() / refb.norm()).item() < 2e-5
