"""The IEEE-half operand format of the frozen pre-LN encoder layers (SC_GEMM_F16 / SC_LN_OUT_F16 / SC_ATTN_F16; speechclip_amd/module/hubert.py `_PRELN_F16`):
every kernel that takes it, against fp32 torch on the same (half-rounded) inputs, and the end-to-end effect on a pre-LN tower.

What it replaces in the reference: the HuBERT-large transformer layers under fp16 autocast (config/speechCLIP/model_large/coco/spchclp_p.yaml:122 `precision: 16`;
avssl/module/speech_encoder_plus.py:49-56 -> fairseq TransformerSentenceEncoderLayer [3P], layer_norm_first).  Tolerances: an f16 result carries 11 significand
bits -- the 16-bit outputs are held to 2e-3 (8x tighter than the bf16 tests' 2e-2), fp32 outputs to 1e-3."""
import pytest
import torch

pytestmark = pytest.mark.gpu

f16 = torch.float16


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


@pytest.mark.parametrize("M,N,K,lda", [(256 * 37, 1024, 1024, None), (256 * 20 + 77, 3072, 128, None), (19000, 512, 1536, 1024), (9001, 2304, 192, None)])
@pytest.mark.parametrize("act,res,f32", [(0, False, False), (1, False, False), (0, True, False), (2, True, False), (0, True, True), (1, False, True), (2, True, True)])
def test_gemm8p_f16_epilogue_variants_vs_fp32(M, N, K, lda, act, res, f32):
    """gemm8p_pers_kernel<ACT, RES, F32, F16 = true>: QKV (plain), fc1 (GELU), out-proj / fc2 (fp32 residual stream) of a pre-LN layer + the other epilogues."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    lib().sc_debug_set_gemm_mode(16)
    try:
        g = torch.Generator(device="cpu").manual_seed(M + N + K + act)
        ld = lda or K
        flat = (torch.randn(M * ld + K + 8, generator=g) * 0.5).to("cuda", f16)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", f16)
        bias = torch.randn(N, generator=g).cuda()
        r = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else f16) if res else None
        y = ops.gemm(flat, w, bias, act, r, out_f32=f32, M=M, K=K, lda=ld)
        assert lib().sc_gemm_last_path() == 3
        assert y.dtype == (torch.float32 if f32 else f16)
        a = torch.as_strided(flat, (M, K), (ld, 1))
        want = _ref(a, w, bias, act, r)
        tol = 1e-3 if f32 else 2e-3
        # (GELU into a half output: the packed-half polynomial's own error, <= ~1.5e-3 absolute, on top of the output rounding; fp32 outputs use the fp32 polynomial)
        torch.testing.assert_close(y.float(), want, atol=4e-3 if (act == 1 and not f32) else tol, rtol=tol)
        # and the format is what makes the difference: the same call on bf16-rounded operands is 4-8x further from the fp32 product of the ORIGINAL values
        if act == 0 and not res and not f32 and lda is None:
            yb = ops.gemm(a.to(torch.bfloat16).contiguous(), w.to(torch.bfloat16), bias, act, None)
            e16, eb = (y.float() - want).abs().mean().item(), (yb.float() - want).abs().mean().item()
            assert e16 * 4 < eb, (e16, eb)
    finally:
        lib().sc_debug_set_gemm_mode(-1)


@pytest.mark.parametrize("M,N,K", [(100, 1024, 1024), (4096, 520, 768), (300, 64, 128), (8192, 768, 768)])
@pytest.mark.parametrize("act,res,f32", [(0, False, False), (1, False, False), (0, True, False), (0, True, True), (1, False, True)])
def test_small_shapes_f16_take_the_128_row_kernel(M, N, K, act, res, f32):
    """Shapes outside gemm8p's rules (few rows, N % 256 != 0, < 128 tiles): gemm_bf16_kernel<.., F16 = true>."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    g = torch.Generator().manual_seed(M + N)
    a = (0.5 * torch.randn(M, K, generator=g)).to("cuda", f16)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).to("cuda", f16)
    bias = torch.randn(N, generator=g).cuda()
    r = torch.randn(M, N, generator=g).to("cuda", torch.float32 if f32 else f16) if res else None
    y = ops.gemm(a, w, bias, act, r, out_f32=f32)
    assert lib().sc_gemm_last_path() == 0
    tol = 1e-3 if f32 else 2e-3
    torch.testing.assert_close(y.float(), _ref(a, w, bias, act, r), atol=tol, rtol=tol)


def test_f16_gelu_epilogue_on_a_dense_input_grid():
    """fc1's epilogue in the half format: the packed-half polynomial with the last product taken from the fp32 pre-activation (gelu_poly2_x8<true>), RNE to half.
    One-hot products reproduce a dense grid of pre-activations exactly; the error against exact erf-GELU stays below 3e-3 absolute + the output's own
    half-ulp (the degree-4 polynomial's worst error, 2.4e-3 in the negative tail, measured; the bf16 epilogue's test bound is 4e-3 + bf16 rounding), and large arguments saturate: gelu(x) = x for x >= 6, -0 / 0 for x <= -6."""
    from speechclip_amd import ops
    from speechclip_amd._lib import lib
    lib().sc_debug_set_gemm_mode(16)
    try:
        M, N, K = 256 * 128, 256, 128
        xs16 = torch.linspace(-12.0, 12.0, M, dtype=torch.float64).to(f16)       # the grid, representable in half
        a = torch.zeros(M, K, dtype=f16)
        a[:, 0] = xs16                                                          # pre-activation of (row m, column n) = grid[m] * w[n, 0], exact in the fp32 accumulator
        w = torch.zeros(N, K, dtype=f16)
        w[:, 0] = 1.0
        w[N // 2:, 0] = 0.5
        y = ops.gemm(a.cuda(), w.cuda(), None, 1, None)
        assert lib().sc_gemm_last_path() == 3
        x = xs16.double()[:, None] * w[:, 0].double()[None, :]
        want = 0.5 * x * (1.0 + torch.erf(x / 2 ** 0.5))
        excess = (y.double().cpu() - want).abs() - want.abs() * 2.0 ** -11            # beyond the half-ulp of the output format itself
        assert excess.max().item() < 3e-3, excess.max().item()
        big = x >= 6.0
        assert torch.equal(y.cpu()[big].double(), x[big].to(f16).double())
        assert (y.cpu()[x <= -6.0] == 0).all()
    finally:
        lib().sc_debug_set_gemm_mode(-1)


@pytest.mark.parametrize("rows,D", [(4001, 1024), (777, 768), (130, 64)])
def test_layernorm_f32_to_f16(rows, D):
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(rows)
    x = (3.0 * torch.randn(rows, D, generator=g) + 0.7).cuda()
    gam, bet = (1 + 0.2 * torch.randn(D, generator=g)).cuda(), (0.3 * torch.randn(D, generator=g)).cuda()
    y = ops.layernorm(x, gam, bet, out=torch.empty(rows, D, device="cuda", dtype=f16))
    want = torch.nn.functional.layer_norm(x, (D,), gam, bet)
    torch.testing.assert_close(y.float(), want, atol=2e-3, rtol=1e-3)
    assert torch.equal(y, want.to(f16)) or (y.float() - want.to(f16).float()).abs().max().item() <= 4e-3      # the same values up to a last-place rounding


@pytest.mark.parametrize("B,T,H,ragged", [(3, 499, 16, True), (5, 100, 4, True), (2, 300, 16, False)])
def test_attention_f16_vs_fp32(B, T, H, ragged):
    """attn_fwd_kernel<.., F16 = true>, padded and packed entry: fairseq MultiheadAttention with a key-padding mask on half q|k|v."""
    from speechclip_amd import ops
    g = torch.Generator().manual_seed(B * T)
    D = H * 64
    qkv = (torch.randn(B * T, 3 * D, generator=g)).to("cuda", f16)
    lens = [T] * B
    if ragged:
        lens = [max(1, T - 37 * i) for i in range(B)]
    kl = torch.tensor(lens, dtype=torch.int32, device="cuda")
    out = ops.attention(qkv, B, T, H, kl)
    assert out.dtype == f16
    q, k, v = (qkv.float().view(B, T, 3, H, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    s = (q @ k.transpose(-1, -2)) * 0.125
    mask = torch.arange(T, device="cuda")[None, :] >= kl[:, None]
    s = s.masked_fill(mask[:, None, None, :], float("-inf"))
    want = (torch.softmax(s, -1) @ v).permute(0, 2, 1, 3).reshape(B * T, D)
    for b in range(B):
        rows = slice(b * T, b * T + lens[b])
        torch.testing.assert_close(out[rows].float(), want[rows], atol=2e-3, rtol=2e-3)
    # packed rows: utterance b owns lens[b] + 1 rows
    rows_b = [n + 1 for n in lens]
    off = [0]
    for r in rows_b:
        off.append(off[-1] + r)
    packed = torch.cat([qkv[b * T:b * T + min(rows_b[b], T)] if rows_b[b] <= T else torch.cat([qkv[b * T:(b + 1) * T], qkv[:1]]) for b in range(B)])
    off_t = torch.tensor(off, dtype=torch.int32, device="cuda")
    outp = ops.attention_packed(packed.contiguous(), B, max(rows_b), H, kl, off_t)
    assert outp.dtype == f16
    for b in range(B):
        torch.testing.assert_close(outp[off[b]:off[b] + lens[b]].float(), want[b * T:b * T + lens[b]], atol=2e-3, rtol=2e-3)


def test_pre_ln_layers_in_f16_are_closer_to_the_fp32_oracle_than_in_bf16(monkeypatch):
    """Six pre-LN layers at HuBERT-large's widths (d = 1024, ffn 4096, 16 heads) on a ragged batch.  The layers' OWN error is isolated by running the fp32 oracle's
    layers on the engine's hidden[0] (the conv stack and the positional conv in front of it stay bf16 in both runs): with half operands the last hidden state is at
    least 4x closer to it than with bf16 operands (SC_PRELN_F16=0) -- the mechanism behind test_p_large_b64_ragged_vs_oracle's floor."""
    import dataclasses
    from oracle.hubert_ref import HubertModelRef, HubertRefConfig, randomize_norm_affine
    from speechclip_amd.module import hubert as hb
    rc = dataclasses.replace(HubertRefConfig.large(), encoder_layers=6)
    cfg = dataclasses.replace(hb.HubertConfig.from_name("hubert_large_ll60k"), encoder_layers=6)
    torch.manual_seed(5)
    ref = HubertModelRef(rc).eval()
    randomize_norm_affine(ref, torch.Generator().manual_seed(6))
    g = torch.Generator().manual_seed(7)
    lens = [48000, 31000, 40000, 22050]
    wav = torch.zeros(len(lens), max(lens))
    for i, n in enumerate(lens):
        wav[i, :n] = 0.1 * torch.randn(n, generator=g) + 0.01
    errs = {}
    for mode in (True, False):
        monkeypatch.setattr(hb, "_PRELN_F16", mode)
        m = hb.HubertModel(cfg)
        m.load_state_dict(ref.state_dict())
        m = m.cuda().eval()
        hidden, T, Tp, valid = m.extract_all_layers(wav.cuda(), lens)
        assert m._packed["layer_dtype"] == (f16 if mode else torch.bfloat16) and hidden.dtype == torch.float32
        h0, last = hidden[0, :, :T].float().cpu(), hidden[-1, :, :T].float().cpu()
        pad = torch.arange(T)[None, :] >= torch.tensor(valid)[:, None]
        x = h0.transpose(0, 1)                                                          # [T, B, d], as speech_encoder_plus.py:45
        with torch.no_grad():
            for layer in ref.encoder.layers:
                x, _ = layer(x, self_attn_padding_mask=pad, need_weights=False)
        want = x.transpose(0, 1)
        num = den = 0.0
        for b in range(len(lens)):
            num += (last[b, :valid[b]] - want[b, :valid[b]]).pow(2).sum().item()
            den += want[b, :valid[b]].pow(2).sum().item()
        errs[mode] = (num / den) ** 0.5
    print("relative error of six pre-LN layers, half / bf16 operands:", errs)
    assert errs[True] * 4 < errs[False], errs
    assert errs[True] < 1e-3, errs
