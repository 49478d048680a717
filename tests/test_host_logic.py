"""CPU tests (no GPU, no /root/reference): host-side logic of the product, the C-ABI surface, and the N > 1 exchange step
over gloo with world_size 2."""
import ctypes
import os
import pickle
import socket
from argparse import Namespace
from collections import OrderedDict
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from helpers import make_config

GOLD = os.path.join(os.path.dirname(__file__), "golden")


# --------------------------------------------------------------------------------------------------- C ABI
def test_abi_library_loads_and_exports_every_declared_symbol():
    from speechclip_amd import _lib
    L = _lib.lib()                       # raises if the .so is missing: no fallback
    names = _lib.exported_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/speechclip_hip.h but not exported"
    assert L.sc_abi_version() == 1
    assert isinstance(L.sc_last_error(), bytes)


def test_abi_argument_errors_are_codes_not_crashes():
    """Bad arguments return a negative code + message before any launch (no GPU needed)."""
    from speechclip_amd import _lib
    L = _lib.lib()
    rc = L.sc_gemm_bf16(None, 0, None, 0, None, 0, None, None, 0, 16, 16, 48, 0, None)      # K not a multiple of 64
    assert rc < 0 and b"K=48" in L.sc_last_error()
    rc = L.sc_attention_fwd(None, None, None, None, None, 1, 1, 8, 96, 8, 8, ctypes.c_float(1.0), 0, None)
    assert rc < 0 and b"head_dim" in L.sc_last_error()
    rc = L.sc_layernorm(None, 0, None, None, None, 0, 4, 2048, ctypes.c_float(1e-5), 0, None)
    assert rc < 0 and b"D=2048" in L.sc_last_error()


def test_product_refuses_cpu_tensors():
    from speechclip_amd import ops
    from speechclip_amd._lib import SpeechClipHipError
    with pytest.raises(SpeechClipHipError):
        ops.gemm(torch.zeros(4, 64, dtype=torch.bfloat16), torch.zeros(8, 64, dtype=torch.bfloat16))


# --------------------------------------------------------------------------------------------------- config object
def test_ordered_namespace_semantics():
    """Behaviour pinned by the reference's test/test_dict.py (restated)."""
    from speechclip_amd.base import OrderedNamespace
    d = {"a": 1, "b": [2, {"c": 3}], "d": {"e": 4, "f": "g"}, "h": SimpleNamespace(i=5, j={"k": 6})}
    x, y = OrderedNamespace(d), OrderedNamespace(**d)
    assert x.a == x["a"] == 1 and x.b[0] == 2 and x.b[1].c == 3 and x.d.e == x["d"]["e"] == x.d["e"] == 4
    assert x.h.i == 5 and x.h.j.k == 6 and x == y and len(x) == 4 and "a" in x and len(x.keys()) == 4
    assert isinstance(x.pydict, dict) and isinstance(x.odict, OrderedDict) and x.to_dict() == x.pydict and x.keys() == x.pydict.keys()
    assert OrderedNamespace({"a": 1, "b": 2}) == OrderedNamespace(SimpleNamespace(a=1, b=2)) == OrderedNamespace(Namespace(a=1, b=2))
    m = OrderedNamespace([{"a": 1, "c": {"d": 3}}, Namespace(e=4, f=SimpleNamespace(g=5))])
    assert m.a == 1 and m.c.d == 3 and m.e == 4 and m.f.g == 5 and m.f["g"] == 5
    assert m.get("zz", 7) == 7 and not hasattr(m, "zz")
    m.new = 3
    assert m["new"] == 3 and dict(**m.c) == {"d": 3}
    z = pickle.loads(pickle.dumps(x))
    assert z == x and z.d.f == "g"
    import avssl.base
    assert avssl.base.OrderedNamespace is OrderedNamespace


def test_keypadding_mask_golden():
    from speechclip_amd.util import get_keypadding_mask
    g = np.load(os.path.join(GOLD, "small_ops.npz"))
    m = get_keypadding_mask(10, torch.from_numpy(g["kpm_lens"]))
    assert m.dtype == torch.bool and np.array_equal(m.numpy(), g["kpm_mask"])


# --------------------------------------------------------------------------------------------------- HuBERT host geometry
def test_frame_geometry_and_masks_match_reference_table():
    """feat_len (round-half-even, clamp T) and the fairseq chunk-`all` frame mask, against values produced by the reference glue."""
    from speechclip_amd.module.hubert import HubertConfig, HubertModel
    tab = np.load(os.path.join(GOLD, "feat_len.npz"))["table"]
    m = HubertModel(HubertConfig(encoder_layers=1, encoder_embed_dim=64, encoder_ffn_embed_dim=64, encoder_attention_heads=1,
                                 conv_layers=[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2, conv_pos=16, conv_pos_groups=4))
    for lmax, l, T, flen, nvalid in tab.tolist():
        T0, Tg, P0, Tp = m.frame_geometry(lmax)
        assert Tg == T and Tp >= T and P0 % 64 == 0 and P0 >= T0 and P0 // 64 == Tp
        assert m.valid_frames([l], lmax, T)[0] == nvalid
        assert min(round(l / 320), T) == flen


def test_state_dict_keys_match_fairseq_and_openai_names():
    """Same key sets as the oracle's fairseq-/openai-named restatements => reference checkpoints load by name."""
    import dataclasses
    from oracle.clip_ref import ClipRef, ClipRefConfig
    from oracle.hubert_ref import HubertModelRef, HubertRefConfig
    from speechclip_amd.module.clip_model import CLIP, ClipConfig
    from speechclip_amd.module.hubert import HubertConfig, HubertModel
    for large in (False, True):
        rc = HubertRefConfig.tiny(layer_norm_first=large, extractor_mode="layer_norm" if large else "default", conv_bias=large)
        ours = HubertModel(HubertConfig(**dataclasses.asdict(rc))).state_dict()
        ref = HubertModelRef(rc).state_dict()
        assert set(ours) == set(ref)
        assert all(ours[k].shape == ref[k].shape for k in ref)
    cc = ClipRefConfig.tiny()
    ours, ref = CLIP(ClipConfig(**dataclasses.asdict(cc))).state_dict(), ClipRef(cc).state_dict()
    assert set(ours) == set(ref) and all(ours[k].shape == ref[k].shape for k in ref)


def test_model_surface_and_checkpoint_keys():
    from speechclip_amd.model import KWClip_GeneralTransformer
    import dataclasses
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    hc = HubertConfig(**dataclasses.asdict(HubertRefConfig.tiny()))
    cc = ClipConfig(**dataclasses.asdict(ClipRefConfig.tiny()))
    model = KWClip_GeneralTransformer(make_config(d_model=128, branch_heads=4, hubert_config=hc, clip_config=cc, parallel=True, cascaded=True))
    keys = set(model.state_dict())
    for k in ("audio_encoder.encoder.feature_extractor.conv_layers.0.0.weight", "audio_encoder.encoder.feature_extractor.conv_layers.0.2.weight",
              "audio_encoder.encoder.encoder.pos_conv.0.weight_g", "audio_encoder.encoder.encoder.layers.1.self_attn.q_proj.weight",
              "audio_encoder.weightedsum_layer.weights", "clip.model.visual.transformer.resblocks.0.attn.in_proj_weight",
              "clip.model.visual.proj", "criterion.eye_mat", "parallel_branch.cls", "parallel_branch.self_att.model.layers.0.self_attn.in_proj_weight",
              "parallel_branch.self_att.model.norm.weight", "parallel_branch.linear_proj.bias", "cascaded_branch.cls",
              "cascaded_branch.self_att.multihead_attn_layer.in_proj_weight", "cascaded_branch.self_att.attentionBlock_Norm.weight",
              "cascaded_branch.bn_layer.bn_layer.running_mean", "cascaded_branch.vector_quantizer.curr_temp",
              "cascaded_branch.clip.model.ln_final.weight"):
        assert k in keys, k
    assert model.cascaded_branch.cls.shape == (1, 8, 128) and model.parallel_branch.cls.shape == (1, 1, 128)
    for name in ("forward", "training_step", "training_step_end", "validation_step", "validation_step_end", "validation_epoch_end",
                 "compute_loss", "encode_speech", "feature_extractor_s3prl", "forward_audio", "forward_image", "getTrainableParams",
                 "configure_optimizers"):
        assert callable(getattr(model, name))
    with pytest.raises(ValueError):
        model.forward_image(torch.zeros(2, 4, 8, 8))
    with pytest.raises(TypeError):
        model.forward_image("not a tensor")
    opt, sched = model.configure_optimizers()
    n_train = sum(p.numel() for p in model.getTrainableParams())
    assert len(opt) == 1 and sched[0]["interval"] == "step" and n_train > 0
    assert all(not p.requires_grad for p in model.audio_encoder.encoder.parameters())


def _tiny_model(parallel=True, cascaded=True):
    from speechclip_amd.model import KWClip_GeneralTransformer
    import dataclasses
    from oracle.clip_ref import ClipRefConfig
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.module.hubert import HubertConfig
    hc = HubertConfig(**dataclasses.asdict(HubertRefConfig.tiny()))
    cc = ClipConfig(**dataclasses.asdict(ClipRefConfig.tiny()))
    return KWClip_GeneralTransformer(make_config(d_model=128, branch_heads=4, hubert_config=hc, clip_config=cc, parallel=parallel, cascaded=cascaded))


@pytest.mark.parametrize("drop_dup,drop_own", [(False, False), (True, False), (False, True)])
def test_checkpoint_round_trip_with_cascaded_branch(tmp_path, drop_dup, drop_own):
    """reference checkpoints of cascaded models carry the CLIP tower twice (`clip.*` and `cascaded_branch.clip.*`,
    kwClip.py:720).  save -> load_from_checkpoint must work for the full key set and when either spelling is absent."""
    from speechclip_amd.model import KWClip_GeneralTransformer
    torch.manual_seed(3)
    model = _tiny_model()
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    assert any(k.startswith("cascaded_branch.clip.") for k in sd)
    if drop_dup:
        sd = {k: v for k, v in sd.items() if not k.startswith("cascaded_branch.clip.")}
    if drop_own:
        sd = {k: v for k, v in sd.items() if not k.startswith("clip.")}
    path = str(tmp_path / "ckpt.pt")
    # Lightning 1.5 layout (base_task.py:176-193): `callbacks` is keyed by the ModelCheckpoint CLASS and holds Lightning objects -- none of it is
    # importable where the checkpoint is read; util/checkpoint_io.py's restricted unpickler stubs them
    import importlib
    import sys
    pkg = tmp_path / "fakepl"
    (pkg / "fake_lightning").mkdir(parents=True)
    (pkg / "fake_lightning" / "__init__.py").write_text("class ModelCheckpoint:\n    def __init__(self):\n        self.best_model_score = 0.25\n")
    sys.path.insert(0, str(pkg))
    try:
        fl = importlib.import_module("fake_lightning")
        torch.save({"state_dict": sd, "hyper_parameters": {"config": model.config}, "epoch": 3, "global_step": 77, "pytorch-lightning_version": "1.5.10",
                    "callbacks": {fl.ModelCheckpoint: {"monitor": "val_loss", "best": fl.ModelCheckpoint()}}, "optimizer_states": [], "lr_schedulers": []}, path)
    finally:
        sys.path.remove(str(pkg))
        sys.modules.pop("fake_lightning", None)
    torch.manual_seed(4)                                   # different init: the load has to overwrite everything
    back = KWClip_GeneralTransformer.load_from_checkpoint(path)
    ref = model.state_dict()
    for k, v in back.state_dict().items():
        assert torch.equal(v, ref[k]), k
    assert back.cascaded_branch.clip is back.clip
    # a parallel-only model accepts a checkpoint that still has the duplicate tower; an unknown key is an error
    ponly = _tiny_model(cascaded=False)
    from speechclip_amd.model.base_model import load_checkpoint_state
    psd = {k: v for k, v in ponly.state_dict().items()}
    psd.update({k: v for k, v in sd.items() if k.startswith("cascaded_branch.clip.")})
    load_checkpoint_state(ponly, psd, strict=True)
    with pytest.raises(RuntimeError):
        load_checkpoint_state(ponly, dict(psd, **{"parallel_branch.not_a_weight": torch.zeros(1)}), strict=True)


def test_mutual_retrieval_has_no_host_fallback():
    """mutualRetrieval (retrieval.py:6-121) has ONE implementation, sc_retrieval_ranks on the device (golden values: tests/test_kernels_gpu.py::
    test_retrieval_ranks_golden_and_random, also with host score matrices).  Without a GPU it raises instead of re-implementing the ranks on the host."""
    from speechclip_amd._lib import SpeechClipHipError
    from speechclip_amd.module import mutualRetrieval
    if torch.cuda.is_available():
        pytest.skip("GPU present: the device path is exercised by the gpu suite")
    g = np.load(os.path.join(GOLD, "retrieval.npz"))
    aud, img = torch.from_numpy(g["aud"]), torch.from_numpy(g["img"])
    s = aud @ img.t()
    with pytest.raises(SpeechClipHipError):
        mutualRetrieval(s, s.t().contiguous(), torch.from_numpy(g["aud_ids"]), torch.from_numpy(g["img_ids"]), [1, 5, 10])


def test_bench_flop_model_matches_baseline_md():
    import bench
    total, gemm = bench.algorithmic_gflop_per_pair()
    assert abs(total - 158.1) < 0.2        # BASELINE.md section 2, Parallel base at 10 s
    assert bench.conv_lens(160000)[-1] == 499 and bench.conv_lens(102400)[-1] == 319 and bench.conv_lens(16000)[-1] == 49


# --------------------------------------------------------------------------------------------------- N > 1 exchange step
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.speechclip_ref import masked_contrastive_loss
    from speechclip_amd import parallel
    g = torch.Generator().manual_seed(100 + rank)
    B, E = 5, 16
    feats = {"id": torch.tensor([7, 7, 3 + rank, 2 ** 40 + rank, -1 - rank]), "image_feat": torch.randn(B, E, generator=g),
             "parallel_audio_feat": torch.randn(B, E, generator=g)}
    out = parallel.gather_loss_feats(feats)
    loss = masked_contrastive_loss(torch.nn.functional.normalize(out["parallel_audio_feat"], dim=-1),
                                   torch.nn.functional.normalize(out["image_feat"], dim=-1), out["id"]).item()
    # numpy payloads: pickled by value (torch tensors travel as shared-memory handles that die with this process)
    q.put((rank, {k: v.numpy().copy() for k, v in out.items()}, {k: v.numpy().copy() for k, v in feats.items()}, loss))
    dist.destroy_process_group()


def test_gather_loss_feats_gloo_world2():
    """Rank-major concat of every feature + bit-exact int64 ids over ONE collective; every rank sees the same global batch
    (= DataParallel's dim-0 gather order), hence the same loss."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    (_, out0, loc0, loss0), (_, out1, loc1, loss1) = res
    out0, out1, loc0, loc1 = [{k: torch.from_numpy(v) for k, v in d.items()} for d in (out0, out1, loc0, loc1)]
    for k in ("id", "image_feat", "parallel_audio_feat"):
        assert torch.equal(out0[k], out1[k])
        assert torch.equal(out0[k], torch.cat([loc0[k], loc1[k]], 0))
    assert out0["id"].dtype == torch.int64 and abs(loss0 - loss1) < 1e-7


# --------------------------------------------------------------------------------------------------- training tail, host side
def test_lr_schedulers_match_reference_values():
    """get_scheduler reproduces the reference's LambdaLR curves (golden: avssl/optim/scheduler.py run by make_golden.py)."""
    from speechclip_amd.optim import get_scheduler
    g = np.load(os.path.join(GOLD, "small_ops.npz"))
    steps = g["sched_steps"].tolist()
    for name, kw in (("linear_warmup_decay", dict(warmup=10, max_step=100, final_lr=1e-8)), ("noam", dict(warmup=10))):
        prm = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([prm], lr=1e-4)
        sch = get_scheduler(name, opt, **kw)
        lrs = []
        for k in range(max(steps) + 1):
            if k in steps:
                lrs.append(opt.param_groups[0]["lr"])
            opt.step()
            sch.step()
        assert np.allclose(np.array(lrs), g[f"sched_{name}_lr"], rtol=1e-12, atol=0), (name, lrs, g[f"sched_{name}_lr"])


def _train_gather_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.speechclip_ref import masked_contrastive_loss
    from speechclip_amd.train_tail import gather_loss_feats_train
    g = torch.Generator().manual_seed(100 + rank)
    B, E = 5, 16
    a = torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1).requires_grad_(True)
    feats = {"id": torch.tensor([7, 7, 3 + rank, 2 ** 40 + rank, -1 - rank]), "image_feat": torch.nn.functional.normalize(torch.randn(B, E, generator=g), dim=-1),
             "parallel_audio_feat": a}
    out = gather_loss_feats_train(feats)
    loss = masked_contrastive_loss(out["parallel_audio_feat"], out["image_feat"], out["id"])
    loss.backward()
    q.put((rank, a.detach().numpy().copy(), feats["image_feat"].numpy().copy(), feats["id"].numpy().copy(), a.grad.numpy().copy(), loss.item()))
    dist.destroy_process_group()


def test_training_gather_backward_keeps_local_rows_gloo_world2():
    """Every rank evaluates the same global loss on the all-gathered batch; the gradient of its local audio features must be its slice
    of the single-process gradient (the reference's DP gather on device 0, kwClip.py:147-191)."""
    import torch.multiprocessing as mp
    from oracle.speechclip_ref import masked_contrastive_loss
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    res = [tuple(torch.from_numpy(x) if isinstance(x, np.ndarray) else x for x in r) for r in res]
    A = torch.cat([r[1] for r in res]).requires_grad_(True)
    Bm, ids = torch.cat([r[2] for r in res]), torch.cat([r[3] for r in res])
    loss = masked_contrastive_loss(A, Bm, ids)
    loss.backward()
    assert abs(res[0][5] - loss.item()) < 1e-6 and abs(res[1][5] - loss.item()) < 1e-6
    assert torch.allclose(torch.cat([r[4] for r in res]), A.grad, atol=1e-6)


def _one_collective_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechclip_amd import parallel
    from speechclip_amd.train_tail import gather_loss_feats_train
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    g = torch.Generator().manual_seed(7 + rank)
    B = 3
    feats = {"id": torch.tensor([rank, 5, 2 ** 35 + rank]), "image_feat": torch.randn(B, 8, generator=g),
             "parallel_audio_feat": torch.randn(B, 8, generator=g).requires_grad_(True),
             "cascaded_audio_feat": torch.randn(B, 8, generator=g).requires_grad_(True)}
    out = gather_loss_feats_train(feats)
    n_train = len(calls)
    assert not out["image_feat"].requires_grad and not out["id"].requires_grad          # a frozen tower's features stay non-differentiable
    assert out["parallel_audio_feat"].requires_grad and out["cascaded_audio_feat"].requires_grad
    (out["parallel_audio_feat"].sum() * 2 + out["cascaded_audio_feat"][rank * B:(rank + 1) * B].sum() * 3).backward()
    ok_grad = bool(torch.all(feats["parallel_audio_feat"].grad == 2) and torch.all(feats["cascaded_audio_feat"].grad == 3))
    calls.clear()
    others = {"id": feats["id"], "audio_feat": feats["parallel_audio_feat"].detach(), "image_feat": feats["image_feat"],
              "keywords": torch.randn(B, 2, 4, generator=g), "gold_text": torch.arange(B * 5).view(B, 1, 5) + 100 * rank, "note": "x",
              "row_score": torch.randn(B, generator=g),         # a NAMED 1-D per-row float tensor travels in the packed collective too
              "scalar_like": torch.randn(B, generator=g)}       # ... an unnamed 1-D float of length B does not (a [1] entry at B == 1)
    go = parallel.gather_rows_dict(others)
    q.put((rank, n_train, ok_grad, len(calls), {k: (v.numpy().copy() if torch.is_tensor(v) else v) for k, v in go.items()},
           {k: (v.numpy().copy() if torch.is_tensor(v) else v) for k, v in others.items()},
           {k: v.detach().numpy().copy() for k, v in out.items()}))
    dist.destroy_process_group()


def test_train_gather_is_one_collective_and_validation_outputs_are_gathered_gloo_world2():
    """VERDICT r1 weak #12 / ADVICE r1 (medium): the training-time gather is ONE packed all-gather (all float features + bit-cast ids), and
    validation_step_end's `others` are gathered rank-major over all ranks so validation_epoch_end ranks against the full candidate pool."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_one_collective_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for rank, n_train, ok_grad, n_val, go, loc, out in res:
        assert n_train == 1, n_train                       # one collective for ids + every feature
        assert ok_grad
        assert n_val == 3, n_val                           # the local row counts, packed floats+ids, and the one integer tensor (gold_text)
        assert go["note"] == "x"
    for k in ("id", "audio_feat", "image_feat", "keywords", "gold_text", "row_score"):
        want = np.concatenate([res[0][5][k], res[1][5][k]], 0)
        for r in res:
            assert r[4][k].dtype == want.dtype and np.array_equal(r[4][k], want), k
    for r in res:
        assert np.array_equal(r[4]["scalar_like"], r[5]["scalar_like"])      # passed through, not gathered
    for k in ("id", "image_feat", "parallel_audio_feat", "cascaded_audio_feat"):
        assert np.array_equal(res[0][6][k], res[1][6][k])


def _ragged_gather_worker(rank, world, port, q, rows=(3, 2)):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechclip_amd import parallel
    g = torch.Generator().manual_seed(11 + rank)
    B = rows[rank]                                           # DataParallel's scatter of a 5-row last batch over 2 replicas: 3 + 2
    others = {"id": torch.arange(B) + 10 * rank + 2 ** 33, "audio_feat": torch.randn(B, 6, generator=g), "image_feat": torch.randn(B, 6, generator=g),
              "keywords": torch.randn(B, 2, 3, generator=g), "gold_text": torch.arange(B * 4).view(B, 1, 4) + 100 * rank, "row_score": torch.randn(B, generator=g)}
    go = parallel.gather_rows_dict(others)
    q.put((rank, {k: v.numpy().copy() for k, v in go.items()}, {k: v.numpy().copy() for k, v in others.items()}))
    dist.destroy_process_group()


def test_validation_gather_accepts_a_ragged_last_batch_gloo_world2():
    """the reference's DataParallel scatter accepts a last validation batch that does not divide by the replica count
    (kwClip.py:193-269); gather_rows_dict used to raise.  Ranks with 3 and 2 rows: every rank ends up with the 5 rows in rank-major order, pad rows gone,
    ids bit-exact beyond 2^32."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_gather_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for k in ("id", "audio_feat", "image_feat", "keywords", "gold_text", "row_score"):
        want = np.concatenate([res[0][2][k], res[1][2][k]], 0)
        assert want.shape[0] == 5
        for r in res:
            assert r[1][k].dtype == want.dtype and np.array_equal(r[1][k], want), k


def test_validation_gather_accepts_an_empty_rank_gloo_world2():
    """a rank whose last validation batch is EMPTY (2 rows over 2 replicas scatter as 2 + 0 when the loader's shard runs dry) must not raise
    behind the row-count collective (the other rank would block in the packed gather): the gather returns the 2 rows on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ragged_gather_worker, args=(r, 2, port, q, (2, 0))) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    for k in ("id", "audio_feat", "image_feat", "keywords", "gold_text", "row_score"):
        want = res[0][2][k]
        assert want.shape[0] == 2
        for r in res:
            assert r[1][k].dtype == want.dtype and np.array_equal(r[1][k], want), k


def _step_end_worker(rank, world, port, q, train):
    """One rank of the 8-rank global-batch-2048 protocol test: the REAL training_step_end / validation_step_end + compute_loss of the model class
    on this rank's 256-row shard of the golden B = 2048 fixture.  No GPU here, so the criterion is the fp32 oracle's masked InfoNCE standing in
    for sc_infonce_fwd (tests/test_kernels_gpu.py::test_infonce_golden pins the kernel on the same fixture at Bg = 2048)."""
    import types
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from oracle.speechclip_ref import masked_contrastive_loss
    from speechclip_amd.base import OrderedNamespace
    from speechclip_amd.model.kwClip import KWClip_GeneralTransformer, KWClipBase
    g = np.load(os.path.join(GOLD, "loss.npz"))
    Bg = 2048
    B = Bg // world
    sl = slice(rank * B, (rank + 1) * B)
    a = torch.from_numpy(g["fa_2048"][sl]).clone().requires_grad_(train)
    feats = {"id": torch.from_numpy(g["ids_2048"][sl]).clone(), "image_feat": torch.from_numpy(g["fb_2048"][sl]).clone(), "parallel_audio_feat": a}
    m = types.SimpleNamespace()
    m.config = OrderedNamespace({"model_settings": {"cascaded_objective_weight": 0.0, "parallel_objective_weight": 1.0}})
    m.criterion = lambda feat_A, feat_B, index: masked_contrastive_loss(feat_A, feat_B, index)
    m.compute_loss = types.MethodType(KWClip_GeneralTransformer.compute_loss, m)
    m._reduce_metrics = types.MethodType(KWClipBase._reduce_metrics, m)
    logged = {}
    m.log_dict = lambda d, **kw: logged.update({k: float(v) for k, v in d.items()})
    if train:
        out = KWClipBase.training_step_end(m, {"loss_feats": feats, "log_metrics": {"cl_temp": 1 / 0.07}})
        out["loss"].backward()
        q.put((rank, float(out["loss"]), a.grad.numpy().copy(), logged))
    else:
        others = {"id": feats["id"], "audio_feat": feats["parallel_audio_feat"], "image_feat": feats["image_feat"]}
        got = KWClipBase.validation_step_end(m, {"loss_feats": feats, "log_metrics": {"cl_temp": 1 / 0.07}, "others": others})
        q.put((rank, logged["val_loss"], {k: v.numpy().copy() for k, v in got.items()}, logged))
    dist.destroy_process_group()


@pytest.mark.parametrize("train", [True, False])
def test_step_end_hooks_gloo_world8_global_batch_2048(train):
    """BASELINE.json configs[3] protocol without the 8-GPU node: 8 ranks x 256 pairs with DUPLICATE ids run the model's own
    training_step_end / validation_step_end; every rank must see the loss of the single-process computation on the rank-major concatenation
    (= nn.DataParallel's dim-0 gather, kwClip.py:147-191), which is the value the REFERENCE's MaskedContrastiveLoss produced for this very
    batch (tests/golden/loss.npz B = 2048, MAX_EYE patched: losses.py:126,211); in training the local rows' gradient is the rank's slice of the
    single-process gradient; in validation every rank ends up with all 2048 rows in rank-major order."""
    import torch.multiprocessing as mp
    from oracle.speechclip_ref import masked_contrastive_loss
    g = np.load(os.path.join(GOLD, "loss.npz"))
    golden = [c[8] for c in g["cases"] if int(c[0]) == 2048 and c[2] == 0 and c[3] == 0 and c[4] == 1 and c[5] == 1 and abs(c[6] - 1 / 0.07) < 1e-6 and c[7] == 1]
    assert len(golden) == 1
    ids = g["ids_2048"]
    assert len(np.unique(ids)) < len(ids)                      # duplicate ids: the false-negative mask matters
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_step_end_worker, args=(r, world, port, q, train)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=900) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=300)
        assert p.exitcode == 0
    A = torch.from_numpy(g["fa_2048"]).clone().requires_grad_(True)
    Bm, idt = torch.from_numpy(g["fb_2048"]), torch.from_numpy(ids)
    single = masked_contrastive_loss(A, Bm, idt)
    assert abs(single.item() - golden[0]) < 2e-5 * max(1.0, abs(golden[0]))
    for r in res:
        assert abs(r[1] - golden[0]) < 2e-5 * max(1.0, abs(golden[0])), (r[0], r[1], golden[0])      # every rank: the reference's global-batch loss
        assert abs(r[3]["train_loss" if train else "val_loss"] - r[1]) < 1e-6 and abs(r[3][("train" if train else "val") + "_cl_temp"] - 1 / 0.07) < 1e-4
    if train:
        single.backward()
        got = torch.cat([torch.from_numpy(r[2]) for r in res])
        # every rank differentiates the SAME global loss; the optimizer sums the ranks' parameter gradients (FusedAdam all-reduce), so a rank's
        # feature gradient is its row slice of the single-process gradient
        assert torch.allclose(got, A.grad, atol=1e-7, rtol=1e-5)
    else:
        for r in res:
            assert np.array_equal(r[2]["id"], ids) and np.array_equal(r[2]["audio_feat"], g["fa_2048"]) and np.array_equal(r[2]["image_feat"], g["fb_2048"])


def _fairseq_style_hubert_file(tmp_path, hcfg, sd, parametrized=False, cfg_as="dict"):
    """A checkpoint in fairseq's own layout (checkpoint_utils.save_state [3P]): cfg = {model, task, ...} dict (or the legacy `args` Namespace), `model`
    state_dict incl. the pre-training head, and a `task_state` holding an object of a class from a module that is NOT importable when the file is
    read (fairseq.data.dictionary.Dictionary in the real files)."""
    import argparse
    import importlib
    import sys
    pkg = tmp_path / "fakeseq_pkg"
    (pkg / "fakeseq" / "data").mkdir(parents=True, exist_ok=True)
    (pkg / "fakeseq" / "__init__.py").write_text("")
    (pkg / "fakeseq" / "data" / "__init__.py").write_text("")
    (pkg / "fakeseq" / "data" / "dictionary.py").write_text(
        "import enum\nclass Dictionary:\n    def __init__(self):\n        self.symbols = ['<s>', '</s>', 'a']\n        self.indices = {'a': 2}\n"
        "class ChoiceEnum(enum.Enum):\n    default = 'default'\n    layer_norm = 'layer_norm'\n")
    sys.path.insert(0, str(pkg))
    try:
        mod = importlib.import_module("fakeseq.data.dictionary")
        model_cfg = {"_name": "hubert", "extractor_mode": hcfg.extractor_mode if cfg_as == "dict" else mod.ChoiceEnum(hcfg.extractor_mode),
                     "conv_bias": hcfg.conv_bias, "layer_norm_first": hcfg.layer_norm_first, "encoder_layers": hcfg.encoder_layers,
                     "encoder_embed_dim": hcfg.encoder_embed_dim, "encoder_ffn_embed_dim": hcfg.encoder_ffn_embed_dim, "encoder_attention_heads": hcfg.encoder_attention_heads,
                     "conv_feature_layers": "[(%d,10,5)] + [(%d,3,2)] * 4 + [(%d,2,2)] * 2" % ((hcfg.conv_layers[0][0],) * 3), "conv_pos": hcfg.conv_pos,
                     "conv_pos_groups": hcfg.conv_pos_groups, "dropout": 0.1, "attention_dropout": 0.1, "activation_dropout": 0.0, "dropout_input": 0.1,
                     "encoder_layerdrop": 0.05, "feature_grad_mult": 0.1, "final_dim": 256, "untie_final_proj": False}
        task_cfg = {"_name": "hubert_pretraining", "normalize": hcfg.normalize, "sample_rate": 16000, "labels": ["km"]}
        sd = dict(sd)
        if parametrized:
            sd["encoder.pos_conv.0.parametrizations.weight.original0"] = sd.pop("encoder.pos_conv.0.weight_g")
            sd["encoder.pos_conv.0.parametrizations.weight.original1"] = sd.pop("encoder.pos_conv.0.weight_v")
        ck = {"model": sd, "task_state": {"dictionaries": [mod.Dictionary()]}, "optimizer_history": [{"criterion_name": "HubertCriterion"}],
              "extra_state": {"epoch": 3}, "last_optimizer_state": None}
        if cfg_as == "args":
            ck.update(cfg=None, args=argparse.Namespace(**model_cfg, **{k: v for k, v in task_cfg.items() if k != "_name"}))
        else:
            ck.update(cfg={"_name": None, "common": {"seed": 1}, "model": model_cfg, "task": task_cfg, "criterion": {"_name": "hubert"}}, args=None)
        path = tmp_path / ("hubert_%s_%s.pt" % (cfg_as, "par" if parametrized else "wn"))
        torch.save(ck, path)
    finally:
        sys.path.remove(str(pkg))
        for k in [k for k in sys.modules if k == "fakeseq" or k.startswith("fakeseq.")]:
            del sys.modules[k]
    return path


@pytest.mark.parametrize("large_like,parametrized,cfg_as", [(False, False, "dict"), (True, True, "dict"), (True, False, "args")])
def test_fairseq_hubert_checkpoint_loads_without_fairseq(tmp_path, monkeypatch, large_like, parametrized, cfg_as):
    """VERDICT r3 missing-2 / next-6: a file in fairseq's REAL layout -- pickled foreign classes in `task_state` (un-importable here), the model
    configuration in `cfg["model"]` / `cfg["task"]` (or the legacy `args`), both spellings of the positional conv's weight norm, the pre-training head
    present -- loads through the restricted unpickler; the ARCHITECTURE comes from the file (the name says base, the file may say otherwise), every
    weight arrives bit-exact, and a wrong file raises instead of leaving random weights (speech_encoder_plus.py:380-398)."""
    import dataclasses
    import pickle
    from oracle.hubert_ref import HubertModelRef, HubertRefConfig
    from speechclip_amd.module import FairseqSpeechEncoder_Hubert
    from speechclip_amd.module.hubert import HubertConfig
    from speechclip_amd.util.checkpoint_io import load_pickled_checkpoint
    torch.manual_seed(5)
    hcfg = HubertRefConfig.tiny(layer_norm_first=large_like, extractor_mode="layer_norm" if large_like else "default", conv_bias=large_like)
    hcfg = dataclasses.replace(hcfg, normalize=large_like)
    src = HubertModelRef(hcfg)
    sd = {k: v.clone() for k, v in src.state_dict().items()}
    sd.update({"label_embs_concat": torch.randn(504, 256), "final_proj.weight": torch.randn(256, hcfg.encoder_embed_dim), "final_proj.bias": torch.randn(256)})
    hp = _fairseq_style_hubert_file(tmp_path, hcfg, sd, parametrized, cfg_as)
    with pytest.raises((ModuleNotFoundError, pickle.UnpicklingError, AttributeError, RuntimeError)):      # what the round-3 loader did with such a file
        torch.load(hp, map_location="cpu", weights_only=False)
    ck, stubbed = load_pickled_checkpoint(str(hp))
    assert "fakeseq.data.dictionary.Dictionary" in stubbed and ck["task_state"]["dictionaries"][0].symbols == ["<s>", "</s>", "a"]   # inert stub, state kept
    monkeypatch.setenv("SPEECHCLIP_HUBERT_CKPT", str(hp))
    enc = FairseqSpeechEncoder_Hubert(name="hubert", pretrained=True, feat_select_idx="weighted_sum")      # NO hubert_config: the file's cfg decides
    want = HubertConfig(**{**dataclasses.asdict(hcfg), "dropout": 0.1, "attention_dropout": 0.1, "activation_dropout": 0.0, "dropout_input": 0.1,
                           "encoder_layerdrop": 0.05, "feature_grad_mult": 0.1})
    assert enc.encoder.cfg == want, (enc.encoder.cfg, want)
    assert enc.encoder_task.cfg.normalize == large_like
    got = enc.encoder.state_dict()
    for k, v in src.state_dict().items():
        assert k in got and torch.equal(got[k], v), k
    assert all(not p.requires_grad for p in enc.encoder.parameters())
    # an explicit hubert_config that contradicts the file is an error, not a silent override
    with pytest.raises(ValueError):
        FairseqSpeechEncoder_Hubert(name="hubert", pretrained=True, feat_select_idx="weighted_sum", hubert_config=HubertConfig())
    # a file that lacks a weight of the forward, or carries a foreign one, or a wrong shape: raises
    for mutate, exc in ((lambda d: d.pop("encoder.layers.0.fc1.weight"), RuntimeError), (lambda d: d.update({"encoder.layers.0.adapter.weight": torch.zeros(2)}), RuntimeError),
                        (lambda d: d.update({"post_extract_proj.weight": torch.zeros(3, 3)}), RuntimeError)):
        bad = dict(sd)
        mutate(bad)
        monkeypatch.setenv("SPEECHCLIP_HUBERT_CKPT", str(_fairseq_style_hubert_file(tmp_path / "bad", hcfg, bad, parametrized, cfg_as)))
        with pytest.raises(exc):
            FairseqSpeechEncoder_Hubert(name="hubert", pretrained=True, feat_select_idx="weighted_sum")


def _scripted_archive(sd, path):
    """A TorchScript archive whose state_dict has exactly the keys / dtypes of `sd` (the format of openai's released CLIP files)."""
    class Holder(torch.nn.Module):
        pass
    root = Holder()
    for k, v in sd.items():
        parts, m = k.split("."), root
        for p_ in parts[:-1]:
            if not hasattr(m, p_):
                m.add_module(p_, Holder())
            m = getattr(m, p_)
        if v.is_floating_point():
            m.register_parameter(parts[-1], torch.nn.Parameter(v.clone(), requires_grad=False))
        else:
            m.register_buffer(parts[-1], v.clone())
    torch.jit.script(root).save(str(path))


@pytest.mark.parametrize("fmt", ["jit_fp16", "state_dict"])
def test_openai_clip_archive_loads_without_the_clip_package(tmp_path, monkeypatch, fmt):
    """clip_official.py:50 `clip.load(name)`: openai's files are TorchScript archives with fp16 weights and three non-weight buffers.  Loaded through
    torch.jit.load (no `clip` package), cast to fp32, architecture derived from the shapes and checked against the named model; plain state_dict files
    work too; missing / foreign keys raise."""
    import dataclasses
    from oracle.clip_ref import ClipRef, ClipRefConfig
    from speechclip_amd.module import ClipModel
    from speechclip_amd.module.clip_model import ClipConfig
    from speechclip_amd.util.checkpoint_io import clip_config_from_state_dict, load_clip_state_dict
    torch.manual_seed(6)
    ccfg = ClipRefConfig.tiny()
    csrc = ClipRef(ccfg)
    sd = {k: v.detach().clone() for k, v in csrc.state_dict().items()}
    cp = tmp_path / "clip_tiny.pt"
    if fmt == "jit_fp16":
        sd16 = {k: (v.half() if v.is_floating_point() else v) for k, v in sd.items()}
        sd16.update(input_resolution=torch.tensor(ccfg.image_resolution), context_length=torch.tensor(ccfg.context_length), vocab_size=torch.tensor(ccfg.vocab_size))
        _scripted_archive(sd16, cp)
        sd = {k: (v.half().float() if v.is_floating_point() else v) for k, v in sd.items()}      # what survives fp16 storage
    else:
        torch.save(sd, cp)
    got_sd = load_clip_state_dict(str(cp))
    assert all(v.dtype == torch.float32 for v in got_sd.values() if v.is_floating_point())
    want_cfg = ClipConfig(**dataclasses.asdict(ccfg))
    assert clip_config_from_state_dict(got_sd) == want_cfg
    monkeypatch.setenv("SPEECHCLIP_CLIP_CKPT", str(cp))
    clip = ClipModel(name="ViT-B/32", device="cpu", clip_config=want_cfg)
    cgot = clip.model.state_dict()
    for k, v in sd.items():
        assert k in cgot and torch.equal(cgot[k].float(), v.float()), k
    # the named architecture must be the file's: the tiny file is not a ViT-B/32
    with pytest.raises(ValueError):
        ClipModel(name="ViT-B/32", device="cpu")
    # a file without one of the towers' weights raises (round 3: load_state_dict(strict=False) with the missing-key set discarded)
    bad = {k: v for k, v in sd.items() if k != "visual.ln_post.weight"}
    torch.save(bad, tmp_path / "bad.pt")
    monkeypatch.setenv("SPEECHCLIP_CLIP_CKPT", str(tmp_path / "bad.pt"))
    with pytest.raises(RuntimeError):
        ClipModel(name="ViT-B/32", device="cpu", clip_config=want_cfg)
    monkeypatch.setenv("SPEECHCLIP_CLIP_CKPT", str(tmp_path / "nope.pt"))
    with pytest.raises(FileNotFoundError):
        ClipModel(name="ViT-B/32", device="cpu", clip_config=want_cfg)


def test_clip_image_transform_geometry_and_values(tmp_path):
    """ClipModel.prep_image / image_preprocess (clip_official.py:50,151-164) = clip's `_transform`: bicubic resize of the SHORTER side to n_px, centre
    crop, RGB, /255, CLIP mean / std.  Checked against an independent restatement of torchvision's size / crop arithmetic (its PIL path is
    Image.resize + Image.crop), on landscape / portrait / square / already-sized inputs and on L / RGBA / P mode files; a 224 x 224 RGB file is a
    pure known-answer case: (x / 255 - mean) / std."""
    from PIL import Image
    from speechclip_amd.data.image_transforms import CLIP_MEAN, CLIP_STD, clip_preprocess, load_images_u8, normalize_u8
    rng = np.random.default_rng(0)
    specs = [("land.png", (300, 200), "RGB"), ("port.png", (180, 411), "RGB"), ("sq.png", (224, 224), "RGB"), ("gray.png", (500, 333), "L"),
             ("rgba.png", (231, 260), "RGBA"), ("tiny.jpg", (37, 53), "RGB")]
    paths = []
    for name, (w, h), mode in specs:
        ch = {"L": 1, "RGB": 3, "RGBA": 4}[mode]
        arr = rng.integers(0, 256, size=(h, w, ch) if ch > 1 else (h, w), dtype=np.uint8)
        # smooth the noise a little so that JPEG / bicubic do not alias everything away
        Image.fromarray(arr, mode).save(tmp_path / name)
        paths.append(str(tmp_path / name))
    n = 224
    u8 = load_images_u8(paths, n)
    assert u8.shape == (len(paths), n, n, 3) and u8.dtype == torch.uint8
    for i, p in enumerate(paths):          # independent restatement of torchvision.transforms.functional.resize (int size) + center_crop on PIL
        im = Image.open(p)
        w, h = im.size
        short, long_ = (w, h) if w <= h else (h, w)
        new_short, new_long = n, int(n * long_ / short)
        nw, nh = (new_short, new_long) if w <= h else (new_long, new_short)
        if (w, h) != (nw, nh):
            im = im.resize((nw, nh), Image.BICUBIC)
        top, left = int(round((nh - n) / 2.0)), int(round((nw - n) / 2.0))
        want = np.asarray(im.crop((left, top, left + n, top + n)).convert("RGB"))
        assert np.array_equal(u8[i].numpy(), want), p
    x = normalize_u8(u8)
    assert x.shape == (len(paths), 3, n, n) and x.dtype == torch.float32
    sq = np.asarray(Image.open(paths[2]).convert("RGB")).astype(np.float32) / 255.0            # the 224 x 224 file: no resampling at all
    want = (sq - np.array(CLIP_MEAN, dtype=np.float32)) / np.array(CLIP_STD, dtype=np.float32)
    assert np.allclose(x[2].permute(1, 2, 0).numpy(), want, atol=1e-6)
    one = clip_preprocess(n)(Image.open(paths[0]))
    assert torch.equal(one, x[0])


def test_forward_image_type_errors_without_a_gpu():
    """kwClip.py:520-524: ValueError on a bad tensor shape, TypeError on a bad type (checked before anything touches the device)."""
    model = _tiny_model(cascaded=False)
    with pytest.raises(ValueError):
        model.forward_image(torch.zeros(2, 4, 8, 8))
    with pytest.raises(TypeError):
        model.forward_image("a.jpg")


def test_mlp_layers_state_dict_layout_is_the_references():
    """avssl/module/projections.py:6-29: nn.Sequential of [Linear, nonlin, Dropout] triples minus the last two modules -> parameters live at
    `sequential.0`, `sequential.3`, `sequential.6` ...; a checkpoint trained with a projection head loads by key."""
    from speechclip_amd.module import MLPLayers
    m = MLPLayers(units=[16, 32, 24, 8], dropout=0.2)
    assert list(m.state_dict().keys()) == ["sequential.0.weight", "sequential.0.bias", "sequential.3.weight", "sequential.3.bias",
                                           "sequential.6.weight", "sequential.6.bias"]
    assert [tuple(v.shape) for v in m.state_dict().values()] == [(32, 16), (32,), (24, 32), (24,), (8, 24), (8,)]
    assert len(m.sequential) == 7 and isinstance(m.sequential[-1], torch.nn.Linear) and m.sequential[2].p == 0.2
    # the cascaded branch takes `kw_projection` (kwClip.py:757-771): an MLP in place of the single Linear, same attribute name
    from speechclip_amd.model.kwClip import KW_CascadedBranch
    model = _tiny_model()
    cfg = model.config
    d = cfg.model_settings.cascaded_branch.transformer_args.d_model
    from speechclip_amd.base import OrderedNamespace
    cfg.model_settings.cascaded_branch.keyword["kw_projection"] = OrderedNamespace({"dimensions": [d, 48, model.subword_embd_dim], "dropout": 0.1})
    br = KW_CascadedBranch(config=cfg, audio_dim=d, text_dim=model.subword_embd_dim, clip=model.clip)
    assert isinstance(br.linear_proj, MLPLayers) and "linear_proj.sequential.3.weight" in br.state_dict()


def test_trainable_params_refuse_projection_heads_but_not_nested_mlps():
    """getTrainableParams refuses the optional *_projection MLP heads of the model (kwClip.py:1161-1190: eval-only here, README gaps) when the optimizer
    is built -- and ONLY those four attributes: a kw_projection MLP nested inside the cascaded branch must not trip it (that path has its own check)."""
    from speechclip_amd.base import OrderedNamespace
    from speechclip_amd.module import MLPLayers
    model = _tiny_model()
    assert len(model.getTrainableParams()) > 0
    d = model.config.model_settings.cascaded_branch.transformer_args.d_model
    model.cascaded_branch.linear_proj = MLPLayers(units=[d, 48, model.subword_embd_dim], dropout=0.1)       # nested MLP: allowed to build an optimizer
    assert len(model.getTrainableParams()) > 0
    model.img_enc_proj_net = MLPLayers(units=[16, 16], dropout=0.0)
    with pytest.raises(NotImplementedError, match="img_enc_proj_net"):
        model.getTrainableParams()


def test_collate_general_matches_reference_golden():
    """speechclip_amd.data.collate_general vs the reference's collate_general on the same ragged rows (tests/golden/small_ops.npz)."""
    from speechclip_amd.data import collate_general
    g = np.load(os.path.join(GOLD, "small_ops.npz"))
    lens = [int(x) for x in g["collate_lens"]]
    flat = torch.from_numpy(g["collate_wav_flat"])
    imgs = torch.from_numpy(g["collate_image_rows"])
    rows, off = [], 0
    for i, n in enumerate(lens):
        rows.append({"wav": flat[off:off + n].clone(), "image": imgs[i], "id": 7 * i + 1})
        off += n
    out = collate_general(rows)
    assert list(out.keys()) == ["wav", "image", "id", "wav_len"]
    for k, v in out.items():
        ref = torch.from_numpy(g["collate_out_" + k])
        assert v.dtype == ref.dtype and torch.equal(v, ref), k
    # rows without waves: no wav_len key, numbers -> LongTensor
    out2 = collate_general([{"id": 3, "image": imgs[0]}, {"id": 5, "image": imgs[1]}])
    assert list(out2.keys()) == ["id", "image"] and out2["id"].dtype == torch.int64 and out2["image"].shape[0] == 2


def _bench_line(cmd):
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-1500:]                      # rank 0 prints the ONE line
    return json.loads(lines[0])


def test_bench_gpus2_self_launches_and_runs_under_torchrun_dry_run():
    """`python bench.py --gpus 2` with no launcher around it must start its own two ranks (the driver's plain
    `python3 bench.py --gpus 8` used to die on an assert), and the documented external launch must keep working.  CPU/gloo dry run: the
    launcher, rendezvous, packed all-gather, global-batch loss, max-over-ranks timing and the JSON contract are the real code; the towers
    are replaced by random embeddings (no GPU here)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    bench = os.path.join(root, "bench.py")
    d = _bench_line([sys.executable, bench, "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1", "--batch", "4"])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["config"]["global_batch"] == 8 and d["steps"] == 3 and d["warmup"] == 1
    assert abs(d["value"] - 2 * 4 * 1000.0 / d["ms_per_step"]) / d["value"] < 1e-2 and "dry-run" in d["data"]
    d = _bench_line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                     "--master-port", str(_free_port()), bench, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--batch", "4"])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2
    # strong scaling: the GLOBAL batch is fixed and split over the ranks
    d = _bench_line([sys.executable, bench, "--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "0", "--global-batch", "12"])
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 12 and d["config"]["pairs_per_gpu"] == 6 and d["n_gpus"] == 2
    # BASELINE.json configs[4]: Parallel SpeechCLIP large on 4 ranks, global batch 256 -> 64 pairs per rank, E = 768
    d = _bench_line([sys.executable, bench, "--gpus", "4", "--dry-run", "--steps", "2", "--warmup", "0", "--model", "large", "--global-batch", "256"])
    assert d["n_gpus"] == 4 and d["ranks_seen"] == 4 and d["backend"] == "gloo" and d["model"] == "large"
    assert d["scaling"] == "strong" and d["config"]["global_batch"] == 256 and d["config"]["pairs_per_gpu"] == 64
    # a launcher that started the wrong number of ranks is reported, not asserted on
    import subprocess
    r = subprocess.run([sys.executable, bench, "--gpus", "4", "--dry-run"], capture_output=True, text=True, cwd=root,
                       env=dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"))
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


@pytest.mark.skipif(not os.path.isdir("/root/reference/config/speechCLIP"), reason="needs the reference's YAML files (build container only)")
@pytest.mark.parametrize("rel", ["model_base/spchclp_p.yaml", "model_base/spchclp_c.yaml", "model_large/flickr/spchclp_p.yaml",
                                 "model_large/flickr/spchclp_c.yaml", "model_large/coco/spchclp_p.yaml", "model_large/coco/spchclp_c.yaml"])
def test_reference_yaml_builds_through_the_task_entry_point(rel, tmp_path, monkeypatch):
    """each of the six shipped reference configs, unmodified, through `avssl.task.TrainKWClip_GeneralTransformer` exactly as
    run_task.py drives it (add_args -> parse_args -> build_model): the drop-in claim of SURVEY.md section 8(b).  The only thing supplied
    from outside is what the YAML points at on disk (the reduced-vocabulary .npy of the cascaded configs).  Construction only (CPU)."""
    import argparse
    import yaml
    import avssl.task as task_mod                                  # the alias package, as run_task.py imports it
    path = os.path.join("/root/reference/config/speechCLIP", rel)
    cfg = yaml.load(open(path), Loader=yaml.FullLoader)
    vocab_file = cfg["clip"].get("reduce_subword_embbedding")
    if vocab_file:                                                 # relative path in the YAML (spchclp_c.yaml:94): provide the file there
        ids = np.concatenate([[0, 320, 49406, 49407], np.arange(1000, 1000 + 8108)])
        os.makedirs(os.path.join(tmp_path, os.path.dirname(vocab_file)), exist_ok=True)
        np.save(os.path.join(tmp_path, vocab_file), np.stack([ids, np.arange(len(ids))[::-1] + 1], axis=1))
    monkeypatch.chdir(tmp_path)
    t = task_mod.TrainKWClip_GeneralTransformer()
    parser = t.add_args(argparse.ArgumentParser())
    t.parse_args(parser, ["--config", path, "--gpus", "1", "--njobs", "1", "--seed", "7122", "--save_path", str(tmp_path / "exp")])
    from avssl.model import KWClip_GeneralTransformer
    model = t.build_model(KWClip_GeneralTransformer)
    ms = cfg["model_settings"]
    assert (model.parallel_branch is not None) == (ms["parallel_objective_weight"] > 0)
    assert (model.cascaded_branch is not None) == (ms["cascaded_objective_weight"] > 0)
    large = "large" in rel
    assert model.audio_embd_dim == (1024 if large else 768) and model.subword_embd_dim == (768 if large else 512)
    assert model.criterion.temperature_trainable == cfg["cl_loss"]["args"]["temperature_trainable"]
    if vocab_file:
        assert model.clip.model.token_embedding.weight.shape[0] == 8112
    assert model.config.audio_encoder.name == cfg["audio_encoder"]["name"] and model.config.seed == 7122
    opt, sched = model.configure_optimizers()
    assert len(opt) == 1 and len(model.getTrainableParams()) > 0


def test_encoder_fine_tuning_flags_follow_the_reference():
    """speech_encoder_plus.py:416-446 on the host side: `trainable` + `unfreeze_layers` / `reinit_layers` mark exactly the listed transformer
    layers trainable (pos_conv, layer_norm, the feature extractor and post_extract_proj stay frozen, feature_grad_mult = 0), reinit_layers
    re-initialises them, and the argument combinations the reference asserts on are rejected."""
    import dataclasses
    from oracle.hubert_ref import HubertRefConfig
    from speechclip_amd.module import FairseqSpeechEncoder_Hubert
    from speechclip_amd.module.hubert import HubertConfig
    hc = HubertConfig(**dataclasses.asdict(dataclasses.replace(HubertRefConfig.tiny(), encoder_layers=3)))
    torch.manual_seed(0)
    enc = FairseqSpeechEncoder_Hubert("hubert", trainable=True, unfreeze_layers=[1, 2], feat_select_idx="weighted_sum", hubert_config=hc)
    names = [k for k, p in enc.encoder.named_parameters() if p.requires_grad]
    assert len(names) == 32 and all(k.startswith(("encoder.layers.1.", "encoder.layers.2.")) for k in names)
    assert enc.encoder.feature_grad_mult == 0 and enc.train_layers == [1, 2]
    assert len(enc.trainable_params()) == 32 + 1                      # + the layer-mix weights
    torch.manual_seed(0)
    base = FairseqSpeechEncoder_Hubert("hubert", feat_select_idx="weighted_sum", hubert_config=hc)
    torch.manual_seed(0)
    re = FairseqSpeechEncoder_Hubert("hubert", trainable=True, reinit_layers=[2], feat_select_idx="weighted_sum", hubert_config=hc)
    l2b, l2r = base.encoder.encoder.layers[2], re.encoder.encoder.layers[2]
    assert not torch.equal(l2b.fc1.weight, l2r.fc1.weight) and torch.equal(base.encoder.encoder.layers[1].fc1.weight, re.encoder.encoder.layers[1].fc1.weight)
    # `layer.apply(init_weights)` = reset_parameters() of every sub-module, then fairseq MultiheadAttention.reset_parameters on top:
    # fc1 / fc2: torch's kaiming-uniform Linear default (|w| <= 1/sqrt(fan_in), uniform biases); q / k / v: xavier_uniform with gain 1/sqrt(2)
    # (bound sqrt(6 / (2 d)) / sqrt(2)); out_proj: xavier_uniform, zero bias; LayerNorms: ones / zeros
    d = hc.encoder_embed_dim
    assert float(l2r.fc1.weight.abs().max()) <= d ** -0.5 + 1e-6 and float(l2r.fc1.bias.abs().max()) > 0
    assert abs(float(l2r.fc1.weight.std()) - d ** -0.5 / 3 ** 0.5) < 0.1 * d ** -0.5
    qb = (6.0 / (2 * d)) ** 0.5 / 2 ** 0.5
    assert float(l2r.self_attn.q_proj.weight.abs().max()) <= qb + 1e-6 and float(l2r.self_attn.q_proj.weight.abs().max()) > 0.9 * qb
    assert float(l2r.self_attn.out_proj.weight.abs().max()) > qb and float(l2r.self_attn.out_proj.bias.abs().max()) == 0.0
    assert float(l2r.self_attn.q_proj.bias.abs().max()) > 0 and torch.equal(l2r.final_layer_norm.weight, torch.ones(d))
    assert len(base.trainable_params()) == 1 and not any(p.requires_grad for p in base.encoder.parameters())
    # bare trainable=True (speech_encoder_plus.py:399-401): nothing is frozen; the extractor keeps the checkpoint's feature_grad_mult
    full = FairseqSpeechEncoder_Hubert("hubert", trainable=True, feat_select_idx="weighted_sum", hubert_config=hc)
    on = {k for k, p in full.encoder.named_parameters() if p.requires_grad}
    assert full.train_front and full.train_layers == [0, 1, 2] and full.encoder.feature_grad_mult == 0.1
    assert {"feature_extractor.conv_layers.0.0.weight", "feature_extractor.conv_layers.0.2.weight", "feature_extractor.conv_layers.6.0.weight", "layer_norm.bias",
            "post_extract_proj.weight", "encoder.pos_conv.0.weight_g", "encoder.pos_conv.0.weight_v", "encoder.layer_norm.weight",
            "encoder.layers.0.fc1.weight"} <= on and "mask_emb" not in on and len(on) == 18 + 3 * 16
    large = HubertConfig(**dataclasses.asdict(HubertRefConfig.tiny(layer_norm_first=True, extractor_mode="layer_norm", conv_bias=True)))
    fl = FairseqSpeechEncoder_Hubert("hubert_large_ll60k", trainable=True, hubert_config=large)      # the large architecture trains end to end too
    onl = {k for k, p in fl.encoder.named_parameters() if p.requires_grad}
    assert fl.train_front and "encoder.layer_norm.weight" not in onl and "feature_extractor.conv_layers.3.2.1.weight" in onl and "feature_extractor.conv_layers.0.0.bias" in onl
    odd = HubertConfig(**dataclasses.asdict(HubertRefConfig.tiny(layer_norm_first=True)))               # pre-LN layers on a GroupNorm extractor: no released model
    with pytest.raises(NotImplementedError):
        FairseqSpeechEncoder_Hubert("hubert", trainable=True, hubert_config=odd)
    with pytest.raises(AssertionError):
        FairseqSpeechEncoder_Hubert("hubert", trainable=False, unfreeze_layers=[1], hubert_config=hc)
    with pytest.raises(AssertionError):
        FairseqSpeechEncoder_Hubert("hubert", trainable=True, unfreeze_layers=[1], reinit_layers=[2], hubert_config=hc)


def test_vq_temperature_forms_follow_the_reference():
    """my_vector_quantizer.py:28-62: "fixed=x" (buffer), "learnable=x" (parameter), "(max, min, decay)" (scheduled: starts at max, decays with
    set_num_updates, floored at min -- which nothing in the reference ever calls)."""
    from speechclip_amd.module.speechclip_c_modules.vector_quantizers import SimpleVectorQuantizer as VQ
    f = VQ("fixed=0.1")
    assert f.temp_type == "fixed" and abs(f.temperature_value() - 0.1) < 1e-7 and "curr_temp" in dict(f.named_buffers())
    l = VQ("learnable=0.5")
    assert l.temp_type == "learnable" and isinstance(l.curr_temp, torch.nn.Parameter)
    s = VQ("(2.0, 0.5, 0.9)")
    assert s.temp_type == "scheduled" and s.temperature_value() == 2.0
    s.set_num_updates(3)
    assert abs(s.temperature_value() - 2.0 * 0.9 ** 3) < 1e-12
    s.set_num_updates(1000)
    assert s.temperature_value() == 0.5
    f.set_num_updates(5)
    assert abs(f.temperature_value() - 0.1) < 1e-7


def test_packed_geometry_covers_the_receptive_field_halo():
    """Padding-free row allotment (module/hubert.py: packed_geometry): utterance b gets max(valid_b, need_b) + 1 transformer rows and
    2^(6-l) times that at conv layer l.  The frames a kept output depends on must fit the allotment at EVERY conv level, and the frame
    mask / feat_len rules stay those of the padded reference (speech_encoder_plus.py:604-611, fairseq forward_padding_mask)."""
    from speechclip_amd.module.hubert import CONV_LAYERS, HubertConfig, HubertModel, conv_lengths
    enc = HubertModel(HubertConfig(encoder_layers=1))
    rng = np.random.RandomState(0)
    for trial in range(50):
        B = int(rng.randint(1, 9))
        lens = [int(x) for x in rng.randint(400, 160001, size=B)]
        lmax = max(lens)
        T0, T, P0, Tp = enc.frame_geometry(lmax)
        need = [min(round(l / 320), T) for l in lens]
        geo = enc.packed_geometry(lens, lmax, need_rows=need)
        assert geo["valid"] == enc.valid_frames(lens, lmax, T) and geo["scale0"] == 64 and geo["padded_rows"] == B * Tp
        assert geo["row_off"][0] == 0 and geo["row_off"][-1] == geo["total"] == sum(geo["rows"])
        pad_lens = conv_lengths(lmax, CONV_LAYERS)
        for b in range(B):
            F = max(geo["valid"][b], need[b])
            assert geo["rows"][b] == F + 1 and F <= T
            n = F                                            # frames needed at the last conv level, walking down to layer 0
            scale = 1
            for lvl in range(len(CONV_LAYERS) - 1, 0, -1):
                _, k, s = CONV_LAYERS[lvl]
                n = (n - 1) * s + k                          # frames of level lvl-1 that n frames of level lvl read
                scale *= s
                assert n <= scale * geo["rows"][b], (lvl, n, scale * geo["rows"][b])
                assert n <= pad_lens[lvl - 1], "a kept frame never needs a frame the padded layout does not have"
    # equal lengths: nothing to gain, one extra row per utterance at most
    geo = enc.packed_geometry([160000] * 4, 160000, need_rows=[499] * 4)
    assert geo["rows"] == [500] * 4 and geo["total"] == geo["padded_rows"]


def test_product_library_links_no_vendor_blas():
    """The shipped libspeechclip_hip.so contains no vendor GEMM: hipBLASLt is reachable only through the separate comparator library
    (libspeechclip_vendor_cmp.so), dlopen()ed when bench.py's comparator leg registers a workspace."""
    import subprocess
    from speechclip_amd import _lib
    out = subprocess.run(["ldd", _lib.LIB_PATH], capture_output=True, text=True).stdout.lower()
    assert "blas" not in out and "miopen" not in out, out
    syms = subprocess.run(["nm", "-D", "--undefined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "hipblas" not in syms.lower()


def test_restricted_unpickler_runs_no_foreign_code(tmp_path):
    """the unpickler used to pass every `torch.*` / `numpy.*` global to pickle's own find_class, which resolves protocol-4 DOTTED
    names through module attributes -- ("torch.serialization", "os.system") executed os.system.  Now only an explicit list of reconstruction helpers is
    real; gadgets become inert stubs (calling a stub returns a stub), and ordinary tensor / ndarray payloads still load bit-exact."""
    import io
    import pickle
    import numpy as np
    from speechclip_amd.util.checkpoint_io import RestrictedUnpickler, StubObject, load_pickled_checkpoint
    marker = tmp_path / "pwned"
    gadgets = [
        b"\x80\x04\x8c\x13torch.serialization\x8c\x09os.system\x93\x8c" + bytes([len(f"touch {marker}")]) + f"touch {marker}".encode() + b"\x85R.",
        pickle.dumps(("x",), protocol=2).replace(b"\x80\x02", b"\x80\x02", 1),      # harmless control
    ]
    u = RestrictedUnpickler(io.BytesIO(gadgets[0]))
    out = u.load()
    assert isinstance(out, StubObject) and not marker.exists() and "torch.serialization.os.system" in u.stubbed
    for mod, name in (("torch.hub", "load"), ("numpy.testing._private.utils", "runstring"), ("torch.utils.cpp_extension", "load_inline"),
                      ("torch", "load"), ("numpy", "load"), ("os", "system"), ("builtins", "eval"), ("builtins", "exec"), ("torch._utils", "os.system")):
        cls = RestrictedUnpickler(io.BytesIO(b"")).find_class(mod, name)
        assert isinstance(cls, type) and issubclass(cls, StubObject), (mod, name, cls)
    # per-instance stub logs (two loaders do not share state)
    a, b = RestrictedUnpickler(io.BytesIO(b"")), RestrictedUnpickler(io.BytesIO(b""))
    a.find_class("foo.bar", "Baz")
    assert a.stubbed == {"foo.bar.Baz"} and b.stubbed == set()
    # ordinary payloads are untouched
    t = {"w": torch.arange(12, dtype=torch.float32).reshape(3, 4), "h": torch.ones(5, dtype=torch.bfloat16), "n": np.arange(6, dtype=np.int64).reshape(2, 3),
         "s": np.float32(2.5), "p": torch.nn.Parameter(torch.zeros(2))}
    f = tmp_path / "plain.pt"
    torch.save(t, f)
    ck, stubbed = load_pickled_checkpoint(str(f))
    assert not stubbed and torch.equal(ck["w"], t["w"]) and torch.equal(ck["h"], t["h"]) and (ck["n"] == t["n"]).all() and float(ck["s"]) == 2.5
    assert isinstance(ck["p"], torch.nn.Parameter)


def test_clip_loader_does_not_mask_a_corrupt_torchscript_archive(tmp_path):
    """a truncated TorchScript archive must surface its own error instead of falling through to torch.load(weights_only=True)."""
    import zipfile
    from speechclip_amd.util.checkpoint_io import load_clip_state_dict
    bad = tmp_path / "ViT-bad.pt"
    with zipfile.ZipFile(bad, "w") as z:
        z.writestr("archive/constants.pkl", b"\x80\x02).")
        z.writestr("archive/version", b"3\n")
    with pytest.raises(Exception) as ei:
        load_clip_state_dict(str(bad))
    assert "weights_only" not in str(ei.value).lower()
    good = tmp_path / "plain_sd.pt"
    torch.save({"state_dict": {"a": torch.ones(2, dtype=torch.float16)}}, good)
    sd = load_clip_state_dict(str(good))
    assert sd["a"].dtype == torch.float32
