"""Reading the REAL checkpoint files of the reference's backbones without their packages (VERDICT r3 missing-2).

  * fairseq `hubert_base_ls960.pt` / `hubert_large_ll60k.pt` (avssl/module/speech_encoder_plus.py:327-331, :380-398 -> fairseq.checkpoint_utils.
    load_model_ensemble_and_task): a torch pickle `{"cfg": {...}, "model": state_dict, "task_state": {...fairseq Dictionary objects...}, "args", ...}`.
    fairseq is not installed and plain `torch.load` refuses (weights_only) or cannot import the pickled classes.  `load_pickled_checkpoint` unpickles
    with a RESTRICTED unpickler: tensors / containers / numpy / argparse.Namespace are rebuilt, every other global (fairseq.*, omegaconf.*,
    pytorch_lightning.*, ...) becomes an inert stub object that only records its state -- nothing outside an explicit allow-list is imported or run.
    `hubert_config_from_fairseq` then reads the ARCHITECTURE from the checkpoint's own `cfg` (`extractor_mode`, `conv_bias`, `layer_norm_first`,
    sizes, dropouts; `task.normalize`) instead of guessing it from the model name.
  * openai CLIP `ViT-B-32.pt` / `ViT-L-14.pt` (avssl/module/clip_official.py:50 -> clip.load): a TorchScript archive.  `load_clip_state_dict` reads it
    with `torch.jit.load(...).state_dict()` (no `clip` package needed), falls back to a plain state_dict file, casts fp16 weights to fp32 as
    `clip.load(name, "cpu")` does, and `clip_config_from_state_dict` derives the architecture from tensor shapes as openai's `build_model` does.
  * Lightning SpeechCLIP checkpoints (download_ckpts.sh:7-23; `hyper_parameters.config` is a pickled avssl.base OrderedNamespace, `callbacks` is
    keyed by pytorch_lightning classes): the same restricted unpickler, with the OrderedNamespace class mapped to this package's.
Both loaders' callers RAISE on any missing / unexpected key outside an explicit allow-list (`strict_load`)."""
import argparse
import ast
import collections
import io
import pickle
from typing import Dict, Iterable, Tuple

import torch

# ------------------------------------------------------------------------------------------------ restricted unpickling
_SAFE_BUILTINS = {"set", "frozenset", "list", "dict", "tuple", "int", "float", "bool", "str", "bytes", "bytearray", "complex", "slice", "range", "object"}
_SAFE_GLOBALS = {
    ("collections", "OrderedDict"): collections.OrderedDict,
    ("collections", "defaultdict"): collections.defaultdict,
    ("argparse", "Namespace"): argparse.Namespace,
    ("_codecs", "encode"): __import__("_codecs").encode,
}
# Tensor / storage / ndarray reconstruction helpers: an explicit (module, name) list.  (a `torch.*` / `numpy.*` PREFIX rule lets protocol-4
# dotted names through -- ("torch.serialization", "os.system") resolved to os.system -- and exposes torch.hub.load, numpy.testing runstring, ...)
_TORCH_STORAGES = ("FloatStorage", "HalfStorage", "BFloat16Storage", "DoubleStorage", "LongStorage", "IntStorage", "ShortStorage", "CharStorage",
                   "ByteStorage", "BoolStorage", "ComplexFloatStorage", "ComplexDoubleStorage")
_SAFE_HELPERS = {
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._utils", "_rebuild_device_tensor_from_numpy"),
    ("torch._tensor", "_rebuild_from_type_v2"), ("torch", "Size"), ("torch", "device"), ("torch", "Tensor"),
    ("torch.nn.parameter", "Parameter"), ("torch.storage", "UntypedStorage"), ("torch.storage", "TypedStorage"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy.core.multiarray", "scalar"),
    ("numpy._core.multiarray", "scalar"), ("numpy", "ndarray"), ("numpy", "dtype"), ("numpy.core.numeric", "_frombuffer"),
    ("numpy._core.numeric", "_frombuffer"),
} | {("torch", n) for n in _TORCH_STORAGES}


def _safe_helper(module: str, name: str):
    """The real object for an allow-listed reconstruction helper, else None.  `torch.<dtype>` globals (torch.float32, ...) are allowed by TYPE."""
    if "." in name:
        return None
    import importlib
    if (module, name) in _SAFE_HELPERS:
        try:
            return getattr(importlib.import_module(module), name)
        except (ImportError, AttributeError):
            return None
    if module == "torch" and isinstance(getattr(torch, name, None), torch.dtype):
        return getattr(torch, name)
    return None


class StubObject:
    """Inert stand-in for an instance of a class that is not importable here (fairseq Dictionary, omegaconf nodes, Lightning callbacks ...):
    accepts any constructor arguments and any pickled state, runs no code of the original class."""
    _stub_of = "?"

    def __init__(self, *args, **kwargs):
        self._stub_args, self._stub_kwargs = args, kwargs

    def __setstate__(self, state):
        if isinstance(state, dict):
            self.__dict__.update(state)
        else:
            self._stub_state = state

    def __call__(self, *args, **kwargs):             # a pickled FUNCTION reference used as a reducer
        return StubObject(*args, **kwargs)

    def __repr__(self):
        return f"<stub of {self._stub_of}>"

    # list / dict subclasses are pickled through append / extend / __setitem__
    def append(self, x):
        self.__dict__.setdefault("_stub_items", []).append(x)

    def extend(self, xs):
        self.__dict__.setdefault("_stub_items", []).extend(xs)

    def __setitem__(self, k, v):
        self.__dict__.setdefault("_stub_map", {})[k] = v


def _stub_class(module: str, name: str):
    return type(name, (StubObject,), {"_stub_of": f"{module}.{name}", "__module__": __name__})


class RestrictedUnpickler(pickle.Unpickler):
    """find_class: allow-listed globals and the explicit list of tensor / ndarray reconstruction helpers are real, everything else is a stub class.
    `self.stubbed`: the "module.name" globals this load replaced by stubs (per instance: concurrent / nested loads do not share it)."""

    def __init__(self, *args, stubbed=None, **kwargs):
        super().__init__(*args, **kwargs)
        self.stubbed = stubbed if stubbed is not None else set()

    def find_class(self, module, name):
        if (module, name) in _SAFE_GLOBALS:
            return _SAFE_GLOBALS[(module, name)]
        if module == "builtins" and name in _SAFE_BUILTINS:
            return getattr(__import__("builtins"), name)
        if module in ("avssl.base.ordered_namespace", "avssl.base", "speechclip_amd.base.ordered_namespace") and name == "OrderedNamespace":
            from ..base import OrderedNamespace
            return OrderedNamespace
        if (module, name) in (("speechclip_amd.module.hubert", "HubertConfig"), ("speechclip_amd.module.clip_model", "ClipConfig")):
            import importlib                      # this package's own plain dataclasses (test-size architectures stored in a config)
            return getattr(importlib.import_module(module), name)
        helper = _safe_helper(module, name)
        if helper is not None:
            return helper
        self.stubbed.add(f"{module}.{name}")
        return _stub_class(module, name.replace(".", "_"))


def _restricted_pickle_module(stubbed: set):
    """The `pickle_module` torch.load expects (`Unpickler`, `load`, `loads`, a name), bound to ONE load's stub log."""
    class _Unpickler(RestrictedUnpickler):
        def __init__(self, *args, **kwargs):
            super().__init__(*args, stubbed=stubbed, **kwargs)

    class _Module:
        __name__ = "speechclip_amd.util.checkpoint_io.restricted_pickle"
        Unpickler = _Unpickler
        UnpicklingError = pickle.UnpicklingError

        @staticmethod
        def load(f, **kwargs):
            return _Unpickler(f, **kwargs).load()

        @staticmethod
        def loads(b, **kwargs):
            return _Unpickler(io.BytesIO(b), **kwargs).load()

    return _Module


def load_pickled_checkpoint(path: str, map_location="cpu") -> Tuple[dict, set]:
    """torch.load through the restricted unpickler.  -> (checkpoint dict, set of "module.name" globals that were replaced by stubs)."""
    stubbed: set = set()
    ckpt = torch.load(path, map_location=map_location, weights_only=False, pickle_module=_restricted_pickle_module(stubbed))
    return ckpt, stubbed


# ------------------------------------------------------------------------------------------------ fairseq HuBERT
HUBERT_UNUSED_KEYS = ("mask_emb", "final_proj.", "label_embs_concat")     # never touched by customHubertForward (speech_encoder_plus.py:67-107)


def _as_plain(x):
    """cfg sections may arrive as dicts, argparse.Namespaces or stubs of omegaconf containers: -> a plain dict view."""
    if x is None:
        return {}
    if isinstance(x, dict):
        return x
    if isinstance(x, argparse.Namespace):
        return vars(x)
    d = getattr(x, "__dict__", {})
    for k in ("_content", "_stub_map"):                      # omegaconf DictConfig keeps its children under `_content`
        if isinstance(d.get(k), dict):
            return {kk: getattr(v, "_val", v) for kk, v in d[k].items()}
    return d


def _enum_text(v) -> str:
    """`extractor_mode` is a ChoiceEnum in fairseq: a str in dict-form cfgs, an Enum (stubbed here) in older pickles."""
    if isinstance(v, str):
        return v
    for attr in ("_value_", "value", "_name_", "name"):
        t = getattr(v, attr, None)
        if isinstance(t, str):
            return t
    args = getattr(v, "_stub_args", ())
    if args and isinstance(args[0], str):
        return args[0]
    raise ValueError(f"cannot read an enum value from {v!r}")


def _parse_conv_layers(spec) -> list:
    """fairseq's `conv_feature_layers` is a Python EXPRESSION string ("[(512,10,5)] + [(512,3,2)] * 4 + [(512,2,2)] * 2") that fairseq eval()s;
    here it is walked as an AST of lists / tuples / ints joined by + and * only."""
    if not isinstance(spec, str):
        return [tuple(int(v) for v in c) for c in spec]

    def ev(n):
        if isinstance(n, ast.Expression):
            return ev(n.body)
        if isinstance(n, ast.Constant) and isinstance(n.value, int):
            return n.value
        if isinstance(n, (ast.List, ast.Tuple)):
            vals = [ev(e) for e in n.elts]
            return vals if isinstance(n, ast.List) else tuple(vals)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Add):
            return ev(n.left) + ev(n.right)
        if isinstance(n, ast.BinOp) and isinstance(n.op, ast.Mult):
            return ev(n.left) * ev(n.right)
        raise ValueError(f"conv_feature_layers: unsupported expression node {ast.dump(n)}")
    out = ev(ast.parse(spec.strip(), mode="eval"))
    return [tuple(int(v) for v in c) for c in out]


def hubert_config_from_fairseq(ckpt: dict):
    """HubertConfig from the checkpoint's OWN configuration: `cfg["model"]` / `cfg["task"]` (fairseq >= 0.10.2 dict form) or the legacy `args`
    Namespace.  [3P fairseq/models/hubert/hubert.py HubertConfig field names]."""
    from ..module.hubert import HubertConfig
    cfg = ckpt.get("cfg")
    if cfg is not None:
        c = _as_plain(cfg)
        m, t = _as_plain(c.get("model")), _as_plain(c.get("task"))
    elif ckpt.get("args") is not None:
        m = t = _as_plain(ckpt["args"])
    else:
        raise KeyError("fairseq checkpoint has neither `cfg` nor `args`: cannot read the model configuration")
    need = ("encoder_layers", "encoder_embed_dim", "encoder_ffn_embed_dim", "encoder_attention_heads")
    lack = [k for k in need if k not in m]
    if lack:
        raise KeyError(f"fairseq checkpoint cfg['model'] lacks {lack}")
    d = HubertConfig()        # field defaults = fairseq's HubertConfig defaults for the fields a checkpoint may omit
    return HubertConfig(
        extractor_mode=_enum_text(m.get("extractor_mode", d.extractor_mode)), conv_bias=bool(m.get("conv_bias", False)),
        conv_layers=_parse_conv_layers(m.get("conv_feature_layers", d.conv_layers)),
        encoder_layers=int(m["encoder_layers"]), encoder_embed_dim=int(m["encoder_embed_dim"]), encoder_ffn_embed_dim=int(m["encoder_ffn_embed_dim"]),
        encoder_attention_heads=int(m["encoder_attention_heads"]), layer_norm_first=bool(m.get("layer_norm_first", False)),
        conv_pos=int(m.get("conv_pos", 128)), conv_pos_groups=int(m.get("conv_pos_groups", 16)), normalize=bool(t.get("normalize", False)),
        feature_grad_mult=float(m.get("feature_grad_mult", 1.0)), dropout=float(m.get("dropout", 0.1)), attention_dropout=float(m.get("attention_dropout", 0.1)),
        activation_dropout=float(m.get("activation_dropout", 0.0)), dropout_input=float(m.get("dropout_input", 0.0)),
        encoder_layerdrop=float(m.get("encoder_layerdrop", 0.0)))


def normalize_hubert_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Both spellings of the positional conv's weight norm: `weight_g / weight_v` (torch.nn.utils.weight_norm, the released files) and
    `parametrizations.weight.original0 / original1` (torch >= 2.1 parametrize-based weight_norm, files re-saved by newer fairseq)."""
    out = {}
    for k, v in sd.items():
        k = k.replace("pos_conv.0.parametrizations.weight.original0", "pos_conv.0.weight_g").replace("pos_conv.0.parametrizations.weight.original1", "pos_conv.0.weight_v")
        out[k] = v
    return out


def strict_load(module: torch.nn.Module, sd: Dict[str, torch.Tensor], allow_missing: Iterable[str] = (), allow_unexpected: Iterable[str] = (), what: str = "checkpoint"):
    """load_state_dict that RAISES on any key mismatch outside the explicit allow-lists (prefix match) and on any shape mismatch: a wrong file must not
    silently leave random weights behind."""
    am, au = tuple(allow_missing), tuple(allow_unexpected)
    own = module.state_dict()
    bad_shape = [(k, tuple(v.shape), tuple(own[k].shape)) for k, v in sd.items() if k in own and tuple(v.shape) != tuple(own[k].shape)]
    if bad_shape:
        raise RuntimeError(f"{what}: shape mismatch (key, file, model): {bad_shape[:8]}{' ...' if len(bad_shape) > 8 else ''}")
    missing, unexpected = module.load_state_dict({k: v for k, v in sd.items() if k in own}, strict=False)
    unexpected = list(unexpected) + [k for k in sd if k not in own]
    bad_m = [k for k in missing if not k.startswith(am)] if am else list(missing)
    bad_u = [k for k in unexpected if not k.startswith(au)] if au else list(unexpected)
    if bad_m or bad_u:
        raise RuntimeError(f"{what}: missing keys {bad_m[:12]}{' ...' if len(bad_m) > 12 else ''}; unexpected keys {bad_u[:12]}{' ...' if len(bad_u) > 12 else ''}")
    return [k for k in missing if k not in bad_m], [k for k in unexpected if k not in bad_u]


def load_fairseq_hubert(path: str):
    """-> (HubertConfig read from the file, state_dict with this package's key names, stubbed globals)."""
    ckpt, stubbed = load_pickled_checkpoint(path)
    if not isinstance(ckpt, dict) or "model" not in ckpt:
        raise KeyError(f"{path}: not a fairseq checkpoint (no `model` entry; top-level keys: {list(ckpt)[:8] if isinstance(ckpt, dict) else type(ckpt)})")
    sd = normalize_hubert_keys({k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ckpt["model"].items()})
    return hubert_config_from_fairseq(ckpt), sd, stubbed


# ------------------------------------------------------------------------------------------------ openai CLIP
CLIP_NON_WEIGHT_KEYS = ("input_resolution", "context_length", "vocab_size")       # buffers of the TorchScript archive, not parameters (clip/model.py build_model)


def load_clip_state_dict(path: str) -> Dict[str, torch.Tensor]:
    """openai's released files are TorchScript archives: read through torch.jit.load (no `clip` package); a plain state_dict file (what
    `clip.load(jit=False)` users save) is accepted too.  fp16 weights are cast to fp32 as `clip.load(name, "cpu")` does (clip/clip.py: model.float())."""
    import zipfile
    is_script = False
    if zipfile.is_zipfile(path):
        with zipfile.ZipFile(path) as z:
            is_script = any(n.endswith("constants.pkl") for n in z.namelist())      # TorchScript archives carry constants.pkl; torch.save zips do not
    if is_script:
        sd = torch.jit.load(path, map_location="cpu").state_dict()               # a truncated / corrupt archive raises HERE, with its own message
    else:
        obj = torch.load(path, map_location="cpu", weights_only=True)
        sd = obj.get("state_dict", obj) if isinstance(obj, dict) else obj
    return {k: (v.float() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}


def clip_config_from_state_dict(sd: Dict[str, torch.Tensor]):
    """Architecture from tensor shapes, as openai's `build_model` derives it (clip/model.py [3P]); ViT image towers only (every shipped config)."""
    from ..module.clip_model import ClipConfig
    if "visual.proj" not in sd:
        raise NotImplementedError("CLIP checkpoint has no `visual.proj`: ResNet image towers are not built (every shipped config uses ViT-B/32 or ViT-L/14)")
    width = sd["visual.conv1.weight"].shape[0]
    layers = len({k.split(".")[3] for k in sd if k.startswith("visual.transformer.resblocks.") and k.endswith(".attn.in_proj_weight")})
    patch = sd["visual.conv1.weight"].shape[-1]
    grid = round((sd["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    tw = sd["ln_final.weight"].shape[0]
    tl = len({k.split(".")[2] for k in sd if k.startswith("transformer.resblocks.") and k.endswith(".attn.in_proj_weight")})
    return ClipConfig(image_resolution=int(patch * grid), vision_patch=int(patch), vision_width=int(width), vision_layers=int(layers),
                      embed_dim=int(sd["text_projection"].shape[1]), context_length=int(sd["positional_embedding"].shape[0]),
                      vocab_size=int(sd["token_embedding.weight"].shape[0]), text_width=int(tw), text_heads=int(tw // 64), text_layers=int(tl))
