"""BLASER and MuTox heads on the MI355X engine, with the reference's interfaces.

  * `BlaserModel(config, state_dict, device)` -> `forward(src, mt, ref=None)`:
    sonar/models/blaser/model.py:27-125, configs `basic_ref` / `basic_qe` of
    sonar/models/blaser/config.py:36-67; checkpoint layout of handler.py:36-45 (a bare state dict
    or {"model": state dict}, keys `mlp.<i>.weight|bias`).
  * `MutoxClassifier(config, state_dict, device)` -> `forward(inputs, output_prob=False)`:
    sonar/models/mutox/model.py:11-24, factory.py:15-38, handler.py:36-44 (keys `model_all.*`).

The features (normalisation, products, absolute differences) and the MLP run in
`libsonar_mi355.so` (`smi_head_featurize`, `smi_mlp_head_*`); there is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import re
from dataclasses import dataclass, field
from typing import Dict, List, Mapping, Optional, Tuple, Union

import torch

from . import _lib

BLASER_INPUT_FORMS = {"COMET", "QE"}
ACTIVATIONS = {"TANH": 1, "RELU": 0}


@dataclass
class BlaserConfig:
    """sonar/models/blaser/config.py:14-24."""

    input_form: str = "COMET"
    norm_emb: bool = True
    embedding_dim: int = 1024
    output_dim: int = 1
    hidden_dims: List[int] = field(default_factory=lambda: [3072, 1536])
    dropout: float = 0.1
    activation: str = "TANH"
    output_act: bool = False


def get_blaser_config(arch: str) -> BlaserConfig:
    if arch == "basic_ref":
        return BlaserConfig(input_form="COMET")
    if arch == "basic_qe":
        return BlaserConfig(input_form="QE")
    raise ValueError(f"unknown BLASER architecture {arch!r} (basic_ref, basic_qe)")


@dataclass
class MutoxConfig:
    """sonar/models/mutox/config.py:13-18."""

    input_size: int = 1024


def _linear_layers(sd: Mapping[str, torch.Tensor], prefix: str) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    found = []
    for k, v in sd.items():
        m = re.fullmatch(re.escape(prefix) + r"((?:\d+\.)*\d+)\.weight", k)
        if m and v.dim() == 2:
            found.append((tuple(int(x) for x in m.group(1).split(".")), m.group(1)))
    if not found:
        raise ValueError(f"no Linear layers named {prefix}<i>.weight in the checkpoint")
    found.sort()
    return [(sd[f"{prefix}{name}.weight"], sd[f"{prefix}{name}.bias"]) for _, name in found]


def _tv(t: torch.Tensor, keep: list) -> _lib.smi_tensor:
    t = t.detach()
    if t.dtype not in (torch.float16, torch.float32):
        t = t.float()
    t = t.contiguous()
    keep.append(t)
    return _lib.smi_tensor(t.data_ptr(), _lib.SMI_F32 if t.dtype == torch.float32 else _lib.SMI_F16, int(t.is_cuda),
                           t.numel())


class _MlpHead:
    """Owns one `smi_mlp_head` handle."""

    def __init__(self, layers, input_dim: int, hidden_act: int, out_act: int, device: torch.device):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the MI355X SONAR engine needs device='cuda[:i]' (no CPU path)")
        # an index-less "cuda" means the CURRENT device (as the text / speech engines resolve it)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", idx)
        self.input_dim = input_dim
        self.out_dim = int(layers[-1][0].shape[0])
        n = len(layers)
        arr = (_lib.smi_mlp_head_layer * n)()
        keep: list = []
        prev = input_dim
        for i, (w, b) in enumerate(layers):
            if w.shape[1] != prev or b.shape[0] != w.shape[0]:
                raise ValueError(f"layer {i}: weight {tuple(w.shape)} / bias {tuple(b.shape)} do not chain from {prev}")
            arr[i].w, arr[i].b, arr[i].out_dim = _tv(w, keep), _tv(b, keep), int(w.shape[0])
            prev = int(w.shape[0])
        cfg = _lib.smi_mlp_head_config(input_dim, n, hidden_act, out_act)
        self._handle = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_init(self.device.index))
            _lib.check(self.lib.smi_mlp_head_create(C.byref(cfg), arr, C.byref(self._handle)))

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h:
            self.lib.smi_mlp_head_destroy(h)
            self._handle = None

    def featurize(self, form: int, src, mt, ref, norm: bool) -> Tuple[torch.Tensor, int]:
        ts = [t for t in (src, mt, ref) if t is not None]
        dt = torch.float32 if any(t.dtype not in (torch.float16,) for t in ts) else torch.float16
        prep = lambda t: None if t is None else t.to(self.device, dt).contiguous()
        src, mt, ref = prep(src), prep(mt), prep(ref)
        rows, d = src.shape
        for t in (mt, ref):
            if t is not None and t.shape != src.shape:
                raise ValueError("src, mt and ref embeddings must have the same shape")
        blocks = (1, 4, 6)[form]
        if blocks * d != self.input_dim:
            raise ValueError(f"embedding dim {d} does not match the head's input width {self.input_dim}")
        rows_pad = (rows + 127) // 128 * 128
        feats = torch.empty((rows_pad, blocks * d), dtype=torch.float16, device=self.device)
        p = lambda t: t.data_ptr() if t is not None else None
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_head_featurize(form, p(src), p(mt), p(ref),
                                                   _lib.SMI_F32 if dt == torch.float32 else _lib.SMI_F16, rows, d,
                                                   int(norm), feats.data_ptr(), _lib.current_stream_ptr()))
        return feats, rows

    def run(self, feats: torch.Tensor, rows: int, out_act: int = -1) -> torch.Tensor:
        out = torch.empty((rows, self.out_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.smi_mlp_head_forward(self._handle, feats.data_ptr(), rows, out_act, out.data_ptr(),
                                                     _lib.current_stream_ptr()))
        return out


def _unwrap(checkpoint) -> Mapping[str, torch.Tensor]:
    if isinstance(checkpoint, (str, bytes)) or hasattr(checkpoint, "__fspath__"):
        from .cards import resolve_checkpoint

        path, _ = resolve_checkpoint(checkpoint, "")  # card names resolve under $SONAR_CHECKPOINT_DIR
        checkpoint = torch.load(path, map_location="cpu", weights_only=False)
    if "model" in checkpoint and isinstance(checkpoint["model"], Mapping):  # blaser/handler.py:40-45
        return checkpoint["model"]
    return checkpoint


class BlaserModel(torch.nn.Module):
    """A multilayer perceptron over concatenated embeddings of source, translation and optionally
    reference, and their pointwise products and differences (blaser/model.py:27-31)."""

    def __init__(self, config: BlaserConfig, checkpoint, device: Union[str, torch.device] = "cuda:0"):
        super().__init__()
        if config.input_form not in BLASER_INPUT_FORMS:
            raise Exception(f"Unrecognized input format: {config.input_form}")
        if config.activation not in ACTIVATIONS:
            raise Exception(f"Unrecognized activation: {config.activation}")
        self.config = config
        self.input_form = config.input_form
        self.norm_emb = config.norm_emb
        self.embedding_dim = config.embedding_dim
        layers = _linear_layers(_unwrap(checkpoint), "mlp.")
        width = config.embedding_dim * (6 if config.input_form == "COMET" else 4)
        out_act = 1 if (config.output_act and len(layers) > 1) else 0   # model.py:76-80: Tanh only after hidden layers
        self.head = _MlpHead(layers, width, ACTIVATIONS[config.activation], out_act, torch.device(device))

    @torch.inference_mode()
    def forward(self, src: torch.Tensor, mt: torch.Tensor, ref: Optional[torch.Tensor] = None) -> torch.Tensor:
        if self.input_form == "COMET":
            if ref is None:
                raise ValueError("With the COMET input form of BLASER, a reference embedding must be provided.")
            feats, rows = self.head.featurize(2, src, mt, ref, self.norm_emb)
        else:
            feats, rows = self.head.featurize(1, src, mt, None, self.norm_emb)
        return self.head.run(feats, rows)


class MutoxClassifier(torch.nn.Module):
    """mutox/model.py:11-24 over the MLP of mutox/factory.py:15-38."""

    def __init__(self, config: MutoxConfig, checkpoint, device: Union[str, torch.device] = "cuda:0"):
        super().__init__()
        sd = _unwrap(checkpoint)
        sd = {k: v for k, v in sd.items() if k.startswith("model_all.")}  # mutox/handler.py:40-43
        layers = _linear_layers(sd, "model_all.")
        if layers[0][0].shape[1] != config.input_size:
            raise ValueError(f"checkpoint input width {layers[0][0].shape[1]} != config.input_size {config.input_size}")
        self.config = config
        self.head = _MlpHead(layers, config.input_size, ACTIVATIONS["RELU"], 0, torch.device(device))

    @torch.inference_mode()
    def forward(self, inputs: torch.Tensor, output_prob: bool = False) -> torch.Tensor:
        feats, rows = self.head.featurize(0, inputs, None, None, False)
        return self.head.run(feats, rows, 2 if output_prob else 0)


def load_blaser_model(checkpoint, arch: str = "basic_ref", device="cuda:0",
                      config: Optional[BlaserConfig] = None) -> BlaserModel:
    if isinstance(checkpoint, str) and checkpoint in ("blaser_2_0_ref", "blaser_2_0_qe"):
        arch = "basic_ref" if checkpoint.endswith("ref") else "basic_qe"
    return BlaserModel(config or get_blaser_config(arch), checkpoint, device)


def load_mutox_model(checkpoint, device="cuda:0", config: Optional[MutoxConfig] = None) -> MutoxClassifier:
    return MutoxClassifier(config or MutoxConfig(1024), checkpoint, device)
