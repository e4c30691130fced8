"""CPU oracle (TEST INFRASTRUCTURE ONLY -- never imported by the product path) for the embedding
heads: plain fp32 torch restatement of

  * BLASER  sonar/models/blaser/model.py:82-125: `F.normalize` of src / mt / ref when `norm_emb`
    (:89-93), `featurize_input` (:95-125: COMET = [ref, mt, src*mt, ref*mt, |mt-src|, |mt-ref|],
    QE = [src, mt, src*mt, |mt-src|]), then the MLP built at :61-80 (Linear / activation per hidden
    layer, output Linear, optional Tanh); dropout is inert in eval mode;
  * MuTox   sonar/models/mutox/factory.py:15-38 (Linear in-512, ReLU, Linear 512-128, ReLU,
    Linear 128-1) and model.py:18-24 (sigmoid when `output_prob`).

Parity PINNED: tests/golden/heads_reference.pt holds outputs of the reference's own modules run in the
build container (tests/golden/make_golden_heads.py); tests/test_oracle_heads_cpu.py checks this file
against them.
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


def linear_layers(state_dict: Dict[str, torch.Tensor], prefix: str) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    """(weight, bias) of the nn.Linear modules under `prefix`, in module order
    (BLASER: `mlp.<i>.weight`; MuTox: `model_all.<j>.1.weight`)."""
    found = []
    for k in state_dict:
        m = re.fullmatch(re.escape(prefix) + r"((?:\d+\.)*\d+)\.weight", k)
        if m and state_dict[k].dim() == 2:
            found.append((tuple(int(x) for x in m.group(1).split(".")), m.group(1)))
    found.sort()
    return [(state_dict[f"{prefix}{name}.weight"].float(), state_dict[f"{prefix}{name}.bias"].float())
            for _, name in found]


def blaser_features(src, mt, ref, input_form: str, norm_emb: bool) -> torch.Tensor:
    src, mt = src.float(), mt.float()
    ref = ref.float() if ref is not None else None
    if norm_emb:  # model.py:89-93
        src, mt = F.normalize(src), F.normalize(mt)
        ref = F.normalize(ref) if ref is not None else None
    if input_form == "COMET":  # model.py:98-113
        if ref is None:
            raise ValueError("With the COMET input form of BLASER, a reference embedding must be provided.")
        return torch.cat([ref, mt, src * mt, ref * mt, (mt - src).abs(), (mt - ref).abs()], dim=-1)
    if input_form == "QE":  # model.py:114-123
        return torch.cat([src, mt, src * mt, (mt - src).abs()], dim=-1)
    raise ValueError(f"Unrecognized input format: {input_form}")


def blaser_forward(state_dict, src, mt, ref=None, *, input_form="COMET", norm_emb=True, activation="TANH",
                   output_act=False) -> torch.Tensor:
    x = blaser_features(src, mt, ref, input_form, norm_emb)
    layers = linear_layers(state_dict, "mlp.")
    act = torch.tanh if activation == "TANH" else torch.relu
    for i, (w, b) in enumerate(layers):
        x = x @ w.T + b
        if i + 1 < len(layers):
            x = act(x)
    return torch.tanh(x) if output_act and len(layers) > 1 else x


def mutox_forward(state_dict, inputs, output_prob: bool = False) -> torch.Tensor:
    x = inputs.float()
    layers = linear_layers(state_dict, "model_all.")
    for i, (w, b) in enumerate(layers):
        if i:
            x = torch.relu(x)  # factory.py:22-30: ReLU precedes the 2nd and 3rd Linear
        x = x @ w.T + b
    return torch.sigmoid(x) if output_prob else x
