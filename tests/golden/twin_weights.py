"""Deterministic weights of the GPU-sized HuggingFace twins (no transformers needed).

`make_golden_gpu_twin.py` loads these tensors into HuggingFace `M2M100Encoder` / `M2M100Decoder`
(strict state-dict load) and stores only the inputs and HF's outputs; the GPU tests rebuild the same
tensors here, feed them to the HIP engines through the reference's checkpoint converters, and
compare with the stored outputs.  Sizes are the smallest the engines accept (model_dim = heads * 64,
a multiple of 256)."""
import zlib

import torch

D, H, F, L, V, MAXPOS = 256, 4, 512, 2, 300, 64


def _tensor(name: str, shape, seed: int) -> torch.Tensor:
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) ^ seed) & 0x7FFFFFFF)
    t = torch.randn(*shape, generator=g)
    if "layer_norm" in name and name.endswith("weight"):
        return 1.0 + 0.1 * t
    if name == "embed_tokens.weight":
        t = 0.02 * t
        t[2] *= 2.5      # fairseq id 2 = </s>: greedy decoding then stops at different lengths
        return t
    # layers strong against the (tied) embedding, so that generation does not just echo its input
    return (1.5 / shape[-1] ** 0.5) * t if len(shape) == 2 else 0.05 * t


def _layer(prefix: str, cross: bool):
    names = []
    for blk in ["self_attn"] + (["encoder_attn"] if cross else []):
        for p in ("k_proj", "v_proj", "q_proj", "out_proj"):
            names += [(f"{prefix}{blk}.{p}.weight", (D, D)), (f"{prefix}{blk}.{p}.bias", (D,))]
        names += [(f"{prefix}{blk}_layer_norm.weight", (D,)), (f"{prefix}{blk}_layer_norm.bias", (D,))]
    names += [(f"{prefix}fc1.weight", (F, D)), (f"{prefix}fc1.bias", (F,)), (f"{prefix}fc2.weight", (D, F)),
              (f"{prefix}fc2.bias", (D,)), (f"{prefix}final_layer_norm.weight", (D,)),
              (f"{prefix}final_layer_norm.bias", (D,))]
    return names


def hf_state_dict(kind: str, seed: int = 7):
    """State dict of HF M2M100Encoder (kind='encoder') / M2M100Decoder ('decoder'), minus the
    sinusoidal buffer -- which is also the fairseq checkpoint layout the reference converts
    (sonar/models/sonar_text/handler.py:71-82, 139-159)."""
    names = [("embed_tokens.weight", (V, D))]
    for i in range(L):
        names += _layer(f"layers.{i}.", cross=kind == "decoder")
    names += [("layer_norm.weight", (D,)), ("layer_norm.bias", (D,))]
    salt = 0 if kind == "encoder" else 0x5151
    return {n: _tensor(n, s, seed ^ salt) for n, s in names}


def fairseq_checkpoint(kind: str, seed: int = 7):
    sd = hf_state_dict(kind, seed)
    if kind == "decoder":
        sd["output_projection.weight"] = sd["embed_tokens.weight"].clone()   # tied in fairseq checkpoints
    sd["version"] = torch.tensor([3.0])
    sd["embed_positions._float_tensor"] = torch.zeros(1)
    return {"state_dict": sd}


TO_HF = torch.arange(V)
TO_HF[0], TO_HF[1], TO_HF[2], TO_HF[3] = 1, 3, 0, 2      # SONAR pad/unk/bos/eos -> fairseq/HF ids
