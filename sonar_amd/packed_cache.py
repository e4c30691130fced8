"""Packed checkpoint cache.

The released SONAR checkpoints are fp32 pickles in fairseq1 / fairseq2 key layouts (3.1 GB for the
text encoder).  Loading one costs a full unpickle, the key conversion of the reference's handlers
(sonar_text/handler.py:52-94,122-172, sonar_speech/handler.py:46-110) and an fp32 -> fp16 conversion
of every matrix.  The first load therefore writes what the engine actually consumes -- converted key
names, control-token rows already permuted, matrices in fp16 (the engine's GEMM operand type;
round-to-nearest-even, the same rounding `smi_*_create` applies), vectors (biases, LayerNorm,
BatchNorm statistics, relative-position biases) and the depthwise-convolution taps in fp32 -- as one
safetensors file next to a key derived from the source file's name, size and mtime.  Later loads
memory-map that file: no unpickling, no conversion, half the bytes; `smi_*_create` then only does the
device upload and the tile-major re-layout (a few ms per matrix on the GPU).

    SONAR_AMD_CACHE=<dir>   cache directory (default ~/.cache/sonar_amd/packed)
    SONAR_AMD_CACHE=0       disable the cache
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Callable, Dict, Mapping, Optional, Union

import torch

FORMAT_VERSION = 2  # v2: sdpa.u_bias / sdpa.v_bias stay fp32
_FP32_2D_SUFFIXES = ("depthwise_conv.weight", "sdpa.u_bias", "sdpa.v_bias")


def cache_dir() -> Optional[Path]:
    env = os.environ.get("SONAR_AMD_CACHE")
    if env is not None and env.strip() in ("0", "", "off", "false"):
        return None
    return Path(env) if env else Path.home() / ".cache" / "sonar_amd" / "packed"


def cache_file(src: Union[str, Path], kind: str) -> Optional[Path]:
    d = cache_dir()
    if d is None:
        return None
    src = Path(src)
    st = src.stat()
    return d / f"{src.name}.{st.st_size}.{st.st_mtime_ns}.{kind}.v{FORMAT_VERSION}.safetensors"


def pack_state_dict(sd: Mapping[str, object]) -> Dict[str, torch.Tensor]:
    """Engine-ready form of a converted (fairseq2-named) state dict: matrices fp16, the rest fp32."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if not isinstance(v, torch.Tensor):
            continue  # e.g. "version" leftovers
        t = v.detach()
        if not t.is_floating_point():
            continue  # num_batches_tracked and friends
        # fp32 like the engine keeps them: vectors, depthwise taps and the conformer's [heads, head_dim]
        # relative-position biases (u_bias / v_bias -- 2-D, but added to q in fp32, never a GEMM operand)
        if t.dim() >= 2 and not k.endswith(_FP32_2D_SUFFIXES):
            t = t.to(torch.float16)
        else:
            t = t.to(torch.float32)
        out[k] = t.contiguous().clone() if t.data_ptr() == v.data_ptr() else t.contiguous()
    return out


def load_converted(path: Union[str, Path], convert: Callable[[Mapping], Mapping[str, torch.Tensor]],
                   kind: str, stats: Optional[dict] = None) -> Mapping[str, torch.Tensor]:
    """`convert(torch.load(path))`, through the packed cache.  stats["cache"] = "hit" | "miss" | "off"."""
    from safetensors.torch import load_file, save_file

    cf = cache_file(path, kind)
    if cf is not None and cf.is_file():
        try:
            sd = load_file(str(cf))
            if stats is not None:
                stats["cache"], stats["file"] = "hit", str(cf)
            return sd
        except Exception:  # a truncated / foreign file: fall through and rebuild it
            pass
    sd = convert(torch.load(str(path), map_location="cpu", weights_only=False))
    if cf is None:
        if stats is not None:
            stats["cache"] = "off"
        return sd
    packed = pack_state_dict(sd)
    try:
        cf.parent.mkdir(parents=True, exist_ok=True)
        tmp = cf.with_suffix(f".tmp{os.getpid()}")
        save_file(packed, str(tmp))
        os.replace(tmp, cf)
    except OSError:
        pass  # read-only home etc.: the cache is an optimisation only
    if stats is not None:
        stats["cache"], stats["file"] = "miss", str(cf)
    return packed
