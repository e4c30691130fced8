"""xsim cosine / margin mining on the MI355X engine.

The reference names xsim as SONAR's evaluation (README.md:5) and computes
similarities as `F.normalize(x) @ F.normalize(y).T`
(tests/integration_tests/test_text_sonar.py:42-53, examples/sonar_text_demo.ipynb).
Here the similarity matrix is never materialised: `smi_xsim_topk` streams
128x128 score tiles out of the MFMA pipeline into a running top-k.

Multi-GPU (sharded_topk): X rows are sharded over ranks, every rank's Y shard
is all-gathered once over RCCL/xGMI (2 GB for 1M x 1024 fp16 -- small next to
288 GB HBM), then each rank mines its X shard against all of Y with no further
exchange; margin scoring needs one more all-gather of the k-NN means.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib


def _prep(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("xsim runs on a HIP device only (no CPU path); move the embeddings to cuda")
    if t.dim() != 2:
        raise ValueError("embeddings must be [rows, dim]")
    if t.dtype not in (torch.float16, torch.float32):
        t = t.float()
    return t.contiguous()


def normalize_rows(t: torch.Tensor) -> torch.Tensor:
    """fp16 L2-normalised copy, zero-padded to a multiple of 256 rows (engine layout)."""
    t = _prep(t)
    lib = _lib.load()
    rows, d = t.shape
    if rows == 0:
        raise ValueError("empty embedding matrix")
    pad = int(lib.smi_xsim_padded_rows(rows))
    out = torch.empty((pad, d), dtype=torch.float16, device=t.device)
    with torch.cuda.device(t.device):
        _lib.check(lib.smi_xsim_normalize(t.data_ptr(), _lib.SMI_F32 if t.dtype == torch.float32 else _lib.SMI_F16,
                                          rows, d, out.data_ptr(), _lib.current_stream_ptr()))
    return out


def topk_normalized(xn: torch.Tensor, nx: int, yn: torch.Tensor, ny: int, k: int = 1,
                    y_index_offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mine on already normalised + padded matrices (see normalize_rows)."""
    lib = _lib.load()
    d = xn.shape[1]
    if yn.shape[1] != d:
        raise ValueError("x and y must have the same dimension")
    if not 1 <= k <= 8:
        raise ValueError("k must be in [1, 8]")
    ws_bytes = int(lib.smi_xsim_workspace_bytes(nx, ny, k))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xn.device)
    idx = torch.empty((nx, k), dtype=torch.int32, device=xn.device)
    score = torch.empty((nx, k), dtype=torch.float32, device=xn.device)
    with torch.cuda.device(xn.device):
        _lib.check(lib.smi_xsim_topk(xn.data_ptr(), nx, yn.data_ptr(), ny, d, k, y_index_offset,
                                     idx.data_ptr(), score.data_ptr(), ws.data_ptr(),
                                     _lib.current_stream_ptr()))
    return score, idx


def topk(x: torch.Tensor, y: torch.Tensor, k: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """For every row of x: (cosine scores [nx,k] fp32, indices into y [nx,k] int32), best first."""
    xn, yn = normalize_rows(x), normalize_rows(y)
    return topk_normalized(xn, x.shape[0], yn, y.shape[0], k)


def margin_scores(fwd_scores: torch.Tensor, fwd_idx: torch.Tensor, x_knn_mean: torch.Tensor,
                  y_knn_mean: torch.Tensor, margin: str = "ratio") -> torch.Tensor:
    """LASER margin over the k-NN candidates: s(x,y) / (0.5 * (mean_kNN(x) + mean_kNN(y)))."""
    denom = 0.5 * (x_knn_mean.unsqueeze(1) + y_knn_mean[fwd_idx.long().clamp_min(0)])
    if margin == "ratio":
        return fwd_scores / denom
    if margin == "distance":
        return fwd_scores - denom
    raise ValueError(margin)


def xsim_error(x: torch.Tensor, y: torch.Tensor, margin: str = "cosine", k: int = 4) -> Tuple[float, torch.Tensor]:
    """xsim error rate for aligned x[i] <-> y[i]; returns (error rate, predicted index per x row)."""
    if x.shape[0] != y.shape[0]:
        raise ValueError("xsim expects aligned x and y")
    xn, yn = normalize_rows(x), normalize_rows(y)
    n = x.shape[0]
    if margin == "cosine":
        _, idx = topk_normalized(xn, n, yn, n, 1)
        pred = idx[:, 0].long()
    else:
        kk = min(k, n)
        fs, fi = topk_normalized(xn, n, yn, n, kk)
        bs, _ = topk_normalized(yn, n, xn, n, kk)
        m = margin_scores(fs, fi, fs.mean(dim=1), bs.mean(dim=1), margin)
        pred = fi.long().gather(1, m.argmax(dim=1, keepdim=True)).squeeze(1)
    err = (pred != torch.arange(n, device=pred.device)).float().mean().item()
    return err, pred
