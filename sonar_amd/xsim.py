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
    ws_bytes = int(lib.smi_xsim_workspace_bytes(nx, ny, k, d))
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xn.device)
    idx = torch.empty((nx, k), dtype=torch.int32, device=xn.device)
    score = torch.empty((nx, k), dtype=torch.float32, device=xn.device)
    with torch.cuda.device(xn.device):
        _lib.check(lib.smi_xsim_topk(xn.data_ptr(), nx, yn.data_ptr(), ny, d, k, y_index_offset,
                                     idx.data_ptr(), score.data_ptr(), ws.data_ptr(), ws_bytes,
                                     _lib.current_stream_ptr()))
    return score, idx


def topk(x: torch.Tensor, y: torch.Tensor, k: int = 1) -> Tuple[torch.Tensor, torch.Tensor]:
    """For every row of x: (cosine scores [nx,k] fp32, indices into y [nx,k] int32), best first."""
    xn, yn = normalize_rows(x), normalize_rows(y)
    return topk_normalized(xn, x.shape[0], yn, y.shape[0], k)


def merge_topk(part_scores: torch.Tensor, part_idx: Optional[torch.Tensor] = None):
    """k-way merge of per-shard top-k lists [parts, n, k] (smi_xsim_merge_topk) -> (scores [n,k], idx [n,k] | None)."""
    lib = _lib.load()
    ps = part_scores.to(torch.float32).contiguous()
    parts, n, k = ps.shape
    pi = part_idx.to(torch.int32).contiguous() if part_idx is not None else None
    out_s = torch.empty((n, k), dtype=torch.float32, device=ps.device)
    out_i = torch.empty((n, k), dtype=torch.int32, device=ps.device) if pi is not None else None
    with torch.cuda.device(ps.device):
        _lib.check(lib.smi_xsim_merge_topk(ps.data_ptr(), pi.data_ptr() if pi is not None else None, parts, n, k,
                                           out_s.data_ptr(), out_i.data_ptr() if out_i is not None else None,
                                           _lib.current_stream_ptr()))
    return out_s, out_i


def margin_select(fwd_scores: torch.Tensor, fwd_idx: torch.Tensor, bwd_scores: Optional[torch.Tensor],
                  margin: str = "ratio", x_index_offset: int = 0, err_count: Optional[torch.Tensor] = None):
    """LASER margin re-scoring of the k-NN candidates on the device (smi_xsim_margin_select).
    Returns (predicted y index per x row int32 [nx], its margin score fp32 [nx]); `err_count` (device
    int32 [1]) is incremented by the number of rows whose prediction is not row index + x_index_offset."""
    if margin not in _lib.SMI_MARGIN:
        raise ValueError(margin)
    lib = _lib.load()
    fs = fwd_scores.to(torch.float32).contiguous()
    fi = fwd_idx.to(torch.int32).contiguous()
    nx, k = fs.shape
    bs = bwd_scores.to(torch.float32).contiguous() if bwd_scores is not None else None
    if bs is not None and bs.shape[1] != k:
        raise ValueError("forward and backward neighbour lists must have the same k")
    pred = torch.empty((nx,), dtype=torch.int32, device=fs.device)
    pm = torch.empty((nx,), dtype=torch.float32, device=fs.device)
    with torch.cuda.device(fs.device):
        _lib.check(lib.smi_xsim_margin_select(
            fs.data_ptr(), fi.data_ptr(), nx, k, bs.data_ptr() if bs is not None else None,
            bs.shape[0] if bs is not None else 0, _lib.SMI_MARGIN[margin], x_index_offset, pred.data_ptr(),
            pm.data_ptr(), err_count.data_ptr() if err_count is not None else None, _lib.current_stream_ptr()))
    return pred, pm


def xsim_error(x: torch.Tensor, y: torch.Tensor, margin: str = "cosine", k: int = 4) -> Tuple[float, torch.Tensor]:
    """xsim error rate for aligned x[i] <-> y[i] (LASER source/xsim.py); returns (error rate, predicted
    index per x row).  margin: "cosine" | "ratio" | "distance" (k nearest neighbours, LASER default 4)."""
    if x.shape[0] != y.shape[0]:
        raise ValueError("xsim expects aligned x and y")
    xn, yn = normalize_rows(x), normalize_rows(y)
    n = x.shape[0]
    errs = torch.zeros(1, dtype=torch.int32, device=xn.device)
    if margin == "cosine":
        fs, fi = topk_normalized(xn, n, yn, n, 1)
        pred, _ = margin_select(fs, fi, None, "cosine", 0, errs)
    else:
        kk = min(k, n)
        fs, fi = topk_normalized(xn, n, yn, n, kk)
        bs, _ = topk_normalized(yn, n, xn, n, kk)
        pred, _ = margin_select(fs, fi, bs, margin, 0, errs)
    return int(errs.item()) / n, pred.long()
