"""ORACLE -- test infrastructure only.  Never imported by the product package.

xsim cosine mining on the CPU in fp32.  The reference does not implement xsim
(it is only named, README.md:5); the nearest in-tree code is
`F.normalize(x) @ F.normalize(y).T` at tests/integration_tests/test_text_sonar.py:42-53.
The external definition restated here is LASER's xsim: nearest
neighbour by cosine, or by ratio / distance margin with k = 4 -- facebookresearch/LASER
`source/xsim.py` (`xSIM`, `_score_knn`, `_score_margin`; LASER is not a dependency of the reference
and is not vendored; the algorithm restated below is the published one: faiss IndexFlatIP k-NN in
both directions, Avg = mean of the k neighbour cosines, score = margin(cos, (Avg_x + Avg_y) / 2),
prediction = arg-max over the k forward candidates, error = #(prediction != i)).
The reference holds no golden vector or fixture for xsim; the cosine top-k is pinned against
scikit-learn's brute-force cosine NearestNeighbors (tests/golden/xsim_sklearn_twin.pt,
make_golden_xsim.py) and the margin variants against a loop-for-loop numpy statement of LASER's
`_score_margin` over scikit-learn neighbours (tests/golden/xsim_margin_twin.pt,
make_golden_xsim_margin.py).  LASER itself cannot run here (faiss absent): PARITY UNPINNED against
the original program, pinned against its published formula.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def cosine_topk(x: torch.Tensor, y: torch.Tensor, k: int, block: int = 4096):
    """Top-k cosine neighbours in y for every row of x. Returns (scores [nx,k], idx [nx,k]).
    Ties are broken towards the lower index (same total order as the kernel)."""
    xn = F.normalize(x.float(), dim=-1)
    yn = F.normalize(y.float(), dim=-1)
    scores, idxs = [], []
    for i in range(0, xn.shape[0], block):
        s = xn[i:i + block] @ yn.T
        # stable sort descending == (score desc, index asc)
        order = torch.sort(s, dim=1, descending=True, stable=True)
        scores.append(order.values[:, :k])
        idxs.append(order.indices[:, :k])
    return torch.cat(scores), torch.cat(idxs)


def xsim_error_rate(x: torch.Tensor, y: torch.Tensor, margin: str = "cosine", k: int = 4) -> float:
    """Fraction of rows i whose best match in y is not row i (x[i] <-> y[i] aligned)."""
    xn = F.normalize(x.float(), dim=-1)
    yn = F.normalize(y.float(), dim=-1)
    s = xn @ yn.T
    if margin == "ratio":
        kx = s.topk(min(k, s.shape[1]), dim=1).values.mean(dim=1)  # x -> y neighbourhood
        ky = s.topk(min(k, s.shape[0]), dim=0).values.mean(dim=0)  # y -> x neighbourhood
        s = s / (0.5 * (kx.unsqueeze(1) + ky.unsqueeze(0)))
    elif margin != "cosine":
        raise ValueError(margin)
    pred = s.argmax(dim=1)
    return float((pred != torch.arange(s.shape[0])).float().mean())


def laser_xsim(x: torch.Tensor, y: torch.Tensor, margin: str = "ratio", k: int = 4):
    """LASER source/xsim.py `_score_knn`: (error count, predicted y index per x row [n]).
    Restricts the re-scoring to the k forward neighbours, exactly as LASER does (the dense
    `xsim_error_rate` above re-scores ALL pairs and can differ when the best margin pair is not among
    the k nearest by cosine)."""
    n = x.shape[0]
    if margin == "cosine":
        _, idx = cosine_topk(x, y, 1)
        pred = idx[:, 0]
    else:
        kk = min(k, n)
        cos_xy, idx_xy = cosine_topk(x, y, kk)      # idx_y.search(x, k)
        cos_yx, _ = cosine_topk(y, x, kk)           # idx_x.search(y, k)
        avg_xy = cos_xy.mean(dim=1)
        avg_yx = cos_yx.mean(dim=1)
        b = 0.5 * (avg_xy.unsqueeze(1) + avg_yx[idx_xy])
        if margin == "ratio":
            scores = cos_xy / b
        elif margin == "distance":
            scores = cos_xy - b
        else:
            raise ValueError(margin)
        best = scores.argmax(dim=1)                 # first maximum, as numpy
        pred = idx_xy.gather(1, best.unsqueeze(1)).squeeze(1)
    err = int((pred != torch.arange(n)).sum())
    return err, pred


def synthetic_pairs(n: int, d: int = 1024, noise: float = 0.3, seed: int = 2):
    """Y unit-normalised N(0,1); X = normalise(Y[perm] + noise*N(0,1))."""
    g = torch.Generator().manual_seed(seed)
    y = F.normalize(torch.randn(n, d, generator=g), dim=-1)
    g2 = torch.Generator().manual_seed(seed + 1)
    perm = torch.randperm(n, generator=g2)
    x = F.normalize(y[perm] + noise * torch.randn(n, d, generator=g2) / (d ** 0.5), dim=-1)
    return x, y, perm
