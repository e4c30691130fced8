"""Generate tests/golden/xsim_margin_twin.pt -- pins the oracle's LASER margin scoring
(oracle/xsim.py: laser_xsim) to the formula LASER publishes in source/xsim.py, written out here
loop for loop as LASER writes `_score_margin`, on neighbours found by scikit-learn's brute-force
cosine search (standing in for faiss IndexFlatIP on normalised vectors: the same exact inner-product
k-NN).  float64 numpy throughout.

Run in the build container:  python tests/golden/make_golden_xsim_margin.py
"""
import os

import numpy as np
import torch
from sklearn.neighbors import NearestNeighbors

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xsim_margin_twin.pt")


def knn(q, base, k):
    nn = NearestNeighbors(n_neighbors=k, metric="cosine", algorithm="brute").fit(base)
    dist, idx = nn.kneighbors(q)
    return 1.0 - dist, idx


def score_margin(Dxy, Ixy, Ax, Ay, margin, k):       # LASER xsim.py: _score_margin
    nbex = Dxy.shape[0]
    scores = np.zeros((nbex, k))
    for i in range(nbex):
        for j in range(k):
            jj = Ixy[i, j]
            a = Dxy[i, j]
            b = (Ax[i] + Ay[jj]) / 2
            scores[i, j] = margin(a, b)
    return scores


def score_knn(x, y, k, margin):                       # LASER xsim.py: _score_knn
    nbex = x.shape[0]
    if margin == "cosine":
        _, indices = knn(x, y, 1)
        return indices.reshape(nbex)
    fn = {"ratio": lambda a, b: a / b, "distance": lambda a, b: a - b}[margin]
    Cos_xy, Idx_xy = knn(x, y, k)
    Cos_yx, _ = knn(y, x, k)
    Avg_xy = Cos_xy.mean(axis=1)
    Avg_yx = Cos_yx.mean(axis=1)
    scores = score_margin(Cos_xy, Idx_xy, Avg_xy, Avg_yx, fn, k)
    best = scores.argmax(axis=1)
    indices = np.zeros(nbex, dtype=np.int64)
    for i in range(nbex):
        indices[i] = Idx_xy[i, best[i]]
    return indices


def main():
    rng = np.random.default_rng(5)
    n, d = 400, 64
    y = rng.standard_normal((n, d))
    # hubs: a few y rows that are close to many x rows make cosine and margin retrieval disagree
    y[:8] = y[:8] * 0.2 + rng.standard_normal((1, d))
    x = y + 2.2 * rng.standard_normal((n, d))
    out = {"x": torch.from_numpy(x).float(), "y": torch.from_numpy(y).float(), "k": 4}
    xf, yf = out["x"].double().numpy(), out["y"].double().numpy()     # what the oracle sees (fp32 values)
    for m in ("cosine", "ratio", "distance"):
        idx = score_knn(xf, yf, 4, m)
        out[m + "_pred"] = torch.from_numpy(idx)
        out[m + "_err"] = int(n - np.equal(idx, np.arange(n)).astype(int).sum())
        print(m, "errors", out[m + "_err"], "/", n)
    assert not np.array_equal(out["cosine_pred"].numpy(), out["ratio_pred"].numpy())
    torch.save(out, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
