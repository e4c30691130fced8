"""Generate tests/golden/xsim_sklearn_twin.pt -- pins the oracle's cosine nearest-neighbour mining
(oracle/xsim.py: cosine_topk, the `F.normalize(x) @ F.normalize(y).T` + top-k of
tests/integration_tests/test_text_sonar.py:42-53 at mining scale) against scikit-learn's
brute-force `NearestNeighbors(metric="cosine")`, an independent implementation.  The ratio margin
(LASER's xsim, un-vendored) has no second implementation here and stays unpinned.

Run in the build container:  python tests/golden/make_golden_xsim.py
"""
import os

import numpy as np
import torch
from sklearn.neighbors import NearestNeighbors

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xsim_sklearn_twin.pt")


def main():
    g = torch.Generator().manual_seed(21)
    y = torch.randn(700, 96, generator=g)
    x = y[torch.randperm(700, generator=g)[:300]] + 0.8 * torch.randn(300, 96, generator=g)   # unnormalised on purpose
    nn = NearestNeighbors(n_neighbors=4, metric="cosine", algorithm="brute").fit(y.numpy().astype(np.float64))
    dist, idx = nn.kneighbors(x.numpy().astype(np.float64))
    torch.save({"x": x, "y": y, "idx": torch.from_numpy(idx), "cosine": torch.from_numpy(1.0 - dist)}, OUT)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
