"""Shared statement of "exact token ids" for the decoder parity tests (VERDICT r5 "weak" 2).

A best hypothesis must equal the fp32 oracle's, token for token.  The ONLY excuse is a near-tie measured BY THE ORACLE:
`oracle.text_decoder.beam_search_incremental(..., margins_out=...)` reports, per sentence, the smallest gap between
neighbouring candidates its own beam rules consumed (plus the first unconsumed one) over all steps, and the final score gap
between its best hypothesis and the runner-up.  An implementation whose arithmetic differs from the oracle's by less than
eps / 2 per candidate score can order two candidates differently only where that gap is below eps.  The engine's own report
(smi_text_decoder_last_margins) is kept as a CROSS-CHECK: it must agree with the oracle's decision margin -- an engine
that picked a wrong token and reported a small margin to excuse itself fails here.
"""


def oracle_excuses(omargin, eps):
    """omargin = (decision margin, final margin, decision margin without the cap step) of one sentence, from the oracle."""
    return omargin[0] < eps or omargin[1] < eps


def check_engine_margin(engine_margins, omargin, eps, where=""):
    """The engine's decision margin against the oracle's.  Both are minima over the same candidate rankings: equal within the
    arithmetic noise when the two searches ran the same path, and both below ~eps when they parted at a near-tie."""
    tol = max(2e-2, eps)
    e, o = float(engine_margins[0]), float(omargin[2])   # the engine ranks the forced EOS of the cap step under its FINAL margin
    assert abs(e - o) <= tol or (e < eps + tol and o < eps + tol), (
        f"{where}: engine decision margin {e:.4e} vs oracle {o:.4e} (tolerance {tol:.2e})")
