"""GPU parity of the individual HIP kernels (through the C ABI) against plain
PyTorch fp32 references of the same op on the same fp16-rounded inputs."""
import ctypes as C
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from sonar_amd import _lib

    lib = _lib.load()
    _lib.check(lib.smi_init(0))
    return lib


def _stream():
    return int(torch.cuda.current_stream().cuda_stream)


def _tm_swz(rr):
    q = (rr >> 2) & 3
    return q ^ ((q & 1) << 1)


def to_tile_major(a):
    """Independent (torch index arithmetic) statement of the tile-major layout documented in
    include/sonar_mi355.h: [rows/256][k/32] blocks of [256 rows][4 slots][8], slot = chunk ^ s(row),
    s(row) = q ^ ((q & 1) << 1) with q = (row >> 2) & 3."""
    rows, k = a.shape
    assert rows % 256 == 0 and k % 32 == 0
    blocks = a.view(rows // 256, 256, k // 32, 4, 8).permute(0, 2, 1, 3, 4)  # [rb, kb, rr, chunk, 8]
    rr = torch.arange(256, device=a.device)
    slot = torch.arange(4, device=a.device)
    chunk_of_slot = slot[None, :] ^ _tm_swz(rr)[:, None]                      # [rr, slot] -> chunk
    idx = chunk_of_slot[None, None, :, :, None].expand(rows // 256, k // 32, 256, 4, 8)
    return torch.gather(blocks, 3, idx).contiguous().view(-1)


def from_tile_major(flat, rows, k):
    blocks = flat.view(rows // 256, k // 32, 256, 4, 8)
    rr = torch.arange(256, device=flat.device)
    chunk = torch.arange(4, device=flat.device)
    slot_of_chunk = chunk[None, :] ^ _tm_swz(rr)[:, None]                     # involution
    idx = slot_of_chunk[None, None, :, :, None].expand(rows // 256, k // 32, 256, 4, 8)
    return torch.gather(blocks, 3, idx).permute(0, 2, 1, 3, 4).reshape(rows, k).contiguous()


@pytest.mark.parametrize("rows,k", [(256, 32), (512, 1024), (768, 8192)])
def test_pack_tile_major(lib, rows, k):
    from sonar_amd import _lib

    a = torch.randn(rows, k, device="cuda").half()
    tm = torch.empty(rows * k, device="cuda", dtype=torch.float16)
    _lib.check(lib.smi_pack_tile_major(a.data_ptr(), tm.data_ptr(), rows, k, 0, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(tm, to_tile_major(a))
    assert torch.equal(from_tile_major(tm, rows, k), a)
    back = torch.empty_like(a)
    _lib.check(lib.smi_pack_tile_major(tm.data_ptr(), back.data_ptr(), rows, k, 1, _stream()))
    torch.cuda.synchronize()
    assert torch.equal(back, a)
    assert lib.smi_pack_tile_major(a.data_ptr(), tm.data_ptr(), rows + 1, k, 0, _stream()) != 0


@pytest.mark.parametrize("m,n,k", [(256, 256, 64), (256, 512, 192), (512, 1024, 1024), (256, 256, 8192),
                                   (1024, 768, 256)])
@pytest.mark.parametrize("epi,out_tm", [(0, 0), (0, 1), (1, 1), (2, 0), (3, 0), (4, 0), (5, 1), (6, 0), (8, 0), (8, 1)])
def test_gemm_tn_tile_major(lib, m, n, k, epi, out_tm):
    """Tile-major operands (and output) on both tile engines against the same fp32 reference."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m * 5 + n * 3 + k + epi + out_tm)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    xt, wt = to_tile_major(x), to_tile_major(w)
    ref = x.float() @ w.float().T + bias
    flags = _lib.SMI_GEMM_IN_TM | (_lib.SMI_GEMM_OUT_TM if out_tm else 0)
    for sel in (1, 2):
        if epi in (2, 4):
            resid = torch.randn(m, n, device="cuda", generator=g)
            out = resid.clone()
            want = resid + (ref if epi == 2 else 0.5 * ref)
        elif epi == 8:   # fp16 residual stream; out_tm: the stream itself is tile-major (read-modify-write in place)
            resid = torch.randn(m, n, device="cuda", generator=g).half()
            out = to_tile_major(resid) if out_tm else resid.clone()
            want = resid.float() + ref
        elif epi == 3:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float32)
            want = ref
        elif epi == 6:   # GLU (the conformer's pointwise_conv1 reads a tile-major LayerNorm output), row-major out
            out = torch.full((m, n // 2), float("nan"), device="cuda", dtype=torch.float16)
            r4 = ref.view(m, n // 64, 2, 32)
            want = (r4[:, :, 0] * torch.sigmoid(r4[:, :, 1])).reshape(m, n // 2)
        else:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
            want = torch.relu(ref) if epi == 1 else (torch.nn.functional.silu(ref) if epi == 5 else ref)
        _lib.check(lib.smi_gemm_tn(epi | (sel << 8) | flags, xt.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                   out.data_ptr(), m, n, k, n // 2 if epi == 6 else n, _stream()))
        torch.cuda.synchronize()
        got = (from_tile_major(out.view(-1), m, n) if out_tm else out).float()
        assert torch.isfinite(got).all()
        err = (got - want).abs().max().item()
        scale = max(want.abs().max().item(), 1.0)
        assert err <= (2e-3 if epi in (0, 1, 5, 6, 8) else 2e-5) * scale, (sel, err, scale)
    # unsupported combinations are refused, not mis-computed
    assert lib.smi_gemm_tn(6 | _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM, xt.data_ptr(), wt.data_ptr(), None, out.data_ptr(),
                           m, n, k, n, _stream()) != 0   # GLU has no tile-major OUTPUT
    assert lib.smi_gemm_tn(0 | _lib.SMI_GEMM_OUT_TM, xt.data_ptr(), wt.data_ptr(), None, out.data_ptr(), m, n, k, n,
                           _stream()) != 0


@pytest.mark.parametrize("m,n,k", [(6144, 2048, 1024), (4096, 4096, 256), (32000, 2048, 1024)])
def test_gemm_v2_glu_tile_major(lib, m, n, k):
    """The GLU epilogue of the 4-wave engine (the conformer's pointwise_conv1: out[m][g*32+c] = a * sigmoid(b), a / b = columns
    g*64+c / g*64+32+c), tile-major [m][n/2] output, against the fp32 reference and the 8-wave engine's row-major GLU on the same
    operands; below the engine's tile threshold the combination is refused."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    xt, wt = to_tile_major(x), to_tile_major(w)
    ref = x.float() @ w.float().T + bias
    r4 = ref.view(m, n // 64, 2, 32)
    want = (r4[:, :, 0] * torch.sigmoid(r4[:, :, 1])).reshape(m, n // 2)
    flags = 6 | _lib.SMI_GEMM_IN_TM
    row = torch.full((m, n // 2), float("nan"), device="cuda", dtype=torch.float16)
    _lib.check(lib.smi_gemm_tn(flags | (2 << 8), xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), row.data_ptr(), m, n, k, n // 2, _stream()))
    outs = []
    with _lib.tuning(G2V2=1, G2V2_MIN=1):
        for rep in range(3):
            out = torch.full((m * n // 2,), float("nan"), device="cuda", dtype=torch.float16)
            _lib.check(lib.smi_gemm_tn(flags | _lib.SMI_GEMM_OUT_TM | (2 << 8), xt.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                                       out.data_ptr(), m, n, k, n // 2, _stream()))
            torch.cuda.synchronize()
            outs.append(out)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    got = from_tile_major(outs[0], m, n // 2).float()
    scale = max(want.abs().max().item(), 1.0)
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 2e-3 * scale
    assert (row.float() - want).abs().max().item() <= 2e-3 * scale
    assert (got - row.float()).abs().max().item() <= 2e-3 * scale
    with _lib.tuning(G2V2=0):   # no other engine writes the GLU output tile-major: refused
        assert lib.smi_gemm_tn(flags | _lib.SMI_GEMM_OUT_TM, xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), m, n, k, n // 2,
                               _stream()) != 0
    assert lib.smi_gemm_tn(flags | _lib.SMI_GEMM_OUT_TM, xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), m, n, k, n,
                           _stream()) != 0   # ldo must be n / 2


@pytest.mark.parametrize("m,n,k,valid_n", [(1280, 32768, 1024, 32768 - 50),   # 640 tiles: 2-3 per workgroup, masked last column tile
                                           (256, 65536, 1024, 65536),          # one row tile, every tile full
                                           (512, 16384, 256, 16384 - 255),     # shortest K loop (8 slices); one valid column in the last tile
                                           (1280, 256256, 1024, 256206)])      # the decoder's projection (beam 5 x batch 256)
def test_gemm_v2_tile_stats(lib, m, n, k, valid_n):
    """The logits projection with fused softmax statistics on the 4-wave engine (gemm_v2_stats_kernel) against an fp32
    restatement computed from the kernel's OWN rounded logits (so the check isolates the statistics), the logits against the
    fp32 product, and both against the 8-wave engine's fused pass (bit-equal logits; statistics equal to fp32 rounding)."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + n + k)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.1).half()
    xt, wt = to_tile_major(x), to_tile_major(w)
    scale = 0.7
    res = {}
    for v2 in (0, 1):
        with _lib.tuning(G2V2=v2, G2V2_MIN=1):
            runs = []
            for rep in range(2 if v2 else 1):
                out = torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
                tmax = torch.full((n // 256, m), float("nan"), device="cuda")
                tsum = torch.full((n // 256, m), float("nan"), device="cuda")
                _lib.check(lib.smi_gemm_tn_tile_stats(xt.data_ptr(), wt.data_ptr(), out.data_ptr(), m, n, k, scale, valid_n,
                                                      tmax.data_ptr(), tsum.data_ptr(), _stream()))
                torch.cuda.synchronize()
                runs.append((out, tmax, tsum))
            for r in runs[1:]:
                assert all(torch.equal(a, b) for a, b in zip(r, runs[0]))
            res[v2] = runs[0]
    assert torch.equal(res[0][0], res[1][0])          # no bias: the same K sum in both engines
    logits = from_tile_major(res[1][0], m, n).float()
    for c0 in range(0, m, 256):                       # fp32 product, a row tile at a time (the full matrix is 1.3 GB)
        want = x[c0:c0 + 256].float() @ w.float().T
        assert (logits[c0:c0 + 256] - want).abs().max().item() <= 2e-3 * max(want.abs().max().item(), 1.0)
    del want
    v = (logits * scale)
    v[:, valid_n:] = float("-inf")
    v = v.view(m, n // 256, 256)
    wmax = v.max(dim=2).values                        # [m, tiles]
    wsum = torch.exp(v - wmax.unsqueeze(2)).sum(dim=2)
    for v2 in (0, 1):
        tmax, tsum = res[v2][1].T, res[v2][2].T
        assert torch.isfinite(tmax).all() and torch.isfinite(tsum).all()
        assert (tmax - wmax).abs().max().item() <= 1e-5 * max(wmax.abs().max().item(), 1.0), v2
        assert ((tsum - wsum).abs() / wsum).max().item() <= 2e-5, v2
    assert (res[0][1] - res[1][1]).abs().max().item() <= 1e-5 * max(wmax.abs().max().item(), 1.0)
    assert lib.smi_gemm_tn_tile_stats(xt.data_ptr(), wt.data_ptr(), out.data_ptr(), m, n, k, -1.0, valid_n, tmax.data_ptr(),
                                      tsum.data_ptr(), _stream()) != 0


@pytest.mark.parametrize("m,n,k", [(6144, 4096, 768),      # 384 tiles on 256 workgroups, the shortest K loop the engine takes (24 slices)
                                   (6144, 4096, 1024),     # the encoder's K; some workgroups walk two tiles, some one
                                   (16384, 4096, 1280),    # 1024 tiles: XCD-owned raster, four tiles per workgroup, 40 slices
                                   (2560, 1024, 8192),     # 40 tiles, K = 8192: one tile per workgroup, long loop
                                   (512, 256, 1024)])      # two tiles: most of the chip idle
@pytest.mark.parametrize("epi", [0, 1, 5, 8, 9])
def test_gemm_v2_engine(lib, m, n, k, epi):
    """The 4-wave 256x256 engine (gemm_v2.hip: accumulators in AGPRs, inline-asm K steps, a ring that streams across tiles;
    epi 8 / 9: the tile-major fp16 residual stream, read-modify-write, old tile requested by asm loads under the last K
    steps) against the fp32 reference AND against the 8-wave engine on the same operands; three launches in a row must be
    bit-identical (the counted waits and the cross-tile stream are where a race would show)."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + 3 * n + k + epi)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    xt, wt = to_tile_major(x), to_tile_major(w)
    ref = x.float() @ w.float().T + bias
    resid = None
    if epi in (8, 9):
        resid = torch.randn(m, n, device="cuda", generator=g).half()
        want = resid.float() + (ref if epi == 8 else 0.5 * ref)
    else:
        want = torch.relu(ref) if epi == 1 else (torch.nn.functional.silu(ref) if epi == 5 else ref)
    flags = _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM | (2 << 8)
    outs = {}
    for v2 in (0, 1):
        with _lib.tuning(G2V2=v2, G2V2_MIN=1):
            runs = []
            for rep in range(3 if v2 else 1):
                out = (to_tile_major(resid) if resid is not None
                       else torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16))
                _lib.check(lib.smi_gemm_tn(epi | flags, xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                           m, n, k, n, _stream()))
                torch.cuda.synchronize()
                runs.append(out)
            for r in runs[1:]:
                assert torch.equal(r, runs[0])
            outs[v2] = from_tile_major(runs[0], m, n).float()
    scale = max(want.abs().max().item(), 1.0)
    for v2 in (0, 1):
        assert torch.isfinite(outs[v2]).all()
        err = (outs[v2] - want).abs().max().item()
        assert err <= 2e-3 * scale, (v2, err, scale)
    # the two engines add the bias at different ends of the K sum: equal within one rounding of the fp16 result
    d = (outs[0] - outs[1]).abs().max().item()
    assert d <= 2e-3 * scale, d
    assert (outs[0] != outs[1]).float().mean().item() <= 0.02
    if epi == 8:   # without a bias: the residual kernel's two DMA instructions read a dummy address, the values are not used
        with _lib.tuning(G2V2=1, G2V2_MIN=1):
            out = to_tile_major(resid)
            _lib.check(lib.smi_gemm_tn(epi | flags, xt.data_ptr(), wt.data_ptr(), None, out.data_ptr(), m, n, k, n, _stream()))
            torch.cuda.synchronize()
        got = from_tile_major(out, m, n).float()
        assert (got - (want - bias)).abs().max().item() <= 2e-3 * scale


@pytest.mark.parametrize("m,n,k,ks", [(1280, 8192, 1024, 1),  # the decode step's FFN-inner projection: 8 x 32 = 256 units of 160 rows, relu, tile-major out
                                      (1280, 1024, 8192, 8),  # its FFN-output projection: 8 x 4 x 8 = 256 units, fp16 slabs
                                      (1280, 512, 2048, 4),   # 64 units, 16 slices per unit
                                      (1280, 256, 256, 1),    # 8 units, the shortest K loop the unit takes (8 slices)
                                      (1536, 8192, 1024, 1),  # BASELINE configs[0] (1312 tokens): 8 x 32 units of 192 rows (5-slot ring)
                                      (1536, 1024, 8192, 8),  # ... and its split-K FFN output projection
                                      (1024, 8192, 1024, 1),  # 8 x 32 units of 128 rows
                                      (768, 1024, 4096, 8),   # 192-row units: 4 x 4 x 8 = 128 units
                                      (512, 2048, 512, 2)])   # 128-row units, 8 slices per unit
def test_gemm_v2_lone_units(lib, m, n, k, ks):
    """The 128 / 160 / 192 x 256 lone units (gemm_v2_lone.hip; 1280 rows = the C5 decode step, 1536 = C1's tokens): X pieces that
    straddle the 256-row blocks of the tile-major image, the half piece under an EXEC mask (160), three whole pieces and a
    5-slot ring (192), the run-time slot index -- against the fp32 reference and against the 256-row tiles (DEC_M160=0) on the
    same operands; three launches bit-identical."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + n + 7 * k + ks)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    xt, wt = to_tile_major(x), to_tile_major(w)
    ref = x.float() @ w.float().T + bias
    for epi in ((1, 0) if ks == 1 else (None,)):      # tile-major relu / bias outputs, or fp16 split-K slabs
        outs = {}
        for m160 in (0, 2):       # 2: the lone units for every K loop the ring can run (the default takes them from 32 slices up)
            with _lib.tuning(DEC_M160=m160):
                runs = []
                for rep in range(3 if m160 else 1):
                    if ks == 1:
                        out = torch.full((m * n,), float("nan"), device="cuda", dtype=torch.float16)
                        _lib.check(lib.smi_gemm_tn(epi | _lib.SMI_GEMM_IN_TM | _lib.SMI_GEMM_OUT_TM, xt.data_ptr(), wt.data_ptr(),
                                                   bias.data_ptr(), out.data_ptr(), m, n, k, n, _stream()))
                    else:
                        out = torch.full((ks, m, n), float("nan"), device="cuda", dtype=torch.float16)
                        _lib.check(lib.smi_gemm_tn_splitk(xt.data_ptr(), wt.data_ptr(), bias.data_ptr(), out.data_ptr(), m, n, k, ks, 1,
                                                          _lib.SMI_F16, _stream()))
                    torch.cuda.synchronize()
                    runs.append(out)
                for r in runs[1:]:
                    assert torch.equal(r, runs[0])
                outs[m160] = from_tile_major(runs[0], m, n).float() if ks == 1 else runs[0].float().sum(0)
        want = torch.relu(ref) if epi == 1 else ref
        scale = max(want.abs().max().item(), 1.0)
        for m160 in (0, 2):
            assert torch.isfinite(outs[m160]).all()
            err = (outs[m160] - want).abs().max().item()
            assert err <= (2e-3 if ks == 1 else 4e-3) * scale, (epi, m160, err, scale)
        assert (outs[0] - outs[2]).abs().max().item() <= 4e-3 * scale


@pytest.mark.parametrize("m,n,k,ks,tm", [(1280, 1024, 8192, 8, 1),      # the decode step's FFN output projection: 160 units, 256x256 engine
                                          (1280, 1024, 1024, 2, 0),      # its attention output projection: lone-tile units
                                          (256, 1024, 8192, 8, 1),       # small-batch encoder: 64x64 lone units
                                          (2816, 1024, 8192, 8, 1)])     # more than one round of 256x256 units: the 128x128 family
def test_splitk_f16_slabs_cancellation_and_saturation(lib, m, n, k, ks, tm):
    """fp16 split-K partial sums (smi_text_decoder_set_slab_dtype): (1) with K ranges that CANCEL -- partials of
    magnitude ~2 000 whose sum is O(1) -- the fp16-slab result stays within 2^-11 x sum |partial| of the fp32-slab result (the
    bound the header states; the reference rounds the full sum once and would keep ~1e-3 relative to the SUM); (2) a partial
    outside fp16's range saturates at +-65504 instead of becoming inf, so the folded sum stays finite."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + k + ks)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    kp = k // ks
    # make K parts 0 and 1 cancel: the same large rank-one term with opposite signs in x
    x[:, :kp] = 0
    x[:, kp:2 * kp] = 0
    x[:, 0] = 40.0
    x[:, kp] = -40.0
    w[:, 0] = 50.0
    w[:, kp] = 50.0           # partial 0 = +2000, partial 1 = -2000 everywhere (+ nothing else: the columns are zeroed)
    bias = torch.randn(n, device="cuda", generator=g)
    xa, wa = (to_tile_major(x), to_tile_major(w)) if tm else (x, w)

    def run(dt):
        parts = torch.full((ks, m, n), float("nan"), device="cuda", dtype=torch.float16 if dt == _lib.SMI_F16 else torch.float32)
        _lib.check(lib.smi_gemm_tn_splitk(xa.data_ptr(), wa.data_ptr(), bias.data_ptr(), parts.data_ptr(), m, n, k, ks, tm,
                                          dt, _stream()))
        torch.cuda.synchronize()
        return parts

    p32, p16 = run(_lib.SMI_F32), run(_lib.SMI_F16)
    ref = x.float() @ w.float().T + bias
    s32, s16 = p32.sum(0), p16.float().sum(0)
    assert torch.isfinite(p16).all()
    assert (s32 - ref).abs().max().item() <= 2e-3 * max(ref.abs().max().item(), 1.0)
    bound = p32.abs().sum(0) * 2.0 ** -11 + 1e-6
    assert ((s16 - s32).abs() <= bound).all(), ((s16 - s32).abs() / bound).max().item()
    assert p32[0].abs().min().item() > 1900 and p32[1].abs().min().item() > 1900      # the cancellation is really there
    print(f"split-K fp16 slabs {m}x{n}x{k}/{ks}: max |sum16 - sum32| {(s16 - s32).abs().max().item():.3f} on sums of "
          f"{s32.abs().max().item():.2f} with partials of {p32.abs().max().item():.0f}")

    # saturation: partial 0 = +80 000, partial 1 = -80 000 (both outside fp16), the true sum is small
    x[:, 0] = 200.0
    x[:, kp] = -200.0
    w[:, 0] = 400.0
    w[:, kp] = 400.0
    xa, wa = (to_tile_major(x), to_tile_major(w)) if tm else (x, w)
    p16 = run(_lib.SMI_F16)
    assert torch.isfinite(p16).all(), "an fp16 partial overflowed to inf (MODE.FP16_OVFL not in effect)"
    assert p16[1].float().min().item() == -65504.0
    assert p16[0].float().max().item() == 65504.0
    assert torch.isfinite(p16.float().sum(0)).all()


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 192), (256, 1024, 1024), (512, 2560, 128), (512, 4096, 256),
                                   (1024, 3072, 192), (256, 256, 4096), (256, 512, 512), (768, 256, 1024)])
def test_gemm_lone_units(lib, m, n, k):
    """The lone-tile engine (gemm_lone.hpp, 64x64 units): K loops shorter / equal / longer than the ring (1, 2, 3, 4, 16,
    64 K tiles), one and two workgroups per CU (4 ... 512 units; 768 units: back on the 128x128 ring), row-major and
    tile-major operands, the fp16 / fp32 / read-modify-write epilogues -- against fp32 torch and BIT-identical to
    round 3's ring (same MFMA order over K)."""
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m + n * 3 + k * 7)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    ref = x.float() @ w.float().T + bias
    resid32 = torch.randn(m, n, device="cuda", generator=g)
    resid16 = resid32.half()
    tm_ok = m % 256 == 0 and n % 256 == 0
    xt, wt = (to_tile_major(x), to_tile_major(w)) if tm_ok else (None, None)

    def run(epi, tm, out_tm):
        if epi == 2:
            out = resid32.clone()
        elif epi == 8:
            out = to_tile_major(resid16) if out_tm else resid16.clone()
        elif epi == 3:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float32)
        elif epi == 6:
            out = torch.full((m, n // 2), float("nan"), device="cuda", dtype=torch.float16)
        else:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
        flags = (_lib.SMI_GEMM_IN_TM if tm else 0) | (_lib.SMI_GEMM_OUT_TM if out_tm else 0)
        a, b = (xt, wt) if tm else (x, w)
        _lib.check(lib.smi_gemm_tn(epi | (1 << 8) | flags, a.data_ptr(), b.data_ptr(), bias.data_ptr(), out.data_ptr(),
                                   m, n, k, n // 2 if epi == 6 else n, _stream()))
        torch.cuda.synchronize()
        return from_tile_major(out.view(-1), m, n) if out_tm else out

    cases = [(0, 0, 0), (1, 0, 0), (2, 0, 0), (3, 0, 0), (6, 0, 0), (8, 0, 0)]
    if tm_ok:
        cases += [(0, 1, 1), (1, 1, 1), (3, 1, 0), (8, 1, 1), (6, 1, 0)]
    for epi, tm, out_tm in cases:
        with _lib.tuning(LONE=1, LONE16=0):       # the LDS-ring unit: same MFMA order over K as the ring
            got = run(epi, tm, out_tm)
        with _lib.tuning(LONE=0):
            old = run(epi, tm, out_tm)
        assert torch.equal(got, old), (epi, tm, out_tm)
        # the k-sliced unit (gemm_lone16.hpp: tile-major operands, K per unit 256 / 512 / 1024; everything else falls through to
        # the ring unit): its own summation order, equal within fp32 rounding of the accumulation
        with _lib.tuning(LONE=1, LONE16=1):
            got16 = run(epi, tm, out_tm)
        d16 = (got16.float() - got.float()).abs().max().item()
        assert d16 <= (2e-3 if epi != 3 else 2e-5) * max(got.float().abs().max().item(), 1.0), (epi, tm, out_tm, d16)
        got = got16
        if epi == 2:
            want = resid32 + ref
        elif epi == 8:
            want = resid16.float() + ref
        elif epi == 6:
            r4 = ref.view(m, n // 64, 2, 32)
            want = (r4[:, :, 0] * torch.sigmoid(r4[:, :, 1])).reshape(m, n // 2)
        else:
            want = torch.relu(ref) if epi == 1 else ref
        err = (got.float() - want).abs().max().item()
        scale = max(want.abs().max().item(), 1.0)
        assert err <= (2e-3 if epi in (0, 1, 6, 8) else 2e-5) * scale, (epi, tm, out_tm, err, scale)


@pytest.mark.parametrize("m,n,k", [(128, 128, 64), (256, 384, 256), (384, 1024, 1024), (128, 256, 8192),
                                   (256, 256, 64), (256, 256, 128), (256, 512, 192), (512, 1024, 1024),
                                   (256, 256, 8192), (1024, 768, 256)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4, 5, 6, 7, 8])
def test_gemm_tn(lib, m, n, k, epi):
    from sonar_amd import _lib

    g = torch.Generator(device="cuda").manual_seed(m * 7 + n * 3 + k + epi)
    x = (torch.randn(m, k, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(n, k, device="cuda", generator=g) * 0.05).half()
    bias = torch.randn(n, device="cuda", generator=g)
    base = x.float() @ w.float().T
    # both tile engines: 1 = 128x128, 2 = 256x256 ping-pong (needs multiples of 256)
    for sel in ([1, 2] if m % 256 == 0 and n % 256 == 0 else [1]):
        use_bias = not (epi == 3 and sel == 1)  # fp32 store also runs without a bias (logits GEMM)
        ref = base + bias if use_bias else base.clone()
        ldo = n
        if epi in (2, 4):
            resid = torch.randn(m, n, device="cuda", generator=g)
            out = resid.clone()
            ref = resid + (ref if epi == 2 else 0.5 * ref)
        elif epi == 8:   # fp16 residual stream: out = f16(float(out) + x.w^T + b), one rounding
            resid = torch.randn(m, n, device="cuda", generator=g).half()
            out = resid.clone()
            ref = resid.float() + ref
        elif epi in (5, 7):
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
            ref = torch.nn.functional.silu(ref) if epi == 5 else torch.tanh(ref)
        elif epi == 6:   # GLU over 64-column groups [32 values | 32 gates] (rows of W pre-interleaved)
            ldo = n // 2
            out = torch.full((m, ldo), float("nan"), device="cuda", dtype=torch.float16)
            r4 = ref.view(m, n // 64, 2, 32)
            ref = (r4[:, :, 0] * torch.sigmoid(r4[:, :, 1])).reshape(m, ldo)
        elif epi == 3:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float32)
        else:
            out = torch.full((m, n), float("nan"), device="cuda", dtype=torch.float16)
            if epi == 1:
                ref = torch.relu(ref)
        _lib.check(lib.smi_gemm_tn(epi | (sel << 8), x.data_ptr(), w.data_ptr(), bias.data_ptr() if use_bias else None,
                                   out.data_ptr(), m, n, k, ldo, _stream()))
        torch.cuda.synchronize()
        got = out.float()
        assert torch.isfinite(got).all()
        err = (got - ref).abs().max().item()
        scale = max(ref.abs().max().item(), 1.0)
        # fp16 output: half-ulp rounding of the result; fp32 outputs: accumulation order only
        allowed = (2e-3 if epi in (0, 1, 5, 6, 7, 8) else 2e-5) * scale
        assert err <= allowed, (sel, err, scale)


@pytest.mark.parametrize("d", [256, 512, 768, 1024, 2048])
@pytest.mark.parametrize("tm", [0, 1])
def test_layernorm(lib, d, tm):
    from sonar_amd import _lib

    rows = 517
    pad = (rows + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(d)
    x = torch.randn(rows, d, device="cuda", generator=g) * 3 + 0.7
    w = torch.randn(d, device="cuda", generator=g)
    b = torch.randn(d, device="cuda", generator=g)
    out = torch.zeros(pad, d, device="cuda", dtype=torch.float16)
    _lib.check(lib.smi_layernorm(x.data_ptr(), w.data_ptr(), b.data_ptr(), 1e-5, out.data_ptr(), rows, d, tm,
                                 _stream()))
    torch.cuda.synchronize()
    if tm:
        out = from_tile_major(out.view(-1), pad, d)
    ref = torch.nn.functional.layer_norm(x, (d,), w, b, 1e-5)
    assert (out[:rows].float() - ref).abs().max().item() <= 4e-3 * ref.abs().max().item()
    assert (out[rows:] == 0).all()


def _attention_ref(qkv, lens, d, heads):
    t = qkv.shape[0]
    out = torch.zeros(t, d, device=qkv.device)
    start = 0
    dh = d // heads
    for L in lens:
        if L == 0:
            continue
        blk = qkv[start:start + L].float()
        q = blk[:, :d].view(L, heads, dh).transpose(0, 1)
        k = blk[:, d:2 * d].view(L, heads, dh).transpose(0, 1)
        v = blk[:, 2 * d:].view(L, heads, dh).transpose(0, 1)
        a = torch.softmax(q @ k.transpose(1, 2) * dh ** -0.5, dim=-1)
        out[start:start + L] = (a @ v).transpose(0, 1).reshape(L, d)
        start += L
    return out


@pytest.mark.parametrize("lens,heads", [([128] * 3, 4), ([1, 5, 64, 65, 127, 129, 200, 33], 4),
                                        ([514, 300, 7], 2), ([31, 32, 33, 63], 16)])
@pytest.mark.parametrize("tm", [0, 1, 2, 3])
def test_attention(lib, lens, heads, tm):
    from sonar_amd import _lib

    d = heads * 64
    t = sum(lens)
    pad = (t + 255) // 256 * 256
    g = torch.Generator(device="cuda").manual_seed(t + heads)
    qkv = (torch.randn(t, 3 * d, device="cuda", generator=g) * 1.5).half()
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    ctx = torch.full((pad, d), float("nan"), device="cuda", dtype=torch.float16)
    qkv_in = qkv
    if tm & 2:   # qkv handed over tile-major (K = 3d), rows padded to 256
        qkv_in = to_tile_major(torch.cat([qkv, torch.zeros(pad - t, 3 * d, device="cuda", dtype=torch.float16)]))
    _lib.check(lib.smi_attention(qkv_in.data_ptr(), cu.data_ptr(), ctx.data_ptr(), len(lens), max(lens), d, heads, tm,
                                 _stream()))
    torch.cuda.synchronize()
    ref = _attention_ref(qkv, lens, d, heads)
    got = (from_tile_major(ctx.view(-1), pad, d) if tm & 1 else ctx)[:t].float()
    assert torch.isfinite(got).all()
    assert (got - ref).abs().max().item() <= 6e-3, (got - ref).abs().max().item()


def _relpos_attention_ref(qkv, lens, d, heads, rp, rp_zero, u, v):
    """scores[i][j] = ((q_i + u) . k_j + (q_i + v) . rp[rp_zero + i - j]) / 8, per clip and head, fp32.  The engine forms q + u and
    q + v in fp16 (as an fp16 model would), so the reference rounds them the same way."""
    out = torch.zeros(sum(lens), d, device=qkv.device)
    o = 0
    for n in lens:
        q, k, vv = (qkv[o:o + n, i * d:(i + 1) * d].float().view(n, heads, 64).transpose(0, 1) for i in range(3))
        qu = (q + u.view(heads, 1, 64)).half().float()
        qv = (q + v.view(heads, 1, 64)).half().float()
        idx = (rp_zero + torch.arange(n, device=qkv.device)[:, None] - torch.arange(n, device=qkv.device)[None, :]).clamp(0, rp.shape[0] - 1)
        r = rp.float().view(rp.shape[0], heads, 64)                     # [rows, H, 64]
        pos = torch.einsum("hid,ijhd->hij", qv, r[idx])                  # [H, n, n]
        att = torch.softmax((qu @ k.transpose(1, 2) + pos) * 0.125, dim=-1)
        out[o:o + n] = (att @ vv).transpose(0, 1).reshape(n, d)
        o += n
    return out


@pytest.mark.parametrize("lens,heads", [([40], 2), ([129, 7, 300], 4), ([499] * 6, 2), ([64] * 40, 2), ([1, 33, 128, 257], 16),
                                        ([300] * 30, 4), ([97] * 100, 8),   # 360 / 800 workgroups: more than one per CU
                                        ([1999, 700], 2)])                 # 63 key blocks: the position-row ring wraps 12 times
def test_relpos_attention(lib, lens, heads):
    """The conformer's relative-position attention through `smi_relpos_attention` against an fp32 restatement: both kernels (per-wave
    global loads of the position rows + fp32 score pad; LDS ring + fp16 pad), row-major and tile-major output, clips shorter than
    a key block, more workgroups than the chip holds at once (the LDS-ring kernel shares a CU three ways), and twice in a row
    bit-identical."""
    from sonar_amd import _lib

    d = heads * 64
    t = sum(lens)
    pad = (t + 255) // 256 * 256
    tmax = max(lens)
    g = torch.Generator(device="cuda").manual_seed(t + heads)
    qkv = (torch.randn(t, 3 * d, device="cuda", generator=g) * 1.2).half()
    rp_rows = (2 * tmax - 1 + 127) // 128 * 128
    rp = (torch.randn(rp_rows, d, device="cuda", generator=g) * 0.8).half()
    u = torch.randn(d, device="cuda", generator=g) * 0.3
    v = torch.randn(d, device="cuda", generator=g) * 0.3
    cu = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32, device="cuda")
    ref = _relpos_attention_ref(qkv, lens, d, heads, rp, tmax - 1, u, v)
    qkv_tm = to_tile_major(torch.cat([qkv, torch.zeros(pad - t, 3 * d, device="cuda", dtype=torch.float16)]))
    for ring in (0, 1):
        for tm in ((0, 1, 2, 3) if ring else (0, 1)):   # bit 1: q | k | v handed over tile-major (the LDS-ring kernel only)
            outs = []
            for rep in range(2):
                ctx = torch.full((pad, d), float("nan"), device="cuda", dtype=torch.float16)
                with _lib.tuning(SPEECH_RP_LDS=ring):
                    _lib.check(lib.smi_relpos_attention((qkv_tm if tm & 2 else qkv).data_ptr(), cu.data_ptr(), rp.data_ptr(), tmax - 1,
                                                        rp_rows, u.data_ptr(), v.data_ptr(), ctx.data_ptr(), len(lens), tmax, d, heads, tm,
                                                        _stream()))
                torch.cuda.synchronize()
                outs.append((from_tile_major(ctx.view(-1), pad, d) if tm & 1 else ctx)[:t])
            assert torch.equal(outs[0], outs[1]), (ring, tm)
            got = outs[0].float()
            assert torch.isfinite(got).all(), (ring, tm)
            err = (got - ref).abs().max().item()
            assert err <= (8e-3 if ring else 6e-3), (ring, tm, err)
    with _lib.tuning(SPEECH_RP_LDS=0):   # the round-5 kernel has no tile-major q | k | v path: refused, not mis-read
        assert lib.smi_relpos_attention(qkv_tm.data_ptr(), cu.data_ptr(), rp.data_ptr(), tmax - 1, rp_rows, u.data_ptr(), v.data_ptr(),
                                        ctx.data_ptr(), len(lens), tmax, d, heads, 2, _stream()) != 0

