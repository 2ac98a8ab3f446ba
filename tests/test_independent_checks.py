"""Independent checks against the common-mode risk VERDICT r01 named: the forward transform networks and the static rate table are
produced by this repo's own generators (tools/gen_txfm.py, tools/gen_tables.py) for BOTH the oracle and the HIP path, and dav1d only
pins the inverse side.  Here the oracle's forward transforms are compared with floating-point DCT / ADST bases (scipy / numpy, no
generated code involved) and the rate table with -log2 of the probabilities read straight from the CDF tables."""
import ctypes as C
import numpy as np
import pytest

DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, IDTX = 0, 1, 2, 3, 9


def _dct_basis(n):
    k, i = np.mgrid[0:n, 0:n]
    b = np.cos(np.pi * (2 * i + 1) * k / (2 * n)) * np.sqrt(2.0 / n)
    b[0] /= np.sqrt(2.0)
    return b                                               # orthonormal DCT-II, rows = basis functions


def _adst_basis(n):
    k, i = np.mgrid[0:n, 0:n]
    if n == 4:                                             # AV1's 4-point ADST is the DST-VII
        return np.sin(np.pi * (2 * k + 1) * (i + 1) / (2 * n + 1)) * 2.0 / np.sqrt(2 * n + 1)
    return np.sin(np.pi * (2 * k + 1) * (2 * i + 1) / (4 * n)) * np.sqrt(2.0 / n)     # DST-IV


@pytest.mark.parametrize('txs', [0, 1, 2, 3, 4])
@pytest.mark.parametrize('txtype', [DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST])
def test_forward_transform_against_float_bases(oracle, txs, txtype):
    n = 4 << txs
    if txtype != DCT_DCT and n > 16:
        pytest.skip('ADST exists up to 16 points')
    L = oracle.lib()
    L.av1o_fwd_txfm2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(txs * 7 + txtype)
    res = rng.integers(-255, 256, size=(n, n)).astype(np.int16)
    cs = min(n, 32)
    out = np.zeros((cs, cs), dtype=np.int32)
    L.av1o_fwd_txfm2d(res.ctypes.data, n, out.ctypes.data, txs, txtype, 8)
    col = _adst_basis(n) if txtype in (ADST_DCT, ADST_ADST) else _dct_basis(n)
    row = _adst_basis(n) if txtype in (DCT_ADST, ADST_ADST) else _dct_basis(n)
    ref = col @ res.astype(np.float64) @ row.T             # orthonormal 2-D transform
    # AV1's forward scaling: orthonormal x 8 for 4x4 .. 16x16, x 4 for 32x32, x 2 for 64x64 (the decoder divides it back out)
    scale = {4: 8.0, 8: 8.0, 16: 8.0, 32: 4.0, 64: 2.0}[n]
    ref = (ref * scale)[:cs, :cs]
    err = np.abs(out - ref)
    assert err.max() <= 2.0 + 0.002 * np.abs(ref).max(), (err.max(), np.abs(ref).max())   # integer butterflies: rounding noise only


@pytest.mark.parametrize('txs', [0, 1, 2, 3])
def test_forward_then_normative_inverse_reconstructs(oracle, txs):
    """fwd (encoder side) followed by the spec's inverse (pinned by dav1d) returns the residual up to rounding."""
    n = 4 << txs
    L = oracle.lib()
    L.av1o_fwd_txfm2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.av1o_inv_txfm2d_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(txs)
    for txtype in ([DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST, IDTX] if n <= 16 else [DCT_DCT, IDTX]):
        res = rng.integers(-200, 201, size=(n, n)).astype(np.int16)
        coef = np.zeros((n, n), dtype=np.int32)
        L.av1o_fwd_txfm2d(res.ctypes.data, n, coef.ctypes.data, txs, txtype, 8)
        dst = np.full((n, n), 512, dtype=np.uint16)
        L.av1o_inv_txfm2d_add(coef.ctypes.data, dst.ctypes.data, n, txs, txtype, 10)
        assert np.abs(dst.astype(int) - 512 - res).max() <= 2


def test_rate_table_is_minus_log2_of_the_default_cdfs(oracle):
    """cost[sym] == round-down of -log2(P(sym)) in 1/512 bit for every row the search prices (checked through av1o_cost_from_icdf)."""
    L = oracle.lib()
    L.av1o_cost_from_icdf.argtypes = [C.POINTER(C.c_uint16), C.c_int, C.c_int]; L.av1o_cost_from_icdf.restype = C.c_uint32
    rng = np.random.default_rng(1)
    for nsyms in (2, 3, 4, 5, 7, 13, 16):
        for _ in range(50):
            p = rng.integers(1, 1000, size=nsyms).astype(np.float64); p = p / p.sum()
            cdf = np.minimum(np.round(np.cumsum(p) * 32768), 32768).astype(int); cdf[-1] = 32768
            cdf = np.maximum.accumulate(np.maximum(cdf, np.arange(1, nsyms + 1)))
            icdf = (C.c_uint16 * (nsyms + 1))(*([32768 - c for c in cdf] + [0]))
            for s in range(nsyms):
                pr = (cdf[s] - (cdf[s - 1] if s else 0)) / 32768.0
                if pr <= 0:
                    continue
                want = -np.log2(pr) * 512.0
                got = L.av1o_cost_from_icdf(icdf, s, nsyms)
                assert abs(got - want) <= 1.5, (got, want)


@pytest.mark.parametrize('code,w,h', [(5, 4, 8), (6, 8, 4)])
@pytest.mark.parametrize('txtype', [DCT_DCT, ADST_DCT, DCT_ADST, ADST_ADST])
def test_two_to_one_forward_transforms_against_float_bases(oracle, code, w, h, txtype):
    """The 4x8 / 8x4 networks (PARTITION_HORZ / VERT blocks, round 3) come from the same generator for the oracle and the HIP path; dav1d pins only their inverse.
    Forward output == orthonormal 2-D transform x 8 (the 2:1 transforms carry the extra 1/sqrt(2): 4 x 2 x sqrt(2) / 2 x 5793/4096), to rounding noise."""
    L = oracle.lib()
    L.av1o_fwd_txfm2d.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.av1o_inv_txfm2d_add.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    rng = np.random.default_rng(code * 11 + txtype)
    for _ in range(20):
        res = rng.integers(-255, 256, size=(h, w)).astype(np.int16)
        out = np.zeros((h, w), dtype=np.int32)
        L.av1o_fwd_txfm2d(res.ctypes.data, w, out.ctypes.data, code, txtype, 8)
        col = _adst_basis(h) if txtype in (ADST_DCT, ADST_ADST) else _dct_basis(h)
        row = _adst_basis(w) if txtype in (DCT_ADST, ADST_ADST) else _dct_basis(w)
        ref = (col @ res.astype(np.float64) @ row.T) * 8.0
        err = np.abs(out - ref)
        assert err.max() <= 2.5 + 0.002 * np.abs(ref).max(), (err.max(), np.abs(ref).max())
        dst = np.full((h, w), 512, dtype=np.uint16)                       # ... and the normative inverse (dav1d-pinned) undoes it
        L.av1o_inv_txfm2d_add(out.ctypes.data, dst.ctypes.data, w, code, txtype, 10)
        assert np.abs(dst.astype(int) - 512 - res).max() <= 2


def test_segment_quantiser_index_rule_against_float(oracle):
    """R-2: a segment whose scale is s times the frame's mean gets the index whose AC step is nearest base_step / sqrt(s) in the log domain (never index 0).
    Float restatement over every base index, both bit depths and deviations of +-3 octaves; the oracle's integer logs are 1/2048-octave floor values, so
    an exact tie-break may differ by one table entry -- and then the two steps must be equally near."""
    import math
    L = oracle.lib()
    L.av1o_seg_qidx_for_dev.argtypes = [C.c_int] * 3; L.av1o_seg_qidx_for_dev.restype = C.c_int
    L.av1o_ac_step.argtypes = [C.c_int] * 2; L.av1o_ac_step.restype = C.c_int
    for bd in (8, 10):
        steps = [L.av1o_ac_step(bd, q) for q in range(256)]
        assert all(steps[i] <= steps[i + 1] for i in range(255)) and steps[0] == 4                   # the spec's Ac_Qlookup: monotone, 4 at index 0 (lossless) at every depth
        for base in range(1, 256, 7):
            for dev in range(-768, 769, 37):                                  # 1/256 octave
                got = L.av1o_seg_qidx_for_dev(base, bd, dev)
                target = math.log2(steps[base]) - dev / 512.0                 # log2(base_step / sqrt(2^(dev/256)))
                want = min(range(1, 256), key=lambda q: (abs(math.log2(steps[q]) - target), q))
                assert got >= 1
                if got != want:
                    assert abs(abs(math.log2(steps[got]) - target) - abs(math.log2(steps[want]) - target)) < 3.0 / 2048, (bd, base, dev, got, want)


def test_psy_boost_q14_against_float(oracle):
    """Tune::Psychovisual's SSIM-like boost (rav1e dist.rs cdef_dist_kernel as recalled): 4033/16384 * (s + d + 16384) / sqrt(4033^2 + s * d), in Q14 with exact
    integer rounding -- against the formula in floating point over a log-spaced grid of variances."""
    import math
    L = oracle.lib()
    L.av1o_psy_boost_q14.argtypes = [C.c_uint32] * 2; L.av1o_psy_boost_q14.restype = C.c_uint32
    grid = [0, 1, 7, 64, 500, 4033, 16384, 100000, 1 << 20, (1 << 22) - 1]
    for s in grid:
        for d in grid:
            want = 4033.0 * (s + d + 16384) / math.sqrt(4033.0 ** 2 + float(s) * d)
            got = L.av1o_psy_boost_q14(s, d)
            den = math.isqrt(16265089 + s * d)
            assert abs(got - 4033.0 * (s + d + 16384) / den) <= 0.5 + 1e-6          # exact rounding of the quotient by the integer square root
            assert abs(got - want) <= 1.0 + want * 2.0 / max(den, 1), (s, d, got, want)
    assert L.av1o_psy_boost_q14(0, 0) == 16384                                        # flat source, flat reconstruction: no boost
