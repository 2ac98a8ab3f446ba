"""Segmentation (oracle/av1o_segment.c): the integer helpers against independent restatements, the fit on real activity maps,
and the decoder-visible part -- segmentation_params, per-segment dequantisers, segment-id prediction -- pinned by dav1d."""
import ctypes as C
import math
import numpy as np
import pytest
from tests.helpers.images import planes


def _lib(oracle):
    L = oracle.lib()
    L.av1o_ilog2_q11.argtypes = [C.c_uint32]; L.av1o_ilog2_q11.restype = C.c_int
    L.av1o_seg_bucket.argtypes = [C.c_uint32]; L.av1o_seg_bucket.restype = C.c_int
    L.av1o_seg_symbol.argtypes = [C.c_int] * 3; L.av1o_seg_symbol.restype = C.c_int
    L.av1o_seg_pred.argtypes = [C.c_int] * 3 + [C.POINTER(C.c_int)]; L.av1o_seg_pred.restype = C.c_int
    return L


def test_ilog2_q11_against_float_log2(oracle):
    L = _lib(oracle)
    assert [L.av1o_ilog2_q11(1 << k) for k in range(0, 31, 5)] == [k << 11 for k in range(0, 31, 5)]
    rng = np.random.default_rng(5)
    xs = [3, 5, 7, 1000, 16384, 16385, 65535, (1 << 22) - 1] + [int(x) for x in rng.integers(1, 1 << 31, 300)]
    prev = -1
    for x in sorted(xs):
        v = L.av1o_ilog2_q11(x)
        assert v >= prev; prev = v                                      # monotone
        assert -2.0 <= v - math.log2(x) * 2048 <= 0.001, (x, v)        # floor-type: never above the real log, within two units below
    assert L.av1o_seg_bucket(16384) == (8 << 11) >> 3 and L.av1o_seg_bucket(1) == 0 and L.av1o_seg_bucket(0xFFFFFFFF) == 4095


def _neg_deinterleave(diff, ref, mx):                                   # AV1 spec 5.11.9, restated from the text
    if ref == 0: return diff
    if ref >= mx - 1: return mx - diff - 1
    if 2 * ref < mx:
        if diff <= 2 * ref: return ref + ((diff + 1) >> 1) if diff & 1 else ref - (diff >> 1)
        return diff
    if diff <= 2 * (mx - ref - 1): return ref + ((diff + 1) >> 1) if diff & 1 else ref - (diff >> 1)
    return mx - (diff + 1)


def test_segment_symbol_inverts_neg_deinterleave(oracle):
    L = _lib(oracle)
    for mx in range(1, 9):
        for pred in range(mx):
            syms = [L.av1o_seg_symbol(s, pred, mx) for s in range(mx)]
            assert sorted(syms) == list(range(mx))
            assert [_neg_deinterleave(d, pred, mx) for d in syms] == list(range(mx))
            assert syms[pred] == 0                                      # the predicted id is the cheapest symbol


def test_segment_prediction_rules(oracle):
    L = _lib(oracle)
    ctx = C.c_int()
    cases = {(-1, -1, -1): (0, 0), (-1, 2, -1): (2, 0), (-1, -1, 1): (1, 0), (3, 3, 3): (3, 2), (1, 1, 2): (1, 1), (1, 2, 1): (1, 1), (0, 1, 1): (1, 1), (0, 1, 2): (2, 0)}
    for (ul, u, l), (pred, c) in cases.items():
        assert L.av1o_seg_pred(ul, u, l, C.byref(ctx)) == pred and ctx.value == c, (ul, u, l)


def test_fit_on_textured_and_flat_frames(oracle):
    pl = planes(136, 200, seed=336, bd=10)
    r = oracle.encode_planes(oracle.make_config(200, 136, 10, False, 121, 4), pl)
    n, q = r['seg_n'], r['seg_qidx']
    assert 3 <= n <= 8
    assert all(q[i] <= q[i + 1] for i in range(n - 1)) and q[0] < q[n - 1]           # segment 0 = largest scale = finest quantiser
    assert q[0] <= r['base_q_idx'] <= q[n - 1] and min(q[:n]) >= 1
    seg = r['m_skip'] >> 1
    assert seg.max() <= n - 1 and len(np.unique(seg)) >= 2                              # several segments in use
    # one scale everywhere: nothing to fit, segmentation stays off (and Tune::Psnr has no scales at all)
    flat = [np.full((64, 64), 300, np.uint16)] * 3
    assert oracle.encode_planes(oracle.make_config(64, 64, 10, False, 121, 4), flat)['seg_n'] == 0
    assert oracle.encode_planes(oracle.make_config(200, 136, 10, False, 121, 4, tune_psnr=1), pl)['seg_n'] == 0


@pytest.mark.parametrize('w,h,bd,speed,q,mono,tiles', [(200, 136, 10, 4, 121, False, 0), (129, 101, 8, 6, 60, False, 0), (256, 200, 10, 4, 66, True, 0),
                                                       (300, 270, 10, 2, 150, False, 4), (136, 72, 8, 4, 10, False, 0), (200, 120, 8, 9, 235, False, 0)])
def test_dav1d_decodes_segmented_streams(oracle, avifdec, w, h, bd, speed, q, mono, tiles):
    """segmentation_params, the ALT_Q dequantisers and read_segment_id (prediction, skipped blocks, contexts): all decoder-visible."""
    pl = planes(h, w, seed=w + h, bd=bd, mono=mono)
    r = oracle.encode_planes(oracle.make_config(w, h, bd, mono, q, speed, tiles=tiles), pl)
    assert r['seg_n'] >= 3
    d = avifdec.decode(oracle.container(r['obu'], None, w, h, bd, mono_color=int(mono)))
    for a, b in zip(d['planes'], r['recon']):
        assert np.array_equal(a, b)
