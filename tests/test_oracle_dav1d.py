"""The oracle's pin: every stream it emits decodes in dav1d 1.5.3 (through the bundled libavif) to
exactly the oracle's own reconstruction, across the tool sets of the BASELINE configs."""
import numpy as np
import pytest
from tests.helpers.images import planes

CASES = [
    # w, h, bd, speed, quantizer, mono, tiles
    (64, 64, 8, 10, 121, False, 0),
    (64, 64, 8, 4, 121, False, 0),
    (128, 85, 8, 10, 121, False, 0),      # config 1 geometry (128x85 fixture size)
    (129, 101, 10, 4, 121, False, 0),     # odd sizes, 10-bit
    (200, 120, 10, 1, 121, False, 0),     # speed 1: blocks up to 64x64, full tx set, complex modes
    (200, 136, 10, 1, 66, True, 0),       # alpha-like 4:0:0 plane, blocks up to 64x64
    (256, 200, 10, 4, 66, True, 0),
    (300, 270, 10, 4, 121, False, 4),     # 2x2 tiles
    (136, 72, 8, 6, 200, False, 0),       # low quality
    (136, 72, 8, 4, 10, False, 0),        # near-lossless quantizer, Golomb-coded levels
    (72, 136, 8, 8, 160, False, 2),
]


@pytest.mark.parametrize('w,h,bd,speed,q,mono,tiles', CASES)
def test_dav1d_decodes_oracle_recon(oracle, avifdec, w, h, bd, speed, q, mono, tiles):
    pl = planes(h, w, seed=w + h, bd=bd, mono=mono)
    cfg = oracle.make_config(w, h, bd, mono, q, speed, tiles=tiles)
    r = oracle.encode_planes(cfg, pl)
    avif = oracle.container(r['obu'], None, w, h, bd, mono_color=int(mono))
    d = avifdec.decode(avif)
    assert d['depth'] == bd and d['width'] == w and d['height'] == h
    assert len(d['planes']) == len(pl)
    for a, b in zip(d['planes'], r['recon']):
        assert np.array_equal(a, b)
    # sanity: the reconstruction is a faithful picture, not just self-consistent
    mse = np.mean((pl[0].astype(float) - r['recon'][0]) ** 2)
    assert 10 * np.log10(((1 << bd) - 1) ** 2 / max(mse, 1e-9)) > 28


def smooth_planes(h, w, bd, seed):
    """A slowly varying 4:4:4 picture with one soft-edged patch: large blocks win the partition search."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    mx = (1 << bd) - 1
    out = []
    for k in range(3):
        a = 0.5 + 0.4 * np.sin(x / (90.0 + 20 * k) + k) * np.cos(y / (70.0 + 10 * k))
        a[h // 3:h // 3 + 40, w // 4:w // 4 + 50] += 0.2
        out.append(np.clip((a + rng.normal(0, 0.002, (h, w))) * mx, 0, mx).astype(np.uint16))
    return out


@pytest.mark.parametrize('w,h,bd,speed,q,bottomup', [(256, 192, 10, 1, 121, 1), (200, 136, 8, 1, 121, 1), (192, 128, 10, 1, 90, 0), (136, 200, 8, 1, 100, 1)])
def test_64x64_colour_blocks_are_chosen_and_decode(oracle, avifdec, w, h, bd, speed, q, bottomup):
    """R-4: speed <= 1 below the high-quality threshold asks for partition_range (4, 64) (ravif/src/av1encoder.rs:556-566).  A 64x64 block of a
    4:4:4 frame carries four 32x32 chroma transform blocks per plane (spec get_tx_size), each predicted from the ones before it; dav1d pins the
    syntax, the per-transform-block prediction, the contexts and the chroma transform edges of the deblocking filter."""
    pl = smooth_planes(h, w, bd, w + h)
    cfg = oracle.make_config(w, h, bd, False, q, speed, bottomup=bottomup)
    assert (cfg.part_min, cfg.part_max) == (4, 64)
    r = oracle.encode_planes(cfg, pl)
    n64 = int((r['m_bsize'] == 4).sum()) // 256
    assert n64 >= 1, 'no 64x64 colour block was chosen on a smooth picture'
    d = avifdec.decode(oracle.container(r['obu'], None, w, h, bd, mono_color=0))
    for a, b in zip(d['planes'], r['recon']):
        assert np.array_equal(a, b)


def test_identity_rgb_roundtrip_through_pillow(oracle):
    """--color=rgb --depth=8: Identity-matrix 4:4:4 decodes through PIL to RGB == GBR recon exactly."""
    import io
    from PIL import Image
    rgb = np.stack(planes(96, 112, seed=5), -1).astype(np.uint8)
    data, cs, _ = oracle.ravif_encode(rgb, quality=70, speed=6, color_model=1, depth=8)
    im = Image.open(io.BytesIO(data)); im.load()
    assert im.size == (112, 96)
    cfg = oracle.make_config(112, 96, 8, False, oracle.lib().av1o_quality_to_quantizer(70.0), 6, matrix=0)
    r = oracle.encode_planes(cfg, [rgb[:, :, 1], rgb[:, :, 2], rgb[:, :, 0]])
    dec = np.array(im.convert('RGB'))
    assert np.array_equal(dec[:, :, 1], r['recon'][0]) and np.array_equal(dec[:, :, 2], r['recon'][1]) and np.array_equal(dec[:, :, 0], r['recon'][2])
