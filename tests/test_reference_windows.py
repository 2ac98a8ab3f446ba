"""The reference's own tests (ravif/src/lib.rs:43-147, tests/stdio.rs) re-expressed against the oracle: container
well-formedness, payload size windows, determinism across entry points.  (The same windows are applied to the
HIP path in test_gpu_parity.py through byte equality with the oracle.)"""
import io
import numpy as np
from tests.helpers.images import rgba_gradient, rgba_opaque


def test_encode8_with_alpha(oracle, avifdec):
    """lib.rs:43-69: 256x200 RGBA, q22 / alpha q22, 8-bit, speed 1, dirty alpha, 2 threads."""
    img = rgba_gradient()
    data, color, alpha = oracle.ravif_encode(img, quality=22, alpha_quality=22, speed=1, depth=8, alpha_mode=0, threads=2)
    assert 50 < color < 1000, color
    assert 50 < alpha < 1000, alpha
    d = avifdec.decode(data)
    assert d['alpha'] is not None and d['depth'] == 8 and (d['width'], d['height']) == (256, 200)


def test_encode8_opaque(oracle, avifdec):
    """lib.rs:71-119: 129x101 opaque RGBA, q33, speed 10, auto depth (=10), 1 thread; RGB entry point is byte-identical."""
    img = rgba_opaque()
    data, color, alpha = oracle.ravif_encode(img, quality=33, speed=10, depth=0, threads=1)
    assert alpha == 0
    assert 150 < color < 500, 'size = %d; expected ~= 215' % color
    d = avifdec.decode(data)
    assert d['alpha'] is None and d['depth'] == 10 and (d['width'], d['height']) == (129, 101)
    data2, color2, _ = oracle.ravif_encode(img[:, :, :3], quality=33, speed=10, depth=10, threads=1)
    assert data2 == data and color2 == color


def test_stdio_ftyp(oracle):
    """tests/stdio.rs:23: bytes [4..12] == 'ftypavif' for the 128x85 fixture geometry at speed 10."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, size=(85, 128, 3), dtype=np.uint8)
    data, _, _ = oracle.ravif_encode(img, speed=10)
    assert data[4:12] == b'ftypavif'
    from PIL import Image
    im = Image.open(io.BytesIO(data)); im.load()
    assert im.size == (128, 85)


def test_preminmax(oracle):
    """ravif/src/dirtyalpha.rs:126-135 verbatim known answers."""
    import ctypes as C
    L = oracle.lib()
    L.av1o_premultiplied_minmax.argtypes = [C.c_uint8, C.c_uint8, C.POINTER(C.c_uint8), C.POINTER(C.c_uint8)]
    for (px, a), want in {(100, 255): (100, 100), (100, 10): (78, 100), (100, 2): (8, 119), (100, 1): (16, 239), (255, 1): (15, 255)}.items():
        lo, hi = C.c_uint8(), C.c_uint8()
        L.av1o_premultiplied_minmax(px, a, C.byref(lo), C.byref(hi))
        assert (lo.value, hi.value) == want
    assert 100 * 10 // 255 == 78 * 10 // 255


def test_encode8_cleans_alpha(oracle):
    """lib.rs:121-147: Clean vs Dirty on a noisy RGBA image, q66 / alpha q88, speed 6, 1 thread."""
    from tests.helpers.images import rgba_noisy
    img = rgba_noisy()
    _, dcol, dalpha = oracle.ravif_encode(img, quality=66, alpha_quality=88, speed=6, alpha_mode=0, threads=1)
    _, ccol, calpha = oracle.ravif_encode(img, quality=66, alpha_quality=88, speed=6, alpha_mode=1, threads=1)
    assert calpha == dalpha
    assert 200 < calpha < 1000
    assert 2000 < ccol < 6000
    assert ccol < dcol / 2
