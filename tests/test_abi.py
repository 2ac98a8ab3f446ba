"""The C-ABI library loads and exports every symbol include/mi_avif.h declares; without a GPU the product
fails loudly instead of falling back to any CPU path."""
import ctypes as C
import os
import re
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    hdr = open(os.path.join(ROOT, 'include', 'mi_avif.h')).read()
    hdr = re.sub(r'/\*.*?\*/', '', hdr, flags=re.S)
    return sorted(set(re.findall(r'\b(mi_[a-z0-9_]+)\s*\(', hdr)))


def test_library_exports_all_declared_symbols():
    import cavif_rs_amd as m
    L = m.load_library()
    names = _declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(L, n), 'libmi_avif.so does not export %s' % n
    assert b'gfx950' in L.mi_version()


def test_container_matches_oracle(oracle):
    """mi_avif_serialize (avif-serialize stand-in) is byte-identical to the oracle's container writer."""
    import cavif_rs_amd as m
    L = m.load_library()
    color, alpha = bytes(range(200)) * 3, bytes(range(100))
    for a in (None, alpha):
        out = C.POINTER(C.c_uint8)()
        n = L.mi_avif_serialize(color, len(color), a, len(a) if a else 0, 640, 480, 10, 6, 0, None, 0, C.byref(out))
        got = bytes(bytearray(out[:n])); L.mi_free(out)
        assert got == oracle.container(color, a, 640, 480, 10)
        assert got[4:12] == b'ftypavif'


def test_no_gpu_fails_loudly():
    """No HIP device -> NoDevice error, never a silent CPU fallback (only meaningful on the GPU-less CI box)."""
    import cavif_rs_amd as m
    if m.device_count() > 0:
        pytest.skip('a GPU is present')
    img = np.zeros((16, 16, 3), np.uint8)
    with pytest.raises(m.AvifError) as e:
        m.Encoder().encode_rgb(img)
    assert e.value.code in (5, 4)
    with pytest.raises(m.AvifError):
        m.BatchEncoder(m.Encoder(), 1, 16, 16)
    with pytest.raises(m.AvifError):
        m.encode_planes([np.zeros((16, 16), np.uint8)] * 3)


def test_exif_item_round_trips_through_libavif(oracle):
    """Encoder::with_exif (ravif/src/av1encoder.rs:208-218, :470-472): the bytes come back from an independent
    reader (Pillow's libavif) and the image still decodes to the same planes as without the Exif item."""
    import io
    PIL = pytest.importorskip('PIL.Image')
    from tests.helpers import avifdec
    if not avifdec.available():
        pytest.skip('no bundled libavif')
    import cavif_rs_amd as m
    L = m.load_library()
    rng = np.random.default_rng(5)
    planes = [rng.integers(0, 256, (32, 48)).astype(np.uint16) for _ in range(3)]
    obu = oracle.encode_planes(oracle.make_config(48, 32, bit_depth=8, quantizer=100, speed=6), planes)['obu']
    exif = b'MM\x00\x2a\x00\x00\x00\x08\x00\x01\x01\x31\x00\x02\x00\x00\x00\x04abc\x00\x00\x00\x00\x00'   # TIFF header + Software="abc"
    outs = []
    for ex in (None, exif):
        out = C.POINTER(C.c_uint8)()
        n = L.mi_avif_serialize(obu, len(obu), None, 0, 48, 32, 8, 6, 0, ex, len(ex) if ex else 0, C.byref(out))
        outs.append(bytes(bytearray(out[:n]))); L.mi_free(out)
    plain, with_exif = outs
    assert len(with_exif) > len(plain) + len(exif)
    a, b = avifdec.decode(plain), avifdec.decode(with_exif)
    for p, q in zip(a['planes'], b['planes']):
        assert np.array_equal(p, q)
    im = PIL.open(io.BytesIO(with_exif))
    got = im.info.get('exif')
    assert got is not None and exif in bytes(got)       # Pillow prepends its own b'Exif\\0\\0' marker
