"""Committed golden inputs (tests/golden/)."""
import base64, hashlib, io, json, os, zlib
import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'golden')


def testimage():
    """The reference's tests/testimage.png (128x85 palette PNG): returns (PIL 'P' image, RGBA uint8 array)."""
    from PIL import Image
    d = json.load(open(os.path.join(GOLDEN, 'testimage_128x85.json')))
    idx = zlib.decompress(base64.b64decode(d['indices_zlib_b64']))
    im = Image.frombytes('P', (d['width'], d['height']), idx)
    im.putpalette(d['palette'])
    if d.get('transparency') is not None:
        im.info['transparency'] = bytes(d['transparency']) if isinstance(d['transparency'], list) else d['transparency']
    rgba = np.array(im.convert('RGBA'))
    assert hashlib.sha256(rgba.tobytes()).hexdigest() == d['rgba_sha256']
    return im, rgba


def testimage_png_bytes():
    im, rgba = testimage()
    buf = io.BytesIO()
    kw = {}
    if im.info.get('transparency') is not None:
        kw['transparency'] = im.info['transparency']
    im.save(buf, format='PNG', **kw)
    return buf.getvalue(), rgba
