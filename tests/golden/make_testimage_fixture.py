#!/usr/bin/env python3
"""Golden input for BASELINE config 1: the pixels of the reference's own fixture tests/testimage.png (128x85, palette PNG;
used by tests/stdio.rs:4-43).  /root/reference does not travel to the GPU box, so its decoded content is committed here as
palette indices + palette (zlib), with this script; the tests rebuild a palette PNG from it and feed cavif_mi.
Run (in the build container): python tests/golden/make_testimage_fixture.py"""
import json, os, sys, zlib, base64
from PIL import Image
SRC = '/root/reference/tests/testimage.png'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'testimage_128x85.json')
im = Image.open(SRC)
assert im.mode == 'P' and im.size == (128, 85)
pal = im.getpalette()
idx = im.tobytes()
trns = im.info.get('transparency')
rgba = im.convert('RGBA').tobytes()
import hashlib
json.dump({'width': 128, 'height': 85, 'mode': 'P', 'palette': pal, 'transparency': list(trns) if isinstance(trns, bytes) else trns,
           'indices_zlib_b64': base64.b64encode(zlib.compress(idx, 9)).decode(), 'rgba_sha256': hashlib.sha256(rgba).hexdigest(),
           'source': 'kornelski/cavif-rs tests/testimage.png (decoded with Pillow)'}, open(OUT, 'w'))
print('wrote', OUT, len(idx), 'indices', 'rgba sha', hashlib.sha256(rgba).hexdigest()[:16])
