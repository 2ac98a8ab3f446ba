#!/usr/bin/env python3
"""Synthetic PNG inputs of the BASELINE configs: integer-only generator (cavif_rs_amd/synth.py), 8-bit truecolour,
no ancillary chunks, zlib level 1.  Usage: gen_synth_png.py OUTDIR [--configs 2 3 4s 4 5] [--count N]
  cfg2: 1 x 1920x1080 RGB (index 0)   cfg3: 1 x 4096x4096 RGBA (index 3)   cfg4: 256 x 1920x1080 RGB (index 0..255)
  cfg4s: the first 8 images of cfg4   cfg5: 1 x 7680x4320 RGB (index 5)"""
import argparse, os, struct, sys, zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from cavif_rs_amd.synth import synth_image


def write_png(path, img):
    h, w, ch = img.shape
    raw = b''.join(b'\x00' + img[y].tobytes() for y in range(h))
    def chunk(t, d):
        return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, 6 if ch == 4 else 2, 0, 0, 0)) + chunk(b'IDAT', zlib.compress(raw, 1)) + chunk(b'IEND', b''))


SETS = {'2': [(1920, 1080, 0, False)], '3': [(4096, 4096, 3, True)], '4': [(1920, 1080, i, False) for i in range(256)],
        '4s': [(1920, 1080, i, False) for i in range(8)], '5': [(7680, 4320, 5, False)]}
if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('outdir'); ap.add_argument('--configs', nargs='+', default=['2']); ap.add_argument('--count', type=int, default=0)
    a = ap.parse_args()
    for c in a.configs:
        d = os.path.join(a.outdir, 'cfg' + c); os.makedirs(d, exist_ok=True)
        items = SETS[c][:a.count] if a.count else SETS[c]
        for (w, h, idx, alpha) in items:
            write_png(os.path.join(d, 'synth_%04d.png' % idx), synth_image(w, h, index=idx, alpha=alpha))
        print('cfg' + c, len(items), 'files ->', d)
