"""the oracle tries PARTITION_HORZ / PARTITION_VERT on 8x8 nodes (8x4 / 4x8 blocks with 2:1 transforms, their scans, context
tables, tx_depth and partition syntax).  dav1d must decode its streams to exactly the encoder's reconstruction, and the rectangular blocks must
really be chosen; the HIP path's equality with this oracle is the business of the -m gpu parity tests and of tests/test_emu_kernels.py."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SCRIPT = r'''
import sys, json, numpy as np
sys.path.insert(0, %r)
from tests.helpers import oracle, avifdec
from tests.helpers.images import planes
oracle.lib()
out = []
for (w, h, bd, speed, q, mono) in [(129, 101, 8, 4, 121, 0), (200, 120, 10, 1, 121, 0), (256, 200, 10, 4, 66, 1), (300, 270, 10, 4, 121, 0), (136, 72, 8, 4, 10, 0)]:
    pl = planes(h, w, seed=w + h, bd=bd, mono=bool(mono))
    r = oracle.encode_planes(oracle.make_config(w, h, bd, bool(mono), q, speed), pl)
    d = avifdec.decode(oracle.container(r['obu'], None, w, h, bd, mono_color=int(mono)))
    same = all(np.array_equal(a, b) for a, b in zip(d['planes'], r['recon']))
    out.append(dict(case=[w, h, bd, speed, q, mono], same=bool(same), rect_cells=int(np.count_nonzero(np.asarray(r['m_bsize']) > 4)), bytes=len(r['obu'])))
print(json.dumps(out))
'''


def test_rect_partition_oracle_decodes_bit_exactly(avifdec):
    env = dict(os.environ)
    p = subprocess.run([sys.executable, '-c', SCRIPT % ROOT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    rows = json.loads(p.stdout.strip().splitlines()[-1])
    assert all(r['same'] for r in rows), rows
    assert sum(r['rect_cells'] for r in rows) > 200, rows            # the rectangular blocks are really chosen
