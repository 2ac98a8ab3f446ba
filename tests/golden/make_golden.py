#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_golden.json: sha256 of the oracle's OBU bytes + reconstruction for a fixed set of
seeded inputs.  The oracle itself is pinned by dav1d (tests/test_oracle_dav1d.py); these vectors freeze its
decisions so that an unintended change of either the oracle or the HIP path is caught without a decoder.
Run: python tests/golden/make_golden.py"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import oracle
from tests.helpers.images import planes

CASES = [dict(w=64, h=64, bd=8, speed=4, q=121, mono=False, tiles=0), dict(w=129, h=101, bd=10, speed=4, q=121, mono=False, tiles=0),
         dict(w=128, h=85, bd=8, speed=10, q=121, mono=False, tiles=0), dict(w=200, h=120, bd=10, speed=1, q=121, mono=False, tiles=0),
         dict(w=256, h=200, bd=10, speed=4, q=66, mono=True, tiles=0), dict(w=300, h=270, bd=10, speed=4, q=121, mono=False, tiles=4),
         dict(w=136, h=72, bd=8, speed=6, q=200, mono=False, tiles=0), dict(w=200, h=136, bd=10, speed=1, q=66, mono=True, tiles=0)]

def run(c):
    pl = planes(c['h'], c['w'], seed=c['w'] + c['h'], bd=c['bd'], mono=c['mono'])
    cfg = oracle.make_config(c['w'], c['h'], c['bd'], c['mono'], c['q'], c['speed'], tiles=c['tiles'])
    r = oracle.encode_planes(cfg, pl)
    hr = hashlib.sha256()
    for p in r['recon']:
        hr.update(p.tobytes())
    return dict(c, obu_len=len(r['obu']), obu_sha256=hashlib.sha256(r['obu']).hexdigest(), recon_sha256=hr.hexdigest(), base_q_idx=r['base_q_idx'], tiles_out=list(r['tiles']))

if __name__ == '__main__':
    out = [run(c) for c in CASES]
    json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'oracle_golden.json'), 'w'), indent=1)
    for o in out:
        print(o['w'], o['h'], o['obu_len'], o['obu_sha256'][:12])
