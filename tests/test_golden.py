"""Committed golden vectors (tests/golden/oracle_golden.json, made by tests/golden/make_golden.py).  They are SELF-GENERATED: the oracle's own output frozen
at a commit (the oracle is pinned by dav1d and the reference's size windows, not by rav1e bytes -- parity with rav1e is unpinned), so
these tests catch unintended changes of the oracle or the HIP path, nothing more."""
import hashlib, json, os
import pytest
from tests.helpers.images import planes

G = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'oracle_golden.json')))


@pytest.mark.parametrize('g', G, ids=lambda g: '%dx%d_bd%d_s%d_q%d' % (g['w'], g['h'], g['bd'], g['speed'], g['q']))
def test_oracle_reproduces_self_generated_golden(oracle, g):
    pl = planes(g['h'], g['w'], seed=g['w'] + g['h'], bd=g['bd'], mono=g['mono'])
    cfg = oracle.make_config(g['w'], g['h'], g['bd'], g['mono'], g['q'], g['speed'], tiles=g['tiles'])
    r = oracle.encode_planes(cfg, pl)
    assert len(r['obu']) == g['obu_len'] and hashlib.sha256(r['obu']).hexdigest() == g['obu_sha256']
    assert r['base_q_idx'] == g['base_q_idx'] and list(r['tiles']) == g['tiles_out']


@pytest.mark.gpu
@pytest.mark.parametrize('g', G, ids=lambda g: '%dx%d_bd%d_s%d_q%d' % (g['w'], g['h'], g['bd'], g['speed'], g['q']))
def test_hip_reproduces_self_generated_golden(g):
    """The HIP path against the committed vectors alone (no oracle involved at run time)."""
    import cavif_rs_amd as m
    pl = planes(g['h'], g['w'], seed=g['w'] + g['h'], bd=g['bd'], mono=g['mono'])
    obu, rec = m.encode_planes(pl, g['bd'], g['q'], g['speed'], g['mono'], tiles=g['tiles'])
    hr = hashlib.sha256()
    for p in rec:
        hr.update(p.tobytes())
    assert len(obu) == g['obu_len'] and hashlib.sha256(obu).hexdigest() == g['obu_sha256']
    assert hr.hexdigest() == g['recon_sha256']
