#!/usr/bin/env python3
"""The divergence ledger: every deliberate difference between this encoder's search and rav1e's (as recalled), switched to rav1e's form ONE at a
time in the CPU oracle (environment switches, oracle/av1o_search.c `abl_flags`, AV1O_LIVE_CDF, AV1O_NO_SEGMENTATION), against the only numbers the reference itself holds
for this arithmetic -- the `encode8_opaque` payload ("~215 B", ravif/src/lib.rs:90) and the size windows of its three tests -- plus bytes / MSE on four 960x540 synthetic
images at the headline settings.  It does not pin parity (rav1e cannot run here); it tells whoever runs scripts/compare_with_cavif.sh where to look first.
Usage: python tools/divergence_ledger.py [--json out.json]   (oracle only, about two minutes)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [('shipped (HIP == oracle)', {}),
            ('live CDFs in RDO (rav1e: adaptive rates)', {'AV1O_LIVE_CDF': '1'}),
            ('tx type searched after the mode decision', {'AV1O_ABL_SEQ_TXTYPE': '1'}),
            ('segmentation off', {'AV1O_NO_SEGMENTATION': '1'}),
            ('the three open ones together', {'AV1O_LIVE_CDF': '1', 'AV1O_ABL_SEQ_TXTYPE': '1', 'AV1O_NO_SEGMENTATION': '1'}),
            ('CLOSED in round 6, switched BACK: every sub-block of a split transform picks its own tx type (rounds 1-5)', {'AV1O_ABL_SUB_TXTYPE': '1'}),
            ('CLOSED in round 6, switched BACK: 4x4-Hadamard SATD for every block size (rounds 1-5)', {'AV1O_ABL_SATD4': '1'}),
            ('both closed ones switched back = the round-5 encoder', {'AV1O_ABL_SUB_TXTYPE': '1', 'AV1O_ABL_SATD4': '1'})]
def run(env):
    code = (
        "import sys, json, io, numpy as np\nsys.path.insert(0, %r)\n"
        "from tests.helpers import oracle\nfrom tests.helpers.images import rgba_gradient, rgba_opaque, rgba_noisy\n"
        "from cavif_rs_amd.synth import synth_image\nfrom PIL import Image\n"
        "out = {}\n"
        "_, c, a = oracle.ravif_encode(rgba_opaque(), quality=33, speed=10, depth=0, threads=1); out['encode8_opaque'] = c\n"
        "_, c, a = oracle.ravif_encode(rgba_gradient(), quality=22, alpha_quality=22, speed=1, depth=8, alpha_mode=0, threads=2); out['encode8_with_alpha'] = [c, a]\n"
        "_, c, a = oracle.ravif_encode(rgba_noisy(), quality=66, alpha_quality=88, speed=6, alpha_mode=1, threads=1); out['encode8_cleans_alpha'] = [c, a]\n"
        "out['sets'] = []\n"
        "for idx in (0, 2, 5, 7):\n"
        "    img = synth_image(960, 540, index=idx)\n"
        "    data, cs, _ = oracle.ravif_encode(img, quality=80, speed=4, depth=10)\n"
        "    dec = np.asarray(Image.open(io.BytesIO(data)).convert('RGB'), dtype=np.float64)\n"
        "    out['sets'].append([len(data), float(((dec - img.astype(np.float64)) ** 2).mean())])\n"
        "print(json.dumps(out))\n") % ROOT
    p = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=3600)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    return json.loads(p.stdout.strip().splitlines()[-1])
if __name__ == '__main__':
    res = [(name, env, run(env)) for name, env in VARIANTS]
    base = res[0][2]
    print('| oracle variant | `encode8_opaque` colour bytes (reference: "~215", window 150..500) | `encode8_with_alpha` colour / alpha (windows 50..1000) | `encode8_cleans_alpha` colour / alpha (2000..6000 / 200..1000) | 960x540 x 4, speed 4 q80 10-bit: bytes (vs shipped) | mean RGB MSE |')
    print('|---|---|---|---|---|---|')
    for name, env, r in res:
        tot = sum(s[0] for s in r['sets']); tot0 = sum(s[0] for s in base['sets'])
        mse = sum(s[1] for s in r['sets']) / len(r['sets'])
        print('| %s%s | %d | %d / %d | %d / %d | %d (%+.2f %%) | %.2f |' % (name, (' `' + ' '.join(k + '=1' for k in env) + '`') if env else '', r['encode8_opaque'], r['encode8_with_alpha'][0], r['encode8_with_alpha'][1],
                                                                       r['encode8_cleans_alpha'][0], r['encode8_cleans_alpha'][1], tot, 100.0 * (tot - tot0) / tot0, mse))
    if len(sys.argv) > 2 and sys.argv[1] == '--json':
        json.dump({name: r for name, _, r in res}, open(sys.argv[2], 'w'), indent=1)
