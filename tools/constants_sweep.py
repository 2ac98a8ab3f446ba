#!/usr/bin/env python3
"""The recalled-constants sweep: the three encoder-side constants of the search that are RECALLED from rav1e rather than chosen
-- the quantiser's dead-zone offsets, the key-frame offset of the quantiser index rule, lambda's scale -- varied one at a time in the CPU oracle (environment knobs
AV1O_SWEEP_DZ / AV1O_SWEEP_KFQ / AV1O_SWEEP_LAMBDA, oracle only; the product has no such knobs) against the only number the reference holds for this arithmetic, the
`encode8_opaque` payload ("expected ~= 215" B, ravif/src/lib.rs:90), and the size windows of its three tests.  No default changes on one datum: this is the map for whoever
runs scripts/compare_with_cavif.sh on a machine with cargo.   Usage: python tools/constants_sweep.py   (oracle only, about a minute)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SWEEPS = [('dead-zone offsets x', 'AV1O_SWEEP_DZ', [0, 50, 75, 90, 100, 110, 125, 150], 100, '%d %%'),
          ('key-frame log-q offset x', 'AV1O_SWEEP_KFQ', [0, 500, 750, 1000, 1250, 1500, 2000], 1000, '%d / 1000'),
          ('lambda x', 'AV1O_SWEEP_LAMBDA', [50, 75, 90, 100, 110, 125, 150, 200], 100, '%d %%')]
CODE = ("import sys, json\nsys.path.insert(0, %r)\n"
        "from tests.helpers import oracle\nfrom tests.helpers.images import rgba_gradient, rgba_opaque, rgba_noisy\n"
        "out = {}\n"
        "_, c, a = oracle.ravif_encode(rgba_opaque(), quality=33, speed=10, depth=0, threads=1); out['encode8_opaque'] = c\n"
        "_, c, a = oracle.ravif_encode(rgba_gradient(), quality=22, alpha_quality=22, speed=1, depth=8, alpha_mode=0, threads=2); out['encode8_with_alpha'] = [c, a]\n"
        "_, c, a = oracle.ravif_encode(rgba_noisy(), quality=66, alpha_quality=88, speed=6, alpha_mode=1, threads=1); out['encode8_cleans_alpha'] = [c, a]\n"
        "print(json.dumps(out))\n") % ROOT
def run(env):
    p = subprocess.run([sys.executable, '-c', CODE], capture_output=True, text=True, env=dict(os.environ, **env), timeout=1800)
    if p.returncode != 0:
        raise SystemExit(p.stderr[-3000:])
    return json.loads(p.stdout.strip().splitlines()[-1])
if __name__ == '__main__':
    print('| recalled constant | value | `encode8_opaque` colour bytes (reference "~215", window 150..500) | `encode8_with_alpha` colour / alpha (50..1000 each) | `encode8_cleans_alpha` colour / alpha (2000..6000 / 200..1000) | windows hold |')
    print('|---|---|---|---|---|---|')
    for name, var, values, dflt, fmt in SWEEPS:
        for v in values:
            r = run({var: str(v)})
            ok = 150 < r['encode8_opaque'] < 500 and all(50 < x < 1000 for x in r['encode8_with_alpha']) and 2000 < r['encode8_cleans_alpha'][0] < 6000 and 200 < r['encode8_cleans_alpha'][1] < 1000
            print('| %s | %s%s | %d | %d / %d | %d / %d | %s |' % (name, fmt % v, ' (shipped)' if v == dflt else '', r['encode8_opaque'], r['encode8_with_alpha'][0], r['encode8_with_alpha'][1],
                                                           r['encode8_cleans_alpha'][0], r['encode8_cleans_alpha'][1], 'yes' if ok else 'NO'), flush=True)
