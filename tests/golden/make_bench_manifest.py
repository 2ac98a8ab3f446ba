#!/usr/bin/env python3
"""sha256 of the ORACLE's .avif for the synthetic images bench.py encodes (cavif_rs_amd/synth.py indices 0..255 at 1920x1080, speed 4, quality 80,
10-bit: BASELINE config 4's 256 files): bench.py compares every file of every batch slot with it (`output_identity`) without running the oracle.
Re-run after any algorithmic change (minutes on 8 cores):   python tests/golden/make_bench_manifest.py [count]"""
import hashlib, json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'bench_manifest.json')
CFG = dict(width=1920, height=1080, speed=4, quality=80.0, bit_depth=10)


def one(idx):
    from tests.helpers import oracle
    from cavif_rs_amd.synth import synth_image
    img = synth_image(CFG['width'], CFG['height'], index=idx)
    data, cs, _ = oracle.ravif_encode(img, quality=CFG['quality'], speed=CFG['speed'], depth=CFG['bit_depth'])
    return idx, hashlib.sha256(data).hexdigest(), len(data)


if __name__ == '__main__':
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    from tests.helpers import oracle
    oracle.build(); oracle.lib()
    t = time.time()
    with mp.get_context('spawn').Pool(max(1, min(os.cpu_count() or 1, 16))) as pool:
        res = sorted(pool.map(one, range(n)))
    json.dump({'config': CFG, 'generator': 'oracle/ (CPU restatement of this encoder, test infrastructure)', 'sha256': [r[1] for r in res], 'bytes': [r[2] for r in res]},
              open(OUT, 'w'), indent=0)
    print('%d images in %.0f s, mean %.0f bytes' % (n, time.time() - t, sum(r[2] for r in res) / n))
