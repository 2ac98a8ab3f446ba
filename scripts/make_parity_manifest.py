#!/usr/bin/env python3
"""scripts/parity_manifest.json: sha256 of what THIS build emits (the oracle == the HIP path, byte for byte) for the inputs and flags
scripts/compare_with_cavif.sh feeds the reference binary -- so that one `cmp` on a machine with cargo settles byte parity.
Keys: "<cfg>/<file>.avif" for threads unspecified (-j0 on a GPU: uncapped tile target) and "<cfg>/<file>.avif@jT" for the -jT
values listed in THREADS (T bounds the tile target, ravif av1encoder.rs:665-668, so the bytes depend on it).
Run in the build container (oracle only, minutes): python scripts/make_parity_manifest.py"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.helpers import oracle
from cavif_rs_amd.synth import synth_image
OUT = os.path.join(ROOT, 'scripts', 'parity_manifest.json')
THREADS = [0, 8, 16, 32]
CFG = {  # name -> (list of (w, h, index, alpha), speed, quality, depth)
    'cfg2': ([(1920, 1080, 0, False)], 4, 80.0, 10),
    'cfg4s': ([(1920, 1080, i, False) for i in range(8)], 4, 80.0, 10),
    'cfg3': ([(4096, 4096, 3, True)], 4, 80.0, 0),
    'cfg5': ([(7680, 4320, 5, False)], 1, 80.0, 10),
}
if __name__ == '__main__':
    only = sys.argv[1:]
    man = json.load(open(OUT)) if os.path.exists(OUT) else {'files': {}}
    man['note'] = 'produced by the CPU oracle of this build (tests prove HIP == oracle); the reference binary has never been run against it'
    for cfg, (items, speed, q, depth) in CFG.items():
        if only and cfg not in only:
            continue
        aq = min((q + 100.0) / 2.0, q + q / 4.0 + 2.0)          # src/main.rs:115-116
        for (w, h, idx, alpha) in items:
            img = synth_image(w, h, index=idx, alpha=True)       # the CLI always hands RGBA to the encoder (load_rgba); opaque -> a == 255
            if not alpha:
                img[..., 3] = 255
            for T in (THREADS if cfg in ('cfg2', 'cfg4s') else [0]):
                t = time.time()
                data, cs, als = oracle.ravif_encode(img, quality=q, alpha_quality=aq, speed=speed, depth=depth, threads=T, alpha_mode=1)
                key = '%s/synth_%04d.avif%s' % (cfg, idx, '@j%d' % T if T else '')
                man['files'][key] = {'sha256': hashlib.sha256(data).hexdigest(), 'bytes': len(data), 'color_bytes': cs, 'alpha_bytes': als}
                print(key, len(data), '%.1fs' % (time.time() - t), flush=True)
                json.dump(man, open(OUT, 'w'), indent=1, sort_keys=True)
