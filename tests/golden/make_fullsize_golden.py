#!/usr/bin/env python3
"""sha256 of the ORACLE's .avif for BASELINE configs 2 (one 1920x1080 RGB image, speed 4, q80), 3 (4096x4096 RGBA, speed 4, q80) and 5 (7680x4320 RGB, speed 1, q80) on
the synthetic images of cavif_rs_amd/synth.py.  The oracle needs minutes for these, so the vectors are produced once here and
the -m gpu test (tests/test_gpu_parity_cells.py) compares the HIP path with them.  Re-run after any algorithmic change:
    python tests/golden/make_fullsize_golden.py [config3|config5]"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.helpers import oracle
from cavif_rs_amd.synth import synth_image
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'fullsize_golden.json')
CASES = {
    'config2_1920x1080_rgb_s4_q80': dict(w=1920, h=1080, index=0, alpha=False, quality=80.0, alpha_quality=90.0, speed=4, depth=10),
    'config3_4096x4096_rgba_s4_q80': dict(w=4096, h=4096, index=3, alpha=True, quality=80.0, alpha_quality=90.0, speed=4, depth=10),
    'config5_7680x4320_rgb_s1_q80': dict(w=7680, h=4320, index=5, alpha=False, quality=80.0, alpha_quality=90.0, speed=1, depth=10),
}
if __name__ == '__main__':
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    for name, c in CASES.items():
        if len(sys.argv) > 1 and not any(a in name for a in sys.argv[1:]):
            continue
        img = synth_image(c['w'], c['h'], index=c['index'], alpha=c['alpha'])
        t = time.time()
        data, cs, als = oracle.ravif_encode(img, quality=c['quality'], alpha_quality=c['alpha_quality'], speed=c['speed'], depth=c['depth'])
        res[name] = dict(c, avif_len=len(data), avif_sha256=hashlib.sha256(data).hexdigest(), color_byte_size=cs, alpha_byte_size=als, oracle_seconds=round(time.time() - t, 1))
        print(name, res[name])
        json.dump(res, open(OUT, 'w'), indent=1)
