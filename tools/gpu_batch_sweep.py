"""Extended parity sweep of the MULTI-FRAME entry points on the GPU box: random batches (BatchEncoder: several pictures of one shape, colour + alpha frames in one
launch, qualities on both sides of the high / low quality thresholds of av1encoder.rs:556-557) and random streams (encode_many: mixed shapes, one and two workers),
every file against oracle.ravif_encode.  Usage: python tools/gpu_batch_sweep.py [N] [seed]"""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from tests.helpers import oracle
from cavif_rs_amd.synth import synth_image
N = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 7)
QS = [10, 30, 54, 55, 56, 70, 79, 80, 81, 90, 99]           # around low_quality (55) and high_quality (80)
bad = 0
for it in range(N):
    speed = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 6, 8, 10])); q = float(rng.choice(QS)); aq = float(rng.choice(QS))
    depth = int(rng.choice([8, 10])); alpha = bool(rng.integers(0, 2)); passes = 2 if rng.integers(0, 6) == 0 else 1
    e = m.Encoder().with_quality(q).with_alpha_quality(aq).with_speed(speed).with_bit_depth(depth).with_rdo_passes(passes)
    kw = dict(quality=q, alpha_quality=aq, speed=speed, depth=depth, rdo_passes=passes)
    if it % 3 != 2:
        w, h = int(rng.integers(16, 300)), int(rng.integers(16, 220)); n = int(rng.integers(1, 6))
        imgs = [synth_image(w, h, index=int(rng.integers(0, 1000)), alpha=alpha) for _ in range(n)]
        if rng.integers(0, 4) == 0: imgs[0] = rng.integers(0, 256, size=imgs[0].shape, dtype=np.uint8)
        b = m.BatchEncoder(e, n, w, h, 4 if alpha else 3)
        for i, im in enumerate(imgs): b.upload(i, im)
        b.encode()
        got = [b.get(i).avif_file for i in range(n)]
        b.close()
        what = dict(kind='batch', w=w, h=h, n=n)
    else:
        n = int(rng.integers(2, 9))
        imgs = [synth_image(int(rng.integers(16, 260)), int(rng.integers(16, 200)), index=int(rng.integers(0, 1000)), alpha=bool(rng.integers(0, 2)) if alpha else False) for _ in range(n)]
        devs = [0] if rng.integers(0, 2) else [0, 0]
        got = [g.avif_file for g in m.encode_many(e, imgs, devices=devs)]
        what = dict(kind='stream', n=n, workers=len(devs))
    ref = [oracle.ravif_encode(im, **kw)[0] for im in imgs]
    if got != ref:
        bad += 1
        print('MISMATCH', what, dict(speed=speed, q=q, aq=aq, depth=depth, alpha=alpha, passes=passes), [i for i in range(len(ref)) if got[i] != ref[i]], flush=True)
print('batch sweep done: %d cases, %d mismatches' % (N, bad))
