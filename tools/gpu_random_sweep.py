"""One-off extended parity sweep on the GPU box: random sizes / depths / speeds / qualities / alpha modes, HIP == oracle bytes.
Usage: python tools/gpu_random_sweep.py [N] [seed] [max_w max_h] [big]     big: speeds 1..2 only and every other picture smooth (gradients + a soft patch), so that 32x32 and
64x64 blocks are searched and chosen (the 64x64 level of dev_blk64.h)"""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from tests.helpers import oracle
from cavif_rs_amd.synth import synth_image
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1234)
MAXW, MAXH = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (420, 300)
BIG = len(sys.argv) > 5 and sys.argv[5] == 'big'
bad = 0
for i in range(N):
    w, h = int(rng.integers(8, MAXW)), int(rng.integers(8, MAXH))
    speed = int(rng.integers(1, 3)) if BIG else int(rng.integers(1, 11)); q = float(rng.integers(5, 100)); aq = float(rng.integers(5, 100))
    depth = int(rng.choice([8, 10])); cm = int(rng.integers(0, 2)); am = int(rng.integers(0, 3)); alpha = bool(rng.integers(0, 2))
    threads = int(rng.choice([0, 0, 1, 3])); passes = 2 if rng.integers(0, 5) == 0 else 1
    img = synth_image(w, h, index=int(rng.integers(0, 1000)), alpha=alpha)
    if BIG and i % 2 == 0:
        yy, xx = np.mgrid[0:h, 0:w]
        base = np.stack([0.5 + 0.4 * np.sin(xx / (60.0 + 25 * k) + k + i) * np.cos(yy / (50.0 + 15 * k)) for k in range(3)], -1)
        base[h // 3:h // 3 + 30, w // 4:w // 4 + 40] += 0.15
        smooth = np.clip((base + rng.normal(0, 0.003, base.shape)) * 255, 0, 255).astype(np.uint8)
        img = np.concatenate([smooth, img[..., 3:]], -1) if alpha else smooth
    elif rng.integers(0, 3) == 0:
        img = rng.integers(0, 256, size=img.shape, dtype=np.uint8)          # pure noise now and then
    e = m.Encoder().with_quality(q).with_alpha_quality(aq).with_speed(speed).with_bit_depth(depth).with_alpha_color_mode(['dirty', 'clean', 'premultiplied'][am])
    if cm: e = e.with_internal_color_model('rgb')
    if threads: e = e.with_num_threads(threads)
    e = e.with_rdo_passes(passes)
    got = e.encode_rgba(img) if alpha else e.encode_rgb(img)
    ref, cs, als = oracle.ravif_encode(img, quality=q, alpha_quality=aq, speed=speed, color_model=cm, depth=depth, alpha_mode=am, threads=threads, rdo_passes=passes)
    ok = got.avif_file == ref
    if not ok:
        bad += 1
        print('MISMATCH', dict(w=w, h=h, speed=speed, q=q, aq=aq, depth=depth, cm=cm, am=am, alpha=alpha, threads=threads, passes=passes), len(got.avif_file), len(ref), flush=True)
print('sweep done: %d cases, %d mismatches' % (N, bad))
