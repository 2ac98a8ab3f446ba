"""Runs inside a subprocess with MI_AVIF_LIB = the emulator build: the product's host engine + HIP kernels, executed lane by lane
on the CPU, against the oracle.  Prints one JSON line per case.  (TEST INFRASTRUCTURE; see tests/emu/include/hip/hip_runtime.h.)"""
import json
import sys
import time
import numpy as np

sys.path.insert(0, sys.argv[1])
from tests.helpers import oracle                      # noqa: E402
from tests.helpers.images import planes, rgba_noisy, rgba_gradient   # noqa: E402
import cavif_rs_amd as m                              # noqa: E402

from cavif_rs_amd import encoder as _enc
assert 'emu' in _enc.library_path(), _enc.library_path()
which = sys.argv[2] if len(sys.argv) > 2 else 'quick'
oracle.build(); oracle.lib()

if which == 'giveup':                              # with MI_EMU_DROP_PUBLISH=1: no root is ever published -> the waits give up -> the encode fails loudly, in bounded time
    w, h, bd = 136, 136, 8
    pl = planes(h, w, seed=w + h, bd=bd, mono=False)
    t = time.time()
    try:
        m.encode_planes(pl, bd, 121, 10, False)
        outcome = 'returned a stream'
    except m.AvifError as e:
        outcome = 'AvifError: %s' % e
    print(json.dumps({'case': 'giveup', 'ok': outcome.startswith('AvifError'), 'outcome': outcome, 's': round(time.time() - t, 1)}), flush=True)
    sys.exit(0)
PLANE_CASES = {
    'quick': [(64, 64, 8, 4, 121, False, 0), (72, 40, 10, 4, 121, False, 2), (48, 40, 10, 1, 121, False, 0), (64, 48, 8, 10, 121, False, 0), (96, 64, 10, 4, 66, True, 0),
              (136, 136, 8, 4, 121, False, 0)],      # 3 x 3 superblocks in one tile: the work queue's waits on the left neighbour and the row above
    'rect': [(129, 101, 10, 4, 121, False, 0), (136, 72, 8, 4, 10, False, 0), (96, 64, 10, 4, 66, True, 0), (72, 40, 10, 1, 121, False, 0)],
    'full': [(64, 64, 8, 4, 121, False, 0), (64, 64, 8, 10, 121, False, 0), (128, 85, 8, 10, 121, False, 0), (129, 101, 10, 4, 121, False, 0), (200, 120, 10, 1, 121, False, 0),
             (200, 136, 10, 1, 66, True, 0), (256, 200, 10, 4, 66, True, 0), (300, 270, 10, 4, 121, False, 4), (136, 72, 8, 6, 200, False, 0), (136, 72, 8, 4, 10, False, 0),
             (72, 136, 8, 8, 160, False, 2), (8, 8, 8, 4, 121, False, 0), (17, 9, 10, 4, 90, False, 0), (200, 120, 10, 2, 121, False, 0), (200, 120, 8, 3, 170, False, 0)],
}.get(which, [])
ok_all = True
if which == 'quick':
    PLANE_CASES = [c + (1,) for c in PLANE_CASES] + [(136, 136, 8, 4, 121, False, 4, 2), (72, 40, 10, 1, 121, False, 2, 2)]      # + two-pass pricing (rdo_passes = 2)
    NOISE_CASES = [(64, 64, 10, 4, 1)]            # near-lossless noise: a superblock's records exceed a buffer, the entropy stage's producer hands on mid-superblock
else:
    PLANE_CASES = [c + (1,) for c in PLANE_CASES]
for (w, h, bd, speed, q, mono, tiles, passes) in PLANE_CASES:
    pl = planes(h, w, seed=w + h, bd=bd, mono=mono)
    r = oracle.encode_planes(oracle.make_config(w, h, bd, mono, q, speed, tiles=tiles, rdo_passes=passes), pl)
    t = time.time()
    obu, rec = m.encode_planes(pl, bd, q, speed, mono, tiles=tiles, rdo_passes=passes)
    ok = obu == r['obu'] and all(np.array_equal(a, b) for a, b in zip(rec, r['recon']))
    ok_all &= ok
    print(json.dumps({'case': 'planes %dx%d bd%d s%d q%d mono%d tiles%d passes%d' % (w, h, bd, speed, q, int(mono), tiles, passes), 'ok': bool(ok), 'bytes': len(obu), 's': round(time.time() - t, 2)}), flush=True)

if which == 'quick':
    for (w, h, bd, speed, q) in NOISE_CASES:
        rng = np.random.default_rng(q)
        pl = [rng.integers(0, 1 << bd, (h, w)).astype(np.uint16) for _ in range(3)]
        r = oracle.encode_planes(oracle.make_config(w, h, bd, False, q, speed), pl)
        t = time.time()
        obu, rec = m.encode_planes(pl, bd, q, speed, False)
        ok = obu == r['obu'] and all(np.array_equal(a, b) for a, b in zip(rec, r['recon']))
        ok_all &= ok
        print(json.dumps({'case': 'noise %dx%d bd%d s%d q%d' % (w, h, bd, speed, q), 'ok': bool(ok), 'bytes': len(obu), 's': round(time.time() - t, 2)}), flush=True)
if which == 'twodev':                              # MI_EMU_DEVICES=2: two DISTINCT emulated devices (own tables, arenas, budgets, compute-unit counts); the emulator aborts when a
    import ctypes                                  # copy or a kernel argument names the other device's memory
    from cavif_rs_amd.synth import synth_image
    assert m.device_count() == 2, m.device_count()
    L = ctypes.CDLL(_enc.library_path()); L.emu_launch_count.restype = ctypes.c_long
    ok_all = True
    shapes = [(72, 40), (40, 56), (64, 64)]
    imgs = [synth_image(*shapes[k2 % 3], index=k2, alpha=(k2 % 5 == 4)) for k2 in range(10)]
    e = m.Encoder().with_quality(70).with_speed(8)
    ref = [oracle.ravif_encode(im, quality=70, alpha_quality=80, speed=8)[0] for im in imgs]
    t = time.time()
    before = [L.emu_launch_count(0), L.emu_launch_count(1)]
    got = [g.avif_file for g in m.encode_many(e, imgs, devices=[0, 1])]
    used = [L.emu_launch_count(0) - before[0], L.emu_launch_count(1) - before[1]]
    ok = got == ref and used[0] > 0 and used[1] > 0
    ok_all &= ok
    print(json.dumps({'case': 'stream over devices [0, 1], 3 shapes, 10 images', 'ok': bool(ok), 'launches': used, 's': round(time.time() - t, 2)}), flush=True)
    got = [g.avif_file for g in m.encode_many(e, imgs)]           # devices = None: every visible device
    ok = got == ref
    ok_all &= ok
    print(json.dumps({'case': 'stream over every visible device', 'ok': bool(ok), 's': round(time.time() - t, 2)}), flush=True)
    # a batch object and the single-image entry points on device 1 while device 0's tables exist (and the other way round)
    for dev in (1, 0):
        e1 = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10).with_device(dev)
        im2 = [synth_image(136, 72, index=30 + i) for i in range(2)]
        b = m.BatchEncoder(e1, 2, 136, 72, 3)
        for i, im in enumerate(im2): b.upload(i, im)
        t = time.time(); b.encode()
        ok = all(b.get(i).avif_file == oracle.ravif_encode(im, quality=80, speed=4, depth=10)[0] for i, im in enumerate(im2)) and e1.encode_rgb(im2[0]).avif_file == b.get(0).avif_file
        b.close()
        ok_all &= ok
        print(json.dumps({'case': 'batch object + encode_rgb on device %d' % dev, 'ok': bool(ok), 's': round(time.time() - t, 2)}), flush=True)
    sys.exit(0 if ok_all else 1)
if which == 'blk64':                               # the 64x64 level (dev_blk64.h): smooth pictures on which 64x64 blocks win, bottom-up (speed 1) and top-down, 4:4:4 and 4:0:0
    from tests.test_oracle_dav1d import smooth_planes
    ok_all = True
    for (w, h, bd, q, mono, over) in [(136, 200, 8, 100, False, {}),      # (a 64x64 luma transform below coded cells: its all_zero context looks at sixteen neighbour cells, tile_entropy.h k4_txb_ctx)
                                       (192, 128, 10, 90, False, {'encode_bottomup': 0}), (192, 128, 10, 90, True, {'encode_bottomup': 0})]:
        pl = smooth_planes(h, w, bd, w + h)[:1 if mono else 3]
        names = {'encode_bottomup': 'bottomup'}
        r = oracle.encode_planes(oracle.make_config(w, h, bd, mono, q, 1, **{names[k]: v for k, v in over.items()}), pl)
        t = time.time()
        obu, rec = m.encode_planes(pl, bd, q, 1, mono, **over)
        ok = obu == r['obu'] and all(np.array_equal(a, b) for a, b in zip(rec, r['recon']))
        ok_all &= ok
        print(json.dumps({'case': 'blk64 %dx%d bd%d q%d mono%d %s' % (w, h, bd, q, int(mono), over), 'ok': bool(ok), 'n64': int((r['m_bsize'] == 4).sum()) // 256, 'bytes': len(obu), 's': round(time.time() - t, 2)}), flush=True)
    sys.exit(0 if ok_all else 1)
if which == 'rect':
    sys.exit(0 if ok_all else 1)
if which == 'batch':                               # batch API: several images, colour + alpha frames, bottom-up order (work lists spanning frames and block-size classes)
    from cavif_rs_amd.synth import synth_image
    ok_all = True
    # (the 4th and 5th: colour above / alpha below the high-quality threshold and the reverse -- av1encoder.rs:556,576: the frames of one launch differ in rdo_tx_decision,
    # so the launch must not run the kernels that hold the speed-4 switches as constants)
    for (w, h, speed, q, aq, depth, alpha, nimg, passes) in [(200, 136, 4, 80.0, 90.0, 10, False, 3, 1), (136, 100, 4, 60.0, 90.0, 8, True, 2, 2), (72, 72, 2, 80.0, 90.0, 10, False, 1, 1),
                                                             (72, 56, 4, 95.0, 40.0, 10, True, 1, 1), (72, 56, 4, 40.0, 95.0, 10, True, 2, 1)]:
        e = m.Encoder().with_quality(q).with_alpha_quality(aq).with_speed(speed).with_bit_depth(depth).with_rdo_passes(passes)
        imgs = [synth_image(w, h, index=i, alpha=alpha) for i in range(nimg)]
        b = m.BatchEncoder(e, nimg, w, h, 4 if alpha else 3)
        for i, im in enumerate(imgs): b.upload(i, im)
        t = time.time(); b.encode()
        ok = all(b.get(i).avif_file == oracle.ravif_encode(im, quality=q, alpha_quality=aq, speed=speed, depth=depth, rdo_passes=passes)[0] for i, im in enumerate(imgs))
        ok &= all(e.encode_rgba(im).avif_file == b.get(i).avif_file for i, im in enumerate(imgs)) if alpha else ok
        ok_all &= ok
        print(json.dumps({'case': 'batch %dx%d s%d q%d aq%d n%d alpha%d' % (w, h, speed, q, aq, nimg, int(alpha)), 'ok': bool(ok), 's': round(time.time() - t, 2)}), flush=True)
        b.close()
    # the streaming fan-out with a dozen shapes (the emulated device reports little free memory: the worker's eviction path runs) and two workers on one device
    shapes = [(40 + 8 * k2, 24 + 8 * (k2 % 3)) for k2 in range(7)]
    imgs = [synth_image(w, h, index=k2, alpha=(k2 % 4 == 3)) for k2, (w, h) in enumerate(shapes)] + [synth_image(*shapes[1], index=20)]
    e = m.Encoder().with_quality(70).with_speed(8)
    t = time.time()
    ref = [oracle.ravif_encode(im, quality=70, alpha_quality=80, speed=8)[0] for im in imgs]
    ok = [g.avif_file for g in m.encode_many(e, imgs, devices=[0])] == ref and [g.avif_file for g in m.encode_many(e, imgs, devices=[0, 0])] == ref
    ok_all &= ok
    print(json.dumps({'case': 'stream, 7 shapes, 1 and 2 workers', 'ok': bool(ok), 's': round(time.time() - t, 2)}), flush=True)
    sys.exit(0 if ok_all else 1)
# ravif level: RGBA with a used alpha channel, UnassociatedClean (dirty-alpha kernels + front end + colour and alpha frames + container)
img = rgba_noisy()[:40, :56].copy()
e = m.Encoder().with_quality(66).with_alpha_quality(88).with_speed(6).with_num_threads(1).with_alpha_color_mode('clean')
t = time.time()
got = e.encode_rgba(img)
ref, color, alpha = oracle.ravif_encode(img, quality=66, alpha_quality=88, speed=6, depth=0, alpha_mode=1, threads=1)
ok = got.avif_file == ref and (got.color_byte_size, got.alpha_byte_size) == (color, alpha)
ok_all &= ok
print(json.dumps({'case': 'ravif rgba clean 56x40 s6', 'ok': bool(ok), 'bytes': len(got.avif_file), 's': round(time.time() - t, 2)}), flush=True)
sys.exit(0 if ok_all else 1)
