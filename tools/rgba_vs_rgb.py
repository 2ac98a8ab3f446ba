"""Steady-state step time of the batch rotation (3 slots x 32 x 1080p) for RGB8 input vs opaque RGBA8 input (the CLI's case)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
B, W, H, SLOTS, STEPS = 32, 1920, 1080, 3, 9
imgs = [synth_image(W, H, index=i) for i in range(B)]
for ch in (3, 4):
    for mode in (('clean', 'dirty') if ch == 4 else ('clean',)):
        enc = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10).with_alpha_color_mode(mode)
        bts = [m.BatchEncoder(enc, B, W, H, channels=ch) for _ in range(SLOTS)]
        for bt in bts:
            for i in range(B):
                bt.pinned_input(i)[...] = imgs[i] if ch == 3 else np.dstack([imgs[i], np.full((H, W), 255, np.uint8)])
            bt.upload_async(0, B)
        def run(n):
            infl = []; st = []
            for k in range(n):
                bt = bts[k % SLOTS]
                if len(infl) == SLOTS:
                    o = infl.pop(0); o.wait(); st.append(o.stage_ms())
                bt.upload_async(0, B); bt.encode_async(); infl.append(bt)
            for o in infl:
                o.wait(); st.append(o.stage_ms())
            return st
        run(3)
        t = time.perf_counter(); st = run(STEPS); dt = (time.perf_counter() - t) / STEPS
        print('channels', ch, mode, 'ms/step %.1f' % (dt * 1e3), {k: round(v, 1) for k, v in st[-1].items()})
        for bt in bts:
            bt.close()
