"""K4: each tile's duration against its stages' busy cycles (a -DMI_PROFILE=2 library in MI_AVIF_LIB).  The stages of a tile meet at a barrier per buffer
(lockstep): a tile lasts the SUM over buffers of its slowest stage; stages running ahead of each other on a deeper ring could approach the busiest stage's total."""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
B = 32
e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
b = m.BatchEncoder(e, B, 1920, 1080, 3)
for i in range(B): b.upload(i, synth_image(1920, 1080, index=i))
b.encode(); b.encode()
n = b.num_tiles()
full = b.phase_profile().astype(np.float64)[:n]
stage = full[:, 2, :4]                                     # busy cycles: producer, adapter 0, adapter 1, coder (shader clock)
c = b.tile_clocks().astype(np.float64)[:n]
dur = (c[:, 3] - c[:, 2]) / 100e6                          # seconds (100 MHz wall clock)
busy_s = stage / 2.4e9                                     # ~2.4 GHz shader clock
mx = busy_s.max(axis=1)
print('tiles %d  K4 stage %.2f ms' % (n, b.stage_ms()['entropy']))
print('tile duration ms: mean %.2f  p90 %.2f  max %.2f' % (1e3 * dur.mean(), 1e3 * np.percentile(dur, 90), 1e3 * dur.max()))
print('busiest stage of the tile ms: mean %.2f  p90 %.2f  max %.2f' % (1e3 * mx.mean(), 1e3 * np.percentile(mx, 90), 1e3 * mx.max()))
r = dur / mx
print('duration / busiest stage: mean %.3f  median %.3f  p10 %.3f  p90 %.3f' % (r.mean(), np.median(r), np.percentile(r, 10), np.percentile(r, 90)))
top = np.argsort(-dur)[:10]
print('ten longest tiles: duration ms, stages busy ms (producer, adapter 0, adapter 1, coder), ratio')
for t in top: print('  %.2f  [%s]  %.3f' % (1e3 * dur[t], ' '.join('%.2f' % (1e3 * x) for x in busy_s[t]), r[t]))
