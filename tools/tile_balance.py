"""Per-tile K1 / K4 durations of one batch encode (load balance across the one-wave-per-tile launch)."""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
b = m.BatchEncoder(e, B, 1920, 1080, 3)
for i in range(B): b.upload(i, synth_image(1920, 1080, index=i))
b.encode(); b.encode()
c = b.tile_clocks().astype(np.float64)
for name, s, t in (('K1', 0, 1), ('K4', 2, 3)):
    d = (c[:, t] - c[:, s]) / 100e6 * 1e3      # ms at 100 MHz
    span = (c[:, t].max() - c[:, s].min()) / 100e6 * 1e3
    print(name, 'tiles', len(d), 'mean %.1f ms  median %.1f  p90 %.1f  max %.1f  min %.1f  kernel span %.1f ms  mean/max %.2f' % (d.mean(), np.median(d), np.percentile(d, 90), d.max(), d.min(), span, d.mean() / d.max()))
print(b.stage_ms())
# K4 time against the size of what a tile codes
sizes = np.array([len(b.get(i).avif_file) for i in range(B)], dtype=np.float64)
d4 = (c[:, 3] - c[:, 2]) / 100e6 * 1e3
per_img = d4.reshape(B, -1)
print('K4 per image: max-tile ms vs file bytes:', [(round(float(per_img[i].max()), 1), int(sizes[i])) for i in range(min(B, 6))])
print('K4 total tile-ms %.1f for %.0f payload bytes -> %.2f us per byte (~%.0f ns per coded bit)' % (d4.sum(), sizes.sum(), 1e3 * d4.sum() / sizes.sum(), 1e6 * d4.sum() / sizes.sum() / 8))
