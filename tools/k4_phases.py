"""K4 (entropy coder) phase profile: needs a library built with -DMI_PROFILE=2, pointed to by MI_AVIF_LIB."""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
# `tools/k4_phases.py 5`: BASELINE config 5 (one 8K image at speed 1: 8 tiles, the four-adapter launch) on a library built with -DMI_PROFILE=2 -DMI_PROF_MAXN=32
CFG5 = len(sys.argv) > 1 and sys.argv[1] == '5'
if CFG5:
    e = m.Encoder().with_quality(80).with_speed(1).with_bit_depth(10)
    b = m.BatchEncoder(e, 1, 7680, 4320, 3)
    b.upload(0, synth_image(7680, 4320, index=5))
    b.encode()
else:
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
    b = m.BatchEncoder(e, B, 1920, 1080, 3)
    for i in range(B): b.upload(i, synth_image(1920, 1080, index=i))
    b.encode(); b.encode()
full = b.phase_profile().astype(np.float64)
full = full[:b.num_tiles()]
p = full[:, 3, :16]                                        # the producer's phases sit in wave 3's slots: [tiles][16]
stage = full[:, 2, :8]                                     # busy cycles of the stage waves (producer, adapters, coder) in wave 2's
print('stage_ms', b.stage_ms())
names = ['partition', 'block_header', 'tx_staging', 'ctx_phase_P', 'records_R', '(unused)', 'lr', 'walk_other']
tot = p[:, :8].sum(axis=1)
print('mean cycles per tile %.4g  (max %.4g)' % (tot.mean(), tot.max()))
for i, n in enumerate(names): print('%-14s %6.2f%%' % (n, 100 * p[:, i].sum() / tot.sum()))
blocks, txb, empty, coefs = p[:, 8].sum(), p[:, 9].sum(), p[:, 10].sum(), p[:, 11].sum()
print('per tile: blocks %.0f, transform blocks %.0f (empty %.0f), coefficients up to eob %.0f' % (blocks / len(p), txb / len(p), empty / len(p), coefs / len(p)))
print('cycles per block header %.0f, per transform block staged %.0f, P per coded block %.0f, S per coefficient %.1f, signs per coefficient %.1f' % (
    p[:, 1].sum() / blocks, p[:, 2].sum() / txb, p[:, 3].sum() / max(1, txb - empty), p[:, 4].sum() / coefs, p[:, 5].sum() / coefs))
names = ['producer', 'adapter 0', 'adapter 1', 'coder']            # MI_K4_ADAPTERS = 2: four waves per tile (slots 4..7 of the record belong to other probes)
if CFG5: names = ['producer', 'adapter 0', 'adapter 1', 'adapter 2', 'adapter 3', 'coder']
print('stage busy cycles per tile (mean / max): ' + '  '.join('%s %.3g / %.3g' % (names[i], stage[:, i].mean(), stage[:, i].max()) for i in range(len(names))))
