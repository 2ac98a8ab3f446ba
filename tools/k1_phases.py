"""K1 phase profile (needs a library built with -DMI_PROFILE=1, pointed to by MI_AVIF_LIB)."""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
# default: the 32 x 1080p batch at speed 4 (the 16x16 class).  `tools/k1_phases.py 5`: BASELINE config 5 -- one 8K image at speed 1 -- on a library built with
# -DMI_PROFILE=1 -DMI_PROF_MAXN=32 (the 64x64 class; the 64x64 level's own phases land in the same slots, 2:1 blocks in WALKER_other)
if len(sys.argv) > 1 and sys.argv[1] == '5':
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
p = b.phase_profile().astype(np.float64)          # [persistent workgroup][wave][phase] cycles
p = p[p[:, 0, 15] > 0]
names = ['queue_claim_wait', 'stage_src_edges', 'WAIT_barrier', 'satd13', 'sort', 'delta_satd', 'luma_rd', 'luma_commit', 'cfl_alpha', 'chroma_eval', 'chroma_commit', 'final', 'luma_final_pred', 'tx_size_trial', 'walker_area_copies', 'WALKER_other']
tr = p[:, :, 22:32].copy()
sub = p[:, :, 16:22]                               # eval_tx sub-phases (nested inside luma_rd / chroma_eval)
p = p[:, :, :16].copy()
p[:, :, 13] += tr[:, :, :7].sum(axis=2)         # the trial's sub-phases (22..28) are part of tx_size_trial; 29..31 are separate clocks per (block, depth)
p[:, :, 15] -= p[:, :, :15].sum(axis=2)           # slot 15 arrives as the workgroup's life: what is left is the walker outside try_block and the area copies
tot = p.sum(axis=2)                                 # per tile per wave
print('stage_ms', b.stage_ms())
print('mean cycles per wave per tile: %.3g' % tot.mean())
for i, n in enumerate(names):
    print('%-18s %6.2f%%   (waves ' % (n, 100 * p[:, :, i].sum() / tot.sum()) + ' '.join('%.2f%%' % (100 * p[:, w_, i].sum() / tot[:, w_].sum()) for w_ in range(4)) + ')')
if len(sys.argv) > 2 and sys.argv[2] == 'sizes':      # a -DMI_PROFILE=3 library: slots 16..21 = time per block size
    print('by block size, share of the kernel: ' + '  '.join('%s %.2f%%' % (n, 100 * sub[:, :, i].sum() / tot.sum()) for i, n in enumerate(['4x4', '8x8', '16x16', '32x32+', '8x4', '4x8'])))
    sys.exit(0)
print('eval_tx split (share of the time spent inside evaluations):')
for i, n in enumerate(['residual', 'fwd_txfm', 'quantize', 'coef_rate', 'dequant_inverse', 'sse']):
    print('  %-16s %6.2f%%  (%.2f%% of the kernel)' % (n, 100 * sub[:, :, i].sum() / sub.sum(), 100 * sub[:, :, i].sum() / tot.sum()))
print('tx-size trial, share of the kernel: ' + '  '.join('%s %.2f%%' % (n, 100 * tr[:, :, i].sum() / tot.sum()) for i, n in enumerate(['stage0', 'edges+predict(wave 0)', 'wait_pred', 'ctx+eval', 'wait_eval', 'pick+copy', 'commit'])))
print('  by wave, edges+predict: ' + ' '.join('%.2f%%' % (100 * tr[:, w, 1].sum() / tot[:, w].sum()) for w in range(4)) + '   ctx+eval: ' + ' '.join('%.2f%%' % (100 * tr[:, w, 3].sum() / tot[:, w].sum()) for w in range(4)))
print('tx-size trial by (block, depth): 16x16 d1 %.2f%%  16x16 d2 %.2f%%  8x8 d1 %.2f%%' % tuple(100 * tr[:, :, i].sum() / tot.sum() for i in (7, 8, 9)))
