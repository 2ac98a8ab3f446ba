"""K1 phase profile (needs a library built with -DMI_PROFILE=1, pointed to by MI_AVIF_LIB)."""
import sys, numpy as np
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
b = m.BatchEncoder(e, B, 1920, 1080, 3)
for i in range(B): b.upload(i, synth_image(1920, 1080, index=i))
b.encode(); b.encode()
p = b.phase_profile().astype(np.float64)          # [tiles][wave][phase] cycles
names = ['txb_ctx', 'stage_src_edges', 'WAIT_barrier', 'satd13', 'sort', 'delta_satd', 'luma_rd', 'luma_commit', 'cfl_alpha', 'chroma_eval', 'chroma_commit', 'final', 'luma_final_pred', 'tx_size_trial']
bw = p[:, :, 22:32]
sub = p[:, :, 16:22]                               # eval_tx sub-phases (nested inside luma_rd / chroma_eval)
p = p[:, :, :16].copy()
p[:, :, 2] += bw.sum(axis=2)
tot = p.sum(axis=2)                                 # per tile per wave
print('stage_ms', b.stage_ms())
print('mean cycles per wave per tile: %.3g' % tot.mean())
for i, n in enumerate(names):
    print('%-18s %6.2f%%   (wave0 %.2f%%, wave3 %.2f%%)' % (n, 100 * p[:, :, i].sum() / tot.sum(), 100 * p[:, 0, i].sum() / tot[:, 0].sum(), 100 * p[:, 3, i].sum() / tot[:, 3].sum()))
print('eval_tx split (share of the time spent inside evaluations):')
for i, n in enumerate(['residual', 'fwd_txfm', 'quantize', 'coef_rate', 'dequant_inverse', 'sse']):
    print('  %-16s %6.2f%%  (%.2f%% of the kernel)' % (n, 100 * sub[:, :, i].sum() / sub.sum(), 100 * sub[:, :, i].sum() / tot.sum()))
if bw.sum() > 0:
    print('barrier waits by site (share of the kernel; sites in source order, the tenth collects the rest):')
    print('  ' + '  '.join('%.2f%%' % (100 * bw[:, :, i].sum() / tot.sum()) for i in range(10)))
