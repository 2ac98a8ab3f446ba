"""How the entropy kernel's adapter waves share the CDF rows (tile_entropy.h k4_row_owner): the oracle counts the adaptive symbols per row offset (AV1O_SYM_HIST,
oracle/av1o_common.c) on pictures of the two block-size classes, and candidate owner functions are compared by the share of the busiest adapter.  CPU only."""
import os, sys, tempfile, itertools, numpy as np
sys.path.insert(0, '.')
from tests.helpers import oracle
from cavif_rs_amd.synth import synth_image

def hist(img, **kw):
    with tempfile.NamedTemporaryFile(delete=False) as t: path = t.name
    os.environ['AV1O_SYM_HIST'] = path
    try: oracle.ravif_encode(img, **kw)
    finally: del os.environ['AV1O_SYM_HIST']
    h = np.fromfile(path, dtype=np.uint32).reshape(-1, 65536); os.unlink(path)
    return h                                   # [tile][row offset]

def share(h, owner, na):
    rows = np.arange(65536, dtype=np.uint32)
    o = owner(rows) & (na - 1)
    per_tile = np.stack([(h * (o == a)).sum(axis=1) for a in range(na)], 1).astype(np.float64)     # [tile][adapter]
    return (per_tile.max(axis=1) / per_tile.sum(axis=1)).mean(), per_tile.sum(axis=0) / per_tile.sum()

cur = lambda r: r // 5 + r // 210
if __name__ == '__main__':
    oracle.build(); oracle.lib()
    sets = {'config-5 like (1280x832, speed 1, q80, 10-bit)': hist(synth_image(1280, 832, index=5), quality=80, speed=1, depth=10),
            'config-4 like (1920x1080, speed 4, q80, 10-bit)': hist(synth_image(1920, 1080, index=0), quality=80, speed=4, depth=10),
            'config-3 like (1024x1024 RGBA, speed 4, q80)': hist(synth_image(1024, 1024, index=3, alpha=True), quality=80, alpha_quality=90, speed=4)}
    for name, h in sets.items():
        tot = h.sum(axis=0); top = np.argsort(tot)[::-1][:12]
        print(name, ': tiles', h.shape[0], 'symbols', int(tot.sum()), 'hottest rows (offset: share %):', ' '.join('%d: %.1f' % (r, 100.0 * tot[r] / tot.sum()) for r in top))
        for na in (2, 4):
            m, per = share(h, cur, na)
            print('   current owner, %d adapters: busiest (mean over tiles) %.1f %%; shares %s' % (na, 100 * m, np.round(100 * per, 1)))
    # candidates for four adapters: (row / a + row / b + row / c) & 3
    best = []
    for a, b, c in itertools.product([3, 5, 10, 15, 20], [0, 42, 70, 105, 210, 420], [0, 7, 21, 630, 1260]):
        f = lambda r, a=a, b=b, c=c: r // a + (r // b if b else 0) + (r // c if c else 0)
        ms = [share(h, f, 4)[0] for h in sets.values()]
        best.append((max(ms), sum(ms), (a, b, c), ms))
    best.sort()
    for x in best[:12]: print('   four adapters, (row/%d + row/%d + row/%d): busiest %s' % (x[2] + (np.round(100 * np.array(x[3]), 1),)))
