"""Single-image latency of BASELINE configs 2 / 3 / 5 (tools/secondary_latency.py [2] [3] [5]); configs 3 and 5 are also compared with the
oracle's committed sha256 (tests/golden/fullsize_golden.json).  the tile search is a work queue of superblocks for every launch size."""
import hashlib, json, os, sys, time
sys.path.insert(0, '.')
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
gold = json.load(open('tests/golden/fullsize_golden.json'))
CFG = {'2': ('config 2', 1920, 1080, False, 0, 4, 10, 'config2_1920x1080_rgb_s4_q80'), '3': ('config 3', 4096, 4096, True, 3, 4, 10, 'config3_4096x4096_rgba_s4_q80'),
       '5': ('config 5', 7680, 4320, False, 5, 1, 10, 'config5_7680x4320_rgb_s1_q80')}
for key in (sys.argv[1:] or ['2', '3']):
    name, w, h, alpha, index, speed, depth, gname = CFG[key]
    img = synth_image(w, h, index=index, alpha=alpha)
    enc = m.Encoder().with_quality(80.0).with_alpha_quality(90.0).with_speed(speed).with_bit_depth(depth)
    bt = m.BatchEncoder(enc, 1, w, h, channels=4 if alpha else 3)
    bt.pinned_input(0)[...] = img
    best = None
    for _ in range(1 if key == '5' else 3):
        t = time.perf_counter(); bt.upload_async(0, 1); bt.encode_async(); bt.wait(); dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    data = bt.get(0).avif_file
    same = None if gname is None else hashlib.sha256(data).hexdigest() == gold[gname]['avif_sha256']
    print(json.dumps({'workload': name, 'latency_ms': round(best * 1e3, 1), 'MPix_per_s': round(w * h / 1e6 / best, 2), 'tiles': bt.num_tiles(),
                      'bytes': len(data), 'equals_oracle_sha256': same, 'stage_ms': {k: round(v, 1) for k, v in bt.stage_ms().items()}}), flush=True)
    bt.close()
