"""-m gpu: the parity cells VERDICT r01 listed as untested -- raw-planes entry points incl. Limited range, 10-bit GBR,
AlphaColorMode::Premultiplied, BASELINE config 1 on the reference's own fixture through the command line (file + stdin,
tests/stdio.rs:4-43), every tool switch HIP == oracle, and full-size configs 3 / 5 against committed sha256 vectors."""
import hashlib
import json
import os
import subprocess
import numpy as np
import pytest
from tests.helpers.images import planes, rgba_gradient

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, 'cavif_rs_amd', 'cavif_mi')


@pytest.mark.parametrize('over', [dict(lrf=0), dict(rdo_tx_decision=0), dict(fast_deblock=1), dict(tune_psnr=1), dict(cdef=0), dict(sgr_full=1),
                                  dict(rdo_tx_decision=0, inter_tx_split=1), dict(lrf=0, cdef=0, fast_deblock=1, rdo_tx_decision=0, tune_psnr=1), dict(rdo_passes=2), dict(rdo_passes=2, tiles_override=6)])
def test_tool_switches_hip_equals_oracle(oracle, over):
    import cavif_rs_amd as m
    w, h, bd = 264, 200, 10
    pl = planes(h, w, seed=w + h, bd=bd)
    names = {'rdo_tx_decision': 'rdo_tx'}
    cfg = oracle.make_config(w, h, bd, False, 121, 4, **{names.get(k, k): v for k, v in over.items()})
    r = oracle.encode_planes(cfg, pl)
    obu, rec = m.encode_planes(pl, bd, 121, 4, False, **over)
    assert obu == r['obu']
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('w,h,bd,q,mono,smooth,over', [
    (256, 192, 10, 121, False, True, {}),                        # smooth picture: 64x64 colour blocks win (bottom-up walker, what speed 1 uses)
    (200, 136, 8, 121, False, True, {}),
    (192, 128, 10, 90, False, True, dict(encode_bottomup=0)),    # the top-down walker with the 64x64 level
    (136, 200, 8, 100, False, True, {}),
    (200, 120, 10, 121, False, False, {}),                       # textured: the 64x64 level is evaluated and mostly loses
    (328, 264, 10, 100, False, True, dict(tiles_override=4)),    # tiles; 328 = 5 superblocks + 8: a 64x64 block must split at the frame edge
    (200, 136, 10, 66, True, True, {}),                          # 4:0:0 (alpha-like) planes take the same path without the chroma stage
    (192, 192, 10, 110, False, True, dict(rdo_tx_decision=0)),
    (192, 192, 8, 110, False, True, dict(complex_pred_modes=0, fine_directional_intra=0, tune_psnr=1)),
])
def test_64x64_blocks_hip_equals_oracle(oracle, w, h, bd, q, mono, smooth, over):
    """R-4: partition_range (4, 64) (speed <= 1 below the high-quality threshold, ravif/src/av1encoder.rs:556-566).  The 64x64 level of the tile search
    (dev_blk64.h: four 32x32 chroma transform blocks per plane, the tx-size trial through HBM scratch) against oracle/av1o_search.c, bytes + reconstruction;
    on the smooth pictures 64x64 blocks must actually be chosen."""
    import cavif_rs_amd as m
    from tests.test_oracle_dav1d import smooth_planes
    pl = smooth_planes(h, w, bd, w + h) if smooth else planes(h, w, seed=w + h, bd=bd, mono=mono)
    if mono: pl = pl[:1]
    names = {'rdo_tx_decision': 'rdo_tx', 'encode_bottomup': 'bottomup', 'complex_pred_modes': 'complex_modes', 'fine_directional_intra': 'fine_directional', 'tiles_override': 'tiles'}
    cfg = oracle.make_config(w, h, bd, mono, q, 1, **{names.get(k, k): v for k, v in over.items()})
    assert cfg.part_max == 64
    r = oracle.encode_planes(cfg, pl)
    if smooth: assert int((r['m_bsize'] == 4).sum()) >= 256, 'no 64x64 block was chosen'
    obu, rec = m.encode_planes(pl, bd, q, 1, mono, **over)
    assert obu == r['obu']
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('kind', ['textured', 'flat', 'two_regions', 'noise_tiles'])
def test_segmentation_hip_equals_oracle(oracle, kind):
    """segment_kernel's fit (histogram, k-means, indices, thresholds), the per-block quantiser switch in K1 and the segment ids K4 codes: the frame
    header carries the fit, so equal bytes mean equal fits; `flat` is the one-scale frame that leaves segmentation off."""
    import cavif_rs_amd as m
    w, h, bd, tiles = 264, 200, 10, 0
    if kind == 'textured': pl = planes(h, w, seed=7, bd=bd)
    elif kind == 'flat': pl = [np.full((h, w), v, np.uint16) for v in (300, 512, 700)]
    elif kind == 'two_regions':
        rng = np.random.default_rng(3)
        pl = [np.where(np.arange(w)[None, :] < w // 2, 400, rng.integers(0, 1024, (h, w))).astype(np.uint16) for _ in range(3)]
    else:
        rng = np.random.default_rng(4); tiles = 4
        pl = [(rng.integers(0, 1024, (h, w)) >> rng.integers(0, 6, (h // 8 + 1, w // 8 + 1)).repeat(8, 0).repeat(8, 1)[:h, :w]).astype(np.uint16) for _ in range(3)]
    r = oracle.encode_planes(oracle.make_config(w, h, bd, False, 100, 4, tiles=tiles), pl)
    assert (r['seg_n'] == 0) == (kind == 'flat')
    obu, rec = m.encode_planes(pl, bd, 100, 4, False, tiles=tiles)
    assert obu == r['obu']
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('bd,q', [(10, 1), (8, 3), (10, 20)])
def test_entropy_record_buffers_hand_on_mid_superblock(oracle, bd, q):
    """Pure noise at near-lossless quantisers: every coefficient carries a long Golomb tail (a record per bit), a superblock needs several times
    the records one buffer holds, and K4's producer hands its buffer on from inside the superblock walk (tile_entropy.h k4_room / k4_handoff)."""
    import cavif_rs_amd as m
    w, h = 136, 72
    rng = np.random.default_rng(bd * 100 + q)
    pl = [rng.integers(0, 1 << bd, (h, w)).astype(np.uint16) for _ in range(3)]
    r = oracle.encode_planes(oracle.make_config(w, h, bd, False, q, 4), pl)
    assert len(r['obu']) > w * h * 3 * bd // 8 // 2          # (it really is near-lossless noise: more than half the raw size)
    obu, rec = m.encode_planes(pl, bd, q, 4, False)
    assert obu == r['obu']
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('depth,full_range,with_alpha', [(8, 0, False), (8, 1, True), (10, 0, True), (10, 1, False)])
def test_raw_planes_entry_points(oracle, avifdec, depth, full_range, with_alpha):
    """encode_raw_planes_8_bit / _10_bit (av1encoder.rs:366,390) incl. PixelRange::Limited == oracle frames + container."""
    import cavif_rs_amd as m
    w, h = 150, 90
    pl = planes(h, w, seed=11 + depth, bd=depth)
    if not full_range:                                         # studio swing samples
        lo, hi = (16, 235) if depth == 8 else (64, 940)
        pl = [np.clip(p, lo, hi).astype(np.uint16) for p in pl]
    alpha = planes(h, w, seed=5, bd=depth, mono=True)[0] if with_alpha else None
    dt = np.uint8 if depth == 8 else np.uint16
    yuv = np.stack(pl, -1).astype(dt)
    e = m.Encoder().with_quality(70).with_alpha_quality(85).with_speed(5)
    fn = e.encode_raw_planes_8_bit if depth == 8 else e.encode_raw_planes_10_bit
    got = fn(w, h, yuv, None if alpha is None else alpha.astype(dt), color_pixel_range=full_range, matrix_coefficients=6)
    L = oracle.lib()
    cq, aq = L.av1o_quality_to_quantizer(70.0), L.av1o_quality_to_quantizer(85.0)
    rc = oracle.encode_planes(oracle.make_config(w, h, depth, False, cq, 5, matrix=6, full_range=full_range), pl)
    ra = oracle.encode_planes(oracle.make_config(w, h, depth, True, aq, 5), [alpha]) if with_alpha else None
    want = oracle.container(rc['obu'], ra['obu'] if ra else None, w, h, depth, cp=1, tc=13, mc=6, full_range=1)
    assert got.color_byte_size == len(rc['obu']) and got.alpha_byte_size == (len(ra['obu']) if ra else 0)
    assert got.avif_file == want
    d = avifdec.decode(got.avif_file)
    # (the container's colr box always says full range -- ravif never forwards the range to avif-serialize, :457-473 --
    #  so the range is carried by the AV1 sequence header alone; the header bit is covered by the byte comparison above)
    for a, b in zip(d['planes'], rc['recon']):
        assert np.array_equal(a, b)


def test_ten_bit_gbr_equals_oracle(oracle):
    """--color=rgb at depth 10: to_ten + G,B,R plane order (av1encoder.rs:485-498) against the oracle, RGB and RGBA."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    e = m.Encoder().with_quality(75).with_speed(6).with_bit_depth(10).with_internal_color_model('rgb')
    rgb = synth_image(176, 112, index=9)
    assert e.encode_rgb(rgb).avif_file == oracle.ravif_encode(rgb, quality=75, speed=6, color_model=1, depth=10)[0]
    rgba = rgba_gradient(120, 88)
    assert e.encode_rgba(rgba).avif_file == oracle.ravif_encode(rgba, quality=75, speed=6, color_model=1, depth=10)[0]


def test_premultiplied_equals_oracle(oracle):
    import cavif_rs_amd as m
    rgba = rgba_gradient(136, 72)
    rgba[:10, :, 3] = 255; rgba[10:20, :, 3] = 0
    e = m.Encoder().with_quality(60).with_alpha_quality(70).with_speed(7).with_alpha_color_mode('premultiplied')
    got = e.encode_rgba(rgba)
    ref, cs, als = oracle.ravif_encode(rgba, quality=60, alpha_quality=70, speed=7, alpha_mode=2)
    assert got.avif_file == ref and b'prem' in ref and als > 0


def test_config1_reference_fixture_through_the_cli(oracle, tmp_path):
    """BASELINE config 1: tests/testimage.png (committed as tests/golden/testimage_128x85.json), `--speed=10`, path -> `-o -`
    and stdin -> stdout exactly as tests/stdio.rs:4-43 drives the reference binary; bytes == oracle on the same pixels."""
    from tests.helpers.fixtures import testimage_png_bytes
    png, rgba = testimage_png_bytes()
    (tmp_path / 'testimage.png').write_bytes(png)
    aq = min((80.0 + 100.0) / 2.0, 80.0 + 80.0 / 4.0 + 2.0)                     # src/main.rs:115-116
    ref, cs, als = oracle.ravif_encode(rgba, quality=80.0, alpha_quality=aq, speed=10, alpha_mode=1)
    r = subprocess.run([CLI, str(tmp_path / 'testimage.png'), '--speed=10', '-o', '-'], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout[4:12] == b'ftypavif' and r.stdout == ref
    r = subprocess.run([CLI, '-', '--speed=10'], input=png, capture_output=True)
    assert r.returncode == 0 and r.stdout == ref
    # the same through the 8-bit path the survey's config line names (--depth=8, one thread)
    r = subprocess.run([CLI, '-', '--speed=10', '--depth=8', '-j1'], input=png, capture_output=True)
    assert r.returncode == 0 and r.stdout == oracle.ravif_encode(rgba, quality=80.0, alpha_quality=aq, speed=10, alpha_mode=1, depth=8, threads=1)[0]


G = os.path.join(ROOT, 'tests', 'golden', 'fullsize_golden.json')


@pytest.mark.skipif(not os.path.exists(G), reason='tests/golden/fullsize_golden.json not generated yet (tests/golden/make_fullsize_golden.py)')
@pytest.mark.parametrize('name', ['config2_1920x1080_rgb_s4_q80', 'config3_4096x4096_rgba_s4_q80', 'config5_7680x4320_rgb_s1_q80'])
def test_full_size_configs_equal_oracle_vectors(avifdec, name):
    """BASELINE configs 3 and 5 at full size: the HIP path == the oracle's output, which was produced once on the host
    (minutes of scalar C) and committed as sha256 (tests/golden/make_fullsize_golden.py)."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    allg = json.load(open(G))
    if name not in allg:
        pytest.skip('%s not in tests/golden/fullsize_golden.json yet' % name)
    g = allg[name]
    img = synth_image(g['w'], g['h'], index=g['index'], alpha=g['alpha'])
    e = m.Encoder().with_quality(g['quality']).with_alpha_quality(g['alpha_quality']).with_speed(g['speed']).with_bit_depth(g['depth'])
    b = m.BatchEncoder(e, 1, g['w'], g['h'], channels=4 if g['alpha'] else 3)
    b.upload(0, img)
    b.encode()
    got = b.get(0)
    assert got.color_byte_size == g['color_byte_size'] and got.alpha_byte_size == g['alpha_byte_size']
    assert len(got.avif_file) == g['avif_len'] and hashlib.sha256(got.avif_file).hexdigest() == g['avif_sha256']
    # size-independent properties at the full size: tile plan, dav1d conformance, decoder output == the encoder's reconstruction
    if name.startswith('config2'):
        assert b.num_tiles() == 32       # 256-px minimum tile size at speed 4: 1920*1080 / 256^2 = 31.6 -> 32 (a sparse launch: per-root synchronisation in the tile search)
    elif name.startswith('config5'):
        assert b.num_tiles() == 8        # 2048-px minimum tile size at speed 1 (ravif/src/av1encoder.rs:598-604): 7680*4320 / 2048^2 = 7.9 -> 8
    else:
        assert b.num_tiles() == 512      # 256-px minimum tile size at speed 4: 256 colour + 256 alpha tiles
    d = avifdec.decode(got.avif_file)
    assert (d['width'], d['height'], d['depth']) == (g['w'], g['h'], g['depth'])
    for a, r in zip(d['planes'], b.recon(0)):
        assert np.array_equal(a, r)
    if g['alpha']:
        assert np.array_equal(d['alpha'], b.recon(0, alpha=True)[0])
    b.close()


def test_pooled_batch_objects_give_the_same_bytes(oracle):
    """The one-call entry points reuse pooled batch objects (mi_avif.hip pool_acquire): a loop of encode_rgba calls with the same
    settings, interleaved with another shape / other settings / Exif, must give the same files as fresh objects."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    L = m.load_library()
    L.mi_release_cached.restype = None
    e = m.Encoder().with_quality(72).with_speed(6)
    a, b2, c = synth_image(160, 96, index=11, alpha=True), synth_image(160, 96, index=12, alpha=True), synth_image(96, 64, index=13)
    opaque = np.dstack([c, np.full(c.shape[:2], 255, np.uint8)])
    first = [e.encode_rgba(a).avif_file, e.encode_rgba(b2).avif_file, e.encode_rgba(opaque).avif_file]
    for im, want in zip((a, b2), first):
        ref, _, _ = oracle.ravif_encode(im, quality=72, alpha_quality=80, speed=6)
        assert want == ref
    # same shape again: opaque image through an object that last carried an alpha frame, and the other way round
    opaque160 = a.copy(); opaque160[..., 3] = 255
    got = e.encode_rgba(opaque160)
    assert got.alpha_byte_size == 0
    ref, _, _ = oracle.ravif_encode(opaque160, quality=72, alpha_quality=80, speed=6)
    assert got.avif_file == ref
    again = [e.encode_rgba(a).avif_file, e.encode_rgba(b2).avif_file, e.encode_rgba(opaque).avif_file]
    assert again == first
    with_exif = e.with_exif(b'Exif\x00\x00II*\x00\x08\x00\x00\x00\x00\x00').encode_rgba(a).avif_file
    assert with_exif != first[0] and b'Exif' in with_exif
    assert e.encode_rgba(a).avif_file == first[0]                      # the pooled object does not keep the Exif of its last user
    L.mi_release_cached()
    assert e.encode_rgba(a).avif_file == first[0]
    L.mi_release_cached()


def test_near_lossless_noise_grows_the_packed_buffers(oracle):
    """quality 100 on noise: the payload exceeds the initial 1/16 share of the worst-case packed size (mi_batch_wait grows the
    device / pinned pair); bytes still equal the oracle's."""
    import cavif_rs_amd as m
    rng = np.random.default_rng(5)
    im = rng.integers(0, 256, size=(96, 160, 3), dtype=np.uint8)
    e = m.Encoder().with_quality(100).with_speed(8).with_bit_depth(10)
    got = e.encode_rgb(im)
    assert got.color_byte_size > 96 * 160 * 3 * 2 // 16
    ref, _, _ = oracle.ravif_encode(im, quality=100, speed=8, depth=10)
    assert got.avif_file == ref
    assert e.encode_rgb(im).avif_file == ref                     # the grown pair is kept (pooled object)


@pytest.mark.parametrize('q,aq', [(95, 40), (40, 95), (81, 80), (80, 81)])
def test_colour_and_alpha_frames_with_different_switches(oracle, q, aq):
    """A picture's colour and alpha frames are searched in one launch, each with its own quality: rdo_tx_decision is `speed <= 4 && !high_quality`
    (ravif/src/av1encoder.rs:556,576), so the two frames of one launch can differ in it.  The launch must then run the general kernels, not the ones that hold the
    speed-4 switches as constants (tile_search.h Tools; found by tools/gpu_random_sweep.py in round 4: 3 of 150 cases)."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(200, 136, index=q + aq, alpha=True)
    e = m.Encoder().with_quality(q).with_alpha_quality(aq).with_speed(4).with_bit_depth(10)
    ref, col, al = oracle.ravif_encode(img, quality=q, alpha_quality=aq, speed=4, depth=10, alpha_mode=1)
    assert e.encode_rgba(img).avif_file == ref
    b = m.BatchEncoder(e, 2, 200, 136, 4)
    for i in range(2): b.upload(i, img)
    b.encode()
    assert b.get(0).avif_file == ref and b.get(1).avif_file == ref
    b.close()


@pytest.mark.parametrize('depth', [10, 8])
def test_saturated_primaries_through_the_front_end_kernel(oracle, avifdec, depth):
    """a-4 on the device: ravif's rgb_to_ycbcr rounds half away from zero and saturates only at the integer type (ravif/src/av1encoder.rs:504-523), so pure red
    gives Cr = 1024 at depth 10 -- one above the sample range -- and Cb / Cr = 255 (saturated) at depth 8.  Flat 64x64 patches of the
    primaries, white, black, mid grey and (1,2,3) go through mi_ravif_encode_rgb at quality 100 (quantizer 0: the finest steps, so the flat patches survive):
    bytes == oracle, and dav1d's picture carries the known answers clipped to the sample range in the middle of every patch."""
    import cavif_rs_amd as m
    cols = [(255, 255, 255), (0, 0, 0), (255, 0, 0), (0, 255, 0), (0, 0, 255), (128, 128, 128), (1, 2, 3), (255, 255, 0)]
    want10 = [(1023, 512, 512), (0, 512, 512), (306, 339, 1024), (601, 173, 84), (117, 1023, 429), (514, 512, 512), (7, 515, 510)]
    want8 = {2: (76, 85, 255), 4: (29, 255, 107)}
    img = np.zeros((128, 256, 3), np.uint8)
    for i, c in enumerate(cols):
        img[(i // 4) * 64:(i // 4) * 64 + 64, (i % 4) * 64:(i % 4) * 64 + 64] = c
    e = m.Encoder().with_quality(100).with_speed(4).with_bit_depth(depth)
    got = e.encode_rgb(img)
    ref, _, _ = oracle.ravif_encode(img, quality=100, speed=4, depth=depth)
    assert got.avif_file == ref
    mx = (1 << depth) - 1
    d = avifdec.decode(got.avif_file)
    assert d['depth'] == depth
    for i, c in enumerate(cols):
        cy, cx = (i // 4) * 64 + 32, (i % 4) * 64 + 32
        dec = tuple(int(d['planes'][p][cy, cx]) for p in range(3))
        exp = None
        if depth == 10 and i < len(want10): exp = want10[i]
        if depth == 8 and i in want8: exp = want8[i]
        if exp is not None:
            # quantizer 0 is not lossless: allow the finest step's rounding around the clipped known answer
            assert all(abs(a - min(b, mx)) <= 2 for a, b in zip(dec, exp)), (i, dec, exp)


def test_wide_picture_beyond_4096_columns_of_4x4_cells(oracle):
    """the deblock stages queue edge lines as packed (4x4 column, line, filter size) words; the column field was 12 bits, so pictures of 16384 samples
    and more across filtered the wrong lines.  16400 x 16 (4100 cells across, one superblock row, several tiles by the 4096-sample tile-width limit): HIP == oracle."""
    import cavif_rs_amd as m
    w, h, bd = 16400, 16, 10
    pl = planes(h, w, seed=w + h, bd=bd)
    cfg = oracle.make_config(w, h, bd, False, 121, 6)
    r = oracle.encode_planes(cfg, pl)
    obu, rec = m.encode_planes(pl, bd, 121, 6, False)
    assert obu == r['obu']
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


def test_cli_without_j_takes_the_host_core_count(tmp_path):
    """`threads: None` of the reference resolves to rayon::current_num_threads() = the host's logical cores (ravif/src/av1encoder.rs:665-668), and that bounds the tile
    target.  cavif_mi without -j (and with -j0) == cavif_mi -jN == the library with with_num_threads(N), N = the CPUs this process can use (affinity mask and cgroup quota, `bench.host_cores()`); RAYON_NUM_THREADS overrides N as it
    does for rayon's global pool.  768x512 at speed 4 asks for min(N, 6) tiles: on hosts with fewer than six CPUs the bound is live."""
    Image = pytest.importorskip('PIL.Image')
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(768, 512, index=11)
    Image.fromarray(img, 'RGB').save(tmp_path / 'a.png')
    import bench
    n = bench.host_cores()                # affinity and cgroup quota: Rust's available_parallelism()
    outs = {}
    for name, args, env in (('none', [], {}), ('j0', ['-j0'], {}), ('jN', ['-j%d' % min(n, 255)], {}), ('rayon3', [], {'RAYON_NUM_THREADS': '3'}), ('j3', ['-j3'], {})):
        r = subprocess.run([CLI, '-f', '-q', '-o', str(tmp_path / (name + '.avif'))] + args + [str(tmp_path / 'a.png')], capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs[name] = (tmp_path / (name + '.avif')).read_bytes()
    assert outs['none'] == outs['j0'] == outs['jN']
    assert outs['rayon3'] == outs['j3']
    aq = min((80.0 + 100.0) / 2.0, 80.0 + 80.0 / 4.0 + 2.0)
    rgba = np.dstack([img, np.full(img.shape[:2], 255, np.uint8)])
    want_n = m.Encoder().with_quality(80).with_alpha_quality(aq).with_speed(4).with_num_threads(min(n, 255)).encode_rgba(rgba).avif_file
    want_3 = m.Encoder().with_quality(80).with_alpha_quality(aq).with_speed(4).with_num_threads(3).encode_rgba(rgba).avif_file
    assert outs['jN'] == want_n and outs['j3'] == want_3 and want_3 != want_n or n <= 3
