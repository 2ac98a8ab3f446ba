"""-m gpu: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs (bit-exact),
against dav1d (conformance + recon match), through the ravif-level entry points and the batch API, and at
BASELINE's full 1080p size through size-independent properties."""
import io
import numpy as np
import pytest
from tests.helpers.images import planes, rgba_gradient, rgba_opaque

pytestmark = pytest.mark.gpu

CASES = [
    (64, 64, 8, 4, 121, False, 0), (64, 64, 8, 10, 121, False, 0), (128, 85, 8, 10, 121, False, 0), (129, 101, 10, 4, 121, False, 0),
    (200, 120, 10, 1, 121, False, 0), (200, 136, 10, 1, 66, True, 0), (256, 200, 10, 4, 66, True, 0), (300, 270, 10, 4, 121, False, 4),
    (136, 72, 8, 6, 200, False, 0), (136, 72, 8, 4, 10, False, 0), (72, 136, 8, 8, 160, False, 2), (8, 8, 8, 4, 121, False, 0),
    (17, 9, 10, 4, 90, False, 0), (640, 360, 10, 2, 121, False, 0), (640, 360, 8, 3, 170, False, 6),
]


@pytest.mark.parametrize('w,h,bd,speed,q,mono,tiles', CASES)
def test_hip_equals_oracle(oracle, avifdec, w, h, bd, speed, q, mono, tiles):
    import cavif_rs_amd as m
    pl = planes(h, w, seed=w + h, bd=bd, mono=mono)
    cfg = oracle.make_config(w, h, bd, mono, q, speed, tiles=tiles)
    r = oracle.encode_planes(cfg, pl)
    obu, rec = m.encode_planes(pl, bd, q, speed, mono, tiles=tiles)
    assert obu == r['obu'], 'bitstream differs (%d vs %d bytes)' % (len(obu), len(r['obu']))
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)
    d = avifdec.decode(oracle.container(obu, None, w, h, bd, mono_color=int(mono)))
    for a, b in zip(d['planes'], rec):
        assert np.array_equal(a, b)


def test_ravif_entry_points(oracle, avifdec):
    """encode_rgb / encode_rgba (opaque and with alpha) == oracle; reference windows of lib.rs:71-119 hold for the HIP path."""
    import cavif_rs_amd as m
    img = rgba_opaque()
    e = m.Encoder().with_quality(33).with_speed(10).with_num_threads(1)
    a = e.encode_rgba(img)
    ref, color, alpha = oracle.ravif_encode(img, quality=33, speed=10, depth=0, threads=1)
    assert a.avif_file == ref and a.alpha_byte_size == 0 and 150 < a.color_byte_size < 500
    b = e.with_bit_depth(10).encode_rgb(img[:, :, :3])
    assert b.avif_file == a.avif_file
    g = rgba_gradient()
    e2 = m.Encoder().with_quality(22).with_alpha_quality(22).with_speed(1).with_bit_depth(8).with_alpha_color_mode('dirty').with_num_threads(2)
    c = e2.encode_rgba(g)
    ref2, color2, alpha2 = oracle.ravif_encode(g, quality=22, alpha_quality=22, speed=1, depth=8, alpha_mode=0, threads=2)
    assert c.avif_file == ref2 and (c.color_byte_size, c.alpha_byte_size) == (color2, alpha2)
    assert 50 < c.color_byte_size < 1000 and 50 < c.alpha_byte_size < 1000
    d = avifdec.decode(c.avif_file)
    assert d['alpha'] is not None and d['depth'] == 8


def test_clean_alpha_matches_oracle(oracle):
    """UnassociatedClean (blurred_dirty_alpha on the GPU) == oracle; reference windows of lib.rs:121-147 hold."""
    import cavif_rs_amd as m
    from tests.helpers.images import rgba_noisy
    img = rgba_noisy()
    e = m.Encoder().with_quality(66).with_alpha_quality(88).with_speed(6).with_num_threads(1)
    clean = e.with_alpha_color_mode('clean').encode_rgba(img)
    dirty = e.with_alpha_color_mode('dirty').encode_rgba(img)
    rc, ccol, calpha = oracle.ravif_encode(img, quality=66, alpha_quality=88, speed=6, alpha_mode=1, threads=1)
    rd, dcol, dalpha = oracle.ravif_encode(img, quality=66, alpha_quality=88, speed=6, alpha_mode=0, threads=1)
    assert clean.avif_file == rc and dirty.avif_file == rd
    assert clean.alpha_byte_size == dirty.alpha_byte_size and 200 < clean.alpha_byte_size < 1000
    assert 2000 < clean.color_byte_size < 6000 and clean.color_byte_size < dirty.color_byte_size / 2
    # an image whose alpha is strictly {0, 255} is not touched by the cleaner
    hard = img.copy(); hard[:, :, 3] = np.where(hard[:, :, 3] > 10, 255, 0)
    assert e.with_alpha_color_mode('clean').encode_rgba(hard).avif_file == e.with_alpha_color_mode('dirty').encode_rgba(hard).avif_file


def test_rgb_identity_model(oracle):
    import cavif_rs_amd as m
    from PIL import Image
    rgb = np.stack(planes(96, 112, seed=5), -1).astype(np.uint8)
    got = m.Encoder().with_quality(70).with_speed(6).with_internal_color_model('rgb').with_bit_depth(8).encode_rgb(rgb)
    ref, _, _ = oracle.ravif_encode(rgb, quality=70, speed=6, color_model=1, depth=8)
    assert got.avif_file == ref
    im = Image.open(io.BytesIO(got.avif_file)); im.load()
    assert im.size == (112, 96)


def test_batch_matches_single_and_oracle(oracle):
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
    imgs = [synth_image(320, 200, index=i) for i in range(5)]
    b = m.BatchEncoder(e, len(imgs), 320, 200, channels=3)
    for i, im in enumerate(imgs):
        b.upload(i, im)
    b.encode()
    first = [b.get(i).avif_file for i in range(len(imgs))]
    b.encode()                                    # idempotent: a second pass over the resident batch gives the same bytes
    for i, im in enumerate(imgs):
        assert b.get(i).avif_file == first[i]
        ref, _, _ = oracle.ravif_encode(im, quality=80, speed=4, depth=10)
        assert first[i] == ref
        assert e.encode_rgb(im).avif_file == ref
    st = b.stage_ms()
    assert st['tile_search'] > 0
    b.close()


def test_full_size_1080p_properties(avifdec):
    """BASELINE config 2 size: dav1d decodes the HIP stream to exactly the HIP reconstruction; PSNR sane; deterministic."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(1920, 1080, index=0)
    e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
    b = m.BatchEncoder(e, 1, 1920, 1080, channels=3)
    b.upload(0, img)
    b.encode()
    out = b.get(0)
    rec = b.recon(0)
    d = avifdec.decode(out.avif_file)
    assert d['depth'] == 10 and (d['width'], d['height']) == (1920, 1080)
    for a, r in zip(d['planes'], rec):
        assert np.array_equal(a, r)
    b.encode()
    assert b.get(0).avif_file == out.avif_file
    assert b.num_tiles() == 32
    b.close()
    from PIL import Image
    dec = np.array(Image.open(io.BytesIO(out.avif_file)).convert('RGB')).astype(float)
    psnr = 10 * np.log10(255 ** 2 / np.mean((dec - img) ** 2))
    assert psnr > 30, psnr


def test_full_size_1080p_equals_oracle(oracle):
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(1920, 1080, index=1)
    got = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10).encode_rgb(img)
    ref, _, _ = oracle.ravif_encode(img, quality=80, speed=4, depth=10)
    assert got.avif_file == ref


def test_config3_shape_rgba_with_alpha(oracle, avifdec):
    """BASELINE config 3 shape at reduced size: RGBA with radial alpha ramp, speed 4 q80 (alpha q90 as the CLI derives), clean alpha."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(512, 384, index=7, alpha=True)
    e = m.Encoder().with_quality(80).with_alpha_quality(90).with_speed(4)
    got = e.encode_rgba(img)
    ref, col, al = oracle.ravif_encode(img, quality=80, alpha_quality=90, speed=4, depth=0, alpha_mode=1)
    assert got.avif_file == ref and got.alpha_byte_size == al and al > 0
    d = avifdec.decode(got.avif_file)
    assert d['alpha'] is not None and d['depth'] == 10 and (d['width'], d['height']) == (512, 384)


def test_async_pipeline_two_batches(oracle):
    """encode_async / wait on two resident batches driven alternately (bench.py's loop) gives the same bytes as the blocking call."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    e = m.Encoder().with_quality(80).with_speed(4).with_bit_depth(10)
    imgs = [synth_image(256, 192, index=20 + i) for i in range(3)]
    slots = [m.BatchEncoder(e, len(imgs), 256, 192, 3) for _ in range(2)]
    for b in slots:
        for i, im in enumerate(imgs):
            b.upload(i, im)
    ref = [oracle.ravif_encode(im, quality=80, speed=4, depth=10)[0] for im in imgs]
    slots[0].encode_async(); slots[1].encode_async()
    slots[0].wait(); slots[0].encode_async(); slots[1].wait(); slots[0].wait()
    for b in slots:
        for i in range(len(imgs)):
            assert b.get(i).avif_file == ref[i]
        b.close()


def test_with_exif_embeds_item_and_leaves_payload_unchanged(oracle):
    """Encoder::with_exif (ravif/src/av1encoder.rs:208-218): same AV1 payload, plus an Exif item an independent reader returns."""
    import cavif_rs_amd as m
    PIL = pytest.importorskip('PIL.Image')
    img = rgba_gradient(96, 64)[..., :3]
    exif = b'II\x2a\x00\x08\x00\x00\x00\x01\x00\x31\x01\x02\x00\x04\x00\x00\x00xyz\x00\x00\x00\x00\x00'
    e = m.Encoder().with_quality(70).with_speed(6)
    plain, tagged = e.encode_rgb(img), e.with_exif(exif).encode_rgb(img)
    assert plain.color_byte_size == tagged.color_byte_size and len(tagged.avif_file) > len(plain.avif_file)
    ref, _, _ = oracle.ravif_encode(img, quality=70, speed=6)
    assert plain.avif_file == ref
    im = PIL.open(io.BytesIO(tagged.avif_file))
    assert exif in bytes(im.info.get('exif') or b'')
    assert np.array_equal(np.asarray(im.convert('RGB')), np.asarray(PIL.open(io.BytesIO(plain.avif_file)).convert('RGB')))


def _random_cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        w, h = int(rng.integers(8, 220)), int(rng.integers(8, 180))
        out.append((w, h, int(rng.choice([8, 10])), int(rng.integers(1, 11)), int(rng.integers(5, 251)), bool(rng.integers(0, 4) == 0), int(rng.choice([0, 0, 2, 4]))))
    return out


@pytest.mark.parametrize('w,h,bd,speed,q,mono,tiles', _random_cases(24, 20260924))
def test_random_configuration_sweep(oracle, w, h, bd, speed, q, mono, tiles):
    """Seeded sweep over ragged sizes, both depths, every speed preset, the whole quantizer range, colour / monochrome, tile counts."""
    import cavif_rs_amd as m
    pl = planes(h, w, seed=1000 + w * 7 + h, bd=bd, mono=mono)
    r = oracle.encode_planes(oracle.make_config(w, h, bd, mono, q, speed, tiles=tiles), pl)
    obu, rec = m.encode_planes(pl, bd, q, speed, mono, tiles=tiles)
    assert obu == r['obu'], 'bitstream differs (%d vs %d bytes)' % (len(obu), len(r['obu']))
    for a, b in zip(rec, r['recon']):
        assert np.array_equal(a, b)


def test_config3_full_size_4096_rgba_properties(avifdec):
    """BASELINE config 3 at its full size (4096x4096 RGBA, speed 4, q80): both planes' streams decode (dav1d) to exactly the
    encoder's reconstruction, the call is deterministic, and the alpha plane went through the second encode."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(4096, 4096, index=3, alpha=True)
    e = m.Encoder().with_quality(80).with_alpha_quality(90).with_speed(4)
    b = m.BatchEncoder(e, 1, 4096, 4096, channels=4)
    b.upload(0, img)
    b.encode()
    out = b.get(0)
    assert out.alpha_byte_size > 0 and out.color_byte_size > 0
    d = avifdec.decode(out.avif_file)
    assert (d['width'], d['height'], d['depth']) == (4096, 4096, 10) and d['alpha'] is not None
    for a, r in zip(d['planes'], b.recon(0)):
        assert np.array_equal(a, r)
    assert np.array_equal(d['alpha'], b.recon(0, alpha=True)[0])
    b.encode()
    assert b.get(0).avif_file == out.avif_file
    b.close()


# (BASELINE config 5 at full size -- conformance, reconstruction identity, tile plan and the oracle's sha256 -- is one test in
#  tests/test_gpu_parity_cells.py::test_full_size_configs_equal_oracle_vectors: the 8K speed-1 bottom-up encode takes ~50 s.)


@pytest.mark.parametrize('w,h', [(1, 1), (3, 2), (5, 7), (4, 64), (65, 3)])
def test_tiny_and_sliver_images(oracle, avifdec, w, h):
    """Frames smaller than one block / one superblock, and one-block-wide slivers: edge replication and the forced splits at the frame boundary."""
    import cavif_rs_amd as m
    rng = np.random.default_rng(w * 100 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    got = m.Encoder().with_quality(75).with_speed(5).encode_rgb(img)
    ref, _, _ = oracle.ravif_encode(img, quality=75, speed=5)
    assert got.avif_file == ref
    d = avifdec.decode(got.avif_file)
    assert (d['width'], d['height']) == (w, h)


def test_entry_points_are_reentrant(oracle):
    """The reference calls the encoder from several rayon threads at once (encode_color || encode_alpha, av1encoder.rs:454, and
    one task per file, src/main.rs:223): concurrent calls through the C ABI must not share state."""
    import threading
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    imgs = [synth_image(192 + 16 * i, 128, index=40 + i, alpha=(i % 2 == 1)) for i in range(6)]
    e = m.Encoder().with_quality(60).with_speed(6)
    ref = [oracle.ravif_encode(im, quality=60, alpha_quality=80, speed=6)[0] for im in imgs]
    out = [None] * len(imgs)
    def work(i):
        for _ in range(3):
            out[i] = (e.encode_rgba(imgs[i]) if imgs[i].shape[2] == 4 else e.encode_rgb(imgs[i])).avif_file
    th = [threading.Thread(target=work, args=(i,)) for i in range(len(imgs))]
    for t in th: t.start()
    for t in th: t.join()
    assert out == ref


@pytest.mark.parametrize('quality,depth,speed,idx', [(50, 8, 4, 11), (95, 10, 4, 12), (65, 10, 3, 13), (35, 8, 2, 14), (88, 10, 6, 15)])
def test_quarter_hd_across_qualities_equals_oracle(oracle, quality, depth, speed, idx):
    """960x540 synthetic photographs across the quality range (low quality switches LRF/CDEF flags and 64x64 partitions on,
    high quality caps blocks at 16x16 and doubles the tile size), both depths, speeds around the bench preset."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    img = synth_image(960, 540, index=idx)
    got = m.Encoder().with_quality(quality).with_speed(speed).with_bit_depth(depth).encode_rgb(img)
    ref, _, _ = oracle.ravif_encode(img, quality=quality, speed=speed, depth=depth)
    assert got.avif_file == ref
