"""CPU side of the cavif command line: the PNG reader against Pillow for every colour type / bit depth /
tRNS / Adam7, and the CLI's argument, path and failure rules (src/main.rs:24-43, 137-200, 244-252) up to the point where
a GPU is needed -- without one every file must fail loudly (no CPU fallback)."""
import ctypes as C
import io
import os
import struct
import subprocess
import zlib
import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, 'cavif_rs_amd', 'cavif_mi')
Image = pytest.importorskip('PIL.Image')


def _decode(data):
    import cavif_rs_amd as m
    L = m.load_library()
    L.mi_png_decode_rgba.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    out = C.POINTER(C.c_uint8)(); w = C.c_uint32(); h = C.c_uint32()
    st = L.mi_png_decode_rgba(data, len(data), C.byref(out), C.byref(w), C.byref(h))
    if st:
        return st, None
    a = np.ctypeslib.as_array(out, shape=(h.value, w.value, 4)).copy()
    L.mi_free(out)
    return 0, a


def _png_bytes(img, **kw):
    b = io.BytesIO(); img.save(b, 'PNG', **kw); return b.getvalue()


def _chunk(t, body):
    return struct.pack('>I', len(body)) + t + body + struct.pack('>I', zlib.crc32(t + body) & 0xffffffff)


def _raw_png(w, h, depth, ctype, rows_by_pass, interlace, extra=b''):
    raw = b''.join(b'\x00' + r for rows in rows_by_pass for r in rows)
    return b'\x89PNG\r\n\x1a\n' + _chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, ctype, 0, 0, interlace)) + extra + _chunk(b'IDAT', zlib.compress(raw)) + _chunk(b'IEND', b'')


@pytest.mark.parametrize('mode', ['RGB', 'RGBA', 'L', 'LA', 'P', '1'])
def test_png_reader_matches_pillow(mode):
    rng = np.random.default_rng(len(mode) * 7)
    w, h = 37, 23
    if mode == 'P':
        img = Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8), 'RGB').quantize(60)
    elif mode == '1':
        img = Image.fromarray((rng.integers(0, 2, (h, w), dtype=np.uint8) * 255), 'L').convert('1')
    else:
        ch = {'RGB': 3, 'RGBA': 4, 'L': 1, 'LA': 2}[mode]
        a = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
        img = Image.fromarray(a[..., 0] if ch == 1 else a, mode)
    st, got = _decode(_png_bytes(img))
    assert st == 0
    assert np.array_equal(got, np.asarray(img.convert('RGBA')))


def test_png_reader_palette_transparency_and_filters():
    rng = np.random.default_rng(3)
    img = Image.fromarray(rng.integers(0, 256, (40, 50, 3), dtype=np.uint8), 'RGB').quantize(16)
    data = _png_bytes(img, transparency=bytes([0, 128, 255, 7]), optimize=True)       # tRNS for the first four entries, 4-bit indices
    st, got = _decode(data)
    assert st == 0 and np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)).convert('RGBA')))
    smooth = np.add.outer(np.arange(64), np.arange(80)).astype(np.uint8)              # gradients make the encoder use Sub/Up/Average/Paeth
    data = _png_bytes(Image.fromarray(np.stack([smooth, smooth.T[:64, :80] if False else smooth // 2, 255 - smooth], -1), 'RGB'))
    st, got = _decode(data)
    assert st == 0 and np.array_equal(got[..., :3], np.asarray(Image.open(io.BytesIO(data)).convert('RGB')))


def test_png_reader_16_bit_keeps_high_byte():
    """load_rgba: RGB16 / RGBA16 / GRAY16 -> (c >> 8) as u8 (src/main.rs:272-278)."""
    rng = np.random.default_rng(11)
    w, h = 9, 5
    v = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)
    rows = [b''.join(struct.pack('>HHH', *px) for px in row) for row in v]
    st, got = _decode(_raw_png(w, h, 16, 2, [rows], 0))
    assert st == 0 and np.array_equal(got[..., :3], (v >> 8).astype(np.uint8)) and (got[..., 3] == 255).all()
    g = rng.integers(0, 65536, (h, w), dtype=np.uint16)
    rows = [b''.join(struct.pack('>H', x) for x in row) for row in g]
    st, got = _decode(_raw_png(w, h, 16, 0, [rows], 0))
    assert st == 0 and np.array_equal(got[..., 0], (g >> 8).astype(np.uint8)) and np.array_equal(got[..., 0], got[..., 2])


def test_png_reader_adam7():
    rng = np.random.default_rng(5)
    w, h = 21, 13
    a = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    passes = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]
    rows_by_pass = []
    for x0, y0, dx, dy in passes:
        sub = a[y0::dy, x0::dx]
        rows_by_pass.append([r.tobytes() for r in sub] if sub.size else [])
    data = _raw_png(w, h, 8, 6, rows_by_pass, 1)
    st, got = _decode(data)
    assert st == 0 and np.array_equal(got, a)
    assert np.array_equal(got, np.asarray(Image.open(io.BytesIO(data)).convert('RGBA')))


def test_png_reader_rejects_garbage():
    assert _decode(b'\xff\xd8\xff\xe0' + b'\0' * 64)[0] == 2                  # a JPEG: unsupported, not a crash
    good = _png_bytes(Image.new('RGB', (8, 8), (1, 2, 3)))
    assert _decode(good[:40])[0] in (2, 3)
    bad = bytearray(good); bad[-20] ^= 0xff
    assert _decode(bytes(bad))[0] in (0, 3)                                    # corrupt zlib stream / CRC is not checked, never a crash


def _run(args, **kw):
    return subprocess.run([CLI] + args, capture_output=True, **kw)


@pytest.mark.skipif(not os.path.exists(CLI), reason='CLI not built')
def test_cli_argument_rules(tmp_path):
    assert _run([]).returncode == 1
    for bad in (['-Q', '0', 'x.png'], ['-Q', '101', 'x.png'], ['-s', '0', 'x.png'], ['-s', '11', 'x.png'], ['--color', 'cmyk', 'x.png'], ['--depth', '12', 'x.png']):
        r = _run(bad)
        assert r.returncode == 1 and b'error' in r.stderr
    r = _run([str(tmp_path / 'missing.png')])
    assert r.returncode == 1 and b'Unable to read input image' in r.stderr
    # an existing .avif among the inputs is ignored with a warning (src/main.rs:145-156)
    (tmp_path / 'old.avif').write_bytes(b'x')
    r = _run([str(tmp_path / 'old.avif')])
    assert r.returncode == 1 and b"already an AVIF" in r.stderr and b'No PNG/JPEG files specified' in r.stderr


@pytest.mark.skipif(not os.path.exists(CLI), reason='CLI not built')
def test_cli_skips_existing_output_and_fails_loudly_without_gpu(tmp_path):
    import cavif_rs_amd as m
    p = tmp_path / 'a.png'
    Image.new('RGB', (16, 16), (10, 200, 30)).save(p)
    (tmp_path / 'a.avif').write_bytes(b'existing')
    r = _run([str(p)])
    assert r.returncode == 1 and b'already exists; skipping' in r.stderr              # src/main.rs:195-200
    assert (tmp_path / 'a.avif').read_bytes() == b'existing'
    if m.device_count() == 0:
        r = _run(['-f', str(p)])
        assert r.returncode == 1 and b'no HIP device' in r.stderr                      # no silent CPU path
        assert (tmp_path / 'a.avif').read_bytes() == b'existing'


@pytest.mark.skipif(not os.path.exists(CLI), reason='CLI not built')
def test_cli_exit_handoff_reports_what_the_foreground_exit_reports(tmp_path):
    """CAVIF_MI_BACKGROUND_EXIT=1 (opt-in): the work runs in a child whose status reaches the caller through a pipe (device teardown off the caller's clock): same exit
    status, same stderr / stdout, streams closed when the command returns as in the default one-process mode, for usage errors (the child returns from
    main before reporting), per-file failures (it reports 1) and a child that dies (signal -> 128 + signo)."""
    import signal, time
    p = tmp_path / 'a.png'
    Image.new('RGB', (16, 16), (10, 200, 30)).save(p)
    (tmp_path / 'a.avif').write_bytes(b'existing')
    for args in ([], ['-Q', '0', str(p)], [str(p)], [str(tmp_path / 'missing.png')]):
        a = _run(args, env=dict(os.environ, CAVIF_MI_BACKGROUND_EXIT='1'))
        b = _run(args)
        assert (a.returncode, a.stdout, a.stderr) == (b.returncode, b.stdout, b.stderr) and a.returncode == 1
    # a child that is killed: the parent returns 128 + the signal (stdin keeps the child waiting inside read_all)
    pr = subprocess.Popen([CLI, '-o', str(tmp_path / 'o.avif'), '-'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, CAVIF_MI_BACKGROUND_EXIT='1'))
    time.sleep(0.3)
    kids = subprocess.run(['ps', '-o', 'pid=', '--ppid', str(pr.pid)], capture_output=True).stdout.split()
    assert len(kids) == 1
    os.kill(int(kids[0]), signal.SIGTERM)
    assert pr.wait(timeout=20) == 128 + signal.SIGTERM
    pr.stdin.close(); pr.stdout.close(); pr.stderr.close()


def test_png_reader_survives_corruption():
    """Mutated / truncated PNGs: mi_png_decode_rgba returns a status, never crashes, never allocates a claimed canvas
    that the compressed data cannot back."""
    import ctypes as C, io, zlib, struct
    import numpy as np
    from PIL import Image
    import cavif_rs_amd as m
    L = m.load_library()
    L.mi_png_decode_rgba.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    rng = np.random.default_rng(7)
    seeds = []
    for mode, shape in (('RGB', (23, 31, 3)), ('RGBA', (16, 16, 4)), ('L', (9, 40)), ('P', (12, 12))):
        arr = rng.integers(0, 256, size=shape, dtype=np.uint8)
        im = Image.fromarray(arr if mode != 'P' else arr % 7, mode)
        if mode == 'P':
            im.putpalette([int(x) for x in rng.integers(0, 256, size=21)])
        b = io.BytesIO(); im.save(b, format='PNG', interlace=bool(len(seeds) & 1)); seeds.append(b.getvalue())

    def run(data):
        buf = (C.c_uint8 * max(1, len(data))).from_buffer_copy(data if data else b'\\0')
        out = C.POINTER(C.c_uint8)(); w = C.c_uint32(); h = C.c_uint32()
        st = L.mi_png_decode_rgba(buf, len(data), C.byref(out), C.byref(w), C.byref(h))
        if st == 0:
            assert 0 < w.value <= 65536 and 0 < h.value <= 65536
            L.mi_free(out)
        return st
    ok = 0
    for s in seeds:
        assert run(s) == 0
        for _ in range(120):
            d = bytearray(s)
            kind = rng.integers(0, 4)
            if kind == 0:
                d = d[:int(rng.integers(0, len(d)))]
            elif kind == 1:
                for _ in range(int(rng.integers(1, 6))):
                    d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
            elif kind == 2:
                p = int(rng.integers(8, len(d))); d[p:p] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))
            else:
                struct.pack_into('>II', d, 16, int(rng.integers(1, 1 << 16)), int(rng.integers(1, 1 << 16)))     # lie about the canvas
            ok += run(bytes(d)) == 0
    # a 100-byte file that claims 65536 x 65536 must be refused without allocating 17 GB
    ihdr = struct.pack('>IIBBBBB', 65536, 65536, 8, 6, 0, 0, 0)
    def chunk(t, dd): return struct.pack('>I', len(dd)) + t + dd + struct.pack('>I', zlib.crc32(t + dd) & 0xffffffff)
    bomb = b'\\x89PNG\\r\\n\\x1a\\n' + chunk(b'IHDR', ihdr) + chunk(b'IDAT', zlib.compress(b'\\0' * 64)) + chunk(b'IEND', b'')
    assert run(bomb) != 0
