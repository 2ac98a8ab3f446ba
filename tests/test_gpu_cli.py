"""-m gpu: the cavif command line replica (cavif_rs_amd/cli/cavif_mi.cpp over the C ABI) and mi_ravif_encode_batch
against the library entry points they wrap: same bytes, the reference's report line, path and stdin/stdout rules
(src/main.rs:110-252)."""
import io
import os
import re
import subprocess
import numpy as np
import pytest
from tests.helpers.images import rgba_gradient

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, 'cavif_rs_amd', 'cavif_mi')
Image = pytest.importorskip('PIL.Image')


def _cli_encoder(quality=80.0, speed=4, dirty=False, depth=0, color=0):
    import cavif_rs_amd as m
    aq = min((quality + 100.0) / 2.0, quality + quality / 4.0 + 2.0)           # src/main.rs:115
    e = m.Encoder().with_quality(quality).with_alpha_quality(aq).with_speed(speed).with_alpha_color_mode('dirty' if dirty else 'clean')
    if depth:
        e = e.with_bit_depth(depth)
    if color:
        e = e.with_internal_color_model('rgb')
    return e


def test_encode_many_equals_single_calls_and_oracle(oracle):
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    imgs = [synth_image(160, 96, index=1), synth_image(160, 96, index=2), synth_image(96, 160, index=3), synth_image(128, 64, index=4, alpha=True), synth_image(160, 96, index=5)]
    e = m.Encoder().with_quality(70).with_speed(6)
    got = m.encode_many(e, imgs, devices=[0])
    for im, g in zip(imgs, got):
        ref, _, _ = oracle.ravif_encode(im, quality=70, alpha_quality=80, speed=6)
        assert g.avif_file == ref
    again = m.encode_many(e, imgs)                                              # every visible device
    assert [g.avif_file for g in again] == [g.avif_file for g in got]
    with pytest.raises(m.AvifError):
        m.encode_many(e, imgs, devices=[m.device_count() + 3])


def test_encode_many_with_a_dozen_shapes_and_two_workers_on_one_device(oracle):
    """A directory of differently sized files: every shape gets its own batch object, the worker gives the least recently used ones back when its
    memory budget says so, runs follow the arrival order.  devices=[0, 0] drives the multi-device code path (two host workers, one shared cursor,
    independent slots) on a single GPU; the bytes do not depend on which worker took an image."""
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image
    shapes = [(64 + 8 * k, 48 + 16 * (k % 5)) for k in range(13)]
    imgs = [synth_image(w, h, index=k, alpha=(k % 4 == 3)) for k, (w, h) in enumerate(shapes)]
    imgs += [synth_image(*shapes[2], index=40), synth_image(*shapes[2], index=41), synth_image(*shapes[0], index=42)]      # shapes that come back
    e = m.Encoder().with_quality(75).with_speed(6)
    ref = [oracle.ravif_encode(im, quality=75, alpha_quality=80, speed=6)[0] for im in imgs]
    got = m.encode_many(e, imgs, devices=[0])
    assert [g.avif_file for g in got] == ref
    two = m.encode_many(e, imgs, devices=[0, 0])
    assert [g.avif_file for g in two] == ref


def test_cli_matches_library_and_reports_like_the_reference(tmp_path):
    rgb = (np.add.outer(np.arange(72), np.arange(120))[..., None] * np.array([1, 2, 3])).astype(np.uint8)
    rgba = rgba_gradient(96, 80)
    Image.fromarray(rgb, 'RGB').save(tmp_path / 'one.png')
    Image.fromarray(rgba, 'RGBA').save(tmp_path / 'two.png')
    r = subprocess.run([CLI, str(tmp_path / 'one.png'), str(tmp_path / 'two.png')], capture_output=True)
    assert r.returncode == 0, r.stderr
    e = _cli_encoder()
    want_one = e.encode_rgba(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]))
    want_two = e.encode_rgba(rgba)
    assert (tmp_path / 'one.avif').read_bytes() == want_one.avif_file
    assert (tmp_path / 'two.avif').read_bytes() == want_two.avif_file
    lines = sorted(r.stdout.decode().strip().splitlines())
    m1 = re.fullmatch(r'(.*one\.avif): (\d+)KB \((\d+)B color, (\d+)B alpha, (\d+)B HEIF\)', lines[0])
    assert m1 and int(m1.group(2)) == -(-len(want_one.avif_file) // 1000) and int(m1.group(3)) == want_one.color_byte_size and int(m1.group(4)) == 0
    assert int(m1.group(5)) == len(want_one.avif_file) - want_one.color_byte_size
    m2 = re.fullmatch(r'(.*two\.avif): (\d+)KB \((\d+)B color, (\d+)B alpha, (\d+)B HEIF\)', lines[1])
    assert m2 and int(m2.group(4)) == want_two.alpha_byte_size > 0
    # second run: outputs exist -> skipped, exit 1, files untouched; -f overwrites
    r = subprocess.run([CLI, '-q', str(tmp_path / 'one.png')], capture_output=True)
    assert r.returncode == 1 and r.stdout == b'' and r.stderr == b''
    r = subprocess.run([CLI, '-f', '-Q', '50', '-s', '7', '--depth', '8', str(tmp_path / 'one.png')], capture_output=True)
    assert r.returncode == 0
    want = _cli_encoder(50.0, 7, depth=8).encode_rgba(np.dstack([rgb, np.full(rgb.shape[:2], 255, np.uint8)]))
    assert (tmp_path / 'one.avif').read_bytes() == want.avif_file


def test_cli_files_are_complete_when_the_command_returns_in_both_exit_modes(tmp_path):
    """Default: one process, device teardown included.  CAVIF_MI_BACKGROUND_EXIT=1 (opt-in): the work runs in a child that reports through a pipe once every
    file is written (the device teardown is off the caller's clock).  Same bytes, same report, every file complete at return."""
    import os
    files = []
    for i in range(5):
        Image.fromarray(rgba_gradient(64 + 8 * i, 48), 'RGBA').save(tmp_path / ('f%d.png' % i)); files.append(str(tmp_path / ('f%d.png' % i)))
    outs = []
    for env in ({}, {'CAVIF_MI_BACKGROUND_EXIT': '1'}):
        r = subprocess.run([CLI, '-f'] + files, capture_output=True, env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr
        outs.append(([(tmp_path / ('f%d.avif' % i)).read_bytes() for i in range(5)], sorted(r.stdout.splitlines())))   # read at once: nothing is still being written
        for i in range(5): (tmp_path / ('f%d.avif' % i)).unlink()
    assert outs[0] == outs[1] and all(len(b) > 100 for b in outs[0][0])


def test_cli_output_directory_stdio_and_dirty_alpha(tmp_path):
    rgba = rgba_gradient(64, 48)
    Image.fromarray(rgba, 'RGBA').save(tmp_path / 'g.png')
    Image.fromarray(rgba[::-1].copy(), 'RGBA').save(tmp_path / 'h.png')
    outdir = tmp_path / 'out' / 'nested'
    r = subprocess.run([CLI, '-o', str(outdir), '--dirty-alpha', '--color', 'rgb', str(tmp_path / 'g.png'), str(tmp_path / 'h.png')], capture_output=True)
    assert r.returncode == 0, r.stderr
    e = _cli_encoder(dirty=True, color=1)
    assert (outdir / 'g.avif').read_bytes() == e.encode_rgba(rgba).avif_file
    assert (outdir / 'h.avif').read_bytes() == e.encode_rgba(rgba[::-1].copy()).avif_file
    # stdin -> stdout, nothing else on stdout
    r = subprocess.run([CLI, '-'], input=(tmp_path / 'g.png').read_bytes(), capture_output=True)
    assert r.returncode == 0 and r.stdout == _cli_encoder().encode_rgba(rgba).avif_file
    # single input with -o file
    r = subprocess.run([CLI, '-o', str(tmp_path / 'named.avif'), str(tmp_path / 'g.png')], capture_output=True)
    assert r.returncode == 0 and (tmp_path / 'named.avif').read_bytes() == _cli_encoder().encode_rgba(rgba).avif_file
    im = Image.open(io.BytesIO((tmp_path / 'named.avif').read_bytes()))
    assert im.size == (64, 48)



def test_cli_keeps_going_past_a_bad_file(tmp_path):
    """One unreadable input among many: the reference reports it, carries on with the rest and exits 1 (src/main.rs:158-186).
    The streamed path must hand every other image its own file (no slot mix-up after the gap)."""
    from cavif_rs_amd.synth import synth_image
    e = _cli_encoder()
    names = []
    for i in range(7):
        p = tmp_path / ('im%d.png' % i)
        if i in (2, 5):
            p.write_bytes(b'\x89PNG\r\n\x1a\n' + b'garbage' * 9 if i == 2 else b'')
        else:
            Image.fromarray(synth_image(96 + 32 * (i & 1), 64, index=i), 'RGB').save(p)
        names.append(str(p))
    r = subprocess.run([CLI] + names, capture_output=True)
    assert r.returncode == 1
    assert r.stderr.count(b'error') >= 2 and b'im2.png' in r.stderr and b'im5.png' in r.stderr
    for i in range(7):
        out = tmp_path / ('im%d.avif' % i)
        if i in (2, 5):
            assert not out.exists()
        else:
            im = synth_image(96 + 32 * (i & 1), 64, index=i)
            assert out.read_bytes() == e.encode_rgba(np.dstack([im, np.full(im.shape[:2], 255, np.uint8)])).avif_file
