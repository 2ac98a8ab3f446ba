"""The shipped HIP kernels + host engine, executed lane by lane on the CPU (tests/emu/: a SIMT emulator, test infrastructure),
must produce the oracle's bytes.  This is what keeps a kernel edit honest when no MI355X is attached; the `-m gpu` tests remain the
parity tests proper.  The emulated library is a separate build (tests/emu/_build/), never the product library."""
import json
import os
import subprocess
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def emu_env(oracle):
    from tests import emu
    return emu.env()


def _run(emu_env, which, timeout, **extra):
    emu_env = dict(emu_env, **extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'emu', 'emu_cases.py'), ROOT, which], env=emu_env, capture_output=True, text=True, timeout=timeout)
    rows = [json.loads(l) for l in p.stdout.splitlines() if l.startswith('{')]
    return p, rows


def test_emulated_kernels_equal_oracle(emu_env):
    p, rows = _run(emu_env, 'quick', 900)
    assert rows, p.stderr[-2000:]
    bad = [r['case'] for r in rows if not r['ok']]
    assert not bad and p.returncode == 0, 'emulated HIP path differs from the oracle: %s\n%s' % (bad, p.stderr[-2000:])
    assert len(rows) >= 8


def test_emulated_kernels_do_not_depend_on_lane_order(emu_env):
    """Between two meeting points the emulator runs the lanes of a wavefront one after another; with MI_EMU_REVERSE they (and the waves of
    a workgroup) run in the opposite order.  A kernel with an unsynchronised LDS / global exchange between lanes gives different bytes."""
    p, rows = _run(emu_env, 'quick', 900, MI_EMU_REVERSE='1')
    bad = [r['case'] for r in rows if not r['ok']]
    assert rows and not bad and p.returncode == 0, 'lane-order dependence: %s\n%s' % (bad, p.stderr[-2000:])


def test_emulated_kernels_rect_partition_cases(emu_env):
    """Inputs on which PARTITION_HORZ / PARTITION_VERT of 8x8 nodes are chosen often (dev_rect.h: 8x4 / 4x8 blocks, 2:1 transforms), including the full
    mode set of speed 1 that keeps the one-candidate-per-wavefront path."""
    p, rows = _run(emu_env, 'rect', 900)
    assert rows and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])


def test_emulated_kernels_64x64_level(emu_env):
    """dev_blk64.h without a GPU: pictures on which 64x64 blocks are chosen (asserted from the oracle's block map), bottom-up and top-down walkers, 4:4:4 with its
    four 32x32 chroma transform blocks per plane and 4:0:0; once more with the lanes and waves of the emulator running in reverse order."""
    for extra in ({}, {'MI_EMU_REVERSE': '1'}):
        p, rows = _run(emu_env, 'blk64', 900, **extra)
        assert len(rows) == 3 and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])
        assert sum(r['n64'] for r in rows) >= 2, rows


def test_emulated_batch_api_equals_oracle(emu_env):
    """The batch entry points under the emulator's thread pool: work lists that span several frames (colour and colour + alpha images, two block-size
    classes in one encode, top-down and bottom-up order, colour and alpha frames on different sides of the high-quality threshold) equal the oracle byte for byte."""
    p, rows = _run(emu_env, 'batch', 1200)
    assert len(rows) == 6 and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])


def test_a_dependency_wait_that_never_ends_fails_the_encode_in_bounded_time(emu_env):
    """tile_search.h root_wait: bounded polls.  The emulator drops every root publish (MI_EMU_DROP_PUBLISH): the first wait of the frame runs into its bound and sets
    the frame's sticky error word, the other waits leave at their next check, the entropy stage fails every tile, the host returns MI_ENCODING_ERROR -- no hang,
    no stream whose reconstruction the search did not see."""
    p, rows = _run(emu_env, 'giveup', 900, MI_EMU_DROP_PUBLISH='1')
    assert len(rows) == 1 and rows[0]['ok'] and p.returncode == 0, (rows, p.stderr[-2000:])
    assert 'ncod' in rows[0]['outcome'] or 'rror' in rows[0]['outcome'], rows


def test_two_distinct_emulated_devices(emu_env):
    """the multi-device fan-out (mi_ravif_encode_stream: one host thread per device, shared cursor, per-device tables / arenas / memory budgets) had only
    ever run with devices=[0, 0].  MI_EMU_DEVICES=2 gives the emulator two devices that differ in compute units and free memory and that refuse each other's memory
    (a copy or a kernel argument naming the other device's allocation aborts).  Both devices must do work; every file == oracle.  Unmeasured on real multi-GPU hardware."""
    p, rows = _run(emu_env, 'twodev', 1200, MI_EMU_DEVICES='2')
    assert len(rows) == 4 and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])


def test_product_library_is_not_the_emulator():
    """The product library is built by hipcc for gfx950 and knows nothing of the emulator; without a GPU it reports no device."""
    import cavif_rs_amd as m
    from cavif_rs_amd import encoder
    assert 'emu' not in os.path.basename(encoder.library_path())
    with open(encoder.library_path(), 'rb') as fh:
        blob = fh.read()
    assert b'emu_switch' not in blob and b'gfx950' in blob
    import __graft_entry__ as g
    csrc = os.path.join(ROOT, 'cavif_rs_amd', 'csrc')
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(ROOT, 'include', 'mi_avif.h')]
    stamp = encoder.library_path() + '.stamp'
    if os.path.exists(stamp) and not os.environ.get('MI_AVIF_LIB'):
        assert open(stamp).read().strip() == g._digest(srcs, g.HIPCC_FLAGS), 'libmi_avif.so is older than its sources or was built with other flags: python __graft_entry__.py'

