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


# The emulator runs are independent processes and most of this suite's time: the ones the selected tests need are started together (three at a time) when the
# first of them is asked for, and each test waits for its own.  job -> (case set of tests/emu/emu_cases.py, timeout, extra environment); test -> its jobs.
_JOBS = {'quick': ('quick', 900, {}), 'quick_reverse': ('quick', 900, {'MI_EMU_REVERSE': '1'}), 'rect': ('rect', 900, {}),
         'blk64': ('blk64', 900, {}), 'blk64_reverse': ('blk64', 900, {'MI_EMU_REVERSE': '1'}), 'batch': ('batch', 1200, {}),
         'giveup': ('giveup', 900, {'MI_EMU_DROP_PUBLISH': '1'}), 'twodev': ('twodev', 1200, {'MI_EMU_DEVICES': '2'})}
_TEST_JOBS = {'test_emulated_kernels_equal_oracle': ['quick'], 'test_emulated_kernels_do_not_depend_on_lane_order': ['quick_reverse'],
              'test_emulated_kernels_rect_partition_cases': ['rect'], 'test_emulated_kernels_64x64_level': ['blk64', 'blk64_reverse'],
              'test_emulated_batch_api_equals_oracle': ['batch'], 'test_a_dependency_wait_that_never_ends_fails_the_encode_in_bounded_time': ['giveup'],
              'test_two_distinct_emulated_devices': ['twodev']}


@pytest.fixture(scope='module')
def emu_jobs(request, emu_env):
    from concurrent.futures import ThreadPoolExecutor
    wanted = []
    for item in request.session.items:
        for j in _TEST_JOBS.get(getattr(item, 'originalname', None) or item.name, []):
            if j not in wanted: wanted.append(j)
    wanted.sort(key=lambda j: -_JOBS[j][1])                      # the long ones first
    ex = ThreadPoolExecutor(max_workers=3)
    futs = {j: ex.submit(_run, emu_env, _JOBS[j][0], _JOBS[j][1], **_JOBS[j][2]) for j in wanted}
    yield futs
    ex.shutdown(wait=True)


def test_emulated_kernels_equal_oracle(emu_jobs):
    p, rows = emu_jobs['quick'].result()
    assert rows, p.stderr[-2000:]
    bad = [r['case'] for r in rows if not r['ok']]
    assert not bad and p.returncode == 0, 'emulated HIP path differs from the oracle: %s\n%s' % (bad, p.stderr[-2000:])
    assert len(rows) >= 8


def test_emulated_kernels_do_not_depend_on_lane_order(emu_jobs):
    """Between two meeting points the emulator runs the lanes of a wavefront one after another; with MI_EMU_REVERSE they (and the waves of
    a workgroup) run in the opposite order.  A kernel with an unsynchronised LDS / global exchange between lanes gives different bytes."""
    p, rows = emu_jobs['quick_reverse'].result()
    bad = [r['case'] for r in rows if not r['ok']]
    assert rows and not bad and p.returncode == 0, 'lane-order dependence: %s\n%s' % (bad, p.stderr[-2000:])


def test_emulated_kernels_rect_partition_cases(emu_jobs):
    """Inputs on which PARTITION_HORZ / PARTITION_VERT of 8x8 nodes are chosen often (dev_rect.h: 8x4 / 4x8 blocks, 2:1 transforms), including the full
    mode set of speed 1 that keeps the one-candidate-per-wavefront path."""
    p, rows = emu_jobs['rect'].result()
    assert rows and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])


def test_emulated_kernels_64x64_level(emu_jobs):
    """dev_blk64.h without a GPU: pictures on which 64x64 blocks are chosen (asserted from the oracle's block map), bottom-up and top-down walkers, 4:4:4 with its
    four 32x32 chroma transform blocks per plane and 4:0:0; once more with the lanes and waves of the emulator running in reverse order."""
    for job in ('blk64', 'blk64_reverse'):
        p, rows = emu_jobs[job].result()
        assert len(rows) == 3 and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])
        assert sum(r['n64'] for r in rows) >= 2, rows


def test_emulated_batch_api_equals_oracle(emu_jobs):
    """The batch entry points under the emulator's thread pool: work lists that span several frames (colour and colour + alpha images, two block-size
    classes in one encode, top-down and bottom-up order, colour and alpha frames on different sides of the high-quality threshold) equal the oracle byte for byte."""
    p, rows = emu_jobs['batch'].result()
    assert len(rows) == 6 and all(r['ok'] for r in rows) and p.returncode == 0, (rows, p.stderr[-2000:])


def test_a_dependency_wait_that_never_ends_fails_the_encode_in_bounded_time(emu_jobs):
    """tile_search.h root_wait: bounded polls.  The emulator drops every root publish (MI_EMU_DROP_PUBLISH): the first wait of the frame runs into its bound and sets
    the frame's sticky error word, the other waits leave at their next check, the entropy stage fails every tile, the host returns MI_ENCODING_ERROR -- no hang,
    no stream whose reconstruction the search did not see."""
    p, rows = emu_jobs['giveup'].result()
    assert len(rows) == 1 and rows[0]['ok'] and p.returncode == 0, (rows, p.stderr[-2000:])
    assert 'ncod' in rows[0]['outcome'] or 'rror' in rows[0]['outcome'], rows


def test_two_distinct_emulated_devices(emu_jobs):
    """the multi-device fan-out (mi_ravif_encode_stream: one host thread per device, shared cursor, per-device tables / arenas / memory budgets) had only
    ever run with devices=[0, 0].  MI_EMU_DEVICES=2 gives the emulator two devices that differ in compute units and free memory and that refuse each other's memory
    (a copy or a kernel argument naming the other device's allocation aborts).  Both devices must do work; every file == oracle.  Unmeasured on real multi-GPU hardware."""
    p, rows = emu_jobs['twodev'].result()
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

