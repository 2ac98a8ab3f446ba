#!/usr/bin/env python3
"""bench.py -- MPix/s of the AVIF still-image encode hot path (speed=4, quality=80, 1080p batch) on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 launched through
torch.distributed.run, one rank per GPU.  A step = one pass of the hot path (K0 front end -> activity mask -> K1 tile
search -> K2a/K2 deblock level search + filter -> K3 CDEF -> K5 loop restoration -> K4 entropy coding -> pack/D2H -> host
OBU+container assembly) over one batch of synthetic images that is ALREADY RESIDENT in HBM (RGB8, uploaded before the
timed region): that is `value`.  The same run then repeats the timed loop with the H2D of every batch (pinned host memory ->
HBM, async on the batch's stream, overlapped with the other slots' compute) inside the region: `value_pcie_inclusive`, the
"encode-only" clock of SURVEY.md 8(d) (RGBA8/RGB8 in pinned host memory -> .avif bytes in host memory).
Images shard across ranks with no data-path collective (weak scaling: --batch images per GPU); every resident batch slot
holds DIFFERENT images.  Rank 0 prints one JSON line; `roofline` is for the dominant kernel (tile search) from HIP events of
a launch with nothing else in flight, `cpu_baseline` is the scalar C oracle (a port, not the reference) on this box's host
cores, `cpu_baseline_standin` is libaom (through Pillow) at its matching speed on the same inputs -- both stand-ins: the
reference itself (rav1e) cannot be built in this image.  `--secondary` adds single-image latency lines for BASELINE
configs 2, 3 and 5; `--end-to-end N` adds the PNG-file -> .avif-file clock of the command line on N synthetic PNGs (256 by default at one GPU).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
HBM_COPY_PEAK_GBS = 6290.0     # the same guide's measured device-to-device copy rate (what a kernel can actually reach)
MANIFEST_IMAGES = 256          # tests/golden/bench_manifest.json: oracle sha256 of synth images 0..255 at the default workload
ALGO_BYTES_PER_PX = {10: 10.0, 8: 7.0}   # 4 B RGBA8 in + 3*s source samples read once


def hbm_traffic_bytes(kernel, cfg):
    """HBM-side bytes per launch of `kernel` from the committed PMC passes (profiles/hbm_counters.json), when they were
    taken on this very workload; None otherwise.  (2 * FETCH_SIZE + WRITE_SIZE) KB -- the gfx950 FETCH_SIZE correction
    of MI355X_MICROARCH.md; counters come from separate rocprofv3 --pmc passes, see tools/final_profile.sh.)"""
    try:
        with open(os.path.join(ROOT, 'profiles', 'hbm_counters.json')) as fh:
            d = json.load(fh)
        if any(d['config'].get(k_) != v_ for k_, v_ in cfg.items()):
            return None
        c = d['kernels'][kernel]
        return (2.0 * c['FETCH_SIZE_KB'] + c['WRITE_SIZE_KB']) * 1024.0
    except Exception:
        return None


VALU_CLOCK_GHZ = 2.4           # MI355X_MICROARCH.md: 256 CUs x 4 SIMDs at 2.4 GHz max clock; a SIMD issues at most one VALU instruction per cycle to one of its waves


def valu_roofline(kernel, cfg, launch_ms, pixels):
    """What actually bounds the tile search: VALU instruction issue.  From the committed SQ_* counter passes (profiles/valu_counters.json, tools/final_profile.sh; taken on
    this workload) and THIS run's launch duration: wave-level VALU instructions issued per second against what the chip's 1024 SIMDs can issue at four clocks per
    instruction, the share of lanes those instructions had active, and the VALU work per pixel.  None when the counters are for another workload."""
    try:
        with open(os.path.join(ROOT, 'profiles', 'valu_counters.json')) as fh:
            d = json.load(fh)
        if any(d['config'].get(k_) != v_ for k_, v_ in cfg.items()):
            return None
        c = d['kernels'][kernel]
        insts, thr = c.get('SQ_INSTS_VALU', c['SQ_ACTIVE_INST_VALU']), c['SQ_THREAD_CYCLES_VALU']
        peak = 256 * 4 * VALU_CLOCK_GHZ / 4.0                                            # G wave-instructions / s: 1024 SIMDs, one VALU instruction per 4 clocks each
        achieved = insts / (launch_ms / 1e3) / 1e9
        out = {"bound": "valu", "kernel": kernel, "achieved": round(achieved, 2), "peak": round(peak, 2), "unit": "G VALU wave-instructions/s",
               "frac": round(achieved / peak, 4), "lane_utilisation": round(thr / (64.0 * c['SQ_ACTIVE_INST_VALU']), 4),
               "valu_wave_instructions_per_launch": insts, "valu_wave_instructions_per_pixel": round(insts / pixels, 2),
               "counters_launch_ms": c.get('avg_launch_ms'), "launch_ms": round(launch_ms, 3),
               "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 4 clocks: on gfx950 the multiplies, compares, selects, min / max, left shifts, DPP, packed and dot forms and anything with an SGPR operand issue in ~4 clocks per wave, adds / logic / right shifts in ~2 (tools/probe/valu_rates.hip, profiles/r05_valu_rates.txt, profiles/r06_valu_rates.txt: float min / max / compare are slow too): frac 1.0 is reachable only by a kernel made of the first kind; lane_utilisation = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU); counters from profiles/valu_counters.json (own rocprofv3 --pmc passes), launch time from this run"}
        if 'SQ_WAIT_ANY' in c and 'SQ_WAVE_CYCLES' in c:
            out["wave_time_waiting"] = round(c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'], 4)
        if 'SQ_INSTS_SALU' in c:
            out["salu_wave_instructions_per_pixel"] = round(c['SQ_INSTS_SALU'] / pixels, 2)
        return out
    except Exception:
        return None


def _synth_cache_dir():
    """Per-user cache of the synthetic inputs, keyed by the generator's source: a changed cavif_rs_amd/synth.py (or somebody else's files
    under a shared /tmp) can never stand in for the images the sha256 manifest was made from."""
    import hashlib
    with open(os.path.join(ROOT, 'cavif_rs_amd', 'synth.py'), 'rb') as fh:
        gen = hashlib.sha256(fh.read()).hexdigest()[:16]
    base = os.environ.get('MI_SYNTH_CACHE') or os.path.join(os.environ.get('XDG_CACHE_HOME') or os.path.join(os.path.expanduser('~'), '.cache'), 'mi_synth_cache')
    d = os.path.join(base, gen)
    try:
        os.makedirs(d, mode=0o700, exist_ok=True)
    except OSError:
        d = os.path.join(tempfile.gettempdir(), 'mi_synth_cache_%d' % os.getuid(), gen)
        os.makedirs(d, mode=0o700, exist_ok=True)
    return d


def _synth_to_cache(job):
    """One synthetic image into the .npy cache (worker process) with its sha256 beside it: the generator is 0.6 s of numpy per 1080p image."""
    path, w, h, idx = job
    sys.path.insert(0, ROOT)
    import hashlib
    import numpy as np
    from cavif_rs_amd.synth import synth_image
    img = synth_image(w, h, index=idx)
    tmp = '%s.%d.tmp.npy' % (path, os.getpid())
    np.save(tmp, img)
    with open(tmp + '.sha', 'w') as fh:
        fh.write(hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest())
    os.replace(tmp + '.sha', path + '.sha')
    os.replace(tmp, path)
    return path


def _load_cached(path):
    """A cached image, or None when the file or its checksum is missing or does not match (it is then regenerated)."""
    import hashlib
    import numpy as np
    try:
        img = np.load(path)
        with open(path + '.sha') as fh:
            want = fh.read().strip()
        return img if hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == want else None
    except Exception:
        return None


def synth_images(w, h, indices):
    """The synthetic inputs `synth_image(w, h, index=i)`, generated on all host cores and kept as checksummed .npy files in a per-user cache
    directory named after the generator's source hash (_synth_cache_dir; $MI_SYNTH_CACHE overrides the base), so that repeated runs on one
    box do not regenerate them."""
    d = _synth_cache_dir()
    paths = {i: os.path.join(d, 'synth_%dx%d_%05d.npy' % (w, h, i)) for i in indices}
    out = {i: _load_cached(paths[i]) for i in indices}
    missing = [(paths[i], w, h, i) for i in indices if out[i] is None]
    if len(missing) > 2:
        import multiprocessing as mp
        with mp.get_context('spawn').Pool(max(1, min(os.cpu_count() or 1, 16, len(missing)))) as pool:
            pool.map(_synth_to_cache, missing)
    else:
        for job in missing:
            _synth_to_cache(job)
    for (path, _, _, i) in missing:
        out[i] = _load_cached(path)
        if out[i] is None:
            raise SystemExit('bench.py: could not produce %s' % path)
    return out


def check_identity(batches, image_index, B, cfg, manifest=None):
    """sha256 of every .avif the batch slots hold (their last encode) against the committed oracle manifest (or --manifest: same format, made by a test)."""
    import hashlib
    try:
        with open(manifest or os.path.join(ROOT, 'tests', 'golden', 'bench_manifest.json')) as fh:
            man = json.load(fh)
    except Exception as e:
        return {"status": "unchecked: no manifest (%s)" % e}
    if any(man['config'].get(k_) != v_ for k_, v_ in cfg.items()):
        return {"status": "unchecked: the manifest is for %s" % man['config']}
    checked = equal = 0
    bad = []
    for s_, bt in enumerate(batches):
        for i in range(B):
            idx = image_index(s_, i)
            if idx >= len(man['sha256']):
                continue
            checked += 1
            if hashlib.sha256(bt.get(i).avif_file).hexdigest() == man['sha256'][idx]:
                equal += 1
            elif len(bad) < 8:
                bad.append(idx)
    r = {"status": "every file of every slot vs the CPU oracle's sha256 (tests/golden/bench_manifest.json)", "checked": checked, "equal": equal}
    if bad:
        r["first_mismatches"] = bad
    return r


def _oracle_worker(jobs):
    """One worker process: its images first (the generator is numpy, not the encoder under test), then the encodes back to back; returns the wall-clock stamps."""
    sys.path.insert(0, ROOT)
    from cavif_rs_amd.synth import synth_image
    from tests.helpers import oracle
    oracle.lib()
    imgs = [synth_image(w, h, index=idx) for idx, w, h, _, _, _ in jobs]
    t0 = time.time(); n = 0
    for img, (idx, w, h, speed, quality, depth) in zip(imgs, jobs):
        data, cs, _ = oracle.ravif_encode(img, quality=quality, speed=speed, depth=depth)
        n += len(data)
    return t0, time.time(), n, len(jobs)


def _aom_worker(jobs):
    import io
    sys.path.insert(0, ROOT)
    from PIL import Image
    from cavif_rs_amd.synth import synth_image
    ims = [Image.fromarray(synth_image(w, h, index=idx), 'RGB') for idx, w, h, _, _, _ in jobs]
    t0 = time.time(); n = 0
    for im, (idx, w, h, speed, quality, depth) in zip(ims, jobs):
        buf = io.BytesIO()
        im.save(buf, format='AVIF', quality=int(quality), speed=speed, codec='aom', subsampling='4:4:4', max_threads=1)
        n += buf.tell()
    return t0, time.time(), n, len(jobs)


def host_cores():
    """The CPUs this process can actually use: its affinity mask bounded by the cgroup CPU quota (v2 cpu.max along the process's cgroup path, v1 cfs quota) -- what Rust's
    std::thread::available_parallelism(), and with it rayon::current_num_threads() of the reference, returns.  (The MI355X boxes of this pool show 256 logical CPUs and a
    quota of 16: one worker per logical CPU ran 25 x slower per image than sixteen workers.)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    quota = None
    try:
        path = ''
        with open('/proc/self/cgroup') as fh:
            for ln in fh:
                if ln.startswith('0::'):
                    path = ln.strip()[3:]
        d = os.path.normpath('/sys/fs/cgroup/' + path.lstrip('/'))
        while d.startswith('/sys/fs/cgroup'):
            f = os.path.join(d, 'cpu.max')
            if os.path.exists(f):
                lim, per = open(f).read().split()[:2]
                if lim != 'max' and int(per) > 0:
                    q = int(lim) // int(per)
                    quota = q if quota is None else min(quota, q)
            if d == '/sys/fs/cgroup':
                break
            d = os.path.dirname(d)
    except Exception:
        pass
    if quota is None:
        try:
            q, per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read()), int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0 and per > 0:
                quota = q // per
        except Exception:
            pass
    return max(1, min(n, quota)) if quota is not None else max(1, n)


def _pool_baseline(worker, w, h, speed, quality, depth, per_worker, kind, what):
    """Image-parallel over all the CPUs the host gives this process (host_cores(): affinity and cgroup quota -- what the reference's files.into_par_iter() fans out
    over, src/main.rs:223): one worker process per core,
    `per_worker` images each.  The clock runs from the first worker's first encode to the last worker's last (process start, imports and the synthetic image
    generator are outside it)."""
    import multiprocessing as mp
    cores = host_cores()
    chunks = [[(k * per_worker + i, w, h, speed, quality, depth) for i in range(per_worker)] for k in range(cores)]
    with mp.get_context('spawn').Pool(cores) as pool:
        res = pool.map(worker, chunks, chunksize=1)
    wall = max(r[1] for r in res) - min(r[0] for r in res)
    n = sum(r[3] for r in res)
    return {"value": round(n * w * h / 1e6 / wall, 4), "unit": "MPix/s", "cores": cores, "kind": kind,
            "sample": "%d x %dx%d synthetic images, speed %d q%g depth %d, %s, one process per usable CPU (affinity and cgroup quota) x %d images each, %.1f s from the first encode's start to the last one's end (%.1f s mean per image, %.0f bytes mean)"
                      % (n, w, h, speed, quality, depth, what, per_worker, wall, sum(r[1] - r[0] for r in res) / n, sum(r[2] for r in res) / n)}


def cpu_baseline(w, h, speed, quality, depth):
    """Oracle (kind 'port') on the host cores: one image per worker process, bounded sample."""
    try:
        from tests.helpers import oracle
        oracle.lib()
    except Exception as e:      # oracle not built: report nothing rather than a fake number
        return {"value": None, "unit": "MPix/s", "cores": 0, "kind": "port", "sample": "oracle unavailable: %s" % e}
    return _pool_baseline(_oracle_worker, w, h, speed, quality, depth, 2, "port", "oracle/ scalar C restatement of THIS encoder (not rav1e)")


def cpu_standin(w, h, speed, quality, depth):
    """libaom through Pillow's bundled libavif: a different AV1 encoder at a comparable speed setting (SURVEY 8(d)(ii) stand-in)."""
    try:
        from PIL import features
        import PIL._avif as _avif
        if not _avif.encoder_codec_available('aom'):
            raise RuntimeError('Pillow has no aom encoder')
        r = _pool_baseline(_aom_worker, w, h, speed, quality, 8, 2, "stand-in", "libaom via Pillow (8-bit 4:4:4, single-threaded per image; NOT the reference)")
        r["note"] = "a different encoder (libaom, not rav1e) at its own speed %d; comparable settings, not comparable output" % speed
        return r
    except Exception as e:
        return {"value": None, "unit": "MPix/s", "cores": 0, "kind": "stand-in", "sample": "unavailable: %s" % e}


def single_image_line(m, name, w, h, alpha, index, speed, quality, alpha_quality, depth, device, reps=2):
    from cavif_rs_amd.synth import synth_image
    img = synth_image(w, h, index=index, alpha=alpha)
    enc = m.Encoder().with_quality(quality).with_alpha_quality(alpha_quality).with_speed(speed).with_bit_depth(depth).with_device(device)
    bt = m.BatchEncoder(enc, 1, w, h, channels=4 if alpha else 3)
    bt.pinned_input(0)[...] = img
    best = None
    for _ in range(reps + 1):                                   # first pass warms up
        t = time.perf_counter()
        bt.upload_async(0, 1); bt.encode_async(); bt.wait()
        dt = time.perf_counter() - t
        best = dt if best is None or dt < best else best
    st = bt.stage_ms()
    out = {"workload": name, "latency_ms": round(best * 1e3, 2), "MPix_per_s": round(w * h / 1e6 / best, 2), "tiles": bt.num_tiles(),
           "bytes": len(bt.get(0).avif_file), "stage_ms": {k_: round(v_, 2) for k_, v_ in st.items()}}
    bt.close()
    return out


def threads_line(m, T, imgs, B, w, h, speed, quality, depth, device, default_tiles=0, steps=3):
    """The same batch with the tile target bounded by T threads (ravif `with_num_threads(T)` = `cavif -jT`; with T = this box's host cores: what
    the reference's `threads = None` resolves to through rayon::current_num_threads(), ravif/src/av1encoder.rs:665-668).  One batch slot, `steps` timed steps;
    the first eight files are checked against the oracle's sha256 in scripts/parity_manifest.json when it holds that T (keys cfg4s/...@jT)."""
    import hashlib
    enc = m.Encoder().with_quality(quality).with_speed(speed).with_bit_depth(depth).with_device(device).with_num_threads(T)
    bt = m.BatchEncoder(enc, B, w, h, channels=3)
    for i in range(B):
        bt.pinned_input(i)[...] = imgs[i]
    bt.upload_async(0, B); bt.encode_async(); bt.wait()           # warm-up
    t = time.perf_counter()
    k1 = []
    for _ in range(steps):
        bt.encode_async(); bt.wait(); k1.append(bt.stage_ms()['tile_search'])
    dt = (time.perf_counter() - t) / steps
    out = {"threads": T, "tiles_per_image": bt.num_tiles() // B, "ms_per_step": round(dt * 1e3, 3), "MPix_per_s": round(B * w * h / 1e6 / dt, 3),
           "tile_search_ms": round(sum(k1) / len(k1), 3), "slots": 1,
           "what": "the default batch with with_num_threads(%d) (cavif -j%d): tile target min(T, w*h/min_tile_size^2); one batch slot, so compare with a --pipeline 1 run of the default line" % (T, T)}
    try:
        with open(os.path.join(ROOT, 'scripts', 'parity_manifest.json')) as fh:
            man = json.load(fh)['files']
        checked = equal = 0
        for i in range(min(B, 8)):
            e = man.get('cfg4s/synth_%04d.avif@j%d' % (i, T))
            if e is None:
                continue
            checked += 1
            equal += hashlib.sha256(bt.get(i).avif_file).hexdigest() == e['sha256']
        out["output_identity"] = {"checked": checked, "equal": equal, "status": "vs the CPU oracle's sha256 for -j%d (scripts/parity_manifest.json)" % T if checked else "unchecked: no oracle entries for T=%d in scripts/parity_manifest.json" % T}
        if not checked:
            # T at or above the uncapped tile target asks for the same tiles as `threads = None`: the default manifest applies (the bytes do not depend on T beyond the tile split)
            with open(os.path.join(ROOT, 'tests', 'golden', 'bench_manifest.json')) as fh:
                dm = json.load(fh)
            if default_tiles and bt.num_tiles() // B == default_tiles and dm['config'] == {"width": w, "height": h, "speed": speed, "quality": quality, "bit_depth": depth}:
                eq = sum(hashlib.sha256(bt.get(i).avif_file).hexdigest() == dm['sha256'][i] for i in range(min(B, len(dm['sha256']))))
                out["output_identity"] = {"checked": min(B, len(dm['sha256'])), "equal": int(eq), "status": "T = %d does not bound the tile target here (%d tiles per image, as with threads = None): checked against tests/golden/bench_manifest.json" % (T, default_tiles)}
    except Exception as e:
        out["output_identity"] = {"status": "unchecked: %s" % e}
    bt.close()
    return out


def end_to_end(n_files, w, h, speed, quality, depth, encode_only_s=None):
    """PNG files -> .avif files through the command line (cavif_mi), the clock comparable with `cavif` itself."""
    from scripts.gen_synth_png import write_png
    from cavif_rs_amd.synth import synth_image
    cli = os.path.join(ROOT, 'cavif_rs_amd', 'cavif_mi')
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'in')); os.makedirs(os.path.join(d, 'out'))
        imgs = synth_images(w, h, list(range(n_files)))
        for i in range(n_files):
            write_png(os.path.join(d, 'in', 'synth_%04d.png' % i), imgs[i])
        del imgs
        files = sorted(os.path.join(d, 'in', f) for f in os.listdir(os.path.join(d, 'in')))
        cmd = [cli, '-s', str(speed), '-Q', '%g' % quality, '--depth', str(depth), '-f', '-q', '-o', os.path.join(d, 'out')] + files
        runs = []
        # the command as a user runs it (one process: the device teardown is inside the measured time), then with the opt-in exit hand-off (CAVIF_MI_BACKGROUND_EXIT=1:
        # the work runs in a child that reports through a pipe once every file is written; the kernel's unpin / free / queue teardown of that child is off the caller's clock)
        for env in ({}, {'CAVIF_MI_BACKGROUND_EXIT': '1'}):
            t = time.perf_counter()
            r = subprocess.run(cmd, capture_output=True, env=dict(os.environ, CAVIF_MI_TIMING='1', **env))
            dt = time.perf_counter() - t
            runs.append((dt, r, len(os.listdir(os.path.join(d, 'out')))))
            for f in os.listdir(os.path.join(d, 'out')): os.unlink(os.path.join(d, 'out', f))
    (dt, r, n_out), (dt_bg, r_bg, n_bg) = runs
    out = {"files": n_files, "ok": r.returncode == 0 and n_out == n_files and r_bg.returncode == 0 and n_bg == n_files,
           "phases": [l for l in r.stderr.decode().splitlines() if l.startswith('[timing]') and 'unix time' not in l], "seconds": round(dt, 3), "MPix_per_s": round(n_files * w * h / 1e6 / dt, 2),
           "seconds_background_exit": round(dt_bg, 3), "phases_background_exit": [l for l in r_bg.stderr.decode().splitlines() if l.startswith('[timing]') and 'unix time' not in l],
           "what": "cavif_mi -s%d -Q%g --depth %d -o out/ in/*.png (no -j: the tile target is bounded by this host's logical cores, like the reference's `threads: None`): PNG decode on the host cores + "
                   "RGBA8 upload + encode + file writes, process start AND device teardown included; the clock stops when the command returns. seconds_background_exit: the opt-in "
                   "CAVIF_MI_BACKGROUND_EXIT=1 mode, in which the command returns when every file is complete on disk and the worker's device teardown goes on unattended" % (speed, quality, depth)}
    if encode_only_s:
        out["encode_only_seconds"] = round(encode_only_s, 3); out["vs_encode_only"] = round(dt / encode_only_s, 3); out["vs_encode_only_background_exit"] = round(dt_bg / encode_only_s, 3)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=32, help='images per GPU per step (BASELINE config 4: 256 images / 8 GPUs)')
    ap.add_argument('--width', type=int, default=1920)
    ap.add_argument('--height', type=int, default=1080)
    ap.add_argument('--speed', type=int, default=4)
    ap.add_argument('--quality', type=float, default=80.0)
    ap.add_argument('--depth', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-identity-check', action='store_true')
    ap.add_argument('--manifest', default=None, help='sha256 manifest to check the outputs against (default: tests/golden/bench_manifest.json, the default workload)')
    ap.add_argument('--no-pcie-loop', action='store_true', help='skip the second timed loop (H2D inside the region)')
    ap.add_argument('--secondary', action='store_true', help='also time BASELINE configs 2, 3 and 5 (single images; config 5 takes a while)')
    ap.add_argument('--end-to-end', type=int, default=-1, metavar='N', help='PNG files -> .avif files through the cavif_mi command line on N synthetic PNGs (default: 256 at N=1 GPU -- BASELINE config 4 is a batch of 256 files --, 0 = skip)')
    ap.add_argument('--threads', type=int, default=0, help='ravif with_num_threads / cavif -j: T bounds the tile target (av1encoder.rs:665-668); 0 = unspecified (None): uncapped on a GPU')
    ap.add_argument('--no-threads-line', action='store_true', help='skip the secondary line at T = host cores (what `cavif -j0` would ask for on this box)')
    ap.add_argument('--pipeline', type=int, default=4, help='resident batch slots driven in rotation (one batch entropy-codes and filters while the others search). Measured on resident inputs, 20 steps (profiles/r06s_pipeline_sweep.txt): 1 slot 495.5, 2 slots 482.6, 3 slots 483.3, 4 slots 497.9 ... 502.7 MPix/s; at the end of the round (profiles/r06Y_pipeline_sweep.txt): 1 / 2 / 3 / 4 / 5 / 6 / 8 slots 537 / 519 / 525 / 533 ... 543 / 534 / 538 / 540 -- two or three slots interleave one search with the other slots\' filter / entropy kernels badly. The product stream path (mi_ravif_encode_stream, MI_STREAM_SLOTS_DEFAULT = 2) keeps two objects per image shape because on a 256-file job every further object costs 0.07 s of allocation, more than its overlap returns (profiles/r05zk_e2e_knobs.txt); its clock is the end_to_end line')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # Not under a launcher: become one.  `python bench.py --gpus N` alone must measure N GPUs, one rank per GPU (the driver's own launch line is the
        # same command under torch.distributed.run and takes the other branch).  MI_BENCH_SHARE_DEVICES=1 lets ranks share devices (CPU tests against
        # the emulator library: one emulated device).
        import socket
        import cavif_rs_amd as m0
        ndev0 = m0.device_count()
        if ndev0 < args.gpus and os.environ.get('MI_BENCH_SHARE_DEVICES') != '1':
            raise SystemExit('bench.py: --gpus %d asked for, %d HIP device(s) visible' % (args.gpus, ndev0))
        with socket.socket() as so:
            so.bind(('127.0.0.1', 0)); port = so.getsockname()[1]
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != max(1, args.gpus):
        raise SystemExit('bench.py: --gpus %d but the launcher started %d rank(s)' % (args.gpus, world))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    dist = None
    if world > 1:
        # the path shards by image: the only cross-rank traffic is the barrier and the max-time reduction,
        # so the CPU-side gloo group is enough (no RCCL collective exists on this data path).
        import torch.distributed as dist_mod
        dist = dist_mod
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(backend='gloo', rank=rank, world_size=world)

    import numpy as np
    import cavif_rs_amd as m
    from cavif_rs_amd.synth import synth_image

    ndev = m.device_count()
    if ndev < 1:
        raise SystemExit('bench.py: no HIP device visible; the HIP path is mandatory (no CPU fallback)')
    if ndev < world and os.environ.get('MI_BENCH_SHARE_DEVICES') != '1':
        raise SystemExit('bench.py: %d ranks but %d HIP device(s) visible: one rank per GPU' % (world, ndev))
    device = local_rank % ndev
    enc = m.Encoder().with_quality(args.quality).with_speed(args.speed).with_bit_depth(args.depth).with_device(device)
    if args.threads > 0:
        enc = enc.with_num_threads(args.threads)
    w, h, B = args.width, args.height, args.batch
    depth_q = max(1, min(args.pipeline, args.steps))         # a slot that no timed step would use is not created (its outputs could not be checked either)
    batches = [m.BatchEncoder(enc, B, w, h, channels=3) for _ in range(depth_q)]
    batch = batches[0]
    first = None
    # slot s of rank r holds images (r * slots + s) * B ... + B - 1: every slot (and every rank) encodes different pictures.
    # The pictures are written into the batches' pinned host staging once; H2D happens per step (second loop) or here (first loop).
    def image_index(s_, i):
        return ((rank * depth_q + s_) * B + i) % MANIFEST_IMAGES
    imgs = synth_images(w, h, sorted({image_index(s_, i) for s_ in range(depth_q) for i in range(B)}))
    for s_, bt in enumerate(batches):
        for i in range(B):
            img = imgs[image_index(s_, i)]
            if s_ == 0 and i == 0:
                first = img
            bt.pinned_input(i)[...] = img
        bt.upload_async(0, B)
    # setup, not warm-up: every slot runs once before anything is timed, so that the one-time work of a batch object's first encode (work-list and snapshot-pool
    # allocations, kernel attributes) never lands inside the timed region when --warmup is smaller than the number of slots (the default run: 2 < 4)
    for bt in batches:
        bt.encode_async()
    for bt in batches:
        bt.wait()

    def barrier():
        if dist is not None:
            dist.barrier()

    def run_steps(n, with_h2d):
        """n steps; step k runs on slot k % depth_q; a slot is waited for right before it is reused and at the end."""
        stats = []
        inflight = []
        for k_ in range(n):
            bt = batches[k_ % depth_q]
            if len(inflight) == depth_q:
                old = inflight.pop(0); old.wait(); stats.append(old.stage_ms())
            if with_h2d:
                bt.upload_async(0, B)    # pinned host -> HBM on the slot's stream, ahead of its front end; overlaps the other slots' kernels
            bt.encode_async(); inflight.append(bt)
        for old in inflight:
            old.wait(); stats.append(old.stage_ms())   # returns after the stream is drained and the .avif bytes are on the host
        return stats

    def timed(with_h2d):
        run_steps(args.warmup, with_h2d)
        barrier()
        t0 = time.perf_counter()
        stats = run_steps(args.steps, with_h2d)
        own_done[0] = time.perf_counter()
        barrier()
        elapsed = time.perf_counter() - t0
        per_rank = [elapsed]
        if dist is not None:
            import torch
            # the clock the contract asks for: barrier, K steps, barrier, MAX over ranks.  Each rank's own time up to its last wait (before the closing
            # barrier) is gathered too, so that a straggler is visible in the line.
            tt = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt[0])
            mine = torch.tensor([own_done[0] - t0], dtype=torch.float64)
            allr = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
            dist.all_gather(allr, mine)
            per_rank = [float(x[0]) for x in allr]
        return elapsed, stats, per_rank

    own_done = [0.0]
    elapsed, stats, per_rank = timed(False)                      # `value`: inputs resident in HBM
    elapsed_pcie = None
    if not args.no_pcie_loop:
        elapsed_pcie, _, _ = timed(True)                            # `value_pcie_inclusive`: H2D from pinned host memory inside the region
    search_ms, stage_acc = [], {}
    for st in stats:
        search_ms.append(st['tile_search'])
        for k_, v_ in st.items():
            stage_acc[k_] = stage_acc.get(k_, 0.0) + v_

    # one more step with nothing else in flight (outside the timed region): the tile search's launch duration without the
    # stretching that overlapping launches of the other batch slots cause.  This is the kernel's busy time (it agrees with
    # the rocprofv3 kernel trace, profiles/), and it is what roofline.achieved / frac are computed from.
    batches[0].encode_async(); batches[0].wait()
    isolated_k1_ms = batches[0].stage_ms()['tile_search']
    # every file of every slot of this rank against the oracle's sha256 manifest (tests/golden/bench_manifest.json); no oracle run here
    identity = None
    if not args.no_identity_check:
        identity = check_identity(batches, image_index, B, {"width": w, "height": h, "speed": args.speed, "quality": args.quality, "bit_depth": args.depth}, args.manifest)
        if dist is not None:
            import torch
            tt = torch.tensor([identity.get("checked", 0), identity.get("equal", 0)], dtype=torch.int64)
            dist.all_reduce(tt, op=dist.ReduceOp.SUM)
            if "checked" in identity:
                identity["checked"], identity["equal"] = int(tt[0]), int(tt[1])
    if rank == 0:
        total_px = world * B * w * h * args.steps
        value = total_px / 1e6 / elapsed
        k1_overlapped = sum(search_ms) / len(search_ms)                  # ms per launch inside the timed region (slots overlap)
        algo = ALGO_BYTES_PER_PX.get(args.depth, 10.0) * B * w * h        # algorithmic HBM-read bytes per launch
        achieved = algo / (isolated_k1_ms / 1e3) / 1e9
        out = {
            "metric": "MPix/s encoded at speed=4 q=80, 1080p batch; bit-exact vs CPU oracle",
            "value": round(value, 3), "unit": "MPix/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u16 samples / i32 transform / i64 RD cost", "data": "synthetic",
            "config": {"workload": "batch of %d synthetic %dx%d RGB8 images per GPU, speed=%d quality=%g depth=%d, 4:4:4 BT.601, %d tiles per step per GPU"
                                   % (B, w, h, args.speed, args.quality, args.depth, batch.num_tiles()) + (", T=%d" % args.threads if args.threads > 0 else ", T unspecified"),
                       "images_per_gpu": B, "width": w, "height": h, "speed": args.speed, "quality": args.quality, "bit_depth": args.depth,
                       "threads": args.threads,
                       "tile_target": ("T = %d threads (with_num_threads / cavif -j%d): target min(T, w*h/min_tile_size^2) -> %d tiles per image" % (args.threads, args.threads, batch.num_tiles() // B)) if args.threads > 0
                                      else "T = threads unspecified (ravif None): target w*h/min_tile_size^2 uncapped -> %d tiles per image" % (batch.num_tiles() // B),
                       "tools": "partition 4..16, 13 modes + angle deltas, tx-type + tx-size RDO (TX_MODE_SELECT), CfL, Tune::Psychovisual, deblock level search, CDEF search, sgrproj loop restoration (reduced sets)",
                       "parallelism": "images sharded across %d GPU(s), no collective; %d resident batch slot(s) per GPU, each holding different images, driven in rotation (the product's stream path keeps 2 per image shape: allocation time on a file job, see end_to_end; on resident inputs 1 / 2 / 3 / 4 slots measure 495 / 483 / 483 / 498-503 MPix/s, profiles/r06s_pipeline_sweep.txt)" % (world, depth_q)},
            "roofline": {"bound": "hbm", "kernel": "tile_search_kernel", "achieved": round(achieved, 4), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 8), "frac_of_measured_copy_peak": round(achieved / HBM_COPY_PEAK_GBS, 8),
                         "traffic": hbm_traffic_bytes("tile_search_kernel", {"images_per_gpu": B, "width": w, "height": h, "speed": args.speed,
                                                                             "quality": args.quality, "bit_depth": args.depth}),
                         "traffic_note": "bytes per launch, (2*FETCH_SIZE+WRITE_SIZE) from profiles/hbm_counters.json (separate rocprofv3 --pmc passes); far above the algorithmic bytes: the search's own trial commits, the partition walker's area snapshots, partial-line coefficient writes and register spills (profiles/r06_kernel_resources.txt: 544 B of scratch per lane in the kernel, 80 ... 272 B of it in the block searches since the paired chains of round 6 -- profiles/r06u_k1_traffic_by_variant.txt attributes a third of the traffic to them), served by L2 / Infinity Cache; not what limits the kernel -- see roofline_valu; TCP / TCC request, hit and stall counters of the same run: profiles/r06_final_pmc_summary.json",
                         "frac_of_step": round(algo / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 8),
                         "frac_note": "frac = algorithmic bytes / the dominant kernel's launch time; frac_of_step = the same bytes / the driver-timed step (every kernel of the step: the 4 B/px RGB read belongs to the front-end kernel, the 6 B/px plane reads to the tile search)",
                         "algorithmic_bytes_per_launch": algo, "avg_launch_ms": round(isolated_k1_ms, 3),
                         "launch_ms_note": "HIP events on the batch stream around the launch, one step with nothing else in flight (= rocprofv3 kernel-trace average)",
                         "overlapped_launch_ms": round(k1_overlapped, 3)},
            "roofline_valu": valu_roofline("tile_search_kernel", {"images_per_gpu": B, "width": w, "height": h, "speed": args.speed, "quality": args.quality, "bit_depth": args.depth}, isolated_k1_ms, B * w * h),
            "ms_per_step_per_rank": {"min": round(min(per_rank) / args.steps * 1e3, 3), "max": round(max(per_rank) / args.steps * 1e3, 3), "ranks": len(per_rank),
                                     "note": "each rank's own clock from the opening barrier to its last wait; ms_per_step is the MAX over ranks around both barriers"},
            "stage_ms_per_step": {k_: round(v_ / args.steps, 3) for k_, v_ in stage_acc.items()},
        }
        if elapsed_pcie is not None:
            out["value_pcie_inclusive"] = round(total_px / 1e6 / elapsed_pcie, 3)
            out["pcie_note"] = "same loop with every batch's H2D (pinned host -> HBM, %.1f MB per step, async on the slot's stream) inside the timed region" % (B * w * h * 3 / 1e6)
        if not args.no_identity_check:
            out["output_identity"] = identity
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, h, args.speed, args.quality, args.depth)
            out["cpu_baseline_standin"] = cpu_standin(w, h, args.speed, args.quality, args.depth)
    tiles_per_image = batch.num_tiles() // B
    for bt in batches:
        bt.close()
    if rank == 0:
        if world == 1 and not args.no_threads_line and B <= MANIFEST_IMAGES:
            T = host_cores()
            timgs = synth_images(w, h, list(range(B)))
            out["threads_host_cores"] = threads_line(m, T, timgs, B, w, h, args.speed, args.quality, args.depth, device, default_tiles=tiles_per_image)
            if T != 16:       # a host of 16 threads (what round 4's CPU baseline ran on): the regime where T does bound the tile target -- 16 tiles per 1080p image, 512 per launch
                out["threads_16"] = threads_line(m, 16, timgs, B, w, h, args.speed, args.quality, args.depth, device, default_tiles=tiles_per_image)
            del timgs
        if args.secondary:
            aq = min((args.quality + 100.0) / 2.0, args.quality + args.quality / 4.0 + 2.0)
            out["secondary"] = [
                single_image_line(m, "config 2: 1 x 1920x1080 RGB, speed 4, q80, 10-bit", 1920, 1080, False, 0, 4, 80.0, aq, 10, device),
                single_image_line(m, "config 3: 1 x 4096x4096 RGBA (alpha plane = second frame), speed 4, q80", 4096, 4096, True, 3, 4, 80.0, aq, 10, device),
                single_image_line(m, "config 5: 1 x 7680x4320 RGB, speed 1, q80, 10-bit (reference asks for <= 7 tiles -> 8)", 7680, 4320, False, 5, 1, 80.0, aq, 10, device, reps=1),
            ]
        n_e2e = args.end_to_end if args.end_to_end >= 0 else (256 if world == 1 else 0)
        if n_e2e:
            out["end_to_end"] = end_to_end(n_e2e, w, h, args.speed, args.quality, args.depth, encode_only_s=n_e2e / float(args.batch) * out["ms_per_step"] / 1e3)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
