"""The command line's end-to-end clock on N synthetic 1080p PNG files under a list of environment variants (the GPU box):
tools/e2e_knobs.py N REPS 'NAME:VAR=V,VAR=V' ...   -> per run: wall clock, process start, HIP runtime up, first run enqueued, encoded, written, exit."""
import os, sys, subprocess, tempfile, time
sys.path.insert(0, '.')
import bench
from scripts.gen_synth_png import write_png
def main():
    n, reps = int(sys.argv[1]), int(sys.argv[2])
    variants = []
    for a in sys.argv[3:]:
        name, _, kv = a.partition(':')
        variants.append((name, dict(x.split('=', 1) for x in kv.split(',') if x)))
    cli = os.path.join(bench.ROOT, 'cavif_rs_amd', 'cavif_mi')
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'in')); os.makedirs(os.path.join(d, 'out'))
        imgs = bench.synth_images(1920, 1080, list(range(n)))
        for i in range(n): write_png(os.path.join(d, 'in', 'synth_%04d.png' % i), imgs[i])
        files = sorted(os.path.join(d, 'in', f) for f in os.listdir(os.path.join(d, 'in')))
        for rep in range(reps):
            for name, env in variants:
                for f in os.listdir(os.path.join(d, 'out')): os.unlink(os.path.join(d, 'out', f))
                e = dict(os.environ, CAVIF_MI_TIMING='1', MI_AVIF_TIMING='1', **env)
                if 'E2E_LIB_DIR' in e: e['LD_LIBRARY_PATH'] = e['E2E_LIB_DIR'] + ':' + e.get('LD_LIBRARY_PATH', '')
                t = time.perf_counter(); w0 = time.time()
                r = subprocess.run([cli, '-s', '4', '-Q', '80', '--depth', '10', '-f', '-q', '-o', os.path.join(d, 'out')] + files, capture_output=True, env=e)
                dt = time.perf_counter() - t; w1 = time.time()
                n_out = len(os.listdir(os.path.join(d, 'out')))
                ph = {}
                for l in r.stderr.decode().splitlines():
                    w = l.split()
                    if l.startswith('[timing] main entered'): ph['start'] = float(w[4].rstrip(',')) - w0; ph['exit'] = w1 - float(w[7])
                    elif l.startswith('[timing]'): ph[w[1] if w[1] != 'HIP' else 'hip'] = float(w[-2])
                    elif 'enqueued' in l and 'first' not in ph: ph['first'] = float(w[1]) / 1e3
                    elif 'worker done' in l: ph['worker'] = float(w[1]) / 1e3
                print('%-24s wall %.3f s  rc %d files %d | start %.3f  HIP up %.3f  first run enqueued +%.3f  worker done +%.3f  encoded %.3f  written %.3f  exit %.3f' % (
                    name, dt, r.returncode, n_out, ph.get('start', -1), ph.get('hip', -1), ph.get('first', -1), ph.get('worker', -1), ph.get('encoded', -1), ph.get('written', -1), ph.get('exit', -1)), flush=True)
                if ph.get('worker', 0) > 1.8: print('\n'.join('    ' + l[:200] for l in r.stderr.decode().splitlines() if 'done' in l or 'ready' in l or 'batch_create' in l), flush=True)
                time.sleep(float(os.environ.get('E2E_GAP', '0')))
if __name__ == '__main__':
    main()
