"""End-to-end timeline of the command line on N synthetic 1080p PNG files (the GPU box): the worker's own log (MI_AVIF_TIMING) next to the phase times."""
import os, sys, subprocess, tempfile, time
sys.path.insert(0, '.')
import bench
from scripts.gen_synth_png import write_png
def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    cli = os.path.join(bench.ROOT, 'cavif_rs_amd', 'cavif_mi')
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, 'in')); os.makedirs(os.path.join(d, 'out'))
        imgs = bench.synth_images(1920, 1080, list(range(n)))
        for i in range(n): write_png(os.path.join(d, 'in', 'synth_%04d.png' % i), imgs[i])
        files = sorted(os.path.join(d, 'in', f) for f in os.listdir(os.path.join(d, 'in')))
        for rep in range(2):
            t = time.perf_counter(); w0 = time.time()
            r = subprocess.run([cli, '-s', '4', '-Q', '80', '--depth', '10', '-f', '-q', '-o', os.path.join(d, 'out')] + files, capture_output=True, env=dict(os.environ, CAVIF_MI_TIMING='1', MI_AVIF_TIMING='1', LD_LIBRARY_PATH=os.environ.get('E2E_LIB_DIR', '') + ':' + os.environ.get('LD_LIBRARY_PATH', '')))      # E2E_LIB_DIR: a directory with an experiment's libmi_avif.so
            w1 = time.time()
            print('run %d: %.3f s' % (rep, time.perf_counter() - t))
            for l in r.stderr.decode().splitlines():
                if l.startswith('[timing] main entered'):
                    a, b = float(l.split()[4].rstrip(',')), float(l.split()[7])
                    print('  spawn -> main %.3f s, main -> leaving %.3f s, leaving -> parent has the exit status %.3f s' % (a - w0, b - a, w1 - b))
            print(r.stderr.decode())


if __name__ == '__main__':
    main()
