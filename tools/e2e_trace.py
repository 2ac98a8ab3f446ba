import os, subprocess, tempfile, time, sys
sys.path.insert(0, ".")
from scripts.gen_synth_png import write_png
from cavif_rs_amd.synth import synth_image
d = tempfile.mkdtemp()
N = int(sys.argv[1])
for i in range(N): write_png(os.path.join(d, "s%03d.png" % i), synth_image(1920,1080,index=i))
files = sorted(os.path.join(d,f) for f in os.listdir(d))
for rep in range(2):
    t = time.perf_counter()
    r = subprocess.run(["cavif_rs_amd/cavif_mi","-f","-q"]+files, capture_output=True, env=dict(os.environ, CAVIF_MI_TIMING="1", MI_AVIF_TIMING="1"))
    dt = time.perf_counter() - t
    print("wall %.3f s -> %.1f MPix/s" % (dt, N*1920*1080/1e6/dt))
print(r.stderr.decode()[-3500:])
