"""PNG loader throughput per host core (mi_png_decode_rgba on synthetic 1080p PNGs written like bench.py --end-to-end writes them): the host budget of the file fan-out
(one MI355X takes ~240 files/s).  Usage: python tools/loader_rate.py"""
import sys, time, ctypes as C, os, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cavif_rs_amd as m
from cavif_rs_amd.synth import synth_image
from scripts.gen_synth_png import write_png
L = m.load_library()
L.mi_png_decode_rgba.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
L.mi_free.argtypes = [C.c_void_p]
d = tempfile.mkdtemp()
datas = []
for i in range(4):
    p = os.path.join(d, 'a%d.png' % i); write_png(p, synth_image(1920, 1080, index=i)); datas.append(open(p, 'rb').read())
print('png bytes', [len(x) for x in datas])
n = 0; t = time.time()
while time.time() - t < 8:
    for x in datas:
        out = C.POINTER(C.c_uint8)(); w = C.c_uint32(); h = C.c_uint32()
        st = L.mi_png_decode_rgba(x, len(x), C.byref(out), C.byref(w), C.byref(h)); assert st == 0
        L.mi_free(out); n += 1
dt = time.time() - t
print('%d decodes in %.2f s: %.1f files/s per core, %.1f ms per 1080p file, %.1f MPix/s per core' % (n, dt, n / dt, 1e3 * dt / n, n * 1920 * 1080 / 1e6 / dt))
