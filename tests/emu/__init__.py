"""SIMT emulator for the HIP sources (TEST INFRASTRUCTURE): see tests/emu/include/hip/hip_runtime.h."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, '_build', 'libmi_avif_emu.so')
LIB_RECT = os.path.join(HERE, '_build', 'libmi_avif_emu_rect.so')       # the same sources with -DMI_RECT_PART=1 (groundwork, off in the product)


LIB_QUEUE = os.path.join(HERE, '_build', 'libmi_avif_emu_queue.so')     # -DMI_K1_QUEUE_KERNEL=1: the tile search as a work queue (run with MI_K1_QUEUE=1)


LIB_PIPE = os.path.join(HERE, '_build', 'libmi_avif_emu_pipe.so')       # -DMI_K4_PIPE=1: entropy coder as a walker wave + a range-coder wave per tile
LIB_PIPE3 = os.path.join(HERE, '_build', 'libmi_avif_emu_pipe3.so')     # -DMI_K4_PIPE=2: walker wave | four CDF-adapter waves | range-coder wave per tile
LIB_PIPEK = os.path.join(HERE, '_build', 'libmi_avif_emu_pipek.so')     # -DMI_K4_PIPE=3: the same three stages as three kernels, the record stream in HBM
LIB_DIET = os.path.join(HERE, '_build', 'libmi_avif_emu_diet.so')       # -DMI_K1_LDS_DIET=1: the tile search in 32 KB of LDS per workgroup


def build(force=False, rect=False, queue=False, pipe=False, diet=False):
    """g++ over the UNCHANGED product sources (cavif_rs_amd/csrc/mi_avif.hip) with tests/emu/include shadowing <hip/hip_runtime.h>."""
    csrc = os.path.join(ROOT, 'cavif_rs_amd', 'csrc')
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, 'emu_runtime.cpp'), os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'),
                                                               os.path.join(ROOT, 'include', 'mi_avif.h')]
    lib = LIB_RECT if rect else (LIB_QUEUE if queue else (({2: LIB_PIPE3, 3: LIB_PIPEK}.get(int(pipe), LIB_PIPE)) if pipe else (LIB_DIET if diet else LIB)))
    if not force and os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(['g++'] + (['-DMI_RECT_PART=1'] if rect else []) + (['-DMI_K1_QUEUE_KERNEL=1'] if queue else []) + (['-DMI_K4_PIPE=%d' % int(pipe)] if pipe else []) + (['-DMI_K1_LDS_DIET=1'] if diet else []) + ['-O2', '-g', '-rdynamic', '-fno-extern-tls-init', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-w',
                           '-I', os.path.join(HERE, 'include'), '-I', os.path.join(ROOT, 'include'), '-x', 'c++',
                           os.path.join(csrc, 'mi_avif.hip'), os.path.join(HERE, 'emu_runtime.cpp'), '-o', lib, '-lz', '-lpthread', '-ldl'])
    return lib


def env():
    e = dict(os.environ)
    e['MI_AVIF_LIB'] = build()
    return e
