"""SIMT emulator for the HIP sources (TEST INFRASTRUCTURE): see tests/emu/include/hip/hip_runtime.h."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(HERE, '_build', 'libmi_avif_emu.so')



def build(force=False):
    """g++ over the UNCHANGED product sources (cavif_rs_amd/csrc/mi_avif.hip) with tests/emu/include shadowing <hip/hip_runtime.h>."""
    csrc = os.path.join(ROOT, 'cavif_rs_amd', 'csrc')
    srcs = [os.path.join(csrc, f) for f in os.listdir(csrc)] + [os.path.join(HERE, 'emu_runtime.cpp'), os.path.join(HERE, 'include', 'hip', 'hip_runtime.h'),
                                                               os.path.join(ROOT, 'include', 'mi_avif.h')]
    lib = LIB
    if not force and os.path.exists(lib) and all(os.path.getmtime(s) <= os.path.getmtime(lib) for s in srcs):
        return lib
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    subprocess.check_call(['g++'] + ['-O2', '-g', '-rdynamic', '-fno-extern-tls-init', '-std=c++17', '-ffp-contract=off', '-fPIC', '-shared', '-w',
                           '-I', os.path.join(HERE, 'include'), '-I', os.path.join(ROOT, 'include'), '-x', 'c++',
                           os.path.join(csrc, 'mi_avif.hip'), os.path.join(HERE, 'emu_runtime.cpp'), '-o', lib, '-lz', '-lpthread', '-ldl'])
    return lib


def env():
    e = dict(os.environ)
    e['MI_AVIF_LIB'] = build()
    return e
