"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so) -- test infrastructure only."""
import ctypes as C, os, subprocess, numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_LIB = None

class Av1oConfig(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        'width', 'height', 'bit_depth', 'mono', 'quantizer', 'full_range',
        'has_color_desc', 'color_primaries', 'transfer', 'matrix', 'threads',
        'part_min', 'part_max', 'complex_modes', 'fine_directional', 'rdo_tx', 'reduced_tx_set',
        'fast_deblock', 'cdef', 'lrf', 'sgr_full', 'bottomup', 'tx_domain_rate', 'inter_tx_split',
        'min_tile_size', 'tiles_override', 'tune_psnr', 'rdo_passes')]

class Av1oResult(C.Structure):
    _fields_ = [('obu', C.POINTER(C.c_uint8)), ('obu_len', C.c_size_t),
                ('recon', C.POINTER(C.c_uint16) * 3), ('recon_stride', C.c_int),
                ('mi_cols', C.c_int), ('mi_rows', C.c_int), ('mi_stride', C.c_int),
                ('m_bsize', C.POINTER(C.c_uint8)), ('m_ymode', C.POINTER(C.c_uint8)), ('m_uvmode', C.POINTER(C.c_uint8)),
                ('m_skip', C.POINTER(C.c_uint8)), ('m_txtype', C.POINTER(C.c_uint8)),
                ('base_q_idx', C.c_int), ('tile_cols', C.c_int), ('tile_rows', C.c_int),
                ('total_sse', C.c_int64 * 3), ('lf_level', C.c_int * 4), ('seg_n', C.c_int), ('seg_qidx', C.c_int * 8)]

class RavifEncoder(C.Structure):
    _fields_ = [('quality', C.c_float), ('alpha_quality', C.c_float), ('speed', C.c_int), ('color_model', C.c_int),
                ('depth', C.c_int), ('alpha_mode', C.c_int), ('threads', C.c_int), ('tiles_override', C.c_int), ('rdo_passes', C.c_int)]

class RavifImage(C.Structure):
    _fields_ = [('avif', C.POINTER(C.c_uint8)), ('avif_len', C.c_size_t), ('color_byte_size', C.c_size_t), ('alpha_byte_size', C.c_size_t)]

def build():
    subprocess.check_call(['make', '-s', '-C', os.path.join(ROOT, 'oracle')])

def lib():
    global _LIB
    if _LIB is None:
        path = os.environ.get('MI_ORACLE_LIB') or os.path.join(ROOT, 'oracle', '_build', 'liboracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.av1o_quality_to_quantizer.argtypes = [C.c_float]; L.av1o_quality_to_quantizer.restype = C.c_int
        L.av1o_tweaks_from_preset.argtypes = [C.c_int, C.c_int, C.POINTER(Av1oConfig)]
        L.av1o_encode.argtypes = [C.POINTER(Av1oConfig), C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(Av1oResult)]
        L.av1o_free_result.argtypes = [C.POINTER(Av1oResult)]
        L.av1o_rgb_to_ycbcr.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint16)]
        L.av1o_to_ten.argtypes = [C.c_uint8]; L.av1o_to_ten.restype = C.c_uint16
        for fn in (L.ravif_oracle_encode_rgba, L.ravif_oracle_encode_rgb):
            fn.argtypes = [C.POINTER(RavifEncoder), C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(RavifImage)]
        L.av1o_avif_container.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t] + [C.c_int] * 9 + [C.POINTER(C.POINTER(C.c_uint8))]
        L.av1o_avif_container.restype = C.c_size_t
        L.av1o_free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB

def make_config(w, h, bit_depth=8, mono=False, quantizer=121, speed=4, matrix=6, full_range=1, tiles=0, **over):
    c = Av1oConfig()
    c.width, c.height, c.bit_depth, c.mono, c.quantizer, c.full_range = w, h, bit_depth, int(mono), quantizer, full_range
    c.has_color_desc = 0 if mono else 1
    c.color_primaries, c.transfer, c.matrix = 1, 13, matrix
    assert lib().av1o_tweaks_from_preset(speed, quantizer, C.byref(c)) == 0
    c.tiles_override = tiles
    for k, v in over.items():
        setattr(c, k, v)
    return c

def encode_planes(cfg, planes):
    """planes: list of HxW uint16 arrays. Returns dict(obu=bytes, recon=[arrays], maps...)."""
    L = lib()
    planes = [np.ascontiguousarray(p, dtype=np.uint16) for p in planes]
    ptrs = (C.c_void_p * 3)(*[p.ctypes.data for p in planes] + [None] * (3 - len(planes)))
    strides = (C.c_int * 3)(*[p.shape[1] for p in planes] + [0] * (3 - len(planes)))
    r = Av1oResult()
    st = L.av1o_encode(C.byref(cfg), ptrs, strides, C.byref(r))
    assert st == 0, st
    h, w = planes[0].shape
    out = dict(obu=bytes(bytearray(r.obu[:r.obu_len])),
               recon=[np.ctypeslib.as_array(r.recon[i], shape=(h, w)).copy() for i in range(len(planes))],
               base_q_idx=r.base_q_idx, tiles=(r.tile_cols, r.tile_rows), sse=[r.total_sse[i] for i in range(len(planes))], lf_level=[r.lf_level[i] for i in range(4)], seg_n=r.seg_n, seg_qidx=[r.seg_qidx[i] for i in range(8)])
    n = r.mi_stride * ((r.mi_rows + 15) // 16 * 16)
    for k in ('m_bsize', 'm_ymode', 'm_uvmode', 'm_skip', 'm_txtype'):
        a = np.ctypeslib.as_array(getattr(r, k), shape=(n,)).copy().reshape(-1, r.mi_stride)
        out[k] = a[:r.mi_rows, :r.mi_cols]
    L.av1o_free_result(C.byref(r))
    return out

def container(color, alpha, w, h, depth, mono_color=0, cp=1, tc=13, mc=6, full_range=1, premult=0):
    L = lib()
    outp = C.POINTER(C.c_uint8)()
    cb = (C.c_uint8 * len(color)).from_buffer_copy(color)
    ab = (C.c_uint8 * len(alpha)).from_buffer_copy(alpha) if alpha else None
    n = L.av1o_avif_container(cb, len(color), ab, len(alpha) if alpha else 0, w, h, depth, mono_color, cp, tc, mc, full_range, premult, C.byref(outp))
    data = bytes(bytearray(outp[:n]))
    L.av1o_free(outp)
    return data

def ravif_encode(pixels, quality=80., alpha_quality=80., speed=5, color_model=0, depth=0, alpha_mode=1, threads=0, tiles=0, rdo_passes=1):
    """pixels: HxWx3 or HxWx4 uint8. Returns (avif bytes, color_size, alpha_size)."""
    L = lib()
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    h, w, ch = px.shape
    e = RavifEncoder(quality, alpha_quality, speed, color_model, depth, alpha_mode, threads, tiles, rdo_passes)
    img = RavifImage()
    fn = L.ravif_oracle_encode_rgba if ch == 4 else L.ravif_oracle_encode_rgb
    st = fn(C.byref(e), px.ctypes.data, w, h, w, C.byref(img))
    assert st == 0, st
    data = bytes(bytearray(img.avif[:img.avif_len]))
    L.av1o_free(img.avif)
    return data, img.color_byte_size, img.alpha_byte_size
