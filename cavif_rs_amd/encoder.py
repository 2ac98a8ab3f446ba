"""Host-side mirror of ravif::Encoder (ravif/src/av1encoder.rs:67-397) over the C ABI of libmi_avif.so.

Names, argument meaning and error behaviour follow the reference: builder methods `with_quality`,
`with_alpha_quality`, `with_speed`, `with_bit_depth`, `with_internal_color_model`, `with_num_threads`,
`with_alpha_color_mode`; entry points `encode_rgba`, `encode_rgb`, `encode_raw_planes_8_bit`,
`encode_raw_planes_10_bit`; result `EncodedImage{avif_file, color_byte_size, alpha_byte_size}`.
Out-of-range builder arguments raise (the Rust asserts at :117,146,159,188).
"""
import ctypes as C
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class AvifError(RuntimeError):
    """ravif::Error (ravif/src/error.rs:7-25) + argument/device errors."""
    NAMES = {1: 'TooFewPixels', 2: 'Unsupported', 3: 'EncodingError', 4: 'InvalidArgument', 5: 'NoDevice'}

    def __init__(self, code):
        super().__init__(self.NAMES.get(code, 'status %d' % code))
        self.code = code


class _Av1Config(C.Structure):
    _fields_ = [('width', C.c_uint32), ('height', C.c_uint32), ('bit_depth', C.c_uint8), ('quantizer', C.c_uint8),
                ('speed', C.c_uint8), ('chroma', C.c_uint8), ('pixel_range', C.c_uint8), ('threads', C.c_int32),
                ('has_color_desc', C.c_int8), ('matrix', C.c_uint8), ('transfer', C.c_uint8), ('primaries', C.c_uint8),
                ('part_min', C.c_uint8), ('part_max', C.c_uint8), ('complex_pred_modes', C.c_uint8), ('sgr_full', C.c_uint8),
                ('encode_bottomup', C.c_uint8), ('rdo_tx_decision', C.c_uint8), ('reduced_tx_set', C.c_uint8),
                ('fine_directional_intra', C.c_uint8), ('fast_deblock', C.c_uint8), ('lrf', C.c_uint8), ('cdef', C.c_uint8),
                ('inter_tx_split', C.c_uint8), ('tx_domain_rate', C.c_uint8), ('tx_domain_distortion', C.c_int8),
                ('min_tile_size', C.c_uint16), ('tiles_override', C.c_int32), ('device', C.c_int32), ('tune_psnr', C.c_uint8), ('rdo_passes', C.c_uint8)]


class _RavifEncoder(C.Structure):
    _fields_ = [('quality', C.c_float), ('alpha_quality', C.c_float), ('speed', C.c_uint8), ('color_model', C.c_uint8),
                ('depth', C.c_uint8), ('alpha_mode', C.c_uint8), ('threads', C.c_int32),
                ('exif', C.c_void_p), ('exif_len', C.c_size_t), ('device', C.c_int32), ('tiles_override', C.c_int32), ('rdo_passes', C.c_int32)]


class _EncodedImage(C.Structure):
    _fields_ = [('avif_file', C.POINTER(C.c_uint8)), ('avif_len', C.c_size_t), ('color_byte_size', C.c_size_t), ('alpha_byte_size', C.c_size_t)]


def library_path():
    return os.environ.get('MI_AVIF_LIB') or os.path.join(_HERE, 'libmi_avif.so')


def load_library():
    """Load libmi_avif.so; fails loudly when the HIP extension has not been built."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise ImportError('libmi_avif.so is missing: run `python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950)')
    L = C.CDLL(path)
    L.mi_version.restype = C.c_char_p
    L.mi_quality_to_quantizer.argtypes = [C.c_float]
    L.mi_av1_tweaks_from_preset.argtypes = [C.c_uint8, C.c_uint8, C.POINTER(_Av1Config)]
    L.mi_rgb_to_ycbcr.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.POINTER(C.c_uint16)]
    L.mi_av1_encode_planes.argtypes = [C.POINTER(_Av1Config), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t),
                                       C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_size_t), C.POINTER(C.POINTER(C.c_uint16))]
    L.mi_ravif_encoder_default.argtypes = [C.POINTER(_RavifEncoder)]
    for fn in (L.mi_ravif_encode_rgba, L.mi_ravif_encode_rgb):
        fn.argtypes = [C.POINTER(_RavifEncoder), C.c_void_p, C.c_uint32, C.c_uint32, C.c_size_t, C.POINTER(_EncodedImage)]
    L.mi_ravif_encode_raw_planes_8.argtypes = [C.POINTER(_RavifEncoder), C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint8, C.c_uint8, C.POINTER(_EncodedImage)]
    L.mi_ravif_encode_raw_planes_10.argtypes = L.mi_ravif_encode_raw_planes_8.argtypes
    L.mi_batch_create.argtypes = [C.POINTER(_RavifEncoder), C.c_int, C.c_uint32, C.c_uint32, C.c_int]
    L.mi_batch_create.restype = C.c_void_p
    L.mi_batch_upload.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]
    L.mi_batch_input.argtypes = [C.c_void_p, C.c_int]
    L.mi_batch_input.restype = C.c_void_p
    L.mi_batch_upload_async.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.mi_batch_set_count.argtypes = [C.c_void_p, C.c_int]
    L.mi_batch_encode.argtypes = [C.c_void_p]
    L.mi_batch_encode_async.argtypes = [C.c_void_p]
    L.mi_batch_wait.argtypes = [C.c_void_p]
    L.mi_batch_get.argtypes = [C.c_void_p, C.c_int, C.POINTER(_EncodedImage)]
    L.mi_batch_get_recon.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.POINTER(C.c_uint16))]
    L.mi_batch_stage_ms.argtypes = [C.c_void_p, C.c_int]
    L.mi_batch_stage_ms.restype = C.c_double
    L.mi_batch_num_tiles.argtypes = [C.c_void_p]
    L.mi_batch_tile_clocks.argtypes = [C.c_void_p, C.c_void_p]
    if hasattr(L, 'mi_batch_phase_profile'):
        L.mi_batch_phase_profile.argtypes = [C.c_void_p, C.c_void_p]
    L.mi_batch_destroy.argtypes = [C.c_void_p]
    L.mi_avif_serialize.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint8, C.c_uint8,
                                    C.c_int, C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8))]
    L.mi_avif_serialize.restype = C.c_size_t
    L.mi_free.argtypes = [C.c_void_p]
    _LIB = L
    return L


def device_count():
    return load_library().mi_device_count()


def quality_to_quantizer(quality):
    return load_library().mi_quality_to_quantizer(float(quality))


def tweaks_from_preset(speed, quantizer):
    c = _Av1Config()
    st = load_library().mi_av1_tweaks_from_preset(speed, quantizer, C.byref(c))
    if st:
        raise AvifError(st)
    return {k: getattr(c, k) for k in ('part_min', 'part_max', 'complex_pred_modes', 'sgr_full', 'encode_bottomup', 'rdo_tx_decision',
                                       'reduced_tx_set', 'fine_directional_intra', 'fast_deblock', 'lrf', 'cdef', 'inter_tx_split',
                                       'tx_domain_rate', 'min_tile_size')}


def rgb_to_ycbcr(rgb, depth):
    a = (C.c_uint8 * 3)(*rgb)
    o = (C.c_uint16 * 3)()
    load_library().mi_rgb_to_ycbcr(a, depth, o)
    return tuple(o)


class EncodedImage:
    """EncodedImage (ravif/src/av1encoder.rs:54-61)."""

    def __init__(self, avif_file, color_byte_size, alpha_byte_size):
        self.avif_file = avif_file
        self.color_byte_size = color_byte_size
        self.alpha_byte_size = alpha_byte_size


def _take(img):
    L = load_library()
    data = bytes(bytearray(img.avif_file[:img.avif_len]))
    L.mi_free(img.avif_file)
    return EncodedImage(data, img.color_byte_size, img.alpha_byte_size)


def encode_planes(planes, bit_depth=8, quantizer=121, speed=4, mono=False, matrix=6, device=0, tiles=0, want_recon=True, **over):
    """Level-1 entry (encode_to_av1, :749-771): planes = list of HxW arrays. Returns (obu bytes, [recon planes])."""
    L = load_library()
    h, w = planes[0].shape
    c = _Av1Config()
    c.width, c.height, c.bit_depth, c.quantizer, c.chroma, c.pixel_range = w, h, bit_depth, quantizer, int(mono), 1
    c.has_color_desc = 0 if mono else 1
    c.primaries, c.transfer, c.matrix = 1, 13, matrix
    st = L.mi_av1_tweaks_from_preset(speed, quantizer, C.byref(c))
    if st:
        raise AvifError(st)
    c.tiles_override, c.device = tiles, device
    for k, v in over.items():
        setattr(c, k, v)
    dt = np.uint8 if bit_depth == 8 else np.uint16
    arrs = [np.ascontiguousarray(p, dtype=dt) for p in planes]
    ptrs = (C.c_void_p * 3)(*[a.ctypes.data for a in arrs] + [None] * (3 - len(arrs)))
    strides = (C.c_size_t * 3)(*[a.strides[0] for a in arrs] + [0] * (3 - len(arrs)))
    obu = C.POINTER(C.c_uint8)()
    n = C.c_size_t()
    rec = (C.POINTER(C.c_uint16) * 3)()
    st = L.mi_av1_encode_planes(C.byref(c), ptrs, strides, C.byref(obu), C.byref(n), rec if want_recon else None)
    if st:
        raise AvifError(st)
    data = bytes(bytearray(obu[:n.value]))
    L.mi_free(obu)
    recon = []
    if want_recon:
        for i in range(len(arrs)):
            recon.append(np.ctypeslib.as_array(rec[i], shape=(h, w)).copy())
            L.mi_free(rec[i])
    return data, recon


class Encoder:
    """ravif::Encoder builder (ravif/src/av1encoder.rs:67-219). Defaults per Encoder::new (:88-102)."""

    def __init__(self):
        self.quality, self.alpha_quality, self.speed = 80.0, 80.0, 5
        self.color_model, self.depth, self.alpha_mode, self.threads = 0, 0, 1, None
        self.device, self.tiles_override = 0, 0
        self.rdo_passes = 1
        self.exif = None

    def _copy(self, **kw):
        e = Encoder()
        e.__dict__.update(self.__dict__)
        e.__dict__.update(kw)
        return e

    def with_quality(self, quality):                    # :116 assert!(quality >= 1. && quality <= 100.)
        assert 1.0 <= quality <= 100.0
        return self._copy(quality=float(quality))

    def with_alpha_quality(self, quality):              # :145
        assert 1.0 <= quality <= 100.0
        return self._copy(alpha_quality=float(quality))

    def with_speed(self, speed):                        # :158 assert!(speed >= 1 && speed <= 10)
        assert 1 <= speed <= 10
        return self._copy(speed=int(speed))

    def with_bit_depth(self, depth):                    # :128 BitDepth::{Eight,Ten,Auto}
        assert depth in (8, 10, 0, 'auto')
        return self._copy(depth=0 if depth == 'auto' else depth)

    def with_internal_color_model(self, model):         # :174 ColorModel::{YCbCr,RGB}
        assert model in ('ycbcr', 'rgb')
        return self._copy(color_model=1 if model == 'rgb' else 0)

    def with_num_threads(self, n):                      # :187 assert!(num_threads.is_none_or(|n| n > 0))
        assert n is None or n > 0
        return self._copy(threads=n)

    def with_alpha_color_mode(self, mode):              # :197
        modes = {'dirty': 0, 'clean': 1, 'premultiplied': 2}
        return self._copy(alpha_mode=modes[mode])

    def with_exif(self, exif_data):                     # :208 embeds the bytes as an Exif item (no TIFF-offset prefix expected)
        return self._copy(exif=bytes(exif_data))

    def with_device(self, device):
        return self._copy(device=int(device))

    def with_rdo_passes(self, n):                       # extension (not in ravif): 2 = second search priced against every tile's final CDFs of a first pass
        assert n in (1, 2)
        return self._copy(rdo_passes=int(n))

    def _c(self):
        e = _RavifEncoder()
        e.quality, e.alpha_quality, e.speed, e.color_model, e.depth, e.alpha_mode = self.quality, self.alpha_quality, self.speed, self.color_model, self.depth, self.alpha_mode
        e.threads = self.threads or 0
        e.device, e.tiles_override, e.rdo_passes = self.device, self.tiles_override, self.rdo_passes
        if self.exif:
            self._exif_buf = C.create_string_buffer(self.exif, len(self.exif))   # must outlive every call made with `e`
            e.exif, e.exif_len = C.cast(self._exif_buf, C.c_void_p), len(self.exif)
        return e

    def _encode(self, px, channels):
        L = load_library()
        a = np.ascontiguousarray(px, dtype=np.uint8)
        if a.ndim != 3 or a.shape[2] != channels:
            raise AvifError(4)
        h, w, _ = a.shape
        img = _EncodedImage()
        e = self._c()
        fn = L.mi_ravif_encode_rgba if channels == 4 else L.mi_ravif_encode_rgb
        st = fn(C.byref(e), a.ctypes.data, w, h, w, C.byref(img))
        if st:
            raise AvifError(st)
        return _take(img)

    def encode_rgba(self, rgba):                        # :243
        return self._encode(rgba, 4)

    def encode_rgb(self, rgb):                          # :318
        return self._encode(rgb, 3)

    def _raw(self, fn, dt, yuv, alpha, width, height, color_pixel_range, matrix_coefficients):
        L = load_library()
        a = np.ascontiguousarray(yuv, dtype=dt).reshape(-1)
        if a.size < width * height * 3:
            raise AvifError(1)                          # Error::TooFewPixels
        al = None
        if alpha is not None:
            al = np.ascontiguousarray(alpha, dtype=dt).reshape(-1)
            if al.size < width * height:
                raise AvifError(1)
        img = _EncodedImage()
        e = self._c()
        st = fn(C.byref(e), width, height, a.ctypes.data, al.ctypes.data if al is not None else None, color_pixel_range, matrix_coefficients, C.byref(img))
        if st:
            raise AvifError(st)
        return _take(img)

    def encode_raw_planes_8_bit(self, width, height, planes, alpha, color_pixel_range=1, matrix_coefficients=6):   # :366
        return self._raw(load_library().mi_ravif_encode_raw_planes_8, np.uint8, planes, alpha, width, height, color_pixel_range, matrix_coefficients)

    def encode_raw_planes_10_bit(self, width, height, planes, alpha, color_pixel_range=1, matrix_coefficients=6):  # :390
        return self._raw(load_library().mi_ravif_encode_raw_planes_10, np.uint16, planes, alpha, width, height, color_pixel_range, matrix_coefficients)


class _ImageDesc(C.Structure):
    _fields_ = [('pixels', C.c_void_p), ('width', C.c_uint32), ('height', C.c_uint32), ('stride_px', C.c_size_t), ('channels', C.c_int)]


def encode_many(encoder, images, devices=None):
    """mi_ravif_encode_batch: the reference's files.into_par_iter() (src/main.rs:223) over the node's GPUs.
    images: list of HxWx3 / HxWx4 uint8 arrays (shapes may differ).  Returns a list of EncodedImage."""
    L = load_library()
    L.mi_ravif_encode_batch.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    arrs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
    desc = (_ImageDesc * len(arrs))()
    for d, a in zip(desc, arrs):
        if a.ndim != 3 or a.shape[2] not in (3, 4):
            raise AvifError(4)
        d.pixels, d.width, d.height, d.stride_px, d.channels = a.ctypes.data, a.shape[1], a.shape[0], a.shape[1], a.shape[2]
    out = (_EncodedImage * len(arrs))()
    status = (C.c_int * len(arrs))()
    dev = (C.c_int * len(devices))(*devices) if devices else None
    e = encoder._c()
    st = L.mi_ravif_encode_batch(C.byref(e), len(arrs), desc, out, status, dev, len(devices) if devices else 0)
    res = [_take(o) if s == 0 else None for o, s in zip(out, status)]
    if st:
        raise AvifError(st)
    return res


class BatchEncoder:
    """Device-resident batch (the reference's files.into_par_iter(), src/main.rs:223): upload once, encode many."""

    def __init__(self, encoder, n_images, width, height, channels=3):
        self._L = load_library()
        self._encoder = encoder                      # keeps the Exif buffer referenced by the batch alive
        e = encoder._c()
        self._h = self._L.mi_batch_create(C.byref(e), n_images, width, height, channels)
        if not self._h:
            raise AvifError(5 if self._L.mi_device_count() <= encoder.device else 4)
        self.n, self.w, self.h, self.channels = n_images, width, height, channels

    def upload(self, index, pixels):
        a = np.ascontiguousarray(pixels, dtype=np.uint8)
        assert a.shape == (self.h, self.w, self.channels)
        st = self._L.mi_batch_upload(self._h, index, a.ctypes.data, self.w)
        if st:
            raise AvifError(st)

    def pinned_input(self, index):
        """numpy view of the batch's PINNED host staging of image `index`: fill it in place, then upload_async()."""
        ptr = self._L.mi_batch_input(self._h, index)
        if not ptr:
            raise AvifError(4)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint8)), shape=(self.h, self.w, self.channels))

    def upload_async(self, first=0, count=None):
        """enqueue pinned host -> HBM for a range of images on the batch's stream (returns at once)"""
        st = self._L.mi_batch_upload_async(self._h, first, self.n if count is None else count)
        if st:
            raise AvifError(st)

    def set_count(self, n_images):
        st = self._L.mi_batch_set_count(self._h, n_images)
        if st:
            raise AvifError(st)

    def encode(self):
        st = self._L.mi_batch_encode(self._h)
        if st:
            raise AvifError(st)

    def encode_async(self):
        st = self._L.mi_batch_encode_async(self._h)
        if st:
            raise AvifError(st)

    def wait(self):
        st = self._L.mi_batch_wait(self._h)
        if st:
            raise AvifError(st)

    def get(self, index):
        img = _EncodedImage()
        st = self._L.mi_batch_get(self._h, index, C.byref(img))
        if st:
            raise AvifError(st)
        return _take(img)

    def recon(self, index, alpha=False):
        rec = (C.POINTER(C.c_uint16) * 3)()
        st = self._L.mi_batch_get_recon(self._h, index, int(alpha), rec)
        if st:
            raise AvifError(st)
        out = []
        for i in range(3):
            if rec[i]:
                out.append(np.ctypeslib.as_array(rec[i], shape=(self.h, self.w)).copy())
                self._L.mi_free(rec[i])
        return out

    def stage_ms(self):
        names = ('front_end', 'tile_search', 'deblock', 'cdef', 'entropy', 'pack_d2h', 'host_assembly')
        return {n: self._L.mi_batch_stage_ms(self._h, i) for i, n in enumerate(names)}

    def num_tiles(self):
        return self._L.mi_batch_num_tiles(self._h)

    def tile_clocks(self):
        a = np.zeros((self.num_tiles(), 4), dtype=np.uint64)
        st = self._L.mi_batch_tile_clocks(self._h, a.ctypes.data)
        if st:
            raise AvifError(st)
        return a

    def phase_profile(self):
        a = np.zeros((max(self.num_tiles(), 2048), 4, 32), dtype=np.uint64)      # rows: tile jobs (K4 profile) or persistent workgroups (K1 profile); unused rows stay zero
        st = self._L.mi_batch_phase_profile(self._h, a.ctypes.data)
        if st:
            raise AvifError(st)
        return a

    def close(self):
        if self._h:
            self._L.mi_batch_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
