"""Decode an AVIF file to raw planes with the libavif (+dav1d 1.5.3) that Pillow bundles -- the independent
conformance checker of SURVEY.md section 8(c)/Appendix A.  Struct layout hand-declared for libavif 1.4.1 x86-64."""
import ctypes as C, glob, os, numpy as np

class _RWData(C.Structure):
    _fields_ = [('data', C.c_void_p), ('size', C.c_size_t)]

class _AvifImageHead(C.Structure):
    _fields_ = [('width', C.c_uint32), ('height', C.c_uint32), ('depth', C.c_uint32),
                ('yuvFormat', C.c_int32), ('yuvRange', C.c_int32), ('yuvChromaSamplePosition', C.c_int32),
                ('yuvPlanes', C.c_void_p * 3), ('yuvRowBytes', C.c_uint32 * 3), ('imageOwnsYUVPlanes', C.c_int32),
                ('alphaPlane', C.c_void_p), ('alphaRowBytes', C.c_uint32), ('imageOwnsAlphaPlane', C.c_int32),
                ('alphaPremultiplied', C.c_int32), ('icc', _RWData),
                ('colorPrimaries', C.c_uint16), ('transferCharacteristics', C.c_uint16), ('matrixCoefficients', C.c_uint16)]

_lib = None
def _load():
    global _lib
    if _lib is None:
        import PIL
        cands = glob.glob(os.path.join(os.path.dirname(os.path.dirname(PIL.__file__)), 'pillow.libs', 'libavif-*.so*'))
        if not cands:
            raise RuntimeError('bundled libavif not found')
        L = C.CDLL(cands[0])
        L.avifDecoderCreate.restype = C.c_void_p
        L.avifImageCreateEmpty.restype = C.c_void_p
        L.avifDecoderReadMemory.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.avifDecoderReadMemory.restype = C.c_int
        L.avifResultToString.restype = C.c_char_p; L.avifResultToString.argtypes = [C.c_int]
        L.avifImageDestroy.argtypes = [C.c_void_p]; L.avifDecoderDestroy.argtypes = [C.c_void_p]
        _lib = L
    return _lib

def available():
    try:
        _load(); return True
    except Exception:
        return False

def decode(data):
    """-> dict(planes=[Y,U,V] (uint16 arrays; 1 plane if 4:0:0), alpha=array|None, depth, format, matrix, range)"""
    L = _load()
    dec = L.avifDecoderCreate(); img = L.avifImageCreateEmpty()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    r = L.avifDecoderReadMemory(dec, img, buf, len(data))
    if r != 0:
        msg = L.avifResultToString(r).decode()
        L.avifImageDestroy(img); L.avifDecoderDestroy(dec)
        raise ValueError('libavif: ' + msg)
    h = _AvifImageHead.from_address(img)
    bps = 2 if h.depth > 8 else 1
    def plane(ptr, rowbytes):
        raw = np.ctypeslib.as_array((C.c_uint8 * (rowbytes * h.height)).from_address(ptr)).reshape(h.height, rowbytes)
        raw = raw[:, :h.width * bps]
        return (raw.copy().view(np.uint16) if bps == 2 else raw.astype(np.uint16)).reshape(h.height, h.width)
    planes = []
    n = 1 if h.yuvFormat == 4 else 3
    assert h.yuvFormat in (1, 4), 'only 4:4:4 / 4:0:0 expected here'
    for i in range(n):
        planes.append(plane(h.yuvPlanes[i], h.yuvRowBytes[i]))
    alpha = plane(h.alphaPlane, h.alphaRowBytes) if h.alphaPlane else None
    out = dict(planes=planes, alpha=alpha, depth=h.depth, format=h.yuvFormat, matrix=h.matrixCoefficients,
               range=h.yuvRange, width=h.width, height=h.height)
    L.avifImageDestroy(img); L.avifDecoderDestroy(dec)
    return out
