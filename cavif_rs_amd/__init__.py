"""cavif_rs_amd -- MI355X-native AV1 still-picture (AVIF) encode path behind the ravif::Encoder interface.

The package is a thin ctypes layer over libmi_avif.so (HIP kernels + C ABI, see include/mi_avif.h).
There is no CPU fallback: importing works anywhere, encoding requires a HIP device and the built library.
"""
from .encoder import (Encoder, EncodedImage, AvifError, BatchEncoder, quality_to_quantizer, tweaks_from_preset,
                      rgb_to_ycbcr, encode_planes, encode_many, library_path, load_library, device_count)

__all__ = ['Encoder', 'EncodedImage', 'AvifError', 'BatchEncoder', 'quality_to_quantizer', 'tweaks_from_preset',
           'rgb_to_ycbcr', 'encode_planes', 'encode_many', 'library_path', 'load_library', 'device_count']
