"""Deterministic synthetic benchmark images: integer-only generator,
PRNG = splitmix64 seeded 0xCA71F00D + image_index, so that any implementation produces the same bytes.

pix(x,y,c) = clamp8( G_c(x,y) + S_c(x,y) + N_c(x,y) ) overlaid with 16 axis-aligned solid rectangles.
G = diagonal linear gradient, S = three integer-table sinusoids (periods 64/23/7 px, amplitudes 40/20/10),
N = uniform noise in [-8, 8], R = rectangles (hard edges for the directional predictors).
"""
import numpy as np

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _mix(z):
    with np.errstate(over='ignore'):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


def splitmix64(seed, n, offset=0):
    """n outputs of splitmix64 started at `seed`, skipping `offset` outputs."""
    with np.errstate(over='ignore'):
        idx = np.arange(1 + offset, n + 1 + offset, dtype=np.uint64)
        return _mix(np.uint64(seed) + idx * _GAMMA)


def _sin_table(period, amp):
    k = np.arange(period)
    return np.round(amp * np.sin(2.0 * np.pi * k / period)).astype(np.int64)


def synth_image(width, height, index=0, alpha=False):
    """-> uint8 array (H, W, 3) or (H, W, 4) when alpha=True (radial ramp alpha)."""
    seed = (0xCA71F00D + index) & 0xFFFFFFFFFFFFFFFF
    y, x = np.mgrid[0:height, 0:width].astype(np.int64)
    out = np.empty((height, width, 4 if alpha else 3), dtype=np.uint8)
    hdr = splitmix64(seed, 16 * 7 + 16)                 # rectangle parameters + per-channel phases
    noise = splitmix64(seed, width * height * 3, offset=1024).reshape(height, width, 3)
    t64, t23, t7 = _sin_table(64, 40), _sin_table(23, 20), _sin_table(7, 10)
    for c in range(3):
        ph = int(hdr[112 + c] % np.uint64(64))
        g = ((x * (c + 1) + y * (3 - c)) * 255) // max(1, (width * (c + 1) + height * (3 - c)))
        s = t64[(x + ph) % 64] + t23[(y + 2 * ph) % 23] + t7[(x + y + ph) % 7]
        n = (noise[:, :, c] % np.uint64(17)).astype(np.int64) - 8
        out[:, :, c] = np.clip(g + s + n, 0, 255).astype(np.uint8)
    for r in range(16):
        p = hdr[r * 7:(r + 1) * 7]
        x0 = int(p[0] % np.uint64(width)); y0 = int(p[1] % np.uint64(height))
        rw = 8 + int(p[2] % np.uint64(max(9, width // 6))); rh = 8 + int(p[3] % np.uint64(max(9, height // 6)))
        out[y0:y0 + rh, x0:x0 + rw, 0] = int(p[4] & np.uint64(255))
        out[y0:y0 + rh, x0:x0 + rw, 1] = int(p[5] & np.uint64(255))
        out[y0:y0 + rh, x0:x0 + rw, 2] = int(p[6] & np.uint64(255))
    if alpha:
        cx, cy = (width - 1) / 2.0, (height - 1) / 2.0
        rr = np.sqrt((x - cx) ** 2 + (y - cy) ** 2)
        r_in, r_out = 0.6 * (width / 2.0), np.sqrt(cx * cx + cy * cy)
        a = np.clip((r_out - rr) / max(r_out - r_in, 1e-9), 0.0, 1.0) * 255.0
        out[:, :, 3] = np.where(rr < r_in, 255, np.round(a)).astype(np.uint8)
    return out
