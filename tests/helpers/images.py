"""Small deterministic test images shared by the CPU and GPU tests."""
import numpy as np


def planes(h, w, seed=0, bd=8, mono=False):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = [(x * 2 + y) % 256, (y * 3) % 256, ((x + y) // 2) % 256]
    out = []
    for b in base:
        v = b + rng.integers(-6, 7, size=(h, w)) + 40 * np.sin(x / 9.0 + seed) + 30 * np.cos(y / 7.0)
        v[h // 4:h // 2, w // 3:w // 2] = 200
        v = np.clip(v, 0, 255).astype(np.uint16)
        if bd == 10:
            v = (v << 2) | (v >> 6)
        out.append(v)
    return out[:1] if mono else out


def rgba_gradient(w=256, h=200):
    """ravif/src/lib.rs:45-51 encode8_with_alpha input: r=x, g=y, b=255, a=x+y (wrapping u8)."""
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x % 256, y % 256, np.full_like(x, 255), (x + y) % 256], -1).astype(np.uint8)


def rgba_opaque(w=129, h=101):
    """ravif/src/lib.rs:73-79 encode8_opaque input: (255, 100+x, y, 255) with u8 wrap."""
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([np.full_like(x, 255), (100 + x) % 256, y % 256, np.full_like(x, 255)], -1).astype(np.uint8)


def rgba_noisy(w=256, h=200):
    """ravif/src/lib.rs:123-125 encode8_cleans_alpha input."""
    y, x = np.mgrid[0:h, 0:w]
    a = np.clip(((x + y) & 0x7F).astype(int) - 100, 0, 255)          # (u8 & 0x7F).saturating_sub(100)
    return np.stack([(((x // 5 + y) & 0xF) << 4) & 255, (7 * x + y // 2) & 255, (x * y) & 3, a], -1).astype(np.uint8)
