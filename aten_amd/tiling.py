"""Screen-space sharding used for multi-GPU rendering: the host-side mirror of slot_to_pixel() in
csrc/device/kernels.hpp.  The image is cut into 8x8 pixel tiles; tile t (row-major) belongs to
rank t % world; a rank's local slot s covers local tile s // 64, pixel (s % 8, (s % 64) // 8)."""
import numpy as np


def slots_per_rank(w, h, world):
    n_tiles = ((w + 7) // 8) * ((h + 7) // 8)
    return ((n_tiles + world - 1) // world) * 64


def slot_pixels(w, h, rank, world):
    tx, ty = (w + 7) // 8, (h + 7) // 8
    s = np.arange(slots_per_rank(w, h, world))
    tile = (s >> 6) * world + rank
    in_tile = s & 63
    x = (tile % tx) * 8 + (in_tile & 7)
    y = (tile // tx) * 8 + (in_tile >> 3)
    valid = (tile < tx * ty) & (x < w) & (y < h)
    return x, y, valid


def extract_tiles(full, rank, world):
    """full [h, w, 4] -> this rank's tile buffer [slots, 4] (zeros for padding slots)."""
    h, w = full.shape[:2]
    x, y, valid = slot_pixels(w, h, rank, world)
    out = np.zeros((len(x), full.shape[2]), full.dtype)
    out[valid] = full[y[valid], x[valid]]
    return out


def assemble_tiles(gathered, w, h, world):
    """gathered [world, slots, 4] -> full [h, w, 4]."""
    out = np.zeros((h, w, gathered.shape[-1]), gathered.dtype)
    for r in range(world):
        x, y, valid = slot_pixels(w, h, r, world)
        out[y[valid], x[valid]] = gathered[r][valid]
    return out
