"""Would XCD-affine ray queues help a tree that does not fit a 4 MiB per-XCD L2?  A cache model before the
kernel work: incoherent rays on the atrium (250 K triangles, ~17 MB of records), the byte addresses of the records each ray's
walk reads, 8 LRU caches of 4 MiB with 128-byte lines; rays are dealt to the XCDs either at random (today) or by the cell of
their origin in a 2 x 2 x 2 split of the scene box, and advance in lock step in groups of `inflight` rays per XCD.
usage: python tools/xcd_sim.py [n_rays]"""
import sys
from collections import OrderedDict
import numpy as np
sys.path.insert(0, '/root/repo')
from aten_amd.scene import scenedefs

n_rays = int(sys.argv[1]) if len(sys.argv) > 1 else 40000
fs, cam = scenedefs.atrium()
A = fs.arrays
lists = A["bvh_lists"]
tris = A["triangles"]; pos = A["vtx_pos"]
V = pos[:, :3].astype(np.float64)
tri_idx = tris["idx"]
# byte offsets of the device records: lists 1.., then list 0; inner 32 B, triangle leaf 48 B (device/scene_dev.hpp)
base = {}
off = 0
P = {}
for k in list(range(1, len(lists))) + [0]:
    n = lists[k]
    leaf_tri = (n["f1"] >= 0) & ~((k == 0) & (n["f2"] >= 0))
    size = np.where(leaf_tri, 48, 32)
    o = off + np.concatenate([[0], np.cumsum(size)[:-1]])
    off += int(size.sum())
    P[k] = (n["boxmin"].astype(np.float64), n["boxmax"].astype(np.float64), n["hit"].astype(np.int64), n["miss"].astype(np.int64),
            n["f0"].astype(np.int64), n["f1"].astype(np.int64), n["f2"].view(np.uint32).astype(np.int64), n["f2"], o.astype(np.int64), size)
print("node image %.1f MB, %d lists" % (off / 1e6, len(lists)))


def walk(o, d):
    addr = []
    tmax = np.inf

    def run(k, top):
        nonlocal tmax
        bmin, bmax, hit, miss, f0, f1, f2u, f2, offs, size = P[k]
        inv = 1.0 / (d + 1e-6); oi = -o * inv
        n = 0
        while n >= 0:
            addr.append(offs[n]); 
            if size[n] == 48: addr.append(offs[n] + 32)
            leaf = f0[n] >= 0 or f1[n] >= 0
            if leaf and top and f2[n] >= 0:
                run(int(f2u[n]) & 0x7fff, False)
                n = hit[n]; continue
            if leaf and f1[n] >= 0:
                i0, i1, i2 = tri_idx[int(f1[n])]
                v0 = V[i0]; e1 = V[i1] - v0; e2 = V[i2] - v0
                r = o - v0; u = np.cross(d, e2); v = np.cross(r, e1); den = u @ e1
                if den != 0:
                    iv = 1.0 / den; t = (v @ e2) * iv; b = (u @ r) * iv; g = (v @ d) * iv
                    if 0 <= b <= 1 and 0 <= g <= 1 and b + g <= 1 and 1e-6 < t < tmax: tmax = t
                n = hit[n]; continue
            if leaf: n = miss[n]; continue
            f = bmax[n] * inv + oi; nn = bmin[n] * inv + oi
            h = max(np.minimum(f, nn).max(), 1e-6) <= min(np.maximum(f, nn).min(), tmax)
            n = hit[n] if h else miss[n]
    run(0, True)
    return np.asarray(addr, np.int64) >> 7       # 128-byte lines


rng = np.random.default_rng(5)
lo, hi = V.min(0), V.max(0)
# secondary rays start ON surfaces: random triangle (area-blind: good enough), random point, random direction
t = rng.integers(0, len(tri_idx), n_rays)
w = rng.dirichlet([1, 1, 1], n_rays)
org = (V[tri_idx[t, 0]] * w[:, :1] + V[tri_idx[t, 1]] * w[:, 1:2] + V[tri_idx[t, 2]] * w[:, 2:3])
dirs = rng.normal(size=(n_rays, 3)); dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
org = org + 1e-3 * dirs
lines = [walk(org[i], dirs[i]) for i in range(n_rays)]
print("rays %d, record reads per ray %.1f" % (n_rays, np.mean([len(l) for l in lines])))
mid = (lo + hi) / 2
cell = ((org[:, 0] > mid[0]).astype(int) | ((org[:, 1] > mid[1]).astype(int) << 1) | ((org[:, 2] > mid[2]).astype(int) << 2))
print("rays per origin cell:", np.bincount(cell, minlength=8))


def simulate(assign, cache_lines=4 * 1024 * 1024 // 128, inflight=2048):
    """assign[i] = XCD of ray i.  Each XCD runs its rays `inflight` at a time in lock step (one record read per ray per round)."""
    hits = misses = 0
    for x in range(8):
        mine = [lines[i] for i in np.nonzero(assign == x)[0]]
        cache = OrderedDict()
        for g in range(0, len(mine), inflight):
            grp = mine[g:g + inflight]
            for step in range(max(len(l) for l in grp)):
                for l in grp:
                    if step < len(l):
                        a = int(l[step])
                        if a in cache:
                            cache.move_to_end(a); hits += 1
                        else:
                            misses += 1; cache[a] = 1
                            if len(cache) > cache_lines: cache.popitem(last=False)
    return hits / (hits + misses)


for inflight in (2048, 8192):
    r_rand = simulate(rng.integers(0, 8, n_rays), inflight=inflight)
    r_cell = simulate(cell, inflight=inflight)
    # 8 cells by k-means-free alternative: octant of the DIRECTION
    octant = ((dirs[:, 0] > 0).astype(int) | ((dirs[:, 1] > 0).astype(int) << 1) | ((dirs[:, 2] > 0).astype(int) << 2))
    r_dir = simulate(octant, inflight=inflight)
    print("in flight per XCD %5d: L2 hit rate  random %.3f | origin cell %.3f | direction octant %.3f" % (inflight, r_rand, r_cell, r_dir))
