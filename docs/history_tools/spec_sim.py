"""CPU model of the persistent refill walk (trace_refill + walk_iteration, device/traverse.hpp) at wave level, fed with the
step-kind sequences of real incoherent rays: how many wave-level issue slots does a launch cost per ray with and without
SPECULATION past triangle leaves (a lane that reaches a leaf parks it and keeps stepping from the leaf's successor with
the old t_max; the parked leaf is resolved by the leaf step; an accepted hit rolls the lane back)?

The model charges a wave the instruction count of a block whenever at least one lane executes it (VALU + VMEM issue is
what binds the kernel): inner step C_INNER, leaf step C_LEAF, TLAS step C_TLAS, iteration
overhead C_ITER.   usage: python tools/spec_sim.py [sponza_lod|atrium] [n_rays]"""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from aten_amd.scene import scenedefs

scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_lod"
n_rays = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
fs, cam = getattr(scenedefs, scene)()
A = fs.arrays
lists = A["bvh_lists"]
tris = A["triangles"]; pos = A["vtx_pos"]
I, LEAF, TLAS = 0, 1, 2


def prep(nodes):
    return (nodes["boxmin"].astype(np.float64), nodes["boxmax"].astype(np.float64), nodes["hit"].astype(np.int64),
            nodes["miss"].astype(np.int64), nodes["f0"].astype(np.int64), nodes["f1"].astype(np.int64), nodes["f2"].view(np.uint32).astype(np.int64), nodes["f2"])


P = [prep(n) for n in lists]
tri_idx = np.stack([tris["idx"][:, 0], tris["idx"][:, 1], tris["idx"][:, 2]], 1) if tris["idx"].ndim == 2 else None
V = pos[:, :3].astype(np.float64) if pos.ndim == 2 else np.stack([pos["x"], pos["y"], pos["z"]], 1).astype(np.float64)


def walk(org, d):
    """Returns the ray's walk as a list of events: ('I', taken_hit) inner step; ('L', accepted) leaf; ('T',) TLAS leaf;
    ('E',) list end in BLAS (leave).  Closest-hit semantics of the reference (threaded_bvh_traverser.h)."""
    ev = []
    tmax = np.inf
    best = np.inf

    def run(k, o, dd, top):
        nonlocal tmax, best
        bmin, bmax, hit, miss, f0, f1, f2u, f2 = P[k]
        inv = 1.0 / (dd + 1e-6)
        oi = -o * inv
        n = 0
        while n >= 0:
            leaf = f0[n] >= 0 or f1[n] >= 0
            if leaf and top and f2[n] >= 0:
                ev.append(('T',))
                ex = int(f2u[n]) & 0x7fff      # ATN_EXID_MAIN: low 15 bits
                run(ex, o, dd, False)
                ev.append(('E',))
                n = hit[n]          # simplification: the model does not need the hit/miss exit distinction
                continue
            if leaf and f1[n] >= 0:
                t = int(f1[n])
                i0, i1, i2 = tri_idx[t]
                v0 = V[i0]; e1 = V[i1] - v0; e2 = V[i2] - v0
                r = o - v0
                u = np.cross(dd, e2); v = np.cross(r, e1)
                den = np.dot(u, e1)
                acc = False
                if den != 0:
                    invd = 1.0 / den
                    tt = np.dot(v, e2) * invd; b = np.dot(u, r) * invd; g = np.dot(v, dd) * invd
                    if 0 <= b <= 1 and 0 <= g <= 1 and b + g <= 1 and tt >= 0 and tt < best and tt > 1e-6:
                        best = tt; tmax = tt; acc = True
                ev.append(('L', acc))
                n = hit[n]
                continue
            if leaf:
                n = miss[n]; continue
            f = bmax[n] * inv + oi; nn = bmin[n] * inv + oi
            t1 = min(np.maximum(f, nn).min(), tmax); t0 = max(np.minimum(f, nn).max(), 1e-6)
            h = t0 <= t1
            ev.append(('I', h))
            n = hit[n] if h else miss[n]
    run(0, org, d, True)
    return ev


rng = np.random.default_rng(11)
bb = (V.min(0), V.max(0))
seqs = []
for i in range(n_rays):
    o = rng.uniform(bb[0] * 0.9 + bb[1] * 0.1, bb[1] * 0.9 + bb[0] * 0.1)
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    seqs.append(walk(o, d))
lens = np.array([len(s) for s in seqs])
nI = sum(1 for s in seqs for e in s if e[0] == 'I'); nL = sum(1 for s in seqs for e in s if e[0] == 'L')
nA = sum(1 for s in seqs for e in s if e[0] == 'L' and e[1]); nT = sum(1 for s in seqs for e in s if e[0] == 'T')
LL = sum(1 for s in seqs for a, b in zip(s, s[1:]) if a[0] == 'L' and b[0] == 'L')
print("%s: %d rays, events per ray mean %.1f median %.0f max %d; inner %.1f leaf %.2f (accepted %.2f = %.0f %%) tlas %.2f; leaf followed by leaf %.0f %%"
      % (scene, n_rays, lens.mean(), np.median(lens), lens.max(), nI / n_rays, nL / n_rays, nA / n_rays, 100.0 * nA / max(nL, 1), nT / n_rays, 100.0 * LL / max(nL, 1)))
import pickle
pickle.dump(seqs, open('/tmp/sim/seqs_%s.pkl' % scene, 'wb'))
