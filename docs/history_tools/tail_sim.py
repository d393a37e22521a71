"""CPU model of the persistent refill walk's lane occupancy: per-ray node-visit counts of incoherent rays in sponza_lod
(oracle, one ray per call), rays handed to 64-lane waves exactly like trace_refill (refill when >= 16 lanes idle, from a
shared queue), one node visit per lane per step.  How many wave-steps fall into the tail (queue drained), at what
occupancy, and what would merging the tail's waves 4 / 8 / all per CU save?"""
import sys, numpy as np
sys.path.insert(0, '/root/repo')
from aten_amd.scene import scenedefs
from aten_amd import layout as L
from oracle import orc
fs, cam = scenedefs.sponza_lod()
rng = np.random.default_rng(11)
n = 6000
rays = np.zeros(n, L.RAY)
rays["org"] = rng.uniform([-12, 0.2, -5], [12, 10, 5], (n, 3)).astype(np.float32)
d = rng.normal(size=(n, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=1, keepdims=True)
rays["dir"] = d
lens = np.zeros(n, np.int64)
for i in range(n):
    _, st = orc.trace_closest(fs, rays[i:i+1])
    lens[i] = st[0]
print("visits per ray: mean %.1f median %.0f p90 %.0f p99 %.0f max %d" % (lens.mean(), np.median(lens), np.percentile(lens, 90), np.percentile(lens, 99), lens.max()))

def simulate(n_rays, n_waves, merge_group=1, merge_at=16, seed=0):
    r = np.random.default_rng(seed)
    q = r.choice(lens, n_rays)
    qpos = 0
    rem = np.zeros((n_waves, 64), np.int64)
    steps_bulk = steps_tail = 0
    lanes_bulk = lanes_tail = 0
    alive = np.ones(n_waves, bool)
    while alive.any():
        live = rem > 0
        nlive = live.sum(1)
        # refill
        if qpos < n_rays:
            for w in np.nonzero(alive & (64 - nlive >= 16))[0]:
                idle = np.nonzero(~live[w])[0]
                k = min(len(idle), n_rays - qpos)
                rem[w, idle[:k]] = q[qpos:qpos + k]; qpos += k
                if qpos >= n_rays: break
            live = rem > 0; nlive = live.sum(1)
        drained = qpos >= n_rays
        if drained and merge_group > 1:
            # merge: within each group of merge_group waves, pack live lanes into as few waves as possible when a wave is below merge_at
            for g in range(0, n_waves, merge_group):
                ws = np.arange(g, min(g + merge_group, n_waves))
                if (nlive[ws] > 0).sum() > 1 and (nlive[ws][nlive[ws] > 0] <= merge_at).any():
                    vals = rem[ws][rem[ws] > 0]
                    rem[ws] = 0
                    flat = np.zeros(len(ws) * 64, np.int64); flat[:len(vals)] = vals
                    rem[ws] = flat.reshape(len(ws), 64)
            live = rem > 0; nlive = live.sum(1)
        alive = nlive > 0 if drained else np.ones(n_waves, bool)
        act = nlive > 0
        if drained: steps_tail += act.sum(); lanes_tail += nlive.sum()
        else: steps_bulk += act.sum(); lanes_bulk += nlive.sum()
        rem[live] -= 1
    return steps_bulk, lanes_bulk, steps_tail, lanes_tail

for n_rays, n_waves in ((1_900_000 // 256, 20), (260_000 // 256, 20)):   # one CU's share of a 1.9 M-ray launch / of an 8-way shard's launch
    for mg in (1, 4, 8, 20):
        sb, lb, st_, lt = simulate(n_rays, n_waves, mg)
        print("rays/CU %5d waves %d merge %2d: bulk steps %6d occ %.2f | tail steps %6d occ %.2f | total wave-steps %6d" % (n_rays, n_waves, mg, sb, lb / max(sb,1) / 64, st_, lt / max(st_,1) / 64, sb + st_))

# ---- variant: a drained wave with <= T live walks dumps them to a continuation queue; the dumped walks are packed 64 per
# wave and finished by a second launch (plain walk)
def simulate_dump(n_rays, n_waves, T, seed=0, sort=False):
    r = np.random.default_rng(seed)
    q = r.choice(lens, n_rays); qpos = 0
    rem = np.zeros((n_waves, 64), np.int64)
    steps_bulk = steps_tail = 0
    dumped = []
    alive = np.ones(n_waves, bool)
    while alive.any():
        live = rem > 0; nlive = live.sum(1)
        if qpos < n_rays:
            for w in np.nonzero(alive & (64 - nlive >= 16))[0]:
                idle = np.nonzero(~live[w])[0]
                k = min(len(idle), n_rays - qpos)
                rem[w, idle[:k]] = q[qpos:qpos + k]; qpos += k
                if qpos >= n_rays: break
            live = rem > 0; nlive = live.sum(1)
        drained = qpos >= n_rays
        if drained and T > 0:
            for w in np.nonzero((nlive > 0) & (nlive <= T))[0]:
                dumped.extend(rem[w][rem[w] > 0].tolist()); rem[w] = 0
            live = rem > 0; nlive = live.sum(1)
        alive = nlive > 0 if drained else np.ones(n_waves, bool)
        act = nlive > 0
        if drained: steps_tail += act.sum()
        else: steps_bulk += act.sum()
        rem[live] -= 1
    d = np.array(dumped, np.int64)
    if sort: d = np.sort(d)
    cont = 0
    for i in range(0, len(d), 64): cont += d[i:i+64].max()
    return steps_bulk, steps_tail, cont, len(d)
for n_rays, n_waves in ((7421, 20), (1015, 20)):
    for T in (0, 8, 16, 24, 32, 48):
        sb, st, ct, nd = simulate_dump(n_rays, n_waves, T)
        print("rays/CU %5d dump<=%2d: bulk %5d tail-in-kernel %5d continuation %5d (%d walks) total %5d" % (n_rays, T, sb, st, ct, nd, sb+st+ct))
