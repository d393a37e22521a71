"""Model, second pass: K parked slots, delayed leaf step (run it only when enough lanes need it)."""
import sys, pickle
import numpy as np
scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_lod"
seqs = pickle.load(open('/tmp/sim/seqs_%s.pkl' % scene, 'rb'))
def enc(s):
    m = {'I': 0, 'T': 3, 'E': 4}
    return np.array([m[e[0]] if e[0] != 'L' else (2 if e[1] else 1) for e in s], np.int8)
S = [enc(s) for s in seqs]

def simulate(n_total, n_waves, burst=5, slots=0, c_inner=30, c_spec=4, c_leaf=85, c_tlas=120, c_iter=25, c_fin=40, c_refill=80, seed=0,
             leaf_min=1, refill_at=16):
    """slots = parked-leaf capacity per lane (0 = baseline).  leaf_min: the leaf step runs when at least this many lanes
    want it, or when any lane is BLOCKED (cannot step in a burst because of leaves), or nothing else can progress."""
    rng = np.random.default_rng(seed)
    order = rng.integers(0, len(S), n_total)
    qpos = 0; cost = 0; lane_steps = 0; wave_steps = 0
    st = dict(rb=0, inner=0, leaf=0, leaf_lanes=0, iters=0)
    seq = [[None] * 64 for _ in range(n_waves)]; pos = [[0] * 64 for _ in range(n_waves)]
    park = [[[] for _ in range(64)] for _ in range(n_waves)]
    alive = [True] * n_waves
    while any(alive):
        for wi in range(n_waves):
            if not alive[wi]: continue
            sq, ps, pk = seq[wi], pos[wi], park[wi]
            idle = [l for l in range(64) if sq[l] is None]
            if len(idle) >= refill_at:
                if qpos < n_total:
                    k = min(len(idle), n_total - qpos)
                    for l in idle[:k]:
                        sq[l] = S[order[qpos]]; ps[l] = 0; pk[l] = []; qpos += 1
                    cost += c_refill
                elif len(idle) == 64:
                    alive[wi] = False; continue
            cost += c_iter; st['iters'] += 1
            for k in range(burst):
                act = 0
                for l in range(64):
                    s = sq[l]
                    if s is None: continue
                    p = ps[l]
                    if p >= len(s): continue
                    e = s[p]
                    if e == 0: ps[l] = p + 1; act += 1
                    elif (e == 1 or e == 2) and len(pk[l]) < slots: pk[l].append(p); ps[l] = p + 1; act += 1
                if act:
                    cost += c_inner + (c_spec if slots else 0); st['inner'] += 1; lane_steps += act; wave_steps += 1
            # who wants the leaf step?
            want = 0; blocked = 0; ntlas = 0; can_inner = 0
            for l in range(64):
                s = sq[l]
                if s is None: continue
                p = ps[l]
                e = s[p] if p < len(s) else 4
                if pk[l]:
                    want += 1
                    if e != 0: blocked += 1
                elif e == 1 or e == 2:
                    want += 1
                    if slots == 0: blocked += 1
                if e == 3: ntlas += 1
                if e == 0: can_inner += 1
            run_leaf = want and (want >= leaf_min or blocked * 4 >= want or can_inner == 0 or blocked >= 8)
            if run_leaf:
                n = 0
                for l in range(64):
                    s = sq[l]
                    if s is None: continue
                    if pk[l]:
                        pp = pk[l].pop(0); n += 1
                        if s[pp] == 2:
                            st['rb'] += ps[l] - (pp + 1); ps[l] = pp + 1; pk[l] = []
                        continue
                    p = ps[l]
                    if p < len(s) and (s[p] == 1 or s[p] == 2): ps[l] = p + 1; n += 1
                cost += c_leaf; st['leaf'] += 1; st['leaf_lanes'] += n
            if ntlas:
                for l in range(64):
                    s = sq[l]
                    if s is not None and ps[l] < len(s) and s[ps[l]] == 3 and not pk[l]: ps[l] += 1
                cost += c_tlas
            fin = 0
            for l in range(64):
                s = sq[l]
                if s is None: continue
                while ps[l] < len(s) and s[ps[l]] == 4 and not pk[l]: ps[l] += 1
                if ps[l] >= len(s) and not pk[l]: sq[l] = None; fin += 1
            if fin: cost += c_fin
    return cost / n_total, lane_steps / max(wave_steps, 1) / 64, st

n_total, n_waves = 5000, 20
rows = [("baseline", dict()),
        ("1 slot", dict(slots=1)), ("2 slots", dict(slots=2)), ("3 slots", dict(slots=3)),
        ("1 slot, leaf>=24", dict(slots=1, leaf_min=24)), ("2 slots, leaf>=24", dict(slots=2, leaf_min=24)),
        ("2 slots, leaf>=32", dict(slots=2, leaf_min=32)), ("3 slots, leaf>=32", dict(slots=3, leaf_min=32)), ("3 slots, leaf>=40", dict(slots=3, leaf_min=40)),
        ("2 slots, leaf>=32, burst 4", dict(slots=2, leaf_min=32, burst=4)), ("2 slots, leaf>=32, burst 6", dict(slots=2, leaf_min=32, burst=6)),
        ("baseline refill 8", dict(refill_at=8)), ("2 slots leaf>=32 refill 8", dict(slots=2, leaf_min=32, refill_at=8))]
for name, kw in rows:
    c, occ, s = simulate(n_total, n_waves, **kw)
    print("%-30s cost/ray %7.1f  burst occ %.3f  iters %5d inner %6d leaf %5d (%.1f lanes) rollback %d" % (name, c, occ, s['iters'], s['inner'], s['leaf'], s['leaf_lanes'] / max(s['leaf'], 1), s['rb']))
