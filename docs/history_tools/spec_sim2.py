"""Wave-level issue-cost model of trace_refill with / without speculation past leaves.  Input: /tmp/sim/seqs_<scene>.pkl
from tools/spec_sim.py."""
import sys, pickle
import numpy as np
scene = sys.argv[1] if len(sys.argv) > 1 else "sponza_lod"
seqs = pickle.load(open('/tmp/sim/seqs_%s.pkl' % scene, 'rb'))
# encode: 0 = I, 1 = L not accepted, 2 = L accepted, 3 = T, 4 = E
def enc(s):
    out = []
    for e in s:
        if e[0] == 'I': out.append(0)
        elif e[0] == 'L': out.append(2 if e[1] else 1)
        elif e[0] == 'T': out.append(3)
        else: out.append(4)
    return np.array(out, np.int8)
S = [enc(s) for s in seqs]

def simulate(n_total, n_waves, burst=5, spec=0, c_inner=30, c_inner_spec=4, c_leaf=85, c_tlas=120, c_iter=25, c_fin=40, c_refill=80, seed=0,
             resolve_current=True, leaf_steps=1):
    rng = np.random.default_rng(seed)
    order = rng.integers(0, len(S), n_total)
    qpos = 0
    cost = 0
    lane_steps = 0; wave_steps = 0
    stat = dict(rollback_steps=0, inner_exec=0, leaf_exec=0, leaf_lanes=0, iters=0)
    # per wave lane state
    class W: pass
    waves = []
    for _ in range(n_waves):
        w = W(); w.seq = [None] * 64; w.pos = [0] * 64; w.parked = [-1] * 64   # parked: position of the parked leaf event
        waves.append(w)
    alive = [True] * n_waves
    while any(alive):
        for wi, w in enumerate(waves):
            if not alive[wi]: continue
            idle = [l for l in range(64) if w.seq[l] is None]
            if len(idle) >= 16:
                if qpos < n_total:
                    k = min(len(idle), n_total - qpos)
                    for l in idle[:k]:
                        w.seq[l] = S[order[qpos]]; w.pos[l] = 0; w.parked[l] = -1; qpos += 1
                    cost += c_refill
                elif len(idle) == 64:
                    alive[wi] = False; continue
            cost += c_iter; stat['iters'] += 1
            # burst
            for k in range(burst):
                act = 0
                for l in range(64):
                    s = w.seq[l]
                    if s is None: continue
                    p = w.pos[l]
                    if p >= len(s): continue
                    e = s[p]
                    if e == 0:
                        w.pos[l] = p + 1; act += 1
                    elif spec and (e == 1 or e == 2) and w.parked[l] < 0:
                        w.parked[l] = p; w.pos[l] = p + 1; act += 1
                if act:
                    cost += c_inner + (c_inner_spec if spec else 0); stat['inner_exec'] += 1
                    lane_steps += act; wave_steps += 1
            # leaf / tlas step(s)
            for rep in range(leaf_steps):
                nleaf = ntlas = 0
                for l in range(64):
                    s = w.seq[l]
                    if s is None: continue
                    if w.parked[l] >= 0:
                        pp = w.parked[l]; nleaf += 1
                        if s[pp] == 2:      # accepted: roll back
                            stat['rollback_steps'] += w.pos[l] - (pp + 1)
                            w.pos[l] = pp + 1
                        w.parked[l] = -1
                        continue
                    p = w.pos[l]
                    if p >= len(s): continue
                    e = s[p]
                    if (e == 1 or e == 2) and (not spec or resolve_current):
                        w.pos[l] = p + 1; nleaf += 1
                    elif e == 3 and rep == 0:
                        w.pos[l] = p + 1; ntlas += 1
                if nleaf: cost += c_leaf; stat['leaf_exec'] += 1; stat['leaf_lanes'] += nleaf
                if ntlas: cost += c_tlas
            # list ends + finish
            fin = 0
            for l in range(64):
                s = w.seq[l]
                if s is None: continue
                while w.pos[l] < len(s) and s[w.pos[l]] == 4 and w.parked[l] < 0: w.pos[l] += 1
                if w.pos[l] >= len(s) and w.parked[l] < 0:
                    w.seq[l] = None; fin += 1
            if fin: cost += c_fin
    return cost / n_total, lane_steps / max(wave_steps, 1) / 64, stat

n_total, n_waves = 7400, 20      # one CU's share of a 1.9 M-ray launch
for name, kw in (("baseline burst 5", dict()), ("baseline burst 4", dict(burst=4)), ("baseline burst 6", dict(burst=6)),
                 ("spec burst 5", dict(spec=1)), ("spec burst 4", dict(spec=1, burst=4)), ("spec burst 6", dict(spec=1, burst=6)), ("spec burst 8", dict(spec=1, burst=8)),
                 ("spec burst 5, parked only", dict(spec=1, resolve_current=False)),
                 ("spec burst 5, 2 leaf steps", dict(spec=1, leaf_steps=2)), ("spec burst 8, 2 leaf steps", dict(spec=1, burst=8, leaf_steps=2)),
                 ("spec burst 6 free bookkeeping", dict(spec=1, burst=6, c_inner_spec=0))):
    c, occ, st = simulate(n_total, n_waves, **kw)
    print("%-32s cost/ray %7.1f  burst lane occupancy %.3f  iters %6d inner-steps %6d leaf-steps %5d (%.1f lanes) rollback lane-steps %d"
          % (name, c, occ, st['iters'], st['inner_exec'], st['leaf_exec'], st['leaf_lanes'] / max(st['leaf_exec'], 1), st['rollback_steps']))
