#!/usr/bin/env python
"""What the float tolerance of the parity contract is made of: where the GPU's last bits differ from the CPU
oracle's, by function.  Three ladders, written to one JSON:
  1. libm: ocml's sinf / cosf / atanf / acosf / atan2f / logf / expf / powf on the device (atn_libm_probe) against the oracle
     build's libm (orc_libm_probe) over the argument ranges the renderer uses; sqrtf, division, 1/sqrtf as controls (IEEE: 0 ulp).
  2. BSDF tables: atn_material_table against orc.material_table for the materials of a scene on random (normal, wi, uv, sampler
     state): how often the sampled direction / bsdf / pdf differ in the last bits, by material type.
  3. frames: bit-equal and in-tolerance pixels of the same frame at maxDepth 1 .. 5 (how the per-vertex rate compounds).
The oracle is the checker here (tools/ is test infrastructure).    python tools/ulp_study.py --out gpurun_out/r06_ulp/ulp_study.json"""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ulp_diff(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    ia = a.view(np.int32).astype(np.int64); ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, np.int64(-2**31) - ia, ia)
    ib = np.where(ib < 0, np.int64(-2**31) - ib, ib)
    d = np.abs(ia - ib)
    both_nan = np.isnan(a) & np.isnan(b)
    return np.where(both_nan, 0, d)


def hist(d):
    n = float(d.size)
    return {"n": int(d.size), "differ": float((d > 0).sum() / n), "ulp1": float((d == 1).sum() / n), "ulp2": float((d == 2).sum() / n),
            "ulp3plus": float((d > 2).sum() / n), "max_ulp": int(d.max())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/r06_ulp/ulp_study.json")
    ap.add_argument("--n", type=int, default=1 << 21)
    args = ap.parse_args()
    from aten_amd import layout as L
    from aten_amd.renderer import PathTracing
    from aten_amd.scene import scenedefs
    from aten_amd.scene.camera import create_camera
    from oracle import orc
    rng = np.random.default_rng(6)
    N = args.n
    out = {"n_per_case": N}
    r = PathTracing(0)
    fs_c, cam_c = scenedefs.cornell_box()
    r.UpdateSceneData(fs_c)

    # ---- 1. libm
    u = lambda lo, hi: rng.uniform(lo, hi, N).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, N)
    cases = {
        "sinf  phi in [0, 2 pi)  (ggx / diffuse / IBL sample)": ("sinf", u(0, 2 * np.pi), None),
        "cosf  phi in [0, 2 pi)": ("cosf", u(0, 2 * np.pi), None),
        "sinf  theta in [0, pi/2]  (ggx_sample_m)": ("sinf", u(0, np.pi / 2), None),
        "cosf  theta in [0, pi/2]": ("cosf", u(0, np.pi / 2), None),
        "atanf  a * sqrt(r / (1 - r)), a = 0.3  (ggx.cpp:199-215)": ("atanf", (0.3 * np.sqrt(rng.uniform(0, 1, N) / (1 - rng.uniform(0, 1 - 1e-7, N)))).astype(np.float32), None),
        "acosf  dir.y in [-1, 1]  (background.h:88-127)": ("acosf", u(-1, 1), None),
        "atan2f  (dir.x, dir.z) on the circle  (background.h)": ("atan2f", np.sin(ang).astype(np.float32), np.cos(ang).astype(np.float32)),
        "logf  a^2 of D_GTR1, a in [0.001, 0.1]  (disney_brdf.cpp:148-162)": ("logf", (rng.uniform(0.001, 0.1, N) ** 2).astype(np.float32), None),
        "logf  (0, 1)": ("logf", u(1e-6, 1), None),
        "expf  [-30, 0]  (beckman / velvet)": ("expf", u(-30, 0), None),
        "powf  x in [0, 1], y in [1, 8]": ("powf", u(0, 1), u(1, 8)),
        "sqrtf  control": ("sqrtf", u(0, 100), None),
        "a / b  control": ("div", u(-10, 10), u(0.1, 10)),
        "1 / sqrtf  control (normalize)": ("inversesqrt", u(1e-3, 100), None),
    }
    out["libm"] = {}
    for name, (kind, a, b) in cases.items():
        g = r.libm_probe(kind, a, b)
        o = orc.libm_probe(kind, a, b)
        out["libm"][name] = hist(ulp_diff(g, o))
        print("libm %-70s differ %.4f max %d" % (name, out["libm"][name]["differ"], out["libm"][name]["max_ulp"]), flush=True)

    # ---- 2. BSDF tables
    def tables(tag, fs, ids, n):
        r.UpdateSceneData(fs)
        res = {}
        for mid in ids:
            m = fs.arrays["materials"][mid]
            nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
            wi = rng.normal(size=(n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
            flip = (nrm * wi).sum(1) > 0
            wi[flip] = -wi[flip]                        # incoming direction against the normal, as after the back-face flip
            nrm = nrm.astype(np.float32); wi = wi.astype(np.float32)
            idx = rng.integers(0, 256, n).astype(np.uint32); scr = rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)
            uv = rng.uniform(0, 1, (n, 2)).astype(np.float32)
            gs, ge = r.material_table(mid, nrm, wi, idx, scr, uv)
            os_, oe = orc.material_table(fs, mid, nrm, wi, idx, scr, uv)
            d_dir = ulp_diff(gs[:, 0:3], os_[:, 0:3]).max(1)
            ang_ = np.arccos(np.clip((gs[:, 0:3].astype(np.float64) * os_[:, 0:3]).sum(1) /
                                     np.maximum(np.linalg.norm(gs[:, 0:3], axis=1) * np.linalg.norm(os_[:, 0:3], axis=1), 1e-30), -1, 1))
            same_dir = d_dir == 0
            rec = {"type": int(m["type"]), "n": n,
                   "sampled_dir": hist(d_dir), "sampled_dir_other_lobe_or_branch(angle>1e-3)": float((ang_ > 1e-3).mean()),
                   "sample_bsdf": hist(ulp_diff(gs[:, 3:6], os_[:, 3:6]).max(1)), "sample_pdf": hist(ulp_diff(gs[:, 6], os_[:, 6])),
                   "eval_pdf_at_equal_dir": hist(ulp_diff(ge[same_dir, 0], oe[same_dir, 0])) if same_dir.any() else None,
                   "eval_bsdf_at_equal_dir": hist(ulp_diff(ge[same_dir, 1:4], oe[same_dir, 1:4]).max(1)) if same_dir.any() else None,
                   "any_output_differs": float(((d_dir > 0) | (ulp_diff(gs[:, 3:7], os_[:, 3:7]).max(1) > 0)).mean())}
            res["%s material %d (type %d)" % (tag, mid, int(m["type"]))] = rec
            print("table %-36s dir differs %.4f  bsdf %.4f  pdf %.4f  any %.4f" % ("%s m%d t%d" % (tag, mid, int(m["type"])), rec["sampled_dir"]["differ"],
                  rec["sample_bsdf"]["differ"], rec["sample_pdf"]["differ"], rec["any_output_differs"]), flush=True)
        return res
    n_tab = min(N, 1 << 18)
    fs_a, cam_a = scenedefs.atrium(detail=0.25)
    fs_s, cam_s = scenedefs.sponza_lod()
    out["bsdf_tables"] = {}
    ids_a = [i for i, m in enumerate(fs_a.arrays["materials"]) if int(m["type"]) == L.MTRL_DISNEY][:3]
    ids_s = [i for i, m in enumerate(fs_s.arrays["materials"]) if int(m["type"]) == L.MTRL_GGX][:2]
    out["bsdf_tables"].update(tables("atrium", fs_a, ids_a, n_tab))
    out["bsdf_tables"].update(tables("sponza_lod", fs_s, ids_s, n_tab))
    ids_c = [i for i, m in enumerate(fs_c.arrays["materials"]) if int(m["type"]) == L.MTRL_DIFFUSE][:1]
    out["bsdf_tables"].update(tables("cornell", fs_c, ids_c, n_tab))

    # ---- 3. frames by depth
    out["frames_by_depth"] = {}
    for tag, fs, cam in (("atrium", fs_a, cam_a), ("sponza_lod", fs_s, cam_s)):
        w, h = 256, 144
        c = create_camera(cam["pos"], cam["at"], cam["vfov"], w, h)
        r.UpdateSceneData(fs); r.updateCamera(c); r.initSampler(w, h, 0)
        seeds = orc.init_sampler(w, h, 0)
        rows = {}
        for depth in (1, 2, 3, 5, 8):
            r.reset()
            g = r.render(w, h, depth, 3, frame=0)[..., :3]
            o = orc.render(fs, c, seeds, w, h, depth, 3, frame=0)[..., :3]
            eq = (g.view(np.uint32) == o.view(np.uint32)).all(-1) | (np.isnan(g).any(-1) & np.isnan(o).any(-1))
            inside = np.all(np.abs(g - o) <= 1e-3 * np.maximum(1.0, np.abs(o)), axis=-1)
            d = ulp_diff(g, o).max(-1)
            rows[str(depth)] = {"bit_equal": float(eq.mean()), "within_1e-3": float(inside.mean()), "within_4_ulp": float((d <= 4).mean()),
                                "median_ulp_of_unequal": float(np.median(d[~eq])) if (~eq).any() else 0.0}
            print("frame %-10s depth %d: bit-equal %.4f within 4 ulp %.4f within 1e-3 %.4f" % (tag, depth, eq.mean(), (d <= 4).mean(), inside.mean()), flush=True)
        out["frames_by_depth"][tag + " 256x144 1spp frame 0"] = rows
    r.close()
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(out, open(args.out, "w"), indent=1)


if __name__ == "__main__":
    main()
