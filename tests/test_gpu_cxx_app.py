"""The drop-in boundary is C: tests/cxx/aten_app.cpp is a C++17 program that includes only include/*.h, links only
libaten_amd.so / libaten_amd_scene.so, builds a scene with the host library, renders through atn_create / atn_upload_scene /
atn_update_camera / atn_init_sampler / atn_render (the call sequence of INTEGRATION.md's adapter) and dumps what it built.
The same arrays go to the CPU oracle; the films must agree.  The program then checks by itself: three atn_mgpu shards on one
device give the same film byte for byte, atn_svgf_render gives a finite non-black image, and after a deformation tick
(atn_update_geometry + atn_lbvh_rebuild_list + atn_update_tlas) a ray finds the moved panel."""
import glob
import os
import subprocess

import numpy as np
import pytest

from aten_amd import layout as L
from test_gpu_parity import frame_tolerance_report

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def build_app(tmp_path):
    exe = str(tmp_path / "aten_app")
    lib = os.path.join(ROOT, "aten_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cxx", "aten_app.cpp"), "-L", lib, "-laten_amd", "-laten_amd_scene",
                           "-Wl,-rpath," + lib, "-o", exe])
    return exe


def test_cxx_application_through_the_c_abi(tmp_path, orc):
    from aten_amd.scene.builder import FlatScene
    exe = build_app(tmp_path)
    out = tmp_path / "dump"
    out.mkdir()
    W, H, frames = 96, 80, 3
    res = subprocess.run([exe, str(out), str(W), str(H), str(frames)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stderr
    assert "aten_app: 5 lists" in res.stdout

    def rd(name, dt):
        return np.fromfile(str(out / name), dt)
    lists = [rd(os.path.basename(f), L.BVH_NODE) for f in sorted(glob.glob(str(out / "bvh_*.bin")))]
    cfg = L.SceneRenderingConfig.from_buffer_copy(open(str(out / "config.bin"), "rb").read())
    fs = FlatScene.from_arrays(rd("objects.bin", L.OBJECT_PARAM), rd("matrices.bin", np.float32), rd("materials.bin", L.MATERIAL_PARAM),
                               rd("lights.bin", L.LIGHT_PARAM), rd("triangles.bin", L.TRIANGLE_PARAM), rd("vtx_pos.bin", np.float32),
                               rd("vtx_nml.bin", np.float32), lists, config=cfg, scene_bbox=((-1, 0, -1), (1, 2, 1)))
    cam = rd("camera.bin", L.CAMERA_PARAM)[0]
    got = rd("film.bin", np.float32).reshape(H, W, 4)
    # the camera block the host library computed is the oracle's, byte for byte
    want_cam = orc.create_camera((0.0, 1.0, 3.2), (0.0, 1.0, 0.0), 40.0, W, H)
    assert cam.tobytes() == want_cam.tobytes()
    seeds = orc.init_sampler(W, H, 0)
    film = None
    for f in range(frames):
        film = orc.render(fs, want_cam, seeds, W, H, 5, 3, frame=f, film=film)
    assert (got[..., 3] == frames).all() and (film[..., 3] == frames).all()     # FilmProgressive's sample count
    frac, mean_err = frame_tolerance_report(got, film)
    assert frac >= 0.995 and mean_err <= 5e-3, (frac, mean_err)
    assert got[..., :3].mean() > 0.05
