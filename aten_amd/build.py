"""Build recipes for the native pieces (in-tree, explicit compiler calls).

  libaten_amd_scene.so : host-only BVH builder (g++)
  libaten_amd.so       : HIP kernels + C-ABI for gfx950 (hipcc)
  oracle/liboracle.so  : CPU oracle -- test infrastructure, built by `make -C oracle`
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "aten_amd")
CSRC = os.path.join(PKG, "csrc")

HOST_LIB = os.path.join(PKG, "libaten_amd_scene.so")
HIP_LIB = os.path.join(PKG, "libaten_amd.so")
ORACLE_LIB = os.path.join(ROOT, "oracle", "liboracle.so")

# -ffp-contract=off: the parity contract is per-operation IEEE fp32 rounding.
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
             "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
             "-Wno-unused-result"]


def _newer(target, sources):
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources)


def _walk(d, exts):
    out = []
    for r, _, fs in os.walk(d):
        for f in fs:
            if f.endswith(exts):
                out.append(os.path.join(r, f))
    return out


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


_HIPCC_VERSION = None


def hipcc_version():
    """First line of `hipcc --version` that names the HIP / clang build (the GPU box runs the same image: same string);
    '' when there is no hipcc (then the prebuilt library's own id is all there is to compare with)."""
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            out = subprocess.check_output([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], stderr=subprocess.STDOUT).decode()
            _HIPCC_VERSION = " | ".join(l.strip() for l in out.splitlines() if l.startswith(("HIP version", "AMD clang version")))
        except Exception:
            _HIPCC_VERSION = ""
    return _HIPCC_VERSION


def kernel_sources_sha16():
    """Identity of the code the GPU runs: SHA-256 (first 16 hex digits) over the sources of libaten_amd.so, in path order.
    Compiled into the library (atn_build_id, with the extra compile flags) and recorded in profiles/*counters*.json when the PMC
    passes are collected (tools/pmc_to_json.py); bench.py uses a record only if it names the build id of the LOADED binary and
    that binary was built from the sources in the tree (no .git on the GPU box: a content hash, not a commit id)."""
    import hashlib
    h = hashlib.sha256()
    # what aten_amd.hip is made of: the .hip, device/*.hpp, host/*.hpp and the two headers it includes.  (host/*.cpp and
    # aten_amd_scene.h build libaten_amd_scene.so -- BVH builder, camera, scene ingestion -- and never reach a kernel.)
    files = sorted(f for f in _walk(CSRC, (".hip", ".h", ".hpp"))) + [os.path.join(ROOT, "include", n) for n in ("aten_amd.h", "aten_layout.h")]
    for f in files:
        h.update(os.path.relpath(f, ROOT).encode())
        h.update(open(f, "rb").read())
    h.update(repr(HIP_UNITS).encode())      # the per-unit compiler flags are part of what the GPU runs ...
    h.update(repr(HIP_FLAGS).encode())      # ... and so are the common ones (-O3, -ffp-contract=off, correctly rounded divide) ...
    h.update(hipcc_version().encode())      # ... and the compiler that turned them into machine code
    return h.hexdigest()[:16]


def build_id(extra_flags=()):
    """What libaten_amd.so answers from atn_build_id(): the hash of the kernel sources, the common and per-unit compiler flags
    (HIP_FLAGS, HIP_UNITS) and the compiler's version string + the extra flags a variant build adds (none for the product
    build; tools/build_variants.sh passes its own)."""
    return kernel_sources_sha16() + "|" + " ".join(sorted(extra_flags))


def loaded_build_id():
    from ._lib import lib
    return lib().atn_build_id().decode()


def build_host(force=False):
    srcs = [os.path.join(CSRC, "host", "bvh_builder.cpp"), os.path.join(CSRC, "host", "camera.cpp"),
            os.path.join(CSRC, "host", "obj_ingest.cpp")]
    deps = srcs + _walk(os.path.join(ROOT, "include"), (".h",))
    if force or not _newer(HOST_LIB, deps):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", HOST_LIB] + srcs)
    return HOST_LIB


# libaten_amd.so is two translation units with ONE flag of difference (device/svgf_atrous.hpp says why):
#   aten_amd.hip      -fno-slp-vectorize  (no packed-fp32 pairing: the path-tracing kernels lose 15-25 % of their registers to it)
#   svgf_atrous.hip   vectoriser on       (straight-line tap arithmetic, 22 % faster packed)
#   regen.hip         -fno-slp-vectorize  (the path-regeneration kernels: same sources and flags as aten_amd.hip's, compiled beside them)
#   shade_relaxed.hip  k_shade with the reference GPU build's --use_fast_math rules (opt-in, atn_set_shade_math): contraction, approximate
#                      division / sqrt, flushed denormals (the unit itself redirects sinf / cosf / expf / logf / powf to the hardware forms)
RELAXED_FLAGS = ["-fno-slp-vectorize", "-ffp-contract=fast", "-fno-hip-fp32-correctly-rounded-divide-sqrt", "-fgpu-flush-denormals-to-zero"]
HIP_UNITS = [("aten_amd.hip", ["-fno-slp-vectorize"]), ("regen.hip", ["-fno-slp-vectorize"]), ("shade_relaxed.hip", RELAXED_FLAGS), ("svgf_atrous.hip", [])]


def hip_compile(out_lib, extra_flags=(), objdir=None, hipcc=None):
    """Compiles the translation units of libaten_amd.so in parallel and links them into `out_lib`.  `extra_flags` (variant builds:
    tools/build_variants.sh) go to every unit and into the build id."""
    hipcc = hipcc or os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = objdir or os.path.join(PKG, "_obj")
    os.makedirs(objdir, exist_ok=True)
    tag = os.path.splitext(os.path.basename(out_lib))[0]
    common = [f for f in HIP_FLAGS if f != "-shared"] + list(extra_flags) + ['-DATN_BUILD_ID="%s"' % build_id(extra_flags), "-I", os.path.join(ROOT, "include")]
    procs, objs = [], []
    for src, unit_flags in HIP_UNITS:
        obj = os.path.join(objdir, "%s.%s.o" % (tag, os.path.splitext(src)[0]))
        cmd = [hipcc] + common + unit_flags + ["-c", "-o", obj, os.path.join(CSRC, src)]
        print("+", " ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(obj)
    failed = [(cmd, pr.returncode) for cmd, pr in procs if pr.wait() != 0]      # (every unit is waited for before anything is raised)
    if failed:
        raise subprocess.CalledProcessError(failed[0][1], failed[0][0])
    _run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib] + objs)
    return out_lib


def build_hip(force=False):
    deps = _walk(CSRC, (".hip", ".h", ".hpp", ".cpp")) + _walk(os.path.join(ROOT, "include"), (".h",)) + [os.path.abspath(__file__)]
    if force or not _newer(HIP_LIB, deps):
        hip_compile(HIP_LIB)
    return HIP_LIB


def build_oracle(force=False):
    d = os.path.join(ROOT, "oracle")
    deps = _walk(d, (".cpp", ".h")) + _walk(os.path.join(ROOT, "include"), (".h",))
    if force or not _newer(ORACLE_LIB, deps):
        _run(["make", "-C", d, "-B"])
    return ORACLE_LIB


def build_ref(force=False):
    """oracle/_ref/libatenref.so from the reference's untouched sampler / math sources (oracle/Makefile `_ref`);
    skipped where /root/reference does not exist (the GPU box uses the prebuilt file / the fixtures)."""
    ref = os.environ.get("ATEN_REFERENCE", "/root/reference")
    if not os.path.isdir(os.path.join(ref, "src", "libaten", "sampler")):
        return None
    cmd = ["make", "-C", os.path.join(ROOT, "oracle"), "_ref", "REF=" + ref]
    _run(cmd + (["-B"] if force else []))
    return os.path.join(ROOT, "oracle", "_ref", "libatenref.so")


if __name__ == "__main__":
    what = sys.argv[1:] or ["host", "hip", "oracle"]
    for w in what:
        {"host": build_host, "hip": build_hip, "oracle": build_oracle, "ref": build_ref}[w](force=True)
