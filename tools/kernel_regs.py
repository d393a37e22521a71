#!/usr/bin/env python
"""VGPR / SGPR / scratch / LDS / occupancy of every kernel in a built libaten_amd.so (or a variant .so).
The gfx950 code object is carved out of the fat binary's .hip_fatbin section (clang offload bundle) and its
AMDGPU metadata note is read with llvm-readelf.   usage: tools/kernel_regs.py [lib.so] [name-filter]"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def code_objects(lib):
    """Every gfx950 code object in the library: one clang offload bundle per translation unit."""
    data = open(lib, "rb").read()
    import struct
    out, at = [], data.find(b"__CLANG_OFFLOAD_BUNDLE__")
    while at >= 0:
        n = struct.unpack_from("<Q", data, at + 24)[0]
        p = at + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            triple = data[p + 24:p + 24 + tl].decode()
            p += 24 + tl
            if "gfx950" in triple:
                out.append(data[at + off:at + off + size])
        at = data.find(b"__CLANG_OFFLOAD_BUNDLE__", at + 24)
    if not out:
        raise SystemExit("no gfx950 code object in %s" % lib)
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(__file__), "..", "aten_amd", "libaten_amd.so")
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    notes = ""
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co); f.flush()
            notes += subprocess.check_output([LLVM + "/llvm-readelf", "--notes", f.name]).decode()
    for blk in notes.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\.%s:\s+(\S+)" % k, blk) or [None, "?"])[1]
        name = g("name")
        try:
            name = subprocess.check_output([LLVM + "/llvm-cxxfilt", name]).decode().strip()
        except Exception:
            pass
        name = re.sub(r"^void ", "", name)
        name = re.sub(r"\(.*$", "", name)
        if flt and flt not in name:
            continue
        v = int(g("vgpr_count"))
        waves = min(8, 512 // max(8, (v + 7) // 8 * 8))
        print("%-64s vgpr %3s sgpr %3s scratch %4s lds %6s spill_v %s waves/simd<=%d" % (
            name[:64], v, g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"), g("vgpr_spill_count"), waves))


if __name__ == "__main__":
    main()
