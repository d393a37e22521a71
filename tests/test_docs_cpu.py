"""DESIGN.md is generated (docs/design_parts/assemble.py: text parts + the numbers of profiles/): the committed file must be what
the generator writes from the committed profiles, and stay a document one can read in one sitting."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_md_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "DESIGN.md"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "docs", "design_parts", "assemble.py"), "--out", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "WARNING" not in r.stdout, r.stdout                     # (a regenerated burst whose film differs from the serial loop's)
    want = out.read_bytes()
    assert open(os.path.join(ROOT, "DESIGN.md"), "rb").read() == want
    assert len(want) <= 60 * 1024
    assert b"@@" not in want                                       # no placeholder left unfilled
