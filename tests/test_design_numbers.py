"""DESIGN.md §4.4 is generated from the measurement files in profiles/ (tools/design_numbers.py): the block in the file must be what the
files say today, no line of DESIGN.md may exceed 200 characters, and every profile file the block names must exist."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_design_numbers_block_is_current():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "design_numbers.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_design_md_lines_and_cited_profiles():
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    too_long = [i + 1 for i, l in enumerate(text.splitlines()) if len(l) > 200]
    assert not too_long, f"DESIGN.md lines over 200 characters: {too_long}"
    for name in set(re.findall(r"`(?:profiles/)?(r0\d_[A-Za-z0-9_.]+\.(?:txt|json|csv|log))`", text)):
        assert os.path.exists(os.path.join(ROOT, "profiles", name)), f"DESIGN.md cites profiles/{name}, which does not exist"
