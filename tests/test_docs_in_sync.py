"""The measured-number blocks of DESIGN.md / README.md are generated from profiles/r03 (and the round-2 files it still cites) by tools/fill_docs.py: the
committed documents must be what the generator makes of the committed evidence."""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_number_blocks_match_profiles(tmp_path):
    work = tmp_path / "repo"
    (work / "tools").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tools", "fill_docs.py"), work / "tools" / "fill_docs.py")
    shutil.copytree(os.path.join(ROOT, "profiles"), work / "profiles")
    for doc in ("DESIGN.md", "README.md"):
        shutil.copy(os.path.join(ROOT, doc), work / doc)
    subprocess.run([sys.executable, str(work / "tools" / "fill_docs.py")], check=True, capture_output=True, cwd=str(work))
    for doc in ("DESIGN.md", "README.md"):
        assert (work / doc).read_text() == open(os.path.join(ROOT, doc)).read(), f"{doc}: run tools/fill_docs.py"
