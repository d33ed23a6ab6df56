"""The shelved kernel experiments under tools/dev_kernels/*.patch are evidence (DESIGN.md section 6 quotes their measurements):
each must still apply to the tree, or the evidence is no longer reproducible.  (a patch that no longer applies was silent.)"""
import glob
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATCHES = sorted(glob.glob(os.path.join(ROOT, "tools", "dev_kernels", "*.patch")))


@pytest.mark.skipif(shutil.which("git") is None or not os.path.isdir(os.path.join(ROOT, ".git")), reason="needs the git work tree")
@pytest.mark.parametrize("patch", PATCHES, ids=[os.path.basename(p) for p in PATCHES])
def test_shelved_patch_still_applies(patch):
    r = subprocess.run(["git", "apply", "--check", patch], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, f"{os.path.basename(patch)} no longer applies:\n{r.stderr}"


def test_there_are_patches_to_check():
    assert len(PATCHES) >= 4
