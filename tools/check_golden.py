#!/usr/bin/env python3
"""Re-generate tests/golden/*.npz from the REFERENCE into a temporary directory and compare every array of every re-generated
fixture bit for bit with the committed one (the oracle's pin is only as good as the fixtures; tools/gen_golden.py
writes them, this verifies them).

Runs only where /root/reference exists (the build container); exits 2 when it does not.

    python tools/check_golden.py                 # the quick groups (about a minute on 8 cores)
    python tools/check_golden.py --all           # + deeplab / train / train3 / train8 / decoder (several minutes)
    python tools/check_golden.py msda loss       # named groups of tools/gen_golden.py main()
The minutes-long fixtures (deeplab_big, train_big, deeplab_c3, train_c3, decoder_704, decoder_c5) are checked only when named.
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
QUICK = ["ops", "msda", "loss", "m2f", "metric", "datapath", "encoder"]
ALL = QUICK + ["deeplab", "train", "decoder"]


def compare(tmp):
    """-> (checked fixture names, list of (fixture, array, what differs))."""
    bad, names = [], sorted(f for f in os.listdir(tmp) if f.endswith(".npz"))
    for f in names:
        committed = os.path.join(GOLDEN, f)
        if not os.path.exists(committed):
            bad.append((f, "*", "not committed"))
            continue
        a, b = np.load(os.path.join(tmp, f), allow_pickle=False), np.load(committed, allow_pickle=False)
        if sorted(a.files) != sorted(b.files):
            bad.append((f, "*", f"array names differ: {sorted(set(a.files) ^ set(b.files))}"))
            continue
        for k in a.files:
            x, y = a[k], b[k]
            if x.dtype != y.dtype or x.shape != y.shape:
                bad.append((f, k, f"{x.dtype}{x.shape} vs {y.dtype}{y.shape}"))
            elif x.tobytes() != y.tobytes():          # bit identity, NaNs included
                bad.append((f, k, "bytes differ"))
    return names, bad


def regenerate(groups, tmp):
    """gen_golden.main() with its output directory pointed at `tmp`, in a child interpreter (it patches sys.modules)."""
    code = ("import sys; sys.argv = ['gen_golden.py'] + %r; sys.path.insert(0, %r); import gen_golden; gen_golden.OUT = %r; gen_golden.main()"
            % (list(groups), os.path.join(ROOT, "tools"), tmp))
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.DEVNULL)


def main():
    if not os.path.isdir("/root/reference"):
        print("check_golden: /root/reference is absent (the fixtures can only be re-generated in the build container)")
        return 2
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    groups = args or (ALL if "--all" in sys.argv else QUICK)
    with tempfile.TemporaryDirectory(prefix="mss_golden_") as tmp:
        regenerate(groups, tmp)
        names, bad = compare(tmp)
    for f in names:
        print(("DIFF  " if any(b[0] == f for b in bad) else "same  ") + f)
    for b in bad:
        print("   ", *b)
    print(f"check_golden: {len(names)} fixtures re-generated from {groups}, {len(names) - len({b[0] for b in bad})} bit-identical")
    return 1 if bad or not names else 0


if __name__ == "__main__":
    sys.exit(main())
