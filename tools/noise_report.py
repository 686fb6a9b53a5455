"""GPU: fp32 noise of the HIP path at the reference's checkpoints against the reference's own fp32 noise,
as RMS over the 12 x 256-point batches of tests/golden/noise_*.npz (oracle/gen_noise_fixtures.py).
Truth = the fp64 C oracle (cross-checked here against the fixture's reference-fp64 residual)."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from noise_stats import noise_stats  # noqa: E402
from pde_amd._native import NativeError  # noqa: E402


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--lib":        # a development build instead of pde_amd/libpinnjet.so
        import pde_amd._native as _nat
        _nat.LIB_PATH = os.path.abspath(sys.argv[2])
    dev = torch.device("cuda", 0)
    print("%-16s %-5s %-28s %-28s %-28s" % ("case", "kind", "f: ours / ref32 = ratio", "grad: ours / ref32 = ratio", "loss: ours / ref32 = ratio"))
    for path in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "noise_*.npz"))):
        g = np.load(path)
        for kname, kernel in (("valu", 1), ("mfma", 2)):
            try:
                s = noise_stats(g, kernel, dev)
            except NativeError:
                continue
            print("%-16s %-5s %.2e / %.2e = %-6.2f %.2e / %.2e = %-6.2f %.2e / %.2e = %-6.2f (oracle vs ref fp64 %.1e)" % (
                os.path.basename(path)[6:-4], kname, s["f"], s["ref_f"], s["f"] / s["ref_f"], s["g"], s["ref_g"],
                s["g"] / s["ref_g"], s["l"], s["ref_l"], s["l"] / s["ref_l"], s["oracle_vs_ref64"]))
