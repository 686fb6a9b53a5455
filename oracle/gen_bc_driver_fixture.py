"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Container-only (needs /root/reference).

What does the reference's bc-PINN driver loop hand to `set_training_data`, segment by segment?

The loop lives in the `__main__` block of CH/bc-CH-1soliton.py (:334-473; the copies in 5kdv/, 7kdv/ and NLS/ are
commented out) and cannot be run as shipped: its data file CH/data/CH-201*512.mat is one of the missing blobs
(.MISSING_LARGE_BLOBS) and one segment trains for hours.  This script REPLAYS the block's own argument
construction on synthetic data: it executes the reference's class definitions, then the text of its `__main__`
block up to the end of the segment loop (read from the reference file at run time, never copied into the repo),
with

  * `scipy.io.loadmat` answering with a synthetic travelling wave on a 201 x 128 grid,
  * `pyDOE.lhs` (not installed) answered by pde_amd.drivers.lhs drawing from the global numpy state, which the
    reference seeds with np.random.seed(1234) at import (:21),
  * `train` / `save_model` of both classes turned into no-ops and `predict` into a closed-form function of the
    points and of the segment the model belongs to (so that what one segment hands to the next is reproducible
    without training),

and records every constructor call and every `set_training_data` call: class, lb, ub, x0, u0, t0, u1, tb, Xf, Xh.
tests/test_drivers_cpu.py holds pde_amd.drivers.backward_compatible to that record (same numpy seed, the same
stand-ins applied to the drop-in classes).

Findings recorded in the fixture: the reference stops with a ValueError in its fifth pass (one time row left,
`np.random.choice(1, 25, replace=False)`), after four segments; segments 1-3 pass the segment's first ROW INDEX as
`t0` (:447-448) and the last N0 values of the history prediction as `u0` (:446).

Only data is written.  Run:  python oracle/gen_bc_driver_fixture.py
"""
import hashlib
import os
import sys
import types

sys.dont_write_bytecode = True
import matplotlib

matplotlib.use("Agg")
import numpy as np
import scipy.io

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
from pde_amd import drivers  # noqa: E402

REF = "/root/reference"
SCRIPT = "CH/bc-CH-1soliton.py"
OUT = os.path.join(ROOT, "tests", "golden", "bc_driver_args.npz")
NX, NT = 128, 201


def synthetic_mat():
    """t (1,201), x (1,128), uu (201,128): a travelling sech^2 wave on the script's domain [-15,15] x [-1, 1]"""
    x = np.linspace(-15.0, 15.0, NX)
    t = np.linspace(-1.0, 1.0, NT)
    uu = 1.0 / np.cosh(0.5 * (x[None, :] - 3.0 * t[:, None])) ** 2
    return {"t": t[None, :], "x": x[None, :], "uu": uu}


def stub_predict(X, seg):
    """what a 'trained' model of segment `seg` answers (closed form, float64)"""
    X = np.asarray(X, dtype=np.float64)
    return (np.sin(0.3 * X[:, 0:1] + 0.1 * seg) * np.cos(1.7 * X[:, 1:2]) + 0.05 * seg).astype(np.float64)


def digest(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float64))
    return hashlib.sha256(a.tobytes()).hexdigest()


def main():
    stub = types.ModuleType("pyDOE")
    stub.lhs = lambda n, samples=None: drivers.lhs(n, samples, np.random)
    sys.modules["pyDOE"] = stub
    src = open(os.path.join(REF, SCRIPT), encoding="utf-8").read()
    lines = src.split("\n")
    main_at = next(i for i, ln in enumerate(lines) if ln.startswith("if __name__"))
    # the segment loop ends with `counter += 1` (:473); what follows is evaluation and plotting
    end_at = max(i for i, ln in enumerate(lines) if ln.strip() == "counter += 1")
    head = "\n".join(lines[:main_at])
    body = "\n".join(ln[4:] if ln.startswith("    ") else ln for ln in lines[main_at + 1:end_at + 1])

    ns = {"__name__": "ref_bc_ch", "__file__": os.path.join(REF, SCRIPT)}
    cwd = os.getcwd()
    exec(compile(head, os.path.join(REF, SCRIPT), "exec"), ns)          # imports, seeds, class definitions
    os.chdir(cwd)

    calls = []
    state = {"seg": 0}
    for cname in ("PhysicsInformedNN", "bcPhysicsInformedNN"):
        cls = ns[cname]
        orig_init, orig_set = cls.__init__, cls.set_training_data

        def make(cname, orig_init, orig_set):
            def init(self, layers, lb, ub, *a):
                orig_init(self, layers, lb, ub, *a)
                self._seg = state["seg"]
                state["seg"] += 1
                calls.append({"cls": cname, "layers": list(layers), "lb": np.array(lb, dtype=np.float64),
                              "ub": np.array(ub, dtype=np.float64)})

            def set_training_data(self, *args):
                names = ["x0", "u0", "tb", "Xf"] if cname == "PhysicsInformedNN" else ["x0", "u0", "t0", "u1", "tb", "Xf", "Xh"]
                assert len(args) == len(names), (cname, len(args))
                calls[-1].update({k: (np.array(v, dtype=np.float64) if not np.isscalar(v) else v) for k, v in zip(names, args)})
                calls[-1]["t0_type"] = type(dict(zip(names, args)).get("t0", None)).__name__
                orig_set(self, *args)
            return init, set_training_data
        cls.__init__, cls.set_training_data = make(cname, orig_init, orig_set)
        cls.train = lambda self, *a, **k: None
        cls.save_model = lambda self, *a, **k: None
        cls.predict = lambda self, X: stub_predict(X, self._seg)

    real_loadmat = scipy.io.loadmat
    scipy.io.loadmat = lambda *a, **k: synthetic_mat()
    stopped = "completed"
    try:
        exec(compile(body, os.path.join(REF, SCRIPT) + ":__main__", "exec"), ns)
    except ValueError as ex:
        stopped = "ValueError: %s" % ex
    finally:
        scipy.io.loadmat = real_loadmat
        os.chdir(cwd)

    rec = {"n_segments": np.array(len(calls)), "stopped": np.array(stopped), "nx": np.array(NX), "nt": np.array(NT),
           "source": np.array("%s:%d-%d replayed on a synthetic 201 x 128 travelling wave; train/save no-ops, predict = "
                              "closed form of (points, segment)" % (SCRIPT, main_at + 1, end_at + 1))}
    for i, c in enumerate(calls):
        rec["seg%d_cls" % i] = np.array(c["cls"])
        for k in ("lb", "ub", "x0", "u0", "tb"):
            rec["seg%d_%s" % (i, k)] = c[k]
        for k in ("Xf", "Xh", "u1"):
            if k in c:
                rec["seg%d_%s_shape" % (i, k)] = np.array(c[k].shape)
                rec["seg%d_%s_sha256" % (i, k)] = np.array(digest(c[k]))
                rec["seg%d_%s_head" % (i, k)] = c[k][:4].copy()
        if "t0" in c:
            rec["seg%d_t0" % i] = np.array(float(c["t0"]))
            rec["seg%d_t0_type" % i] = np.array(c["t0_type"])
    np.savez_compressed(OUT, **rec)
    print("wrote %s: %d segments, loop %s" % (OUT, len(calls), stopped))
    for i, c in enumerate(calls):
        print("  segment %d: %s lb %s ub %s t0 %s (%s) tb in [%.3f, %.3f] Xh %s" % (
            i, c["cls"], c["lb"], c["ub"], c.get("t0"), c.get("t0_type"), c["tb"].min(), c["tb"].max(),
            c["Xh"].shape if "Xh" in c else None))


if __name__ == "__main__":
    main()
