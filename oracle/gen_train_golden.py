"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.  Container-only (needs /root/reference).

Golden vectors for `PhysicsInformedNN.train`: what the REFERENCE's own `train()` does to given weights
on given data.  For three scripts (Adam + L-BFGS lr 1; Adam + L-BFGS lr 0.5 / history 150; AdamW) the reference class is
built with seeded xavier weights, fed a small seeded data set through its own `set_training_data`, and trained with its
own `train(nIter, None, 0, LBFGS_max_iter)` — torch.optim.Adam/AdamW for nIter iterations, then one LBFGS.step — once in
float32 (its arithmetic) and once in float64 (truth).  Stored: the data, the initial weights, the weights after the
Adam phase alone and after Adam + L-BFGS in both precisions, the closure counts and the final losses — and the float32
run repeated with the collocation points in another order (the same loss, other rounding): the strong-Wolfe line search
amplifies rounding (Camassa-Holm: 4e-6 -> 1e-4 of the distance travelled), and that run measures by how much; a third
float32 run starts from weights moved by about one rounding (relative 1e-7).

tests/test_gpu_train_golden.py holds the drop-in classes on the HIP path to it: after the Adam phase the weights must be as
close to the float64 trajectory as the reference's own float32 run is (factor 3, floor 1e-6); after the L-BFGS phase
(whose line-search branches amplify rounding) the loss must have come down like the reference's.

Only data is written.  Run:  python oracle/gen_train_golden.py
"""
import os
import sys

sys.dont_write_bytecode = True
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_golden import OUT, flat_theta, load_script, set_dtype  # noqa: E402

N0, NB, NF = 32, 24, 512
N_ADAM, N_LBFGS = 12, 6
CASES = [
    # name, script, layers, lb, ub, two outputs
    ("kdv5_pi", "5kdv/PI-51.py", [2, 50, 50, 50, 50, 1], [-10.0, -2.0], [10.0, 2.0], False),
    ("ch_pi", "CH/PI-CH-1soliton.py", [2, 50, 50, 50, 50, 1], [-15.0, -1.0], [15.0, 1.0], False),
    ("gnss_sw_adamw", "NLS/sw-GNSS-adamW.py", [2, 100, 100, 100, 100, 2], [-5.0, -2.0], [5.0, 0.0], True),
]


def to_dtype(model, dtype):
    """the reference keeps its data as float32 tensors; promote those exact values for the float64 run"""
    set_dtype(model, dtype)
    for k, v in list(vars(model).items()):
        if torch.is_tensor(v) and v.is_floating_point() and k not in ("lb", "ub"):
            w = v.detach().float().to(dtype)
            if k in ("xf", "tf"):
                w.requires_grad_(True)
            setattr(model, k, w)


def main():
    torch.set_num_threads(8)
    rec = {}
    for name, script, layers, lb, ub, two in CASES:
        mod = load_script(script)
        lb, ub = np.array(lb), np.array(ub)
        mod.lb, mod.ub = lb, ub
        rng = np.random.default_rng(99)
        x0 = lb[0] + (ub[0] - lb[0]) * rng.random((N0, 1))
        tb = lb[1] + (ub[1] - lb[1]) * rng.random((NB, 1))
        Xf = lb + (ub - lb) * rng.random((NF, 2))
        u0 = 0.5 / np.cosh(0.5 * x0) ** 2
        v0 = 0.2 * np.tanh(0.3 * x0)
        data = dict(x0=x0, u0=u0, tb=tb, Xf=Xf)
        if two:
            data["v0"] = v0
        out = {}
        for tag, dtype in (("64", torch.float64), ("32", torch.float32)):
            torch.manual_seed(4321)
            model = mod.PhysicsInformedNN(layers, lb, ub)
            theta0 = flat_theta(model).astype(np.float32)
            if two:
                model.set_training_data(x0, u0, v0, tb, Xf)
            else:
                model.set_training_data(x0, u0, tb, Xf)
            to_dtype(model, dtype)
            model.train(N_ADAM, None, 0, 0)             # Adam phase (+ the one closure LBFGS.step(max_iter=0) evaluates)
            out["theta_adam" + tag] = flat_theta(model).astype(np.float64)
            out["iter_adam"] = np.array(model.iter)
            # the same run continued: Adam again is NOT what the reference does, so a second model does Adam + L-BFGS in one call
            torch.manual_seed(4321)
            model2 = mod.PhysicsInformedNN(layers, lb, ub)
            if two:
                model2.set_training_data(x0, u0, v0, tb, Xf)
            else:
                model2.set_training_data(x0, u0, tb, Xf)
            to_dtype(model2, dtype)
            model2.train(N_ADAM, None, 0, N_LBFGS)
            out["theta_end" + tag] = flat_theta(model2).astype(np.float64)
            out["iter_end" + tag] = np.array(model2.iter)
            for p in model2.dnn.parameters():
                p.grad = None
            out["loss_end" + tag] = np.array(float(model2.backward().item()))
            if tag == "32":
                # how far does the reference's OWN float32 run move under a change of rounding alone?  The same run with
                # the collocation points in another order (the same loss, other summation order)
                torch.manual_seed(4321)
                model4 = mod.PhysicsInformedNN(layers, lb, ub)
                Xp = Xf[np.random.default_rng(1).permutation(NF)]
                if two:
                    model4.set_training_data(x0, u0, v0, tb, Xp)
                else:
                    model4.set_training_data(x0, u0, tb, Xp)
                model4.train(N_ADAM, None, 0, N_LBFGS)
                out["theta_end32p"] = flat_theta(model4).astype(np.float64)
                out["iter_end32p"] = np.array(model4.iter)
                # ... and with the initial weights moved by about one float32 rounding (relative 1e-7 * N(0,1))
                torch.manual_seed(4321)
                model5 = mod.PhysicsInformedNN(layers, lb, ub)
                if two:
                    model5.set_training_data(x0, u0, v0, tb, Xf)
                else:
                    model5.set_training_data(x0, u0, tb, Xf)
                with torch.no_grad():
                    for prm in model5.dnn.parameters():
                        prm.mul_(1.0 + 1e-7 * torch.randn_like(prm))
                model5.train(N_ADAM, None, 0, N_LBFGS)
                out["theta_end32q"] = flat_theta(model5).astype(np.float64)
                out["iter_end32q"] = np.array(model5.iter)
            torch.manual_seed(4321)
            model3 = mod.PhysicsInformedNN(layers, lb, ub)
            if two:
                model3.set_training_data(x0, u0, v0, tb, Xf)
            else:
                model3.set_training_data(x0, u0, tb, Xf)
            to_dtype(model3, dtype)
            model3.iter, model3.logfilename = 1, None
            out["loss_start" + tag] = np.array(float(model3.backward().item()))
        for k, v in data.items():
            rec["%s_data_%s" % (name, k)] = np.asarray(v, dtype=np.float64)
        rec[name + "_theta0"] = theta0
        rec[name + "_layers"] = np.array(layers)
        rec[name + "_lb"], rec[name + "_ub"] = lb, ub
        rec[name + "_script"] = np.array(script)
        for k, v in out.items():
            rec["%s_%s" % (name, k)] = v
        ea = np.linalg.norm(out["theta_adam32"] - out["theta_adam64"]) / np.linalg.norm(out["theta_adam64"] - theta0)
        tr = np.linalg.norm(out["theta_end64"] - theta0)
        print("%-14s end: f32 %.2e, f32 with permuted points %.2e, f32 with weights moved by 1e-7 %.2e of the distance travelled "
              "(closures %d, %d)" % (
                  name, np.linalg.norm(out["theta_end32"] - out["theta_end64"]) / tr,
                  np.linalg.norm(out["theta_end32p"] - out["theta_end64"]) / tr,
                  np.linalg.norm(out["theta_end32q"] - out["theta_end64"]) / tr, int(out["iter_end32p"]), int(out["iter_end32q"])))
        print("%-14s Adam phase: |theta32 - theta64| / |theta64 - theta0| = %.2e; closures %d / %d / %d; loss %.4e -> %.4e (f32) %.4e (f64)"
              % (name, ea, int(out["iter_adam"]), int(out["iter_end32"]), int(out["iter_end64"]), float(out["loss_start64"]),
                 float(out["loss_end32"]), float(out["loss_end64"])))
    rec["n_adam"], rec["n_lbfgs"] = np.array(N_ADAM), np.array(N_LBFGS)
    np.savez_compressed(os.path.join(OUT, "train_trajectories.npz"), **rec)


if __name__ == "__main__":
    main()
