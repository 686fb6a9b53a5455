import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pde_amd._native as nat
nat.LIB_PATH = os.path.abspath("build_dev/lib_stamp.so")
nat.SYMBOLS = []
from pde_amd import equations
from bench import WORKLOADS
w = WORKLOADS["kdv7"]; n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda", 0)
lb, ub = np.array(w["lb"], dtype=np.float32), np.array(w["ub"], dtype=np.float32)
eng = nat.NativeEngine(equations.get(w["eq"]), w["layers"], lb, ub, dev, kernel=2)
rng = np.random.default_rng(1); X = (lb + (ub - lb) * rng.random((n, 2))).astype(np.float32)
th = torch.tensor((rng.standard_normal(eng.P) * 0.15).astype(np.float32), device=dev)
x = torch.tensor(X[:, 0].copy(), device=dev); t = torch.tensor(X[:, 1].copy(), device=dev)
buf = torch.zeros(eng.P + 1, device=dev)
for _ in range(3):
    eng.loss_grad(th, x, t, 1.0 / n, buf[eng.P:], buf[:eng.P])
torch.cuda.synchronize()
PER = 57984; XT = 57856
lib = nat.load_library(); lib.pinn_debug_read_jets.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64]
names = {0: "start", 1: "first layer done", 2: "hidden fwd done", 3: "output+residual+cotangent", 4: "top tanh layer reversed",
         5: "hidden reverse done", 6: "first-layer gradients (tile end)"}
for l in (1, 2, 3):
    names[10 + l] = "fwd GEMM layer %d" % l; names[40 + l] = "(reverse layer %d starts)" % l
    names[20 + l] = "  phase A (weight-grad GEMM + tails) layer %d" % l; names[30 + l] = "  phase B (adjoint GEMM) layer %d" % l
acc = {}
order = []
nw = 0
for wave in range(0, 2048, 37):
    a = np.zeros(128, dtype=np.float32)
    lib.pinn_debug_read_jets(eng._h, a.ctypes.data_as(ctypes.c_void_p), (wave * PER + XT) * 4, 128 * 4)
    raw = a.view(np.uint32)
    tm = raw[0::2].astype(np.int64); ids = a[1::2].view(np.int32)
    k = 1
    seq = []
    while k < 64 and not (ids[k] == 0 and k > 0 and tm[k] == 0):
        if ids[k] == 0 and tm[k] == 0: break
        d = (tm[k] - tm[k - 1]) & 0xffffffff
        seq.append((int(ids[k - 1]), int(ids[k]), d))
        if ids[k] == 6: break
        k += 1
    if not seq or seq[-1][1] != 6: continue
    nw += 1
    for i, (a0, b0, d) in enumerate(seq):
        key = (i, a0, b0)
        acc.setdefault(key, []).append(d)
tot = 0
print("waves sampled:", nw)
for key in sorted(acc):
    i, a0, b0 = key
    m = float(np.mean(acc[key])); tot += m
    print("%2d  %-55s %9.0f ticks" % (i, "-> " + names.get(b0, str(b0)), m))
print("tile total %.0f ticks" % tot)
