"""Container-only generator of tests/golden/script_literals.json (TEST INFRASTRUCTURE).

Reads every hot-path script of the reference AS TEXT (no import, nothing executed) and records, per
class definition, the literals that differ from script to script: the IC time `t0`, the boundary-loss
form, the Adam/AdamW choice and weight decay, the L-BFGS settings, and the script's `layers` literal and
residual coefficient set.  tests/test_host_logic.py::test_script_literals_match_reference holds
pde_amd/scripts.py to this file, so a drop-in class cannot silently train with other settings than the
script it replaces.  Re-run:  python oracle/gen_script_literals.py   (needs /root/reference)."""
import glob
import json
import os
import re

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "script_literals.json")

SCRIPTS = sorted(
    glob.glob(os.path.join(REF, "7kdv", "*.py")) + glob.glob(os.path.join(REF, "5kdv", "*.py"))
    + glob.glob(os.path.join(REF, "CH", "*.py"))
    + [os.path.join(REF, "NLS", f) for f in ("PI-GNSS-torch.py", "bc-GNSS-torch.py", "sw-GNSS.py", "sw-GNSS-adamW.py",
                                              "sw-PINN-rogue.py",
                                              "Ibc-test.py", "sw-test.py")]   # scratch copies kept for their residual ids (4th order; regulariser)
    + [os.path.join(REF, "Burgers", "Burgers_equation(PINN).py")])


def live(lines):
    """(line number, text) of the lines that are not comments"""
    return [(i + 1, ln) for i, ln in enumerate(lines) if ln.strip() and not ln.strip().startswith("#")]


def num(s):
    return float(eval(s, {"__builtins__": {}}, {}))          # a numeric literal such as 1e-7 or -2


def parse_class(name, start, body):
    d = {"class": name, "line": start}
    for no, ln in body:
        m = re.match(r"\s*self\.t0 = torch\.full_like\(self\.x0,\s*([^)]+)\)", ln)
        if m:
            arg = m.group(1).strip()
            d["t0"] = "argument" if arg == "t0" else num(arg)
            d["t0_line"] = no
        if re.match(r"\s*self\.t0 = torch\.tensor\(0\*x0\)", ln) or re.match(r"\s*self\.t0 = torch\.zeros_like\(self\.x0", ln):
            d["t0"], d["t0_line"] = 0.0, no
        m = re.match(r"\s*lambda_reg = ([0-9.e+-]+)", ln)
        if m:
            d["lambda_reg"], d["lambda_reg_line"] = num(m.group(1)), no
        m = re.match(r"\s*loss_b = (.*)", ln)
        if m and "bc_form" not in d:
            d["bc_form"] = "periodic" if "u_lb_pred - u_ub_pred" in m.group(1) else "dirichlet"
            d["bc_line"] = no
        m = re.match(r"\s*self\.optimizer_Adam = torch\.optim\.(AdamW?)\((.*)\)", ln)
        if m:
            d["optimizer"] = m.group(1).lower()
            wd = re.search(r"weight_decay=([0-9.e+-]+)", m.group(2))
            d["weight_decay"] = num(wd.group(1)) if wd else None
            d["optimizer_line"] = no
        for key in ("lr", "max_eval", "history_size", "tolerance_grad"):
            m = re.match(r"\s*%s=([^,]+)," % key, ln)
            if m:
                d.setdefault("lbfgs", {})[key] = num(m.group(1))
                d.setdefault("lbfgs_lines", []).append(no)
        m = re.match(r"\s*def set_training_data\(self,\s*(.*)\):", ln)
        if m:
            d["set_training_data"] = [a.strip() for a in m.group(1).split(",")]
        m = re.match(r"\s*message = '(.*)' %", ln)
        if m and "log_format" not in d:
            d["log_format"] = m.group(1)
        # which residual body: the return line of net_f (or its fu line for the two-output nets)
        sq = ln.replace(" ", "")
        if re.match(r"\s*return\s+u_t", ln) or re.match(r"\s*fu = u_t", ln):
            if "140*u*u*u*u_x" in sq:
                eq = "kdv7_1sol"
            elif "(5/98)*u**3*u_x" in sq:
                eq = "kdv7_3sol"
            elif "180*u*u*u_x" in sq:
                eq = "kdv5"
            elif "u_xxt" in sq:
                eq = "ch_bc" if "+2*u_x" in sq else "ch"
            elif "self.nu*u_xx" in sq:
                eq = "burgers"
            elif sq.startswith("fu=u_t+0.5*v_xx+(u**2+v**2)*v+0.0625*(v_xxxx"):
                eq = "lpd"
            elif sq.startswith("fu=u_t+0.5*v_xx+(u**2+v**2)*v"):
                eq = "gnss" if "-0.1*(u_xxx+6*(u**2+v**2)*u_x)" in sq else "nls"
            else:
                eq = "unknown"
            d["equation"], d["equation_line"] = eq, no
    if d.get("equation") == "nls" and "lambda_reg" in d:
        d["equation"] = "nls_reg"        # plain NLS body + the derivative regulariser added to the loss
    return d


def main():
    out = {}
    for path in SCRIPTS:
        rel = os.path.relpath(path, REF)
        lines = open(path, encoding="utf-8", errors="replace").read().split("\n")
        lv = live(lines)
        starts = [(no, re.match(r"class (\w*PhysicsInformedNN)\(", ln).group(1)) for no, ln in lv
                  if re.match(r"class \w*PhysicsInformedNN\(", ln)]
        if not starts:
            continue                                        # e.g. the dead TF1 Allen-Cahn script
        main_at = next((no for no, ln in lv if ln.startswith("if __name__")), len(lines))
        ent = {"classes": []}
        for k, (no, name) in enumerate(starts):
            end = starts[k + 1][0] if k + 1 < len(starts) else main_at
            ent["classes"].append(parse_class(name, no, [(n, l) for n, l in lv if no <= n < end]))
        for no, ln in lv:
            m = re.match(r"\s*layers = (\[[0-9, ]+\])", ln)
            if m and no >= main_at:
                ent["layers"], ent["layers_line"] = json.loads(m.group(1)), no
                break
        out[rel] = ent
    with open(OUT, "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
    print("wrote %s: %d scripts" % (OUT, len(out)))


if __name__ == "__main__":
    main()
