#!/usr/bin/env python3
"""Quick GPU-side parity report over the golden fixtures (development aid)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import msckf_amd
from msckf_amd.api import UpdateEngine
from conftest import golden_cases, load_golden, rel_err

def main():
    eng = UpdateEngine(max_clones=50, max_features=20000, max_track=31)
    for case in golden_cases():
        prob, ref = load_golden(case)
        t0 = time.time()
        try:
            res = eng.update_problem(prob)
        except Exception as e:
            print(f"{case:26s} EXC {e}")
            continue
        dt = time.time() - t0
        g, q = eng.debug_gate()
        acc_ok = np.array_equal(res.accepted, ref["accepted"])
        gerr = np.max(np.abs(g - ref["gamma"]) / np.maximum(1e-300, np.abs(ref["gamma"]))) if prob.F else 0
        line = f"{case:26s} st={res.status}/{int(ref['status'])} acc_ok={acc_ok} gamma_rel={gerr:.2e} "
        if int(ref["status"]) == 0 and res.status == 0:
            T, rn = eng.debug_compressed()
            d = prob.d
            G = np.zeros((d, d)); G[15:, 15:] = T.T @ T
            b = np.zeros(d); b[15:] = T.T @ rn
            line += f"G={rel_err(G, ref['G']):.2e} b={rel_err(b, ref['b']):.2e} "
        line += f"dx={rel_err(res.dx, ref['dx']):.2e} P={rel_err(res.P_new, ref['P_new']):.2e} t={dt*1e3:.1f}ms"
        line += f" leaves={res.stats['n_leaves']} lv={res.stats['n_levels']} dev_us={res.stats['us_total']:.0f}"
        print(line, flush=True)
    eng.close()

if __name__ == "__main__":
    main()
