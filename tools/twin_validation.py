"""Validation of the interior-point step rules OFF the benchmark seed, on the CPU twin:
exo / aero, K = 30 / 50 / 100, flyable variant, several seeds.  Writes a markdown table.
    python tools/twin_validation.py > profiles/r02_ipm_validation.md"""
import os, sys
import numpy as np
from dataclasses import replace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import model
import twin_stats

z = np.load(os.path.join(ROOT, "tests", "golden", "lift_drag_tables.npz"))
aero = model.AeroData(z["drag"], z["lift"], z["torque"])
exo = model.base_prob_scaled()
cases = [
    ("exo K=50 (bench seed)", exo, 512, 14, 20261004),
    ("exo K=50 seed 7", exo, 256, 14, 7),
    ("exo K=50 seed 99", exo, 256, 14, 99),
    ("exo K=50 seed 12345", exo, 256, 14, 12345),
    ("aero K=50 (configs[2] seed)", model.base_prob_scaled(aero), 256, 14, 20261003),
    ("aero K=50 seed 5", model.base_prob_scaled(aero), 128, 14, 5),
    ("exo K=30", replace(exo, K=30), 256, 14, 20261004),
    ("exo K=100 (configs[4] seed)", replace(exo, K=100), 96, 10, 20261005),
    ("aero K=100", replace(model.base_prob_scaled(aero), K=100), 48, 8, 11),
    ("flyable exo K=50 (mdry 0.55, tf_guess 8)", replace(exo, mdry=0.55, tf_guess=8.0), 128, 20, 7),
    ("flyable aero K=50", replace(model.base_prob_scaled(aero), mdry=0.55, tf_guess=8.0), 64, 20, 8),
]
print("# Interior-point solver off the benchmark seed (CPU twin = the device solver core, tol 1e-8, accept 1e-6, refine <= 6)\n")
print("| case | B x steps | IPM its mean / max | status 0 | status 4 | other | merit max | merit p99.9 |")
print("|---|---|---|---|---|---|---|---|")
for name, p, B, steps, seed in cases:
    m, st, it = twin_stats.run(p, B, steps, seed, 1e-8, verbose=False)
    print("| %s | %d x %d | %.2f / %d | %.4f | %.4f | %.4f | %.2e | %.2e |" % (
        name, B, steps, it.mean(), it.max(), (st == 0).mean(), (st == 4).mean(), ((st != 0) & (st != 4)).mean(), m.max(), np.quantile(m, 0.999)), flush=True)
