"""Packs the reference's aerodynamic DATA file aero/lift_drag.csv (11,041 rows: aoa,mach,drag,lift,torque)
into tests/golden/lift_drag_tables.npz as three [n_mach=61][n_aoa=181] float64 tables, in the order
Aerodynamics.load_aerodata reshapes them (aerodynamics.jl:17-21: cos(AoA) fastest, Mach slowest).
It is input data of configs 3/5 (SURVEY.md §2 "Data"), not source code; /root/reference does not
exist on the GPU box, so the tests read this copy.   Run here:  python tests/golden/make_aero_fixture.py
"""
import os
import numpy as np

src = "/root/reference/aero/lift_drag.csv"
d = np.genfromtxt(src, delimiter=",", names=True)
assert d.shape[0] == 181 * 61
aoa = d["aoa"].reshape(61, 181)
mach = d["mach"].reshape(61, 181)
assert np.allclose(aoa[0], np.linspace(-1, 1, 181)) and np.allclose(mach[:, 0], np.arange(61) * 0.025)
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lift_drag_tables.npz")
np.savez_compressed(out, drag=d["drag"].reshape(61, 181), lift=d["lift"].reshape(61, 181),
                    torque=d["torque"].reshape(61, 181))
print(out, os.path.getsize(out))

# The reference's only recorded non-input data: aero/lift_drag_test.csv, 1,694 samples logged from a flight in the game by
# aero/TestFlight.jl (its stream callback :66-90 and the CSV write :111-113): aoa in DEGREES = acosd(dot(dir, vel) / |vel|) (:67),
# mach (:68; one row holds `inf`), and the logged force DIVIDED BY THE AIR DENSITY (:86) projected on the velocity ("drag"), on the
# lift direction ("lift") and on their cross product ("other").  Repacked as data for tests/test_oracle_aero.py.
t = np.genfromtxt("/root/reference/aero/lift_drag_test.csv", delimiter=",", names=True)
out2 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lift_drag_flight_log.npz")
np.savez_compressed(out2, aoa_deg=t["aoa"], mach=t["mach"], drag=t["drag"], lift=t["lift"], other=t["other"])
print(out2, os.path.getsize(out2))
