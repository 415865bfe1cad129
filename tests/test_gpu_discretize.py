"""K1 linearize / K2 propagate on the MI355X vs the CPU oracle (same RK4, fp64).

Tolerance: the kernel and the oracle run the same tableau in fp64 but associate sums differently
(sparse column-wise products vs dense loops, fma contraction on the device), so agreement is to
rounding: 1e-11 absolute on O(1) quantities, stated per assert.
"""
import numpy as np
import pytest

from conftest import random_segments

pytestmark = pytest.mark.gpu


def _cache(p, npts=10):
    from successiveconvexification_amd.dynamics import IntegratorCache
    return IntegratorCache(p, npts=npts)


def _product_problem(aero_tables=None):
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.defns import AtmosphericData
    if aero_tables is None:
        return sp.base_prob_scaled
    d, l, t = aero_tables
    return sp.base_prob_aero_scaled(AtmosphericData(d, l, t))


def _oracle_problem(aero_tables=None):
    from oracle import model
    if aero_tables is None:
        return model.base_prob_scaled()
    d, l, t = aero_tables
    return model.base_prob_scaled(model.AeroData(d, l, t))


@pytest.mark.parametrize("B,K,npts", [(1, 50, 10), (7, 50, 4), (64, 30, 1), (3, 1, 2), (5, 100, 10), (1, 13, 3), (1, 15, 3)])
def test_linearize_matches_oracle_exo(B, K, npts):
    from oracle import dynamics as od
    from successiveconvexification_amd.dynamics import linearize_batch, propagate_batch
    po = _oracle_problem()
    x, u, sigma = random_segments(po, B, K, 20261006 + B)
    dt = 1.0 / (K + 1)
    e_ref, d_ref = od.linearize(od.Params(po), x, u, sigma, dt, npts)
    c = _cache(_product_problem(), npts)
    e, d = linearize_batch(c, x, u, sigma, dt)
    assert np.abs(e - e_ref).max() < 1e-12
    assert np.abs(d - d_ref).max() < 1e-11
    xn = propagate_batch(c, x, u, sigma, dt)
    assert np.abs(xn - e_ref).max() < 1e-12
    # K1's endpoint and K2 are the same map
    assert np.abs(xn - e).max() < 1e-13


def test_linearize_matches_oracle_aero(aero_tables):
    from oracle import dynamics as od
    from successiveconvexification_amd.dynamics import linearize_batch, propagate_batch
    po = _oracle_problem(aero_tables)
    B, K, npts = 16, 50, 10
    x, u, sigma = random_segments(po, B, K, 20261003)
    # put a few nodes on the special branches: v = 0 (ifnz guards) and v parallel to the body axis (no lift)
    x[0, 3, 4:7] = 0.0
    x[1, 5, 7:11] = [1, 0, 0, 0]
    x[1, 5, 4:7] = [-0.2, 0, 0]
    dt = 1.0 / (K + 1)
    e_ref, d_ref = od.linearize(od.Params(po), x, u, sigma, dt, npts)
    c = _cache(_product_problem(aero_tables), npts)
    e, d = linearize_batch(c, x, u, sigma, dt)
    assert np.isfinite(d).all()
    assert np.abs(e - e_ref).max() < 1e-12
    assert np.abs(d - d_ref).max() < 1e-10
    xn = propagate_batch(c, x, u, sigma, dt)
    assert np.abs(xn - e_ref).max() < 1e-12


def test_first_order_taylor_property():
    """Size-independent property at the full batch: endpoint(inp + eps*delta) - endpoint(inp) ~ deriv @ (eps*delta)."""
    from successiveconvexification_amd.dynamics import linearize_batch, propagate_batch
    from oracle import model
    po = model.base_prob_scaled()
    B, K = 8192, 50
    x, u, sigma = random_segments(po, B, K, 20261004)
    dt = 1.0 / (K + 1)
    c = _cache(_product_problem(), 4)
    e, d = linearize_batch(c, x, u, sigma, dt)
    rng = np.random.default_rng(1)
    eps = 1e-6
    dx = rng.normal(size=x.shape) * eps
    du = rng.normal(size=u.shape) * eps
    ds = rng.normal(size=sigma.shape) * eps
    e2 = propagate_batch(c, x + dx, u + du, sigma + ds, dt)
    delta = np.concatenate([dx[:, :-1], du[:, :-1], du[:, 1:], np.broadcast_to(ds[:, None, None], (B, K, 1))], axis=-1)
    pred = np.einsum("bkji,bkj->bki", d, delta)
    err = np.abs(e2 - e - pred).max()
    assert err < 50 * eps * eps * 1e3, err  # second-order remainder


def test_empty_and_errors():
    from successiveconvexification_amd import _lib
    from successiveconvexification_amd.dynamics import linearize_batch
    c = _cache(_product_problem())
    e, d = linearize_batch(c, np.zeros((0, 51, 14)), np.zeros((0, 51, 3)), np.zeros(0), 1 / 51)
    assert e.shape == (0, 50, 14) and d.shape == (0, 50, 21, 14)
    with pytest.raises(ValueError):
        linearize_batch(c, np.zeros((2, 51, 14)), np.zeros((2, 50, 3)), np.zeros(2), 1 / 51)
    with pytest.raises(_lib.ScvxError):
        c.set_npts(0)


@pytest.mark.parametrize("aero", [False, True])
def test_fp32_entry_points_match_the_fp64_oracle_to_the_stated_tolerance(aero, aero_tables):
    """scvx_linearize_f32 / scvx_propagate_f32 (float arithmetic, float storage) against the fp64 C oracle on random
    physical segments (SURVEY 8d law): endpoint 2e-5, derivative 2e-4 relative to the largest entry of its column block --
    the stated tolerance of include/scvx.h.  Also the variant check: the fp64 column-per-lane kernel (SCVX_K1_VARIANT=0,
    the one the float kernel is instantiated from) still agrees with the oracle to 1e-11."""
    from oracle import dynamics as od, model
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache, linearize_batch_f32, propagate_batch_f32
    d, l, t = aero_tables
    po = model.base_prob_scaled(model.AeroData(d, l, t) if aero else None)
    pp = sp.base_prob_aero_scaled(AtmosphericData(d, l, t)) if aero else sp.base_prob_scaled
    B, K = 64, 50
    x, u, s = random_segments(po, B, K, 20261006)
    c = IntegratorCache(pp, npts=10)
    e_ref, d_ref = od.linearize(od.Params(po), x, u, s, 1.0 / (K + 1), 10)
    e32, d32 = linearize_batch_f32(c, x, u, s, 1.0 / (K + 1))
    assert e32.dtype == np.float32 and d32.dtype == np.float32
    assert np.abs(e32 - e_ref).max() < 2e-5 * max(1.0, np.abs(e_ref).max())
    scale = np.abs(d_ref).max(axis=(0, 1, 3), keepdims=True)          # per column of the 14x21 derivative
    assert (np.abs(d32 - d_ref) / np.maximum(scale, 1.0)).max() < 2e-4
    xp32 = propagate_batch_f32(c, x, u, s, 1.0 / (K + 1))
    assert np.abs(xp32 - e_ref).max() < 2e-5 * max(1.0, np.abs(e_ref).max())
    assert np.abs(xp32 - e32).max() < 2e-6                             # K2 and the state part of K1: same arithmetic up to contraction
    c.close()


@pytest.mark.parametrize("aero", [False, True])
def test_closed_form_columns_of_the_linearisation_are_exact(aero, aero_tables):
    """Structure of A_k = d x_{k+1} / d x_k the model implies: nothing depends on position, and without aerodynamics
    nothing but r' = sigma v depends on velocity, so d/dr_k = [0; I; 0; 0; 0] always and d/dv_k = [0; sigma dt I; I; 0; 0]
    for the exo model -- EXACTLY (the exo kernel writes them in closed form, the aero kernel integrates zeros), for both
    K1 variants.  (Skipping these columns in the conic solver's E / E' products was tried: -6 % bytes, +9 % time -- the
    lane-dependent selects stop the loads of a row from batching; profiles/README.md.)"""
    import os
    from oracle import model
    from successiveconvexification_amd import sample_problems as sp
    from successiveconvexification_amd.defns import AtmosphericData
    from successiveconvexification_amd.dynamics import IntegratorCache, linearize_batch
    d, l, t = aero_tables
    po = model.base_prob_scaled(model.AeroData(d, l, t) if aero else None)
    pp = sp.base_prob_aero_scaled(AtmosphericData(d, l, t)) if aero else sp.base_prob_scaled
    B, K = 40, 50
    x, u, s = random_segments(po, B, K, 20261007)
    dt = 1.0 / (K + 1)
    for variant in ("1", "0"):
        os.environ["SCVX_K1_VARIANT"] = variant
        try:
            c = IntegratorCache(pp, npts=10)
            e, D = linearize_batch(c, x, u, s, dt)       # D [B][K][21][14]: D[b, k, col, row]
            c.close()
        finally:
            os.environ.pop("SCVX_K1_VARIANT", None)
        eye = np.zeros((3, 14)); eye[np.arange(3), 1 + np.arange(3)] = 1.0
        assert np.array_equal(D[:, :, 1:4, :], np.broadcast_to(eye, (B, K, 3, 14))), variant
        if not aero:
            want = np.zeros((B, K, 3, 14))
            want[:, :, np.arange(3), 4 + np.arange(3)] = 1.0
            want[:, :, np.arange(3), 1 + np.arange(3)] = (s * dt)[:, None, None]
            assert np.array_equal(D[:, :, 4:7, :], want), variant
