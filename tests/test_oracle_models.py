"""Pins the oracle's model layer (reference rows a2/a4 of SURVEY.md section 8) with checkers that are
independent of it: hand-derived known answers, sympy differentiation of the reference's formulas,
finite differences of the RK4 map and scipy's adaptive integrator.  The reference holds no tests or
golden vectors for this path (parity unpinned), so these are what stands in for them."""
import numpy as np
import pytest
import sympy as sp
from scipy.integrate import solve_ivp


# ---- known answers derived by hand from the reference's formulas
def test_known_answers_3dof(oracle):
    f = oracle.model_f(0, [.001, 0, 0, 0, 0], [0, 0])
    assert np.allclose(f[:3], [-7.751937984496124e-4, 0, 0], rtol=0, atol=1e-15)
    f = oracle.model_f(0, [1.0, 0.1, 0.05, 10, 8], [0, 0])
    assert np.allclose(f[:3], [-0.033789147286821725, -3.793334274421232, -0.683244510394037], rtol=1e-13)
    f = oracle.model_f(0, [1.3, -0.05, -0.2, 20, 15], [0, 0])  # u > 1.25 branch
    assert np.allclose(f[:3], [-0.22772093023255863, 1.099470638057595, 1.3198113153911584], rtol=1e-13)


def test_known_answers_guidance_ca1(oracle):
    ak = np.pi / 2
    x0 = np.array([0.7, 0, 4.0, -ak, -ak, 0, 0, 0])  # scripts/usv_guidance_ca1/main.py:95-109
    assert np.allclose(oracle.model_f(1, x0, [0.0]), [0, 0, -0.7, 0, 0, 0.7, 0, 0], atol=1e-15)
    h, _ = oracle.model_h(1, x0, [4, 4, 4, 7, 4, 12, 4, 20] + [100] * 8)
    assert np.allclose(h, [5.656854249492381, 8.06225774829855, 12.649110640673518, 20.396078054371138]
                       + [141.4213562373095] * 4, rtol=1e-14)
    x = [0.7, 0.05, 1.0, -0.3, -0.2, 2.0, 3.0, 0.4]
    f = oracle.model_f(1, x, [0.1])
    assert np.allclose(f, [0, 0, -0.207323218556612, 0.171206086034624, 0.1, 0.625271778686587, 0.3186458893162,
                           0.171206086034624], rtol=1e-12, atol=1e-15)
    xn, _, _ = oracle.rk4_sens(1, 0.05, x, [0.1])
    assert np.allclose(xn, [0.7, 0.05, 0.989776464336302, -0.291527238388938, -0.195, 2.031195487486528,
                            3.016064999001967, 0.408472761611062], rtol=1e-12)


# ---- sympy restatement of the reference's CasADi expressions
def _dof3_sym(c, u, v, r, Tp, Ts):
    m, Iz, B = 30, sp.Rational(41, 10), sp.Rational(41, 100)
    Xud, Yvd, Yrd, Nvd, Nrd = -2.25, -23.13, -1.31, -16.41, -2.79
    Yvv, Yvr, Nrv, Nrr = -99.99, -5.49, -8.8, -3.49
    Xu = sp.Piecewise((64.55, u > 1.25), (-25, True))
    Xuu = sp.Piecewise((-70.92, u > 1.25), (0, True))
    Yv = 0.5 * (-40 * 1000 * sp.Abs(v)) * (1.1 + 0.0045 * (1.01 / 0.09) - 0.1 * (0.27 / 0.09) + 0.016 * ((0.27 / 0.09) ** 2))
    Nr = (-0.52) * sp.sqrt(u * u + v * v)
    Tu = Tp + c * Ts
    Tr = (Tp - c * Ts) * B / 2
    fu = (Tu - (-m + 2 * Yvd) * v - (Yrd + Nvd) * r * r - (-Xu * u - Xuu * sp.Abs(u) * u)) / (m - Xud)
    fv = (-(m - Xud) * u * r - (-Yv - Yvv * sp.Abs(v) - Yvr * sp.Abs(r)) * v) / (m - Yvd)
    fr = (Tr - (-2 * Yvd * u * v - (Yrd + Nvd) * r * u + Xud * u * r) - (-Nr * r - Nrv * sp.Abs(v) * r - Nrr * sp.Abs(r) * r)) / (Iz - Nrd)
    return fu, fv, fr


def _sym_model(model):
    if model == 0:
        x = sp.symbols("u v r Tp Ts", real=True)
        U = sp.symbols("U0 U1", real=True)
        fu, fv, fr = _dof3_sym(0.78, *x)
        f = [fu, fv, fr, U[0], U[1]]
    elif model == 1:
        x = sp.symbols("u v ye chie psied xned yned psi", real=True)
        U = sp.symbols("U0,", real=True)
        u, v, ye, chie, psied, xned, yned, psi = x
        beta = sp.atan2(v, u + 0.001)
        psie = chie - beta
        f = [0, 0, u * sp.sin(psie) + v * sp.cos(psie), (psied - psie) / 1.0, U[0], u * sp.cos(psi) - v * sp.sin(psi),
             u * sp.sin(psi) + v * sp.cos(psi), (psied - psie) / 1.0]
    else:
        x = sp.symbols("psi sinpsi cospsi u v r ye x1 y1 ak nedx nedy Tp Ts", real=True)
        U = sp.symbols("U0 U1", real=True)
        psi, sinpsi, cospsi, u, v, r, ye, x1, y1, ak, nedx, nedy, Tp, Ts = x
        c = 1.0
        fu, fv, fr = _dof3_sym(c, u, v, r, Tp, Ts)
        chi = psi + sp.atan2(v, u + .001)
        f = [r, sp.cos(chi) * r, -sp.sin(chi) * r, fu, fv, fr,
             -(u * sp.cos(psi) - v * sp.sin(psi)) * sp.sin(ak) + (u * sp.sin(psi) + v * sp.cos(psi)) * sp.cos(ak),
             0, 0, 0, u * sp.cos(psi) - v * sp.sin(psi), u * sp.sin(psi) + v * sp.cos(psi), U[0], U[1] / c]
    f = sp.Matrix(f)
    return x, U, f


@pytest.mark.parametrize("model", [0, 1, 2])
def test_jacobians_against_sympy(oracle, model):
    xs, Us, f = _sym_model(model)
    Jx = sp.lambdify((xs, Us), f.jacobian(sp.Matrix(xs)), "numpy")
    Ju = sp.lambdify((xs, Us), f.jacobian(sp.Matrix(Us)), "numpy")
    fn = sp.lambdify((xs, Us), f, "numpy")
    rng = np.random.default_rng(10 + model)
    nx, nu = oracle.dims(model)
    for trial in range(20):
        x = rng.normal(size=nx)
        if trial % 4 == 0:
            x[0 if model < 2 else 3] = 1.3 + abs(x[0])  # exercise the u > 1.25 branch
        U = rng.normal(size=nu)
        assert np.allclose(oracle.model_f(model, x, U), np.asarray(fn(x, U), dtype=float).ravel(), rtol=1e-12, atol=1e-13)
        jx, ju = oracle.model_jac(model, x, U)
        assert np.allclose(jx, np.asarray(Jx(x, U), dtype=float), rtol=1e-10, atol=1e-11)
        assert np.allclose(ju, np.asarray(Ju(x, U), dtype=float), rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("model", [1, 2])
def test_obstacle_rows_against_sympy(oracle, model):
    px, py, ox, oy = sp.symbols("px py ox oy", real=True)
    d = sp.sqrt((px - ox) * (px - ox) + (py - oy) * (py - oy))
    g = sp.lambdify((px, py, ox, oy), [d, sp.diff(d, px), sp.diff(d, py)], "numpy")
    rng = np.random.default_rng(5)
    nx, _ = oracle.dims(model)
    ipx, ipy = (5, 6) if model == 1 else (10, 11)
    x = rng.normal(size=nx) * 3
    p = rng.normal(size=12) * 4
    h, C = oracle.model_h(model, x, p)
    for i in range(6):
        ref = g(x[ipx], x[ipy], p[2 * i], p[2 * i + 1])
        assert np.allclose([h[i], C[i, 0], C[i, 1]], ref, rtol=1e-13)


@pytest.mark.parametrize("model,dt", [(0, 0.05), (1, 0.05), (2, 0.01)])
def test_rk4_sensitivities_are_the_derivative_of_the_rk4_map(oracle, model, dt):
    rng = np.random.default_rng(3 + model)
    nx, nu = oracle.dims(model)
    x = rng.normal(size=nx) * 0.3
    x[0 if model < 2 else 3] += 0.7
    if model != 1:
        # keep the sway speed in the regime where the explicit RK4 map is non-stiff (|v| < 0.07 at
        # dt = 0.05: the damping Yv = -19890|v| otherwise makes central differences meaningless)
        x[1 if model == 0 else 4] = 0.03
    U = rng.normal(size=nu)
    _, A, B = oracle.rk4_sens(model, dt, x, U)
    e = 1e-6
    for j in range(nx):
        d = np.zeros(nx)
        d[j] = e
        fd = (oracle.rk4_sens(model, dt, x + d, U)[0] - oracle.rk4_sens(model, dt, x - d, U)[0]) / (2 * e)
        assert np.allclose(A[:, j], fd, rtol=1e-6, atol=1e-8)
    for j in range(nu):
        d = np.zeros(nu)
        d[j] = e
        fd = (oracle.rk4_sens(model, dt, x, U + d)[0] - oracle.rk4_sens(model, dt, x, U - d)[0]) / (2 * e)
        assert np.allclose(B[:, j], fd, rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("model,dt", [(0, 0.05), (1, 0.05), (2, 0.01)])
def test_rk4_step_against_adaptive_integrator(oracle, model, dt):
    nx, nu = oracle.dims(model)
    x = np.array([1.0, 0.02, 0.05, 10.0, 8.0]) if model == 0 else (
        np.array([0.7, 0.05, 1.0, -0.3, -0.2, 2.0, 3.0, 0.4]) if model == 1 else
        np.array([0.3, np.sin(0.3), np.cos(0.3), 0.8, 0.02, 0.05, 0.5, 4, -5, np.pi / 2, 3.0, 2.0, 10.0, 8.0]))
    U = np.full(nu, 0.1)
    xn, _, _ = oracle.rk4_sens(model, dt, x, U)
    sol = solve_ivp(lambda t, y: oracle.model_f(model, y, U), (0, dt), x, rtol=1e-12, atol=1e-14)
    # one RK4 step: O(dt^5) local error; the 3-DOF block at dt = 0.05 is mildly stiff (sway damping)
    assert np.allclose(xn, sol.y[:, -1], rtol=0, atol=5e-5 if model == 0 else 5e-6)


# structural identities the HIP kernels rely on (csrc/models.hpp OUT_UNIT / IN_UNIT): checked on the
# oracle's dense RK4 sensitivities, which know nothing about them
STRUCT = {1: dict(out_unit=[0, 1], in_unit_x=[2, 5, 6]),
          2: dict(out_unit=[7, 8, 9], in_unit_x=[1, 2, 6, 7, 8, 10, 11])}


@pytest.mark.parametrize("model,dt", [(1, 0.05), (2, 0.01)])
def test_structural_unit_rows_and_columns(oracle, model, dt):
    rng = np.random.default_rng(42 + model)
    nx, nu = oracle.dims(model)
    for _ in range(10):
        x = rng.normal(size=nx)
        x[0 if model == 1 else 3] += 0.7
        if model == 2:
            x[4] *= 0.05
        U = rng.normal(size=nu)
        _, A, B = oracle.rk4_sens(model, dt, x, U)
        for j in STRUCT[model]["out_unit"]:      # x+_j = x_j exactly
            e = np.zeros(nx); e[j] = 1.0
            assert np.array_equal(A[j], e) and np.array_equal(B[j], np.zeros(nu))
        for c in STRUCT[model]["in_unit_x"]:     # state c feeds nothing but itself
            e = np.zeros(nx); e[c] = 1.0
            assert np.array_equal(A[:, c], e)


# the full pattern of the discrete sensitivities the kernels pack by (csrc/models.hpp SENS / DIAG_ONE, MatPack in
# csrc/params.hpp), z = [u; x]: entries outside it must be EXACT zeros - or the exact 1 on a DIAG_ONE diagonal - in the
# oracle's dense sensitivities, for one RK4 step and for several
_CORE2 = (1 << 0) | (1 << 1) | (1 << 5) | (1 << 6) | (1 << 7) | (1 << 14) | (1 << 15)
_CH1 = (1 << 0) | (1 << 1) | (1 << 2) | (1 << 4) | (1 << 5)
SENS = {0: ([0x7f, 0x7f, 0x7f, 1 << 0, 1 << 1], (1 << 3) | (1 << 4)),
        1: ([0, 0, _CH1, _CH1, 1 << 0, _CH1 | (1 << 8), _CH1 | (1 << 8), _CH1], 0xff & ~(1 << 3)),
        2: ([_CORE2, _CORE2 | 4, _CORE2 | 4, _CORE2, _CORE2, _CORE2, _CORE2 | 4 | (1 << 11), 0, 0, 0, _CORE2 | 4, _CORE2 | 4,
             1 << 0, 1 << 1], 0x3fff & ~((1 << 3) | (1 << 4) | (1 << 5)))}


@pytest.mark.parametrize("model,dt,steps", [(0, 0.05, 1), (1, 0.05, 1), (2, 0.01, 1), (2, 0.05, 5), (1, 0.05, 3)])
def test_sensitivity_pattern(oracle, model, dt, steps):
    rng = np.random.default_rng(7 + model)
    nx, nu = oracle.dims(model)
    sens, diag_one = SENS[model]
    seen = np.zeros((nx, nu + nx), dtype=bool)
    for _ in range(10):
        x = rng.normal(size=nx)
        x[{0: 0, 1: 0, 2: 3}[model]] += 0.7
        if model == 2:
            x[4] *= 0.05
        U = rng.normal(size=nu)
        _, A, B = oracle.erk_sens(model, dt, steps, x, U)
        BA = np.hstack([B, A])
        for j in range(nx):
            for c in range(nu + nx):
                if (sens[j] >> c) & 1:
                    seen[j, c] |= BA[j, c] != 0.0
                elif c == nu + j and (diag_one >> j) & 1:
                    assert BA[j, c] == 1.0, (j, c, BA[j, c])
                else:
                    assert BA[j, c] == 0.0, (j, c, BA[j, c])
        for j in range(nx):   # the diagonal is either stored or exactly one, never both
            assert ((sens[j] >> (nu + j)) & 1) != ((diag_one >> j) & 1)
    # and the pattern is tight: every stored entry is non-zero somewhere
    for j in range(nx):
        for c in range(nu + nx):
            if (sens[j] >> c) & 1:
                assert seen[j, c], (j, c)
