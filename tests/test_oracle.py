"""CPU tests of the oracle: Philox known answers, the reference's edge-case ledger (SURVEY
Appendix A), agreement of the three restatements (C faithful, C fast twin, plain Python), and
the RNG-free Markov-propagation check that ties the sampler to the reference's distribution."""
import math

import numpy as np
import pytest

import py_restatement as PY
from conftest import SIM_SEED, TABLE_SEED

T = 24


# ------------------------------------------------------------------ RNG
def test_philox_known_answers(O):
    # Random123 kat_vectors for philox4x32-10
    assert O.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0]) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_uniform_mapping(O):
    seed, car, step = 0x0123456789ABCDEF, (5 << 32) | 7, 11
    w = O.philox4x32_10([7, 5, step, 0], [0x89ABCDEF, 0x01234567])
    u0, u1 = O.uniforms(seed, car, step, 0)
    assert u0 == ((w[1] << 32 | w[0]) >> 11) * 2.0 ** -53
    assert u1 == ((w[3] << 32 | w[2]) >> 11) * 2.0 ** -53
    assert 0.0 <= u0 < 1.0 and 0.0 <= u1 < 1.0


def test_exp_neg_accuracy(O):
    for y in np.concatenate([np.linspace(0, 2, 201), np.linspace(2, 60, 59), [100.0, 700.0, 744.0]]):
        got, want = O.exp_neg(y), math.exp(-y)
        assert abs(got - want) <= 4e-16 * want + 5e-324, y
    assert O.exp_neg(800.0) == 0.0


# ------------------------------------------------------------------ sampler semantics
def test_initial_placement(O):
    st, tr = O.initializestates(12, 4, T)                       # A-2
    assert st[:, 0].tolist() == [1] * 4 + [2] * 4 + [3] * 4
    assert (st[:, 1:] == 0).all() and (tr == 0).all() and tr.shape == (12, T, 4)
    st2, _ = O.initializestates(5, 4, T, car_offset=6)          # a shard keeps global placement
    assert st2[:, 0].tolist() == [2, 2, 3, 3, 3]


def _uniform_fns(O, seed, car_offset=0):
    ivp = lambda i, t: O.uniforms(seed, car_offset + i - 1, t - 1, 0)
    res = lambda i, t: O.uniforms(seed, car_offset + i - 1, T - 1 + t - 1, 0)
    return ivp, res


@pytest.mark.parametrize("Z,cpz,seed", [(3, 4, 1), (5, 6, SIM_SEED), (7, 3, 99)])
def test_c_oracle_equals_python_restatement(O, Z, cpz, seed):
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, TABLE_SEED + Z), O.synth_p_dest_dense(Z, T, TABLE_SEED + Z)
    p_dest[1, :, 3] = 0.0                                       # a zero row at one hour (A-8)
    p_dest[2, :, :] *= 0.75                                     # rows that fall through (A-7 / D1)
    ivp, res = _uniform_fns(O, seed)
    st, tr = O.initializestates(C, cpz, T)
    init_c = O.solveinitialvalueproblem(st, tr, p_drive, p_dest, C, Z, seed)
    st_p, tr_p = O.initializestates(C, cpz, T)
    init_p = PY.solveinitialvalueproblem(st_p, tr_p, p_drive, p_dest, C, Z, T, ivp)
    assert np.array_equal(init_c, init_p)
    assert np.array_equal(st, st_p) and np.array_equal(tr[:, :, :2], tr_p[:, :, :2])
    st, tr = O.initializestates(C, cpz, T)
    st[:, 0] = init_c
    O.resampling(st, tr, C, Z, p_drive, p_dest, None, None, seed)
    st_p, tr_p = O.initializestates(C, cpz, T)
    st_p[:, 0] = init_p
    PY.resampling(st_p, tr_p, p_drive, p_dest, C, Z, T, res)
    assert np.array_equal(st, st_p) and np.array_equal(tr, tr_p)
    pk, dr, dens = O.histogram(Z, st, tr)
    pk_p, dr_p = PY.histogram(Z, T, st_p, tr_p, C)
    assert np.array_equal(pk, pk_p) and np.array_equal(dr, dr_p)
    assert np.array_equal(dens, pk / C)
    assert (pk.sum(axis=0) == C).all()                          # every car is somewhere every hour
    # fast twin == faithful
    r = O.fast_run(p_drive, O.build_cdf(p_dest), C, seed, np.arange(C) // cpz + 1, want_state=True)
    assert np.array_equal(r["zone0"], init_c) and np.array_equal(r["state"], st)
    assert np.array_equal(r["parking"], pk.astype(np.int64)) and np.array_equal(r["driving"], dr.astype(np.int64))


def test_bernoulli_edges(O):
    """p = 0 never drives (u = 0 has probability 2^-53), p = 1 always, NaN never (A-3, A-6)."""
    Z, cpz = 4, 50
    C = Z * cpz
    p_drive = np.zeros((Z, T), order="F")
    p_drive[1, :] = 1.0
    p_drive[2, :] = np.nan
    p_drive[3, :] = 0.5
    p_dest = np.zeros((Z, Z, T), order="F")
    for o in range(Z):
        p_dest[o, o, :] = 1.0                                   # always "drive" to the own zone
    st, tr = O.initializestates(C, cpz, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, None, None, SIM_SEED)
    pk, dr, _ = O.histogram(Z, st, tr)
    assert (dr[0] == 0).all() and (dr[2] == 0).all()
    assert (dr[1] == cpz).all()
    assert 0 < dr[3].sum() < cpz * T
    assert (pk == cpz).all()                                    # self loops: nobody moves


def test_zero_row_drives_to_origin_and_counts_as_driving(O):
    Z, cpz = 3, 40                                              # A-8
    C = Z * cpz
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")                     # all rows zero
    dm = np.zeros((Z, Z, T, 2), order="F")
    dist = np.ones((Z, Z), order="F")
    st, tr = O.initializestates(C, cpz, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, dm, dist, SIM_SEED)
    assert (tr[:, :, 0] == 1).all()
    assert (st == st[:, [0]]).all()
    assert (tr[:, :, 2] == 300).all() and (tr[:, :, 3] == 1).all()      # resampling.jl:58-60
    assert O.averagedrivingtime(C, 0.0, tr) == pytest.approx(300 * C * T / (C * T * 3600))
    assert O.sum_travel_time_q16(tr) == 300 * 65536 * C * T


def test_categorical_skips_zero_probability_zones_and_hits_boundaries(O):
    """first j with u <= cumsum_j; zero-probability zones are never chosen (A-9)."""
    Z = 6
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[:, 1, :] = 0.25
    p_dest[:, 4, :] = 0.75                                      # zones 2 and 5 (1-based) only
    C = 600
    st, tr = O.initializestates(C, C // Z, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, None, None, 7)
    dests = tr[:, :, 1]
    assert set(np.unique(dests)) == {2.0, 5.0}
    u = np.array([[O.uniforms(7, i, T - 1 + t, 0)[1] for t in range(T)] for i in range(C)])
    assert np.array_equal(dests == 2.0, u <= 0.25)


def test_fall_through_policy_d1(O):
    """Row sums to 0.5: u > 0.5 -> LAST zone with p > 0 (the reference would leave 0 and crash)."""
    Z = 5
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[:, 1, :] = 0.2
    p_dest[:, 3, :] = 0.3                                       # last positive zone is 4 (1-based); zone 5 has p = 0
    C = 500
    st, tr = O.initializestates(C, C // Z, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, None, None, 3)
    u = np.array([[O.uniforms(3, i, T - 1 + t, 0)[1] for t in range(T)] for i in range(C)])
    want = np.where(u <= 0.2, 2.0, 4.0)
    assert np.array_equal(tr[:, :, 1], want)
    r = O.fast_run(p_drive, O.build_cdf(p_dest), C, 3, np.arange(C) // (C // Z) + 1, do_ivp=False, want_state=True)
    assert np.array_equal(r["state"], st)


def test_hour_T_is_sampled_but_not_applied(O):
    Z, cpz = 4, 30                                              # A-11
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, 5), O.synth_p_dest_dense(Z, T, 5)
    st, tr = O.initializestates(C, cpz, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, None, None, SIM_SEED)
    assert st.shape[1] == T and tr[:, T - 1, 0].sum() > 0       # hour-24 flags exist and are counted
    _, dr, _ = O.histogram(Z, st, tr)
    assert dr[:, T - 1].sum() == tr[:, T - 1, 0].sum()
    for t in range(T - 1):
        assert np.array_equal(st[:, t + 1], tr[:, t, 1].astype(np.int64))


# ------------------------------------------------------------------ tables
def test_createpdrive_semantics(O):
    Z = 4
    dm = np.zeros((Z, Z, T, 2), order="F")
    dist = np.full((Z, Z), 2.0, order="F")
    hours = np.arange(1, T + 1, dtype=np.float64)
    dm[0, 1, :, 0] = 100 * hours                                # zone 1: one destination, rising
    dm[0, 2, :, 0] = 300 * hours                                #         a second one (mean over non-zero only, A-4)
    dm[1, 0, :, 0] = 50.0                                       # zone 2: constant -> max == min -> 0/0 -> NaN (A-3)
    dm[2, 3, 5:, 0] = 10 * hours[5:]                            # zone 3: hours 1-5 empty -> NaN poisons -> 0 (A-3)
    p = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    ms = (100 * hours / 2 + 300 * hours / 2) / 2
    want = 0.1 + 0.8 * ((ms - ms.min()) / (ms.max() - ms.min())) ** 0.5
    np.testing.assert_allclose(p[0], want, rtol=1e-15)
    assert p[0, 0] == 0.1 and p[0, T - 1] == 0.9
    assert np.isnan(p[1]).all()
    assert (p[2] == 0).all() and (p[3] == 0).all()


def test_createpdestin_semantics(O):
    Z = 3
    dm = np.zeros((Z, Z, T, 2), order="F")
    hours = np.arange(1, T + 1, dtype=np.float64)
    dm[0, 1, :, 0] = 10 * hours                                 # present all day: min = 10
    dm[0, 2, 12:, 0] = 5 * hours[12:]                           # missing hours -> min = 0 (A-5)
    p = O.createpdestin(dm, Z, T, 2)
    w1 = ((10 * hours - 10) / (240 - 10)) ** 2
    w2 = np.where(hours > 12, (5 * hours / 120) ** 2, 0.0)
    nf = w1 + w2
    np.testing.assert_allclose(p[0, 1, 1:], (w1 / nf)[1:], rtol=1e-15)
    np.testing.assert_allclose(p[0, 2, 1:], (w2 / nf)[1:], rtol=1e-15)
    assert (p[0, :, 0] == 0).all()                              # hour 1: both weights 0 -> row stays 0
    assert (p[1] == 0).all() and (p[2] == 0).all()
    np.testing.assert_allclose(p[0, :, 1:].sum(axis=0), 1.0, rtol=1e-15)
    pf = O.createpdestin(dm, Z, T, 2.0)                          # Float64 exponent: pow(), same to an ulp
    np.testing.assert_allclose(pf, p, rtol=1e-15)
    dm[1, 0, :, 0] = 7.0                                        # constant all day -> 0/0 = NaN (A-5)
    assert np.isnan(O.createpdestin(dm, Z, T, 2)[1, 0]).all()


# ------------------------------------------------------------------ reductions
def test_trafficactivity_and_averagedrivingtime(O):
    Z = 3
    driving = np.zeros((Z, T), order="F")
    driving[0] = np.arange(T)
    driving[2] = 2 * np.arange(T)
    act = O.trafficactivity(driving)                            # A-15
    np.testing.assert_allclose(act, np.arange(T) / (T - 1), rtol=1e-15)
    assert np.isnan(O.trafficactivity(np.ones((Z, T), order="F"))).all()   # flat -> 0/0
    C = 10
    tr = np.zeros((C, T, 4), order="F")
    tr[:, :, 2] = 360.0
    assert O.averagedrivingtime(C, 0.25, tr) == pytest.approx(0.25 + 0.1)  # A-16


def test_correctparameters_else_nesting(O):
    assert O.correctparameters(-0.1, 1.2, 0.1, 0.9) == (0.0, 1.0)
    assert O.correctparameters(0.95, 0.05, 0.1, 0.9) == (0.9, 0.1)
    assert O.correctparameters(0.3, 0.6, 0.1, 0.9) == (0.3, 0.6)
    # p_max < 0 makes p_min_next > p_max true only in the else branch: -1 is clamped to 0 first
    assert O.correctparameters(-1.0, 0.5, 0.1, -0.5) == (0.0, 0.5)


# ------------------------------------------------------------------ RNG-free tie to the reference
def test_sampled_density_matches_markov_propagation(O):
    """pi_{t+1} = pi_t M_t with M_t = (1-p_drive) I + p_drive P_t (README.md:457-468 of the
    reference).  Independent of any RNG contract: the only check that ties the restatement to
    the Julia program's distribution."""
    Z, cpz = 24, 4000
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, 11), O.synth_p_dest_dense(Z, T, 11)
    p_dest[3, :, 5] = 0.0                                       # a zero row: mass stays (src/resampling.jl:35-36)
    r = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, np.arange(C) // cpz + 1, do_ivp=False)
    pis = O.markov_expected_density(p_drive, p_dest, np.full(Z, 1.0 / Z))
    dens = (r["parking"] / C).T                                 # [t][z]
    sigma = np.sqrt(pis * (1 - pis) / C)
    zscore = np.abs(dens - pis) / np.maximum(sigma, 1e-12)
    assert zscore[1:].max() < 5.5, zscore.max()
    assert np.array_equal(dens[0], pis[0])
    exp_drive = (pis * np.nan_to_num(p_drive.T)) * C            # E[driving[z,t]]
    zd = np.abs(r["driving"].T - exp_drive) / np.sqrt(np.maximum(exp_drive, 1.0))
    assert zd.max() < 5.5


def test_truncated_normal_kit_against_scipy(O):
    """The sampler's deterministic f64 kit (no libm: +,-,*,/ and bits, so that the GPU reproduces it bit for bit) against scipy:
    ln, sqrt, erf (Cody 1969), Phi^-1 (Wichura's PPND16) -- and the draw itself: inverse CDF of N(mu, sigma) truncated to
    [0.9 mu, 1.1 mu], monotone in u, inside the window for every sigma."""
    import math
    from scipy import stats
    L = O.lib()
    xs = np.concatenate([np.linspace(1e-6, 0.46875, 200), np.linspace(0.46876, 4, 400), np.linspace(4.0001, 7, 100)])
    assert max(abs(L.orc_erf(float(x)) - math.erf(x)) for x in xs) < 3e-16
    ls = np.concatenate([10.0 ** np.linspace(-300, 0, 400)[:-1], np.linspace(0.001, 0.5, 300)])
    assert max(abs(L.orc_log(float(x)) - math.log(x)) / abs(math.log(x)) for x in ls) < 5e-16
    assert max(abs(L.orc_sqrt(float(x)) - math.sqrt(x)) / math.sqrt(x) for x in 10.0 ** np.linspace(-10, 3, 300)) < 5e-16
    for r in np.concatenate([10.0 ** np.linspace(-15.5, -1.2, 300), np.linspace(0.075, 0.5, 200)[:-1]]):
        q = 0.5 - r                               # Phi^-1(1 - r): the upper tail, conditioned on r itself (0.5 - q is exact)
        want = stats.norm.isf(0.5 - q)
        assert abs(L.orc_ppnd(float(q)) - want) <= 2e-15 * max(1.0, abs(want)), r
        assert L.orc_ppnd(float(-q)) == -L.orc_ppnd(float(q))
    for sigma in (400.0, 100.0, 60.0, 20.0, 5.0):     # windows of +-0.25 ... +-20 sigma (the tails of PPND16 from +-1.44 sigma on)
        mu = 1000.0
        E = L.orc_truncnormal_mass(mu, sigma)
        a = 0.1 * mu / sigma
        assert abs(E - (stats.norm.cdf(a) - stats.norm.cdf(-a))) < 1e-15
        u = (np.arange(20001) + 0.5) / 20001
        x = np.array([L.orc_truncnormal_draw(float(v), mu, sigma, E) for v in u])
        assert x.min() >= 0.9 * mu and x.max() <= 1.1 * mu and (np.diff(x) >= 0).all()
        tn = stats.truncnorm(-a, a, loc=mu, scale=sigma)
        assert np.abs(x - tn.ppf(u)).max() < 1e-9
    assert L.orc_truncnormal_draw(0.3, 5.0, 0.0, 1.0) == 5.0       # sigma 0: the mean


def test_truncated_normal_moments(O):
    """The build's own truncated-normal sampler (the reference's is Distributions.jl, absent):
    window, symmetric mean, and the variance of N(mu, sigma) truncated at +-1 sigma."""
    Z, C = 2, 20000
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[0, 1, :] = 1.0
    p_dest[1, 0, :] = 1.0
    dm = np.zeros((Z, Z, T, 2), order="F")
    dm[:, :, :, 0] = 1000.0
    dm[:, :, :, 1] = 0.0                                        # sigma == 0 -> 0.1 mu (A-13)
    dist = np.full((Z, Z), 10.0, order="F")
    st, tr = O.initializestates(C, C // Z, T)
    O.resampling(st, tr, C, Z, p_drive, p_dest, dm, dist, 5)
    tt, dd = tr[:, :, 2].ravel(), tr[:, :, 3].ravel()
    assert tt.min() >= 900.0 and tt.max() <= 1100.0 and dd.min() >= 9.0 and dd.max() <= 11.0
    assert abs(tt.mean() - 1000.0) < 0.5
    # variance of a standard normal truncated to [-1,1]: 1 - 2*phi(1)/(2*Phi(1)-1) = 0.29112
    assert abs(tt.var() / 100.0 ** 2 - 0.29112) < 0.005
    assert abs(dd.var() / 1.0 ** 2 - 0.29112) < 0.005
