"""GPU parity: the HIP path (through the C ABI) against the CPU oracle, bit-exact for every
integer result.  Run on the GPU box: python -m pytest tests -m gpu."""
import os

import numpy as np
import pytest

from conftest import SIM_SEED, TABLE_SEED

pytestmark = pytest.mark.gpu

KERNELS = [pytest.param(0, id="auto"), pytest.param(1, id="car"), pytest.param(2, id="zone_lds"), pytest.param(5, id="zone_grouped")]


def _set_kernel(s, kernel):
    s.set_kernel(kernel)


def _tables(O, Z, T=24, seed=TABLE_SEED):
    return O.synth_p_drive(Z, T, seed), O.synth_p_dest_dense(Z, T, seed)


def _zone0(C, cpz):
    return np.arange(C, dtype=np.int64) // cpz + 1


def test_device_is_gfx950(cpm):
    info = cpm.device_info(0)
    assert "gfx950" in info["name"], info


def test_cdf_is_the_sequential_sum(cpm, O):
    Z, T = 100, 24  # not a multiple of 16: exercises the row padding
    _, p_dest = _tables(O, Z, T)
    cdf = O.build_cdf(p_dest)
    with cpm.Sampler(Z, T) as s:
        s.set_p_dest(p_dest)
        for (o, t) in [(1, 1), (Z, T), (37, 5), (64, 24), (65, 1)]:
            assert np.array_equal(s.get_cdf_row(o, t), cdf[t - 1, o - 1]), (o, t)


def test_device_synth_tables_equal_oracle_synth(cpm, O):
    Z, T = 130, 24
    p_drive, p_dest = _tables(O, Z, T)
    cdf = O.build_cdf(p_dest)
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED)
        assert np.array_equal(s.get_p_drive(), p_drive)
        for (o, t) in [(1, 1), (Z, T), (77, 13), (128, 2), (129, 24)]:
            assert np.array_equal(s.get_cdf_row(o, t), cdf[t - 1, o - 1]), (o, t)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("Z,cpz", [(5, 3), (37, 11), (64, 64), (256, 100), (513, 7)])
def test_ivp_and_counts_bit_exact(cpm, O, kernel, Z, cpz):
    T, C = 24, Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), want_state=True)
    with cpm.Sampler(Z, T) as s:
        _set_kernel(s, kernel)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert np.array_equal(s.get_state(), _zone0(C, cpz))
        init = s.solve_ivp(SIM_SEED)
        assert np.array_equal(init, ref["zone0"])
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"])
        assert np.array_equal(r["driving"], ref["driving"])
        assert (r["parking"].sum(axis=0) == C).all()
        # the resample leaves the initial state untouched: a second call gives the same counts
        r2 = s.resample(SIM_SEED)
        assert np.array_equal(r2["parking"], ref["parking"]) and np.array_equal(r2["driving"], ref["driving"])
        # densities: same count / C in f64 on both sides -> well inside the 1e-6 relative bound
        dens = r["parking"] / C
        np.testing.assert_allclose(dens, ref["parking"] / C, rtol=1e-6, atol=0)


def test_compat_matrices_equal_the_faithful_oracle(cpm, O):
    """state_matrix / transition_matrix (all four columns, travel time and distance included)
    against the three-pass restatement of src/resampling.jl."""
    Z, T, cpz = 24, 24, 9
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.5)
    p_drive = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    p_dest = O.createpdestin(dm, Z, T, 2)
    st, tr = O.initializestates(C, cpz, T)
    init = O.solveinitialvalueproblem(st, tr, p_drive, p_dest, C, Z, SIM_SEED)
    st, tr = O.initializestates(C, cpz, T)
    st[:, 0] = init
    O.resampling(st, tr, C, Z, p_drive, p_dest, dm, dist, SIM_SEED)
    pk, dr, _ = O.histogram(Z, st, tr)
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.set_datamatrix(dm, dist)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), init)
        r = s.resample(SIM_SEED, travel=True, want_state=True, want_trans=True)
    assert np.array_equal(r["state"], st)
    assert np.array_equal(r["trans"][:, :, 0], tr[:, :, 0])
    assert np.array_equal(r["trans"][:, :, 1], tr[:, :, 1])
    assert np.array_equal(r["trans"][:, :, 2], tr[:, :, 2])  # bit-exact: same +,-,*,/ sequence
    assert np.array_equal(r["trans"][:, :, 3], tr[:, :, 3])
    assert np.array_equal(r["parking"], pk.astype(np.int64))
    assert np.array_equal(r["driving"], dr.astype(np.int64))
    assert r["sum_tt_q16"] == O.sum_travel_time_q16(tr)
    a_ref = O.averagedrivingtime(C, 0.0, tr)
    a_gpu = (r["sum_tt_q16"] / 65536.0) / (C * T * 3600.0)
    assert abs(a_gpu - a_ref) <= 1e-6 * abs(a_ref)


@pytest.mark.parametrize("kernel", KERNELS)
def test_travel_time_sum_bit_exact(cpm, O, kernel):
    Z, T, cpz = 40, 24, 50
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.3)
    p_drive = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    p_dest = O.createpdestin(dm, Z, T, 2)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
    with cpm.Sampler(Z, T) as s:
        _set_kernel(s, kernel)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.set_datamatrix(dm, dist)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        r = s.resample(SIM_SEED, travel=True)
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["sum_tt_q16"] == ref["sum_tt_q16"]


@pytest.mark.parametrize("cpz", [700, 2600, 4500])
def test_travel_time_sum_for_every_block_size_of_the_travel_kernel(cpm, O, cpz):
    """The grouped path's travel kernel runs one wave per (origin zone, hour) for mean buckets up to 2048 cars, 128 threads up to
    4096 and 256 above (cpm::travel_block): the time sum is bit-exact in each, with and without the runs of all hours kept."""
    Z, T = 24, 24
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED + 3, density=0.5)
    p_drive = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    p_dest = O.createpdestin(dm, Z, T, 2)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
    with cpm.Sampler(Z, T) as s:
        s.set_kernel(cpm.CPM_KERNEL_ZONE_GROUPED)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.set_datamatrix(dm, dist)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        r = s.resample(SIM_SEED, travel=True)
        assert s.get_info(1) == cpm.CPM_KERNEL_ZONE_GROUPED
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["sum_tt_q16"] == ref["sum_tt_q16"]


def test_travel_times_when_a_row_of_the_travel_table_does_not_fit_lds(cpm, O):
    """Dense datamatrix at Z = 2,304: a row's non-zero cells (2,304 x 16 B) are more than the 32 KB the travel kernel stages, so the
    grouped path builds the dense origin-major table and gathers from it (k_build_travel_table, k_grouped_travel<false>) -- the
    same time sum as the oracle's, bit for bit."""
    Z, T, cpz = 2304, 3, 40
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED + 5, density=1.0)
    p_drive = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    p_dest = O.createpdestin(dm, Z, T, 2)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
    with cpm.Sampler(Z, T) as s:
        s.set_kernel(cpm.CPM_KERNEL_ZONE_GROUPED)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.set_datamatrix(dm, dist)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        r = s.resample(SIM_SEED, travel=True)
        assert s.get_info(1) == cpm.CPM_KERNEL_ZONE_GROUPED
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["sum_tt_q16"] == ref["sum_tt_q16"]


@pytest.mark.parametrize("kernel", KERNELS)
def test_edge_rows(cpm, O, kernel):
    """Zero rows (dest = origin, still counted as driving: Appendix A-8), p_drive 0 / 1 / NaN zones
    (A-3, A-6), rows that sum to less than one (fall-through, deviation D1), sparse rows with
    leading / trailing zero-probability zones (A-9)."""
    Z, T, cpz = 48, 24, 40
    C = Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    p_drive[0, :] = 0.0
    p_drive[1, :] = 1.0
    p_drive[2, :] = np.nan
    p_dest[3, :, :] = 0.0                    # zero row, every hour
    p_dest[4, :, :] *= 0.5                   # sums to 0.5: half the draws fall through -> last zone with p > 0
    p_dest[4, Z - 3:, :] = 0.0
    p_dest[5, :, :] = 0.0
    p_dest[5, 7, :] = 1.0                    # point mass
    p_dest[6, :10, :] = 0.0                  # leading zeros
    p_dest[6, :, :] /= p_dest[6, :, :].sum(axis=0, keepdims=True)
    p_dest = np.asfortranarray(p_dest)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    # the faithful three-pass form agrees with the fast twin on the same edge cases
    st, tr = O.initializestates(C, cpz, T)
    init = O.solveinitialvalueproblem(st, tr, p_drive, p_dest, C, Z, SIM_SEED)
    assert np.array_equal(init, ref["zone0"])
    with cpm.Sampler(Z, T) as s:
        _set_kernel(s, kernel)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["driving"][2].sum() == 0       # NaN zone never drives


def test_nan_table_is_rejected(cpm, O):
    Z, T = 20, 24
    _, p_dest = _tables(O, Z, T)
    p_dest[3, 4, 5] = np.nan
    with cpm.Sampler(Z, T) as s:
        with pytest.raises(cpm.CpmError) as e:
            s.set_p_dest(p_dest)
        assert e.value.status == -4


def test_call_order_errors(cpm, O):
    Z, T = 8, 24
    p_drive, p_dest = _tables(O, Z, T)
    with cpm.Sampler(Z, T) as s:
        with pytest.raises(cpm.CpmError):
            s.resample(1)                                                   # no tables
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        with pytest.raises(cpm.CpmError):
            s.resample(1)                                                   # no cars
        with pytest.raises(cpm.CpmError):
            s.init_states(Z * 4 + 4, 4)                                     # more cars than zones * cpz
        s.init_states(Z * 4, 4)
        with pytest.raises(cpm.CpmError):
            s.resample(1, travel=True)                                      # no datamatrix
        with pytest.raises(cpm.CpmError):
            s.set_state(np.full(Z * 4, Z + 1, dtype=np.int64))              # zone out of range


@pytest.mark.parametrize("kernel", KERNELS)
def test_shards_sum_to_the_single_run(cpm, O, kernel):
    """Philox is keyed by the global car id: any partition of the cars gives the same total
    (SURVEY 8e); includes an empty shard and ragged shard sizes."""
    from carparkingmaps_amd.distributed import shard_range
    Z, T, cpz = 61, 24, 23
    C = Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    for world in (1, 3, 8):
        pk = np.zeros((Z, T), dtype=np.int64)
        dr = np.zeros((Z, T), dtype=np.int64)
        with cpm.Sampler(Z, T) as s:
            _set_kernel(s, kernel)
            s.set_p_drive(p_drive)
            s.set_p_dest(p_dest)
            for rank in range(world):
                b, n = shard_range(C, rank, world)
                s.init_states(C, cpz, b, n)
                s.solve_ivp(SIM_SEED, want=False)
                r = s.resample(SIM_SEED)
                pk += r["parking"]
                dr += r["driving"]
            s.init_states(C, cpz, C, 0)  # empty shard
            r = s.resample(SIM_SEED)
            assert r["parking"].sum() == 0
        assert np.array_equal(pk, ref["parking"]) and np.array_equal(dr, ref["driving"]), world


def test_table_builders_against_the_oracle(cpm, O):
    """createpdrive / createpdestin on device vs the restatement.  Bit-exact for integer exponents
    (main.jl:38: e_dest = 2); where the reference's `^` goes through libm pow (Float64 exponents) the
    device evaluates 0.5 / 1 / 2 exactly (sqrt, x, x*x) and the rest with its own pow, so the two
    sides agree to a few ulp (glibc pow is not correctly rounded either)."""
    Z, T = 70, 24
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.2)
    dm[5, :, :, :] = 0.0          # zone with no outgoing data: NaN mean -> p_drive 0 all day (A-3)
    dm[6, :, 3, :] = 0.0          # one empty hour -> NaN poisons the whole zone (A-3)
    dm = np.asfortranarray(dm)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        for e_drive in (0.5, 1.0, 2.0, 0.7):
            got = s.build_p_drive(0.1, 0.9, e_drive)
            want = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, e_drive)
            np.testing.assert_allclose(got, want, rtol=4e-16 if e_drive != 0.7 else 1e-14, atol=0, equal_nan=True)
            assert (got[5] == 0).all() and (got[6] == 0).all()
        for e_dest in (2, 2.0, 1, 3, 4.0, 0.5):
            got = s.build_p_dest(e_dest)
            want = O.createpdestin(dm, Z, T, e_dest)
            if isinstance(e_dest, int):  # Float64^Int: repeated multiplication, exact on both sides
                assert np.array_equal(got, want), e_dest
            else:
                np.testing.assert_allclose(got, want, rtol=1e-13, atol=0)


def test_reference_call_surface_main_jl(cpm, O, tmp_path):
    """main.jl:79-102 through the mirrored call surface, against the oracle run the same way."""
    Z, T, cpz = 30, 24, 20
    C = Z * cpz
    cpm.params.cars_per_zone, cpm.params.T, cpm.params.seed = cpz, T, SIM_SEED
    cpm.params.e_drive, cpm.params.e_dest, cpm.params.p_min, cpm.params.p_max = 1.0, 2, 0.1, 0.9
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.4)
    try:
        p_drive = cpm.createpdrive(dm, dist, Z)
        p_dest = cpm.createpdestin(dm, Z)
        assert np.array_equal(p_drive, O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 1.0))
        assert np.array_equal(p_dest, O.createpdestin(dm, Z, T, 2))
        state_matrix, transition_matrix = cpm.initializestates(C)
        st, tr = O.initializestates(C, cpz, T)
        assert np.array_equal(state_matrix, st) and np.array_equal(transition_matrix, tr)
        initial_state = cpm.solveinitialvalueproblem(state_matrix, transition_matrix, p_drive, p_dest, C, Z)
        init = O.solveinitialvalueproblem(st, tr, p_drive, p_dest, C, Z, SIM_SEED)
        assert np.array_equal(initial_state, init)
        state_matrix[:, 0] = initial_state
        st, tr = O.initializestates(C, cpz, T)
        st[:, 0] = init
        state_matrix, transition_matrix = cpm.resampling(state_matrix, transition_matrix, C, Z, p_drive, p_dest, dm, dist)
        O.resampling(st, tr, C, Z, p_drive, p_dest, dm, dist, SIM_SEED)
        assert np.array_equal(state_matrix, st) and np.array_equal(transition_matrix, tr)
        a = cpm.averagedrivingtime(C, 0, transition_matrix)
        assert abs(a - O.averagedrivingtime(C, 0.0, tr)) <= 1e-12 * a
        dens, act = cpm.saveresults(Z, state_matrix, transition_matrix, str(tmp_path), "data.csv", C)
        pk, dr, d_ref = O.histogram(Z, st, tr)
        assert np.array_equal(dens, d_ref)
        np.testing.assert_allclose(act, O.trafficactivity(dr), rtol=1e-15)
        lines = (tmp_path / "results_parkingdensities_data.csv").read_text().splitlines()
        assert lines[0].split(",")[0] == "t = 1h" and len(lines) == Z + 1
        fast = cpm.run_dataset(dm, dist, Z, travel=True)
        assert np.array_equal(fast["parking"], pk.astype(np.int64))
        assert abs(fast["A_drive_increment"] - a) <= 1e-6 * a
    finally:
        cpm.release()
        cpm.params.__init__()


def test_headline_config_full_size(cpm, O):
    """BASELINE.json configs[2] at full size (Z = 4,096, 1,000 cars/zone, C = 4,096,000): tables built
    on the device by cpm_synth_tables, IVP + resample on the device, against the oracle's fast twin
    on oracle-built tables -- post-IVP state and both count tensors bit-exact -- plus the
    size-independent properties (every hour holds all C cars; repeatable; shards add up)."""
    from carparkingmaps_amd.distributed import shard_range
    Z, T, cpz = 4096, 24, 1000
    C = Z * cpz
    p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
    p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED)
    cdf = O.build_cdf(p_dest)
    del p_dest
    ref = O.fast_run(p_drive, cdf, C, SIM_SEED, _zone0(C, cpz))
    del cdf
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED)
        assert np.array_equal(s.get_p_drive(), p_drive)
        s.init_states(C, cpz)
        init = s.solve_ivp(SIM_SEED)
        assert np.array_equal(init, ref["zone0"])
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"])
        assert np.array_equal(r["driving"], ref["driving"])
        assert (r["parking"].sum(axis=0) == C).all()
        np.testing.assert_allclose(r["parking"] / C, ref["parking"] / C, rtol=1e-6, atol=0)
        for kernel in (1, 2):  # the other kernels agree at full size too
            _set_kernel(s, kernel)
            r2 = s.resample(SIM_SEED)
            assert np.array_equal(r2["parking"], ref["parking"]) and np.array_equal(r2["driving"], ref["driving"])
        s.set_kernel(0)
        # two shards of the same fleet add up to the whole (Philox keyed by the global car id)
        pk = np.zeros((Z, T), dtype=np.int64)
        for rank in range(2):
            b, n = shard_range(C, rank, 2)
            s.init_states(C, cpz, b, n)
            s.set_state(init[b:b + n])
            pk += s.resample(SIM_SEED)["parking"]
        assert np.array_equal(pk, ref["parking"])


@pytest.mark.parametrize("cpz", [100, 200])
def test_melbourne_shaped_config(cpm, O, cpz):
    """BASELINE.json configs[0] (100 cars/zone: the reference's own CPU-runnable case, main.jl:41 -- AUTO's one-car-per-thread
    instantiation of the grouped path at full Z) and configs[1]'s shape (here 200 cars/zone to keep the oracle quick; the full
    1,000 are in tests/test_full_size.py): sparse Melbourne-shaped datamatrix -> createpdrive / createpdestin on the device ->
    IVP -> resample with travel times, against the oracle run on the tables the device returned."""
    Z, T = 2357, 24
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        p_drive = s.build_p_drive(0.1, 0.9, 0.5)
        p_dest = s.build_p_dest(2)
        want_pd = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
        np.testing.assert_allclose(p_drive, want_pd, rtol=4e-16, atol=0, equal_nan=True)
        assert np.array_equal(p_dest, O.createpdestin(dm, Z, T, 2))
        ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        assert s.get_info(1) == 5                          # AUTO runs the grouped path at both fleet sizes
        r = s.resample(SIM_SEED, travel=True)
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["sum_tt_q16"] == ref["sum_tt_q16"]


def test_sharded_sampler_stream_ordering(cpm, O):
    """ShardedSampler runs the kernels and the collective on one explicit torch stream: reading the
    count tensor on that stream, with no host synchronisation in between, must see the finished
    resample (a rehearsal caught the all-reduce racing the kernels when torch's default stream,
    handle 0, was handed to the C ABI)."""
    import torch
    from carparkingmaps_amd.distributed import ShardedSampler, split_counts
    Z, T, cpz = 512, 24, 300
    C = Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    ss = ShardedSampler(Z, T, rank=0, world_size=1, device=0)
    try:
        ss.s.set_p_drive(p_drive)
        ss.s.set_p_dest(p_dest)
        ss.init_states(C, cpz)
        ss.s.solve_ivp_async(SIM_SEED)
        for _ in range(3):
            counts = ss.resample_allreduce(SIM_SEED)
            with torch.cuda.stream(ss.stream):
                host = counts.to("cpu", non_blocking=False)
            pk, dr, _ = split_counts(host, Z, T)
            assert np.array_equal(pk, ref["parking"]) and np.array_equal(dr, ref["driving"])
    finally:
        ss.close()


def test_rccl_allreduce_on_the_sampler_stream(cpm):
    """The device all-reduce really executes: a child process creates an RCCL process group (world size 1) before any GPU call,
    then drives ShardedSampler's pipelined and blocking forms (tests/rccl_world1.py).  Its counts must equal a plain Sampler's.
    (Several ranks cannot be run on the one-GPU box; the N = 2 logic runs over gloo in tests/test_host_logic.py.)"""
    import hashlib
    import json
    import socket
    import subprocess
    import sys
    Z, cpz, T = 512, 300, 24
    C = Z * cpz
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    proc = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_world1.py"), str(port), str(Z), str(cpz), hex(TABLE_SEED), hex(SIM_SEED)],
                          capture_output=True, text=True, timeout=600, env=env)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-4000:]
    out = json.loads([l for l in proc.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "nccl" and out["world"] == 1
    want = {}
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        for seed in (SIM_SEED, SIM_SEED + 1):
            r = s.resample(seed)
            want[seed] = (hashlib.sha256(r["parking"].tobytes(order="F")).hexdigest(), hashlib.sha256(r["driving"].tobytes(order="F")).hexdigest())
    assert len(out["steps"]) == 3
    for st in out["steps"]:
        assert st["cars_per_hour_ok"]
        assert (st["parking"], st["driving"]) == want[SIM_SEED + (st["k"] & 1)], st["k"]
    assert (out["sync"]["parking"], out["sync"]["driving"]) == want[SIM_SEED]


@pytest.mark.parametrize("Z,cpz,T", [(700, 300, 24), (130, 1100, 6), (2357, 150, 4)])
def test_fused_hour_equals_two_launches_per_hour(cpm, O, Z, cpz, T):
    """The grouped path's hour as ONE launch (sampler workgroups + the placing blocks of their drivers, handed over inside the
    launch) against two launches per hour and against the oracle: counts, post-IVP state, travel-time sum.  Also the bail-out:
    with placing blocks that give up waiting at once every fused step comes back flagged, the blocking calls repeat it with two
    launches, and the context stays there."""
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.3)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        p_drive = s.build_p_drive(0.1, 0.9, 0.5)
        p_dest = s.build_p_dest(2)
        ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
        s.set_kernel(5)
        # (3: the placing-first form -- the previous hour's placing blocks in front of the hour's samplers; 6, 8: all hours of a run in
        #  ONE launch, k_grouped_day, placing blocks among / in front of the sampler workgroups; 2, 4, 7: the bail-outs)
        for mode, lag in [(1, 1), (0, None), (1, 2), (1, 7), (1, 1 << 20), (3, None), (2, None), (3, None), (4, None), (6, None), (8, None),
                          (7, None), (6, None)]:
            s.set_fused(mode, lag)
            s.init_states(C, cpz)
            assert s.get_info(4) == {0: 0, 1: 1, 2: 1, 3: 3, 4: 3, 6: 6, 7: 6, 8: 6}[mode]
            assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"]), (mode, lag)
            r = s.resample(SIM_SEED, travel=True)
            assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (mode, lag)
            assert r["sum_tt_q16"] == ref["sum_tt_q16"], (mode, lag)
            r = s.resample(SIM_SEED)                             # (without travel times the last hour runs in its plain form)
            assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (mode, lag)
            s.set_zone_order(False)                              # (the one-launch hour in zone order instead of largest-first: a hint, the same counts)
            r = s.resample(SIM_SEED)
            assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (mode, lag)
            s.set_zone_order(True)
            assert s.get_info(4) == {0: 0, 1: 1, 2: 0, 3: 3, 4: 0, 6: 6, 7: 0, 8: 6}[mode]   # after a bail-out the context keeps to two launches
            assert s.get_info(2) == 4                            # ... and did not mistake it for an overflow


@pytest.mark.parametrize("Z,want", [(1536, 0), (3072, 1), (4096, 1), (6144, 0)])
def test_one_launch_per_hour_only_where_it_pays(cpm, Z, want):
    """By default the hour is one launch where that was measured to pay (two rounds of sampler workgroups or more, a row pack that
    leaves five blocks per CU: 3,072 <= Z <= ~5,600 on 256 CUs) and two launches elsewhere; CPM_OPT_FUSED = 1 forces it on."""
    cpz, T = 200, 24
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED)
        s.init_states(Z * cpz, cpz)
        assert s.get_info(1) == cpm.CPM_KERNEL_ZONE_GROUPED
        if cpm.device_info(0)["cu_count"] == 256:
            assert s.get_info(4) == want
        s.set_fused(1)
        assert s.get_info(4) == 1
        s.set_fused(5)
        r = s.resample(SIM_SEED)
        assert (r["parking"].sum(axis=0) == Z * cpz).all()


@pytest.mark.parametrize("deal", ["interleaved", "contiguous"])
def test_two_hip_ranks_on_one_gpu_sum_to_the_single_run(cpm, deal):
    """The sharded path with HIP ranks side by side: two processes, each with its own context on the one GPU and its share of the
    fleet, counts summed over gloo (tests/hip_world2.py) -- the sums must be the single-context run's, for both deals (Philox is
    keyed by the global car id; src/resampling.jl:11-83 reads only car i's row)."""
    import hashlib
    import json
    import socket
    import subprocess
    import sys
    Z, cpz, T, world = 384, 250, 24, 2
    C = Z * cpz
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "hip_world2.py"), str(r), str(world), str(port), str(Z), str(cpz),
                               hex(TABLE_SEED), hex(SIM_SEED), deal], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
             for r in range(world)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so_, se_) in zip(procs, outs):
        assert p.returncode == 0, so_[-2000:] + se_[-4000:]
    out = json.loads([l for l in outs[0][0].splitlines() if l.startswith("RESULT ")][-1][7:])
    assert out["backend"] == "gloo" and out["world"] == world and sum(out["counts"]) == C and min(out["counts"]) > 0
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        for k, st in enumerate(out["steps"]):
            r = s.resample(SIM_SEED + k)
            assert st["cars_per_hour_ok"] and st["own_cars_ok"]
            assert st["parking"] == hashlib.sha256(r["parking"].tobytes(order="F")).hexdigest(), (deal, k)
            assert st["driving"] == hashlib.sha256(r["driving"].tobytes(order="F")).hexdigest(), (deal, k)


@pytest.mark.parametrize("kernel", KERNELS)
def test_extreme_skew_everyone_to_one_zone(cpm, O, kernel):
    """Every row is a point mass on zone 3: after one hour the whole fleet sits in one bucket
    (one workgroup walks 200k cars; the fused kernel's 16-bit rank overflows and the blocking API
    falls back by itself).  Counts stay bit-exact."""
    Z, T, cpz = 64, 24, 3200
    C = Z * cpz
    p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[:, 2, :] = 1.0
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    with cpm.Sampler(Z, T) as s:
        _set_kernel(s, kernel)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
    assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])
    assert r["parking"][2, 5] > 0.9 * C


def test_fixed_stride_overflow_is_reported_and_auto_demotes_itself(cpm, O):
    """Everybody moves into zone 3: its bucket outgrows the fixed-stride region (4x the mean).  The async
    form cannot fall back by itself, it raises the status word; the blocking form repeats the step on the
    exact layout, and a context running AUTO stays on the exact layout afterwards."""
    import torch
    from carparkingmaps_amd.distributed import split_counts
    Z, T, cpz = 64, 24, 3200
    C = Z * cpz
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[:, 2, :] = 1.0
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        counts = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
        for kernel in (5,):
            _set_kernel(s, kernel)
            s.resample_dev(SIM_SEED, counts.data_ptr())
            s.sync()
            assert int(counts[-1].item()) != 0
            with pytest.raises(RuntimeError):
                split_counts(counts, Z, T)
        s.set_kernel(2)
        s.resample_dev(SIM_SEED, counts.data_ptr())
        s.sync()
        pk, _, _ = split_counts(counts, Z, T)
        assert (pk.sum(axis=0) == C).all() and pk[2, 1] == C
        # AUTO: first async step is flagged, the blocking call corrects itself, later async steps are clean
        s.set_kernel(0)
        r = s.resample(SIM_SEED)
        assert (r["parking"].sum(axis=0) == C).all() and r["parking"][2, 1] == C
        s.resample_dev(SIM_SEED, counts.data_ptr())
        s.sync()
        assert int(counts[-1].item()) == 0
        pk2, _, _ = split_counts(counts, Z, T)
        assert np.array_equal(pk2, r["parking"])


def test_run_overflow_of_the_grouped_layout_falls_back(cpm, O):
    """Every car of a zone drives into ONE destination group (a permutation of the groups, so no bucket outgrows its
    region): the zone's fixed-size run of that group (scap = a quarter of the bucket region) overflows as soon as the
    zone holds more than scap cars.  Async form: status word; blocking form: exact layout, bit-exact counts."""
    import torch
    Z, T, cpz = 64, 24, 3200
    C = Z * cpz
    p_drive = np.ones((Z, T), order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    for o in range(Z):
        g = (o // 2 + 1) % 32
        p_dest[o, 2 * g, :] = 0.5
        p_dest[o, 2 * g + 1, :] = 0.5
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    assert ref["parking"].max() < 4 * cpz  # no bucket overflow: it is the run that overflows
    with cpm.Sampler(Z, T) as s:
        s.set_kernel(5)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        counts = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
        s.resample_dev(SIM_SEED, counts.data_ptr())
        s.sync()
        assert int(counts[-1].item()) != 0
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
    assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])


def test_few_cars_per_zone_uses_the_car_kernel_and_matches(cpm, O):
    """AUTO below 32 cars/zone (streaming every row would not pay): still bit-exact."""
    Z, T, cpz = 300, 24, 5
    C = Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
    assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])


def _ref_categorical(cdf_row, k53):
    """first j with u <= cdf[j] after the D1 clamp (oracle semantics), 1-based; 0 for an all-zero row"""
    last = cdf_row[-1]
    if last == 0.0:
        return np.zeros(len(k53), dtype=np.int64)
    u = k53.astype(np.float64) * 2.0 ** -53  # exact: k < 2^53
    ue = np.where(u == 0.0, np.float64(5e-324), u)
    ue = np.minimum(ue, last)
    return np.searchsorted(cdf_row, ue, side="left").astype(np.int64) + 1


def test_high_word_search_equals_the_f64_search_on_ties_and_edges(cpm, O):
    """The grouped sampler searches the 4-byte high words of the CDF and falls back to the f64 row on a tie
    or above the row total (cpm_zone6_kernels.h).  Draws are placed exactly on, just below and just above
    every breakpoint, in rows built to hold many equal high words, zero-probability zones, a row total
    below 1, a row total above 1 and an all-zero row."""
    Z, T = 300, 2
    rng = np.random.default_rng(7)
    p = np.zeros((Z, Z, T), order="F")
    w = rng.random(Z) ** 2
    w[rng.random(Z) < 0.3] = 0.0              # zero-probability zones (leading / trailing ones included)
    w[:3] = 0.0
    w[-2:] = 0.0
    p[0, :, 0] = w / w.sum()
    tiny = np.full(Z, 2.0 ** -40)             # 300 breakpoints sharing few high words (steps of 2^-40)
    tiny[0] = 0.25
    p[1, :, 0] = tiny
    p[2, :, 0] = (w / w.sum()) * 0.5          # row total 0.5: half of the draws lie above it (D1: last zone with p > 0)
    p[3, :, 0] = (w / w.sum()) * 1.5          # row total above 1: saturated high words
    # row 4 stays all zero; row 5: everything in the last zone; row 6: everything in the first zone
    p[5, Z - 1, 0] = 1.0
    p[6, 0, 0] = 1.0
    p[7, :, 0] = 2.0 ** -60                   # every high word 0 except through accumulation: all ties
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(np.full((Z, T), 0.5, order="F"))
        s.set_p_dest(p)
        total_exact = 0
        for o in range(1, 9):
            cdf = s.get_cdf_row(o, 1)
            assert np.array_equal(cdf, np.cumsum(p[o - 1, :, 0]))  # sequential f64 sum (numpy cumsum is sequential)
            t53 = np.floor(np.minimum(cdf, 1.0 - 2.0 ** -53) * 2.0 ** 53).astype(np.uint64)
            ks = [np.array([0, 1, 2, 2 ** 53 - 1, 2 ** 53 - 2, 2 ** 52, 2 ** 21, 2 ** 21 - 1, 2 ** 21 + 1], dtype=np.uint64)]
            for d in (-2 ** 21, -1, 0, 1, 2 ** 21 - 1, 2 ** 21):
                ks.append(np.clip(t53.astype(np.int64) + d, 0, 2 ** 53 - 1).astype(np.uint64))
            ks.append(rng.integers(0, 2 ** 53, size=5000, dtype=np.uint64))
            k53 = np.concatenate(ks)
            got, n_exact = s.debug_categorical(o, 1, k53)
            want = _ref_categorical(cdf, k53)
            assert np.array_equal(got, want), (o, np.flatnonzero(got != want)[:5])
            total_exact += n_exact
        assert total_exact > 1000  # the fallback really ran


def test_unnormalised_weights_and_row_sums_give_the_same_tables(cpm, O):
    """cpm_build_p_dest without a host copy (weights, sequential row sums, division in place: k_pdest_weights / _rowsum / _divide on a
    datamatrix too dense for the compact-row builders) -- the CDF rows, the packs and the tie fallback must be those of the
    normalised table (src/createpdestin.jl:38-46 then src/resampling.jl:39).  Checked row by row against the oracle's
    createpdestin + sequential sum, with draws on, below and above every breakpoint (ties go through the checkpoint walk)."""
    Z, T = 333, 24
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.3)
    dm[7, :, :, :] = 0.0                      # an origin without data: all-zero rows
    dm = np.asfortranarray(dm)
    rng = np.random.default_rng(5)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        for e_dest in (2, 0.5):
            s.build_p_dest(e_dest, want=False)            # weights + row sums stay on the device
            p = O.createpdestin(dm, Z, T, e_dest)
            total_exact = 0
            for (o, t) in [(1, 1), (8, 3), (Z, T), (100, 12), (257, 24)]:
                want_cdf = np.cumsum(p[o - 1, :, t - 1])
                cdf = s.get_cdf_row(o, t)                 # (built on first need from the same weights and sums)
                if isinstance(e_dest, int):
                    assert np.array_equal(cdf, want_cdf), (o, t)
                else:
                    np.testing.assert_allclose(cdf, want_cdf, rtol=1e-12, atol=0)
                t53 = np.floor(np.minimum(cdf, 1.0 - 2.0 ** -53) * 2.0 ** 53).astype(np.int64)
                k53 = np.concatenate([np.clip(t53 + d, 0, 2 ** 53 - 1) for d in (-2 ** 21, -1, 0, 1, 2 ** 21)] +
                                     [rng.integers(0, 2 ** 53, size=3000)]).astype(np.uint64)
                got, n_exact = s.debug_categorical(o, t, k53)
                assert np.array_equal(got, _ref_categorical(cdf, k53)), (e_dest, o, t)
                total_exact += n_exact
            assert total_exact > 500


def test_refresh_tables_and_lazy_cdf_rows(cpm, O):
    """The f64 CDF rows are built on first need (car / exact-layout kernels, cpm_get_cdf_row) or with the packs
    (cpm_refresh_tables(with_f64_cdf)); either way they are the sequential sums, and a refresh changes no result."""
    Z, T, cpz = 130, 5, 40
    C = Z * cpz
    p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
    p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED)
    cdf = O.build_cdf(p_dest)
    ref = O.fast_run(p_drive, cdf, C, SIM_SEED, _zone0(C, cpz))
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED)
        r0 = s.resample(SIM_SEED)                          # grouped path: no f64 rows yet
        for full in (True, False, True):
            s.refresh_tables(with_f64_cdf=full)
            for (o, t) in [(1, 1), (Z, T), (64, 3)]:
                assert np.array_equal(s.get_cdf_row(o, t), cdf[t - 1, o - 1])
            for kernel in (0, 1, 2, 5):
                s.set_kernel(kernel)
                r = s.resample(SIM_SEED)
                assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (full, kernel)
        assert np.array_equal(r0["parking"], ref["parking"])


def test_device_synth_datamatrix_equals_the_oracle(cpm, O):
    Z, T = 97, 24
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED)
    with cpm.Sampler(Z, T) as s:
        s.synth_datamatrix(TABLE_SEED)
        assert np.array_equal(s.get_datamatrix(), dm)
        assert np.array_equal(s.get_distance(), dist)
        assert np.array_equal(s.build_p_dest(2), O.createpdestin(dm, Z, T, 2))


def test_high_word_table_rows_longer_than_a_power_of_two(cpm, O):
    """Z = 2^H and Z = 2^H + 1: the tree's special slots (element Z-1 in tree[0], pads)."""
    rng = np.random.default_rng(11)
    for Z in (64, 65, 127, 128, 129):
        p = np.zeros((Z, Z, 1), order="F")
        w = rng.random(Z)
        p[0, :, 0] = w / w.sum()
        with cpm.Sampler(Z, 1) as s:
            s.set_p_drive(np.full((Z, 1), 0.5, order="F"))
            s.set_p_dest(p)
            cdf = s.get_cdf_row(1, 1)
            t53 = np.floor(np.minimum(cdf, 1.0 - 2.0 ** -53) * 2.0 ** 53).astype(np.int64)
            k53 = np.concatenate([np.clip(t53 + d, 0, 2 ** 53 - 1) for d in (-1, 0, 1)] +
                                 [rng.integers(0, 2 ** 53, size=2000)]).astype(np.uint64)
            got, _ = s.debug_categorical(1, 1, k53)
            assert np.array_equal(got, _ref_categorical(cdf, k53)), Z


# ------------------------------------------------------------------ data formats
def _uber_rows(rng, Z, n, dup=0.02):
    src = rng.integers(0, Z, n)          # 0 .. Z-1: id 0 is remapped to Z (src/createdatamatrix.jl:9-14)
    dst = rng.integers(0, Z, n)
    hod = rng.integers(0, 24, n)         # 0 .. 23: hod 0 -> 24 (:15-17)
    k = int(n * dup)
    if k:                                # repeated (source, dest, hour) keys: the last row must win (:21-22)
        j = rng.integers(0, n, k)
        i = rng.integers(0, n, k)
        src[i], dst[i], hod[i] = src[j], dst[j], hod[j]
    mean = np.round(300 + 2100 * rng.random(n), 2)
    std = np.round(mean * (0.1 + 0.3 * rng.random(n)), 2)
    return np.asfortranarray(np.column_stack([src, dst, hod, mean, std]).astype(np.float64))


def test_createdatamatrix_rows_equals_the_reference_loop(cpm, O):
    Z, T = 61, 24
    rng = np.random.default_rng(21)
    raw = _uber_rows(rng, Z, 40_000, dup=0.2)
    want = O.createdatamatrix(raw, Z, T)
    with cpm.Sampler(Z, T) as s:
        s.createdatamatrix_rows(raw)
        got = s.get_datamatrix()
        assert np.array_equal(got, want)
        s.createdatamatrix_rows(np.zeros((0, 5)))                       # no rows: all zeros (:7)
        assert not s.get_datamatrix().any()
        for bad in ([Z + 1, 1, 1, 1, 1], [1, 1, 25, 1, 1], [1.5, 1, 1, 1, 1], [-1, 1, 1, 1, 1]):
            with pytest.raises(cpm.CpmError):
                s.createdatamatrix_rows(np.array([[1, 1, 1, 5.0, 1.0], bad], dtype=float))


def test_createdatamatrix_csv_to_tables_without_a_host_datamatrix(cpm, O, tmp_path):
    """CSV text -> native reader -> dense datamatrix in HBM -> createpdrive / createpdestin on the device, against the
    oracle fed with numpy's reading of the same file."""
    Z, T = 47, 24
    rng = np.random.default_rng(22)
    raw = _uber_rows(rng, Z, 30_000, dup=0.05)
    p = tmp_path / "city-2019-1-All-HourlyAggregate.csv"
    with open(p, "w") as f:
        f.write("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,"
                "geometric_standard_deviation_travel_time\n")
        for r in raw:
            f.write(f"{int(r[0])},{int(r[1])},{int(r[2])},{float(r[3])!r},{float(r[4])!r},{float(r[3]) * 0.9!r},1.3\n")
    lat = -38.5 + 1.5 * rng.random(Z)
    lon = 144.0 + 2.0 * rng.random(Z)
    dm = O.createdatamatrix(np.loadtxt(p, delimiter=",", skiprows=1, usecols=range(5)), Z, T)
    dist = O.distance_matrix(lat, lon)
    with cpm.Sampler(Z, T) as s:
        assert s.createdatamatrix_csv(str(p)) == raw.shape[0]
        assert np.array_equal(s.get_datamatrix(), dm)
        s.set_distance_from_centroids(lat, lon)
        got = s.get_distance()
        np.testing.assert_allclose(got, dist, rtol=4e-16, atol=0)       # device cos vs glibc cos: <= 2 ulp of the result
        assert np.array_equal(np.diag(got), np.ones(Z)) and np.array_equal(got, got.T)
        p_drive = s.build_p_drive(0.1, 0.9, 0.5)
        p_dest = s.build_p_dest(2)
    # the tables from the device-built inputs equal the oracle's from its own (distance enters p_drive only through a
    # division by values that agree to 2 ulp)
    np.testing.assert_allclose(p_drive, O.createpdrive(dm, got, Z, T, 0.1, 0.9, 0.5), rtol=1e-15, atol=0)
    np.testing.assert_allclose(p_dest, O.createpdestin(dm, Z, T, 2), rtol=1e-15, atol=0)


def test_main_jl_flow_from_files(cpm, O, tmp_path):
    """main.jl:51-109 for one city directory: GeoJSON + CSV in, result CSVs out, dense arrays device-resident throughout."""
    import json
    from carparkingmaps_amd import reference_api as R
    Z, T, cpz = 24, 24, 50
    rng = np.random.default_rng(23)
    city = tmp_path / "cities" / "Testville"
    city.mkdir(parents=True)
    feats = []
    for k in range(Z):                                        # MOVEMENT_IDs 0 .. Z-1: id 0 becomes zone Z
        cx, cy = 144.0 + 0.1 * (k % 6), -38.0 + 0.1 * (k // 6)
        ring = [[cx, cy], [cx + 0.08, cy], [cx + 0.08, cy + 0.07], [cx, cy + 0.07], [cx, cy]]
        feats.append({"type": "Feature", "properties": {"MOVEMENT_ID": str(k)}, "geometry": {"type": "Polygon", "coordinates": [ring]}})
    (city / "zz_testville.json").write_text(json.dumps({"type": "FeatureCollection", "features": feats}))
    raw = _uber_rows(rng, Z, 6000, dup=0.05)
    with open(city / "testville-2019-1.csv", "w") as f:
        f.write("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,geometric_standard_deviation_travel_time\n")
        for r in raw:
            f.write(f"{int(r[0])},{int(r[1])},{int(r[2])},{float(r[3])!r},{float(r[4])!r},1,1\n")
    results_root = str(tmp_path / "results") + "/"
    os.mkdir(results_root)
    R.release()
    R.params.cars_per_zone, R.params.T = cpz, T
    try:
        dataset_list = sorted(os.listdir(city))               # main.jl:51-53: the GeoJSON sorts last
        path_to_results = R.createresultsdirectory(results_root, "Testville")
        dist, number_zones = R.processgeodata(str(city / dataset_list[-1]), str(city), dataset_list[:-1], path_to_results)
        assert number_zones == Z
        C = number_zones * cpz
        datamatrix = R.createdatamatrix(str(city / dataset_list[0]), number_zones)
        out = R.run_dataset(datamatrix, dist, number_zones, travel=True)
        R.saveparameters(path_to_results, T, number_zones, cpz, C, R.params.e_drive, R.params.p_min, R.params.p_max, R.params.e_dest,
                         out["A_drive_increment"])
        # oracle on the same files
        clat, clong = R.polygon_centroids(*R.geojson_vertex_lists(feats))
        odist = O.distance_matrix(clat, clong)
        odm = O.createdatamatrix(raw, Z, T)
        np.testing.assert_allclose(dist.numpy(), odist, rtol=4e-16)
        assert np.array_equal(datamatrix.numpy(), odm)
        coords = np.loadtxt(os.path.join(path_to_results, "zoneID_coordinates.csv"), delimiter=",", skiprows=1)
        assert np.array_equal(coords[:, 0], clat) and np.array_equal(coords[:, 1], clong)
        p_drive = O.createpdrive(odm, dist.numpy(), Z, T, 0.1, 0.9, 0.5)
        p_dest = O.createpdestin(odm, Z, T, 2)
        ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, R.params.seed, _zone0(C, cpz))
        assert np.array_equal(out["parking"], ref["parking"]) and np.array_equal(out["driving"], ref["driving"])
        assert os.path.exists(os.path.join(path_to_results, "sampling_parameters.csv"))
    finally:
        R.release()
        R.params.cars_per_zone, R.params.T = 1000, 24


@pytest.mark.parametrize("Z,T,cpz", [(8192, 3, 48), (12000, 2, 40)])
def test_grouped_path_with_long_rows(cpm, O, Z, T, cpz):
    """Row packs of 34 and 56 KiB (more LDS-DMA instructions per wave; above 48 KiB the kernel has to opt in to its LDS),
    place kernel with 32 and 64 blocks per group, 256 / 512 zones per destination group."""
    C = Z * cpz
    p_drive, p_dest = _tables(O, Z, T)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    with cpm.Sampler(Z, T) as s:
        s.set_kernel(5)
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])
        # the grouped layout really ran (no overflow demotion): the async form reports a clean status
        import torch
        counts = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
        s.resample_dev(SIM_SEED, counts.data_ptr())
        s.sync()
        assert int(counts[-1].item()) == 0


def test_overflow_is_absorbed_by_growing_the_bucket_regions(cpm, O):
    """A popular zone that holds ~6x the mean: overflows the default regions (4x), fits after one doubling.  The blocking calls grow
    and repeat by themselves; afterwards the grouped layout keeps running (no demotion to the exact layout) with clean status."""
    import torch
    Z, T, cpz = 64, 24, 800
    C = Z * cpz
    p_drive = np.full((Z, T), 0.5, order="F")
    p_dest = np.zeros((Z, Z, T), order="F")
    w = np.ones(Z)
    w[7] = 7.0                       # zone 8 attracts 10 % of every hour's drivers: steady state 6.4x the mean population
    p_dest[:, :, :] = (w / w.sum())[None, :, None]
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    assert 4 * cpz < ref["parking"].max() < 8 * cpz
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz)
        assert s.get_info(1) == 5 and s.get_info(2) == 4
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])
        assert s.get_info(1) == 5 and s.get_info(2) == 8       # grew once, still the grouped layout
        counts = torch.zeros(s.counts_words(), dtype=torch.int64, device="cuda:0")
        s.resample_dev(SIM_SEED, counts.data_ptr())
        s.sync()
        assert int(counts[-1].item()) == 0
        from carparkingmaps_amd.distributed import split_counts
        pk, dr, _ = split_counts(counts, Z, T)
        assert np.array_equal(pk, ref["parking"]) and np.array_equal(dr, ref["driving"])


def test_heavy_buckets_are_split_over_several_workgroups(cpm, O):
    """Skewed destination popularity (the oracle's / the library's Zipf-Mandelbrot tables, bit-identical): a few zones hold many times
    the mean.  The first grouped step walks such a bucket with one workgroup and reports its size; from then on the context launches
    the heavy kernel with several blocks per zone (CPM_INFO_PARTS > 1).  Counts stay bit-exact on both sides of the switch, for the
    IVP and for the resample, with and without travel times."""
    Z, T, cpz, q = 192, 24, 1000, 4
    C = Z * cpz
    p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
    p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED, skew_q=q)
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.9)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
    assert ref["parking"].max() > 2 * 4 * 256 + 1000         # a bucket well above twice the 4 x 256 slots of a sampler workgroup
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(TABLE_SEED, skew_q=q)
        for (o, t) in [(1, 1), (Z, T), (77, 13)]:
            assert np.array_equal(s.get_cdf_row(o, t), O.build_cdf(p_dest)[t - 1, o - 1]), (o, t)
        s.set_datamatrix(dm, dist)
        s.set_kernel(5)
        s.init_states(C, cpz)
        assert s.get_info(3) == 1
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        parts_after_ivp = s.get_info(3)
        assert parts_after_ivp > 1                               # the IVP saw the heavy buckets
        for k in range(3):
            r = s.resample(SIM_SEED, travel=(k == 1))
            assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), k
            if k == 1:
                assert r["sum_tt_q16"] == ref["sum_tt_q16"]
            assert s.get_info(3) >= int(np.ceil(ref["parking"].max() / 1024)) - 1
        # and from a fresh context whose very first step is already split (parts carried over by the IVP above are not needed)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])


def test_async_ivp_is_committed_on_the_tables_it_was_enqueued_with(cpm, O):
    """solve_ivp_async, then new tables, then a read of the state: the IVP must have run on the OLD tables whether or not its first
    attempt overflowed the bucket regions (an overflowed attempt is repeated by the library when the state is next needed -- every
    entry that replaces a table commits the pending IVP first)."""
    Z, T, cpz = 64, 24, 800
    C = Z * cpz
    p_drive = np.full((Z, T), 0.5, order="F")
    w = np.ones(Z)
    w[7] = 7.0                       # zone 8 holds ~6.4x the mean: overflows the default regions (4x) during the IVP
    p_dest = np.zeros((Z, Z, T), order="F")
    p_dest[:, :, :] = (w / w.sum())[None, :, None]
    flat_drive, flat_dest = _tables(O, Z, T)     # the tables installed afterwards: no overflow, different results
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz))
    other = O.fast_run(flat_drive, O.build_cdf(flat_dest), C, SIM_SEED, _zone0(C, cpz))
    assert not np.array_equal(ref["zone0"], other["zone0"])
    for swap in ("p_dest", "p_drive", "both"):
        with cpm.Sampler(Z, T) as s:
            s.set_p_drive(p_drive)
            s.set_p_dest(p_dest)
            s.init_states(C, cpz)
            assert s.get_info(2) == 4
            s.solve_ivp_async(SIM_SEED)
            if swap in ("p_dest", "both"):
                s.set_p_dest(flat_dest)
            if swap in ("p_drive", "both"):
                s.set_p_drive(flat_drive)
            assert np.array_equal(s.get_state(), ref["zone0"]), swap
            assert s.get_info(2) == 8            # the first attempt did overflow and was repeated with grown regions


def test_examples_main_py_on_a_city_directory(cpm, O, tmp_path):
    """examples/main.py (main.jl line for line) in both modes on a small synthetic city: same result files."""
    import json
    import subprocess
    import sys
    Z = 12
    rng = np.random.default_rng(31)
    city = tmp_path / "cities" / "Smallville"
    city.mkdir(parents=True)
    feats = []
    for k in range(1, Z + 1):
        cx, cy = 144.0 + 0.1 * (k % 4), -38.0 + 0.1 * (k // 4)
        feats.append({"type": "Feature", "properties": {"MOVEMENT_ID": str(k)},
                      "geometry": {"type": "Polygon", "coordinates": [[[cx, cy], [cx + 0.08, cy], [cx + 0.08, cy + 0.07], [cx, cy + 0.07], [cx, cy]]]}})
    (city / "zz.json").write_text(json.dumps({"type": "FeatureCollection", "features": feats}))
    raw = _uber_rows(rng, Z, 2500, dup=0.02)
    raw[:, :2] += 1  # ids 1 .. Z
    with open(city / "a.csv", "w") as f:
        f.write("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,geometric_standard_deviation_travel_time\n")
        for r in raw:
            f.write(f"{int(r[0])},{int(r[1])},{int(r[2])},{float(r[3])!r},{float(r[4])!r},1,1\n")
    outs = []
    for mode in ([], ["--compat"]):
        res = tmp_path / ("results" + ("_compat" if mode else ""))
        res.mkdir()
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        subprocess.check_call([sys.executable, os.path.join(root, "examples", "main.py"), str(tmp_path / "cities"), str(res), "--cars-per-zone", "40"] + mode)
        d = res / "Smallville"
        outs.append({n: (d / n).read_text() for n in sorted(os.listdir(d))})
    assert set(outs[0]) == {"results_parkingdensities_a.csv", "results_trafficactivity_a.csv", "sampling_parameters.csv", "zoneID_coordinates.csv"}
    for n in ("results_parkingdensities_a.csv", "results_trafficactivity_a.csv", "zoneID_coordinates.csv"):
        assert outs[0][n] == outs[1][n], n
    # A_drive: integer q16 sum on the device vs f64 sum of the host matrix: equal to ~1e-8 relative
    a0 = float(outs[0]["sampling_parameters.csv"].splitlines()[1].split(",")[-1])
    a1 = float(outs[1]["sampling_parameters.csv"].splitlines()[1].split(",")[-1])
    assert abs(a0 - a1) <= 1e-7 * abs(a1)


def test_contiguous_shard_of_an_8_gpu_run_starts_skewed(cpm, O):
    """Rank 0 of 8 owns a contiguous car range: its cars start in an eighth of the zones, 8x the shard's mean bucket.  The default
    regions (4x) overflow in the first IVP hour; the context grows them and stays on the grouped layout (what bench.py --gpus 8 does
    on every rank).  Counts against the oracle run of the same shard."""
    Z, T, world = 64, 24, 8
    cpz = 300 * world
    C = Z * cpz
    count = C // world
    p_drive, p_dest = _tables(O, Z, T)
    zone0 = (np.arange(count, dtype=np.int64)) // cpz + 1          # rank 0: cars 0 .. count-1
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), count, SIM_SEED, zone0, car_offset=0)
    with cpm.Sampler(Z, T) as s:
        s.set_p_drive(p_drive)
        s.set_p_dest(p_dest)
        s.init_states(C, cpz, 0, count)
        assert s.get_info(1) == 5 and s.get_info(2) == 4
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        assert s.get_info(1) == 5 and s.get_info(2) == 8            # grew, did not fall back
        r = s.resample(SIM_SEED)
        assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"])
        assert (r["parking"].sum(axis=0) == count).all()


@pytest.mark.parametrize("seed", range(24))
def test_randomized_small_configurations(cpm, O, seed):
    """Random small problems through the grouped path and AUTO: odd zone counts, T from 1 to 24, one car to hundreds per zone,
    sparse rows, all-zero rows, single-destination rows, unreachable zones, p_drive of 0 / 1 / NaN.  Bit-exact against the oracle."""
    rng = np.random.default_rng(1000 + seed)
    Z = int(rng.integers(2, 150))
    T = int(rng.choice([1, 2, 5, 24]))
    cpz = int(rng.choice([1, 2, 7, 40, 120, 300]))
    C = Z * cpz
    p_drive = np.asfortranarray(rng.random((Z, T)))
    p_drive[rng.random((Z, T)) < 0.05] = 0.0
    p_drive[rng.random((Z, T)) < 0.05] = 1.0
    p_drive[rng.random((Z, T)) < 0.02] = np.nan                      # never drives (Appendix A-3)
    w = rng.random((Z, Z, T)) ** 3
    w[rng.random((Z, Z, T)) < float(rng.choice([0.0, 0.5, 0.9]))] = 0.0  # sparse rows
    w[:, rng.random(Z) < 0.1, :] = 0.0                               # zones nobody drives to
    for o in np.flatnonzero(rng.random(Z) < 0.1):                    # single-destination rows
        w[o, :, :] = 0.0
        w[o, int(rng.integers(0, Z)), :] = 1.0
    w[rng.random(Z) < 0.1, :, :] = 0.0                               # all-zero rows: destination = origin, counted as driving
    p_dest = np.zeros((Z, Z, T), order="F")
    for t in range(T):
        for o in range(Z):
            tot = 0.0
            for v in w[o, :, t]:
                tot += v                                             # the sequential sum of src/createpdestin.jl:31-35
            if tot > 0:
                p_dest[o, :, t] = w[o, :, t] / tot
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED + seed, _zone0(C, cpz))
    for kernel in (5, 0):
        with cpm.Sampler(Z, T) as s:
            s.set_kernel(kernel)
            s.set_p_drive(p_drive)
            s.set_p_dest(p_dest)
            s.init_states(C, cpz)
            assert np.array_equal(s.solve_ivp(SIM_SEED + seed), ref["zone0"]), (kernel, Z, T, cpz)
            r = s.resample(SIM_SEED + seed)
            assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (kernel, Z, T, cpz)


def test_sparse_dataset_tables_equal_the_dense_ones(cpm, O):
    """A sparse datamatrix takes the compact-row builders (csrc/cpm_dataset.h: ONE sweep of the datamatrix, everything else on the 6 % of
    the cells that hold data; sparse row packs in the sampler).  Against the oracle's dense createpdrive / createpdestin
    (src/createpdrive.jl:10-33, src/createpdestin.jl:10-46) and its run on them: the tables a caller gets back, the f64 CDF rows built
    on demand, the categorical draw on, below and above every breakpoint (ties walk the row's cells), the post-IVP state, counts and
    travel-time sum -- for the Int exponent of main.jl:38 and a Float64 one, with an origin without data, a (mean 0, std > 0) cell
    and Z not a multiple of 32."""
    Z, T, cpz = 700, 24, 60
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.06)
    dm[7, :, :, :] = 0.0                      # an origin without data: all-zero rows, p_drive = 0
    dm[11, 13, 5, 0], dm[11, 13, 5, 1] = 0.0, 5.0   # a cell with a std and no mean: never a destination, but a cell of the travel row
    dm = np.asfortranarray(dm)
    rng = np.random.default_rng(6)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        p_drive = s.build_p_drive(0.1, 0.9, 0.5)
        np.testing.assert_allclose(p_drive, O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5), rtol=4e-16, atol=0, equal_nan=True)
        for e_dest in (2, 0.5):
            p_dest = s.build_p_dest(e_dest)
            assert s.get_info(6) > 0                       # the sparse route was taken
            want_p = O.createpdestin(dm, Z, T, e_dest)
            if isinstance(e_dest, int):
                assert np.array_equal(p_dest, want_p)
            else:
                np.testing.assert_allclose(p_dest, want_p, rtol=1e-12, atol=0)
            s.build_p_dest(e_dest, want=False)             # (no dense array this time: the tables alone)
            total_exact = 0
            for (o, t) in [(1, 1), (8, 3), (Z, T), (100, 12), (12, 6)]:
                cdf = np.cumsum(p_dest[o - 1, :, t - 1])
                t53 = np.floor(np.minimum(cdf, 1.0 - 2.0 ** -53) * 2.0 ** 53).astype(np.int64)
                k53 = np.concatenate([np.array([0, 1, 2 ** 53 - 1, 2 ** 21, 2 ** 21 - 1])] +
                                     [np.clip(t53 + d, 0, 2 ** 53 - 1) for d in (-2 ** 21, -1, 0, 1, 2 ** 21)] +
                                     [rng.integers(0, 2 ** 53, size=3000)]).astype(np.uint64)
                got, n_exact = s.debug_categorical(o, t, k53)
                assert np.array_equal(got, _ref_categorical(cdf, k53)), (e_dest, o, t)
                total_exact += n_exact
                assert np.array_equal(s.get_cdf_row(o, t), cdf), (o, t)   # (dense p_destin and its f64 rows, built on demand)
            assert total_exact > 100
            ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, _zone0(C, cpz), datamatrix=dm, dist=dist)
            for kernel in (0, 2, 5):
                s.set_kernel(kernel)
                s.init_states(C, cpz)
                assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"]), (e_dest, kernel)
                r = s.resample(SIM_SEED, travel=True)
                assert np.array_equal(r["parking"], ref["parking"]) and np.array_equal(r["driving"], ref["driving"]), (e_dest, kernel)
                assert r["sum_tt_q16"] == ref["sum_tt_q16"], (e_dest, kernel)
            s.set_kernel(0)
            s.refresh_tables()
            assert s.get_info(6) > 0
            r = s.resample(SIM_SEED)
            assert np.array_equal(r["parking"], ref["parking"])
        # a new datamatrix behind installed sparse tables: they stay whole (a tie walks the TABLE's cells, not the new dataset's)
        dm2, dist2 = O.synth_datamatrix(Z, T, TABLE_SEED + 1, density=0.06)
        s.set_datamatrix(dm2, dist2)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)         # (sweeps the NEW datamatrix into the compact rows)
        s.set_p_drive(p_drive)
        cdf = np.cumsum(p_dest[99, :, 11])
        t53 = np.floor(np.minimum(cdf, 1.0 - 2.0 ** -53) * 2.0 ** 53).astype(np.uint64)
        got, n_exact = s.debug_categorical(100, 12, t53)
        assert np.array_equal(got, _ref_categorical(cdf, t53)) and n_exact > 0
