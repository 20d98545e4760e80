"""Model-selection layer: objectives against loop restatements of the notebook's
formulas (CPU), and a small grid on the device against the oracle (GPU)."""
import numpy as np
import pytest

from conftest import SIM_SEED, TABLE_SEED

T = 24


def test_objectives_match_the_notebook_formulas():
    from carparkingmaps_amd import model_selection as ms
    rng = np.random.default_rng(3)
    Z, C = 9, 900
    driving = rng.integers(0, 60, size=(Z, T)).astype(np.int64)
    parking = rng.integers(50, 150, size=(Z, T)).astype(np.int64)
    parking[4] = 77                                           # a flat zone: left as it is and not validated
    measured_act = rng.uniform(0, 1, T)
    measured_park = rng.uniform(0, 1, (Z, T))
    measured_park[2] = 0                                      # an unmeasured zone
    # README.md:1270-1283
    ta = driving.sum(axis=0) / C
    ta = (ta - ta.min()) / (ta.max() - ta.min())
    err = sum((measured_act[t] - ta[t]) ** 2 for t in range(T)) / 24
    np.testing.assert_allclose(ms.traffic_activity(driving), ta, rtol=1e-14)
    assert ms.traffic_activity_error(ms.traffic_activity(driving), measured_act) == pytest.approx(err, rel=1e-13)
    # README.md:2219-2244
    pc = parking / C
    mn, mx = pc.min(axis=1), pc.max(axis=1)
    for i in range(Z):
        for j in range(T):
            if mx[i] != mn[i]:
                pc[i, j] = (pc[i, j] - mn[i]) / (mx[i] - mn[i])
    ev, counter = np.zeros(Z), 0
    for i in range(Z):
        if measured_park[i].sum() != 0 and mx[i] != mn[i]:
            ev[i] = sum((pc[i, k] - measured_park[i, k]) ** 2 for k in range(T)) / 24
            counter += 1
    assert counter == Z - 2
    assert ms.parking_density_error(parking, C, measured_park) == pytest.approx(ev.sum() / counter, rel=1e-13)
    assert ms.a_drive(int(0.07 * C * T * 3600 * 65536), C, T) == pytest.approx(0.07, rel=1e-9)


def test_search_exponent_and_p_tuning_follow_the_update_rules():
    from carparkingmaps_amd import model_selection as ms
    calls = []

    def err(e):
        calls.append(e)
        return 0.05 + 0.02 * (e - 0.6) ** 2
    best, best_err, hist = ms.search_exponent(err, [2.0, 1.0, 0.5], step_size=10.0, max_iter=3)
    assert hist[0][0] == 2.0 and hist[2][0] == 0.5
    assert hist[3][0] == pytest.approx(0.5 + 10.0 * err(0.5))            # Step 3.1: one step from the best initial value
    assert best_err == min(h[1] for h in hist) and best in [h[0] for h in hist]
    # A_drive roughly proportional to the mean drive probability (p_min + p_max) / 2
    p_min, p_max, A, hist = ms.tune_p_min_max(lambda a, b: 0.4 * (a + b) / 2, A_set=0.07, max_iter=5)
    assert p_min == 0.1 and p_max < 0.9
    assert abs(A - 0.07) < abs(hist[0][2] - 0.07)
    assert all(0 <= a <= b <= 1 for a, b, _ in hist)


def test_points_are_dealt_without_overlap():
    from carparkingmaps_amd import model_selection as ms
    grid = ms.make_grid()
    assert len(grid) == 256
    seen = sorted(i for r in range(8) for i in ms.points_of_rank(len(grid), r, 8))
    assert seen == list(range(256))
    assert max(len(ms.points_of_rank(256, r, 8)) for r in range(8)) == 32
    assert sorted(i for r in range(3) for i in ms.points_of_rank(10, r, 3, order=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0])) == list(range(10))
    assert ms.points_of_rank(10, 0, 3, order=[9, 8, 7, 6, 5, 4, 3, 2, 1, 0]) == [9, 8, 7, 6]


@pytest.mark.gpu
def test_grid_points_on_device_match_the_oracle(cpm, O):
    from carparkingmaps_amd import model_selection as ms
    Z, cpz = 40, 60
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.3)
    rng = np.random.default_rng(5)
    measured_act, measured_park = rng.uniform(0, 1, T), rng.uniform(0, 1, (Z, T))
    grid = [ms.Point(0.5, 0.1, 0.9, 2), ms.Point(1.0, 0.0, 0.8, 2), ms.Point(2.0, 0.2, 1.0, 1), ms.Point(1.0, 0.1, 0.9, 3)]
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        s.build_p_dest(2, want=False)
        s.init_states(C, cpz)
        init = s.solve_ivp(SIM_SEED)                          # once; every point restarts from it (README.md:1180)
        ev = ms.Evaluator(s, C, SIM_SEED, measured_act, measured_park, travel=True)
        full = ms.grid_sweep(ev, grid)
        halves = [ms.grid_sweep(ev, grid, rank=r, world_size=2, gather=False) for r in range(2)]
    for i, pt in enumerate(grid):
        assert (halves[0][i] is None) != (halves[1][i] is None)
        got = full[i]
        half = halves[0][i] if halves[0][i] is not None else halves[1][i]
        assert (half["A_drive"], half["driving_total"]) == (got["A_drive"], got["driving_total"])
        p_drive = O.createpdrive(dm, dist, Z, T, pt.p_min, pt.p_max, pt.e_drive)
        p_dest = O.createpdestin(dm, Z, T, pt.e_dest)
        ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, init, do_ivp=False, datamatrix=dm, dist=dist)
        assert got["driving_total"] == int(ref["driving"].sum())
        assert got["A_drive"] == ms.a_drive(ref["sum_tt_q16"], C, T)
        assert got["activity_error"] == pytest.approx(ms.traffic_activity_error(ms.traffic_activity(ref["driving"]), measured_act), rel=1e-12)
        assert got["parking_error"] == pytest.approx(ms.parking_density_error(ref["parking"], C, measured_park), rel=1e-12)


@pytest.mark.gpu
def test_two_sampler_contexts_in_flight_give_the_same_points(cpm, O):
    """grid_sweep over two Evaluators (two contexts, two streams: a grid point of each on the GPU at a time) returns what one
    context returns, point by point and bit by bit."""
    from carparkingmaps_amd import model_selection as ms
    Z, cpz = 48, 50
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.4)
    grid = [ms.Point(a, 0.1, b, d) for d in (1, 2) for a in (0.5, 1.0, 2.0) for b in (0.8, 1.0)]   # 12 points, two e_dest values
    samplers, lanes = [], []
    try:
        import torch
        for _ in range(2):
            s = cpm.Sampler(Z, T, stream=torch.cuda.Stream())   # (a lane brings its stream: Sampler.__init__)
            samplers.append(s)
            s.set_datamatrix(dm, dist)
            s.build_p_drive(0.1, 0.9, 0.5, want=False)
            s.build_p_dest(2, want=False)
            s.init_states(C, cpz)
            s.solve_ivp(SIM_SEED, want=False)
            lanes.append(ms.Evaluator(s, C, SIM_SEED, travel=True))
        one = ms.grid_sweep(lanes[0], grid, checksums=True)
        two = ms.grid_sweep(lanes, grid, checksums=True)
        odd = ms.grid_sweep(lanes, grid[:5], checksums=True)        # 3 + 2 points: the lanes run dry at different times
    finally:
        for s in samplers:
            s.close()
    keys = ("A_drive", "driving_total", "parking_crc32", "driving_crc32", "hours_hold_all_cars")
    for a, b in zip(one, two):
        assert all(a[k] == b[k] for k in keys)
    for a, b in zip(one[:5], odd):
        assert all(a[k] == b[k] for k in keys)
    assert all(r["hours_hold_all_cars"] for r in two)


@pytest.mark.gpu
def test_pipelined_points_that_overflow_are_evaluated_again(cpm, O):
    """A datamatrix whose trips all end in 6 of 192 zones: those buckets hold ~30 x the mean, the bucket regions (4 x) overflow, and
    the asynchronous steps of the pipelined sweep come back with their status word set.  Evaluator.finish() then evaluates the
    point again through the blocking call (which grows the regions) -- the results must be those of a blocking sweep on a fresh
    context, and the fallbacks are counted."""
    from carparkingmaps_amd import model_selection as ms
    Z, cpz = 192, 120
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.9)
    dm[:, 6:, :, :] = 0.0
    dm = np.asfortranarray(dm)
    grid = [ms.Point(a, 0.1, b, d) for d in (2, 1) for a in (0.5, 2.0) for b in (0.8, 1.0)]
    def lane():
        s = cpm.Sampler(Z, T)
        s.set_datamatrix(dm, dist)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        s.build_p_dest(2, want=False)
        s.init_states(C, cpz)                        # the initial placement (no IVP: its blocking call would grow the regions already):
        return s                                     # mean-sized buckets, default regions ...
    s1 = lane()
    try:
        ev = ms.Evaluator(s1, C, SIM_SEED, travel=True)
        assert s1.get_info(2) == 4
        piped = ms.grid_sweep(ev, grid, checksums=True)   # ... which the first resample hour overflows
        assert ev.fallbacks >= 1 and sum(r["fallback"] for r in piped) == ev.fallbacks
        assert s1.get_info(2) > 4 or s1.get_info(1) == 2
    finally:
        s1.close()
    s2 = lane()
    try:
        ev2 = ms.Evaluator(s2, C, SIM_SEED, travel=True)
        blocking = [ev2.evaluate(pt) for pt in grid]
    finally:
        s2.close()
    import zlib
    for a, pt, b in zip(piped, grid, blocking):
        assert a["A_drive"] == b["A_drive"], pt
        assert a["parking_crc32"] == zlib.crc32(np.ascontiguousarray(b["parking"].ravel(order="F")).tobytes()), pt
        assert a["driving_crc32"] == zlib.crc32(np.ascontiguousarray(b["driving"].ravel(order="F")).tobytes()), pt
        assert a["hours_hold_all_cars"]


def test_lanes_cover_the_ranks_slice_once_and_in_turn():
    """grid_sweep's lane logic without a GPU: stub Evaluators that record the order of begin / finish."""
    from carparkingmaps_amd import model_selection as ms

    class Stub:
        C = 10

        def __init__(self, name, log):
            self.name, self.log, self.open = name, log, {}

        def begin(self, pt, slot):
            assert slot not in self.open, "a slot is reused only after its finish"
            self.open[slot] = pt
            self.log.append((self.name, "begin", pt.e_drive))

        def finish(self, pt, slot):
            assert self.open.pop(slot) is pt
            self.log.append((self.name, "finish", pt.e_drive))
            z = np.full((3, 2), 5, dtype=np.int64)
            return {"e_drive": pt.e_drive, "parking": z, "driving": z, "A_drive": 0.5}

    grid = [ms.Point(float(i), 0.1, 0.9, 2 if i % 2 else 1) for i in range(11)]
    for world in (1, 2):
        for nl in (1, 2, 3):
            got = {}
            for rank in range(world):
                log = []
                lanes = [Stub(f"L{l}", log) for l in range(nl)]
                res = ms.grid_sweep(lanes if nl > 1 else lanes[0], grid, rank=rank, world_size=world, gather=False)
                mine = {i for i, r in enumerate(res) if r is not None}
                assert not (mine & set(got)), "ranks share no point"
                got.update({i: res[i] for i in mine})
                begun = [e for (_, what, e) in log if what == "begin"]
                assert sorted(begun) == sorted(grid[i].e_drive for i in mine) and len(set(begun)) == len(begun)
                assert all(not l.open for l in lanes)
                if nl > 1 and len(mine) >= 2 * nl:   # the lanes advance in turn: every lane has begun a point before any lane finishes one
                    first_finish = next(k for k, (_, what, _) in enumerate(log) if what == "finish")
                    assert {name for (name, what, _) in log[:first_finish]} == {l.name for l in lanes}
            assert sorted(got) == list(range(len(grid)))
            assert all(got[i]["e_drive"] == grid[i].e_drive and got[i]["hours_hold_all_cars"] is False for i in got)
