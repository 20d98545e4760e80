"""BASELINE.json's configurations at their full sizes on the GPU (python -m pytest tests -m gpu):

  configs[1]  Melbourne-shaped (Z = 2,357, 8.68 % dense datamatrix), 1,000 cars/zone, travel times on -- against the oracle
  configs[3]  Z = 8,192 x 4,000 cars/zone dealt over 8 GPUs: what ranks 0 and 7 compute (4,096,000 cars each, T = 24), for the
              interleaved and the contiguous deal -- against SHA-256 checksums the oracle produced in the build container
              (tests/golden/make_golden.py --s8k -> s8k_shard_checksums.json; no oracle run of that size on the GPU box)
  configs[4]  the 256-point model-selection grid on the Melbourne-shaped problem, dealt over 8 (emulated) ranks -- size-
              independent properties for every point, eight sampled points bit-exact against the oracle
(configs[2], the headline S4k run: tests/test_golden.py::test_hip_reproduces_full_size_checksums and
 tests/test_gpu_parity.py::test_headline_config_full_size.)"""
import hashlib
import json
import os
import zlib

import numpy as np
import pytest

from conftest import SIM_SEED, TABLE_SEED

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
T = 24


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_s8k_rank_shards_of_the_8_gpu_configuration(cpm):
    c = json.load(open(os.path.join(G, "s8k_shard_checksums.json")))
    Z, cpz, C = c["Z"], c["cpz"], c["C"]
    with cpm.Sampler(Z, T) as s:
        s.synth_tables(c["table_seed"])
        # (interleaved shards first: the context's bucket regions only ever grow, and the contiguous shards make them grow)
        for name, sh in sorted(c["shards"].items(), key=lambda kv: -kv[1]["car_stride"]):
            s.init_states(C, cpz, sh["car_first"], sh["car_count"], car_stride=sh["car_stride"])
            assert s.get_info(1) == cpm.CPM_KERNEL_ZONE_GROUPED, name           # what AUTO resolves to for a rank's shard
            init = s.solve_ivp(c["sim_seed"])
            assert _sha(init) == sh["initial_state_sha256"], name
            r = s.resample(c["sim_seed"])
            assert (r["parking"].sum(axis=0) == sh["car_count"]).all(), name     # every hour holds the whole shard
            assert _sha(r["parking"].ravel(order="F")) == sh["parking_sha256"], name
            assert _sha(r["driving"].ravel(order="F")) == sh["driving_sha256"], name
            assert int(r["driving"].sum()) == sh["driving_total"], name
            assert [int(x) for x in r["parking"][:8, 23]] == sh["parking_hour24_first8"], name
            assert int(r["parking"].max()) == sh["parking_max"], name
            assert s.get_info(1) == cpm.CPM_KERNEL_ZONE_GROUPED, name           # the grouped layout survived
            if sh["car_stride"] > 1:
                assert s.get_info(2) == 4, name     # interleaved deal: every bucket starts at its mean size, nothing had to grow
            else:
                assert s.get_info(2) >= 8, name     # contiguous deal: 8x the mean bucket in IVP hour 1, regions doubled (once or twice)


def test_melbourne_shaped_full_fleet_with_travel_times(cpm, O):
    """configs[1]: sparse Melbourne-shaped datamatrix -> createpdrive / createpdestin on the device -> IVP -> resample with travel
    times, 1,000 cars/zone (C = 2,357,000), against the oracle run on the tables the device returned."""
    Z, cpz = 2357, 1000
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED)
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        p_drive = s.build_p_drive(0.1, 0.9, 0.5)
        p_dest = s.build_p_dest(2)
        np.testing.assert_allclose(p_drive, O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5), rtol=4e-16, atol=0, equal_nan=True)
        assert np.array_equal(p_dest, O.createpdestin(dm, Z, T, 2))
        cdf = O.build_cdf(p_dest)
        del p_dest
        ref = O.fast_run(p_drive, cdf, C, SIM_SEED, np.arange(C, dtype=np.int64) // cpz + 1, datamatrix=dm, dist=dist)
        s.init_states(C, cpz)
        assert np.array_equal(s.solve_ivp(SIM_SEED), ref["zone0"])
        r = s.resample(SIM_SEED, travel=True)
        s.set_fused(3)                                  # the placing-first form of the hour at full size: the same counts
        assert s.get_info(4) == 3
        r3 = s.resample(SIM_SEED, travel=True)
        assert s.get_info(4) == 3                       # (no sampler workgroup gave up or found its arrivals on another XCD)
    assert np.array_equal(r3["parking"], ref["parking"]) and np.array_equal(r3["driving"], ref["driving"]) and r3["sum_tt_q16"] == ref["sum_tt_q16"]
    assert np.array_equal(r["parking"], ref["parking"])
    assert np.array_equal(r["driving"], ref["driving"])
    assert r["sum_tt_q16"] == ref["sum_tt_q16"]
    assert (r["parking"].sum(axis=0) == C).all()
    np.testing.assert_allclose(r["parking"] / C, ref["parking"] / C, rtol=1e-6, atol=0)


def test_model_selection_grid_of_256_points_at_melbourne_size(cpm, O):
    """configs[4]: make_grid()'s 256 points x the Melbourne-shaped resample (Z = 2,357, 1,000 cars/zone, travel times), dealt over
    8 emulated ranks on one GPU (points are independent: no data-path collective).  Every point: the size-independent
    properties.  Eight points (two per e_dest value): counts, travel-time sum and both errors bit-exact against the oracle's
    fast twin run on the tables the device built for that point, from the same post-IVP state (README.md:1180: the IVP is not
    re-run per point; clamp = src/correctparameters.jl:3-22 is exercised by the searches in test_model_selection.py)."""
    from carparkingmaps_amd import model_selection as ms
    Z, cpz, world = 2357, 1000, 8
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED)
    rng = np.random.default_rng(11)
    measured_act, measured_park = rng.uniform(0, 1, T), rng.uniform(0, 1, (Z, T))
    grid = ms.make_grid()
    assert len(grid) == 256
    with cpm.Sampler(Z, T) as s:
        s.set_datamatrix(dm, dist)
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        s.build_p_dest(2, want=False)
        s.init_states(C, cpz)
        init = s.solve_ivp(SIM_SEED)                                   # once; every point restarts from it
        ev = ms.Evaluator(s, C, SIM_SEED, measured_act, measured_park, travel=True)
        per_rank = [ms.grid_sweep(ev, grid, rank=r, world_size=world, gather=False, checksums=True) for r in range(world)]
        results = []
        for i in range(len(grid)):
            owners = [r for r in range(world) if per_rank[r][i] is not None]
            assert len(owners) == 1, i                                 # every point evaluated by exactly one rank
            results.append(per_rank[owners[0]][i])
        assert sorted(sum(1 for x in pr if x is not None) for pr in per_rank) == [32] * world
        # a rank's points share few e_dest values (the CDF and the row packs are rebuilt only when e_dest changes)
        for r in range(world):
            assert len({results[i]["e_dest"] for i in range(len(grid)) if per_rank[r][i] is not None}) <= 1
        for i, (pt, got) in enumerate(zip(grid, results)):
            assert (got["e_drive"], got["p_min"], got["p_max"], got["e_dest"]) == (pt.e_drive, pt.p_min, pt.p_max, float(pt.e_dest))
            assert got["hours_hold_all_cars"], i                       # sum_z parking[z,t] = C for every hour
            assert 0 <= got["driving_total"] <= C * T
            assert 0.0 <= got["A_drive"] < 1.0 and np.isfinite(got["activity_error"]) and np.isfinite(got["parking_error"])
        # monotone in what it must be monotone in: more driving when the drive probabilities rise (same e_drive, e_dest, p_min)
        by = {(p.e_drive, p.p_min, p.p_max, float(p.e_dest)): r for p, r in zip(grid, results)}
        assert by[(0.5, 0.1, 0.5, 2.0)]["driving_total"] < by[(0.5, 0.1, 0.7, 2.0)]["driving_total"] < by[(0.5, 0.1, 1.0, 2.0)]["driving_total"]
        # eight sampled points against the oracle
        picks = [i for e in (0, 1, 2, 3) for i in (64 * e + 5, 64 * e + 58)]
        assert len({float(grid[i].e_dest) for i in picks}) == 4
        last_e, cdf = None, None
        for i in picks:
            pt = grid[i]
            p_drive = s.build_p_drive(pt.p_min, pt.p_max, pt.e_drive)
            key = (type(pt.e_dest).__name__, float(pt.e_dest))
            if key != last_e:
                cdf = None
                p_dest = s.build_p_dest(pt.e_dest)
                cdf = O.build_cdf(p_dest)
                del p_dest
                last_e = key
            ref = O.fast_run(p_drive, cdf, C, SIM_SEED, init, do_ivp=False, datamatrix=dm, dist=dist)
            got = results[i]
            assert got["driving_total"] == int(ref["driving"].sum()), i
            assert got["parking_crc32"] == zlib.crc32(np.ascontiguousarray(ref["parking"].ravel(order="F")).tobytes()), i
            assert got["driving_crc32"] == zlib.crc32(np.ascontiguousarray(ref["driving"].ravel(order="F")).tobytes()), i
            assert got["A_drive"] == ms.a_drive(ref["sum_tt_q16"], C, T), i
            assert got["activity_error"] == pytest.approx(ms.traffic_activity_error(ms.traffic_activity(ref["driving"]), measured_act), rel=1e-12)
            assert got["parking_error"] == pytest.approx(ms.parking_density_error(ref["parking"], C, measured_park), rel=1e-12)
