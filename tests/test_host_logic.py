"""Host-side logic that needs no GPU: the mirrored call surface's pure parts, car sharding, and the
N > 1 count reduction (world_size 2 over gloo, with the oracle standing in for the device)."""
import os
import socket

import numpy as np
import pytest

from conftest import ROOT, SIM_SEED, TABLE_SEED


def test_shard_ranges_tile_the_cars():
    from carparkingmaps_amd.distributed import shard_range
    for C in (0, 1, 7, 4096000, 32768001):
        for world in (1, 2, 3, 8):
            pos = 0
            sizes = []
            for r in range(world):
                b, n = shard_range(C, r, world)
                assert b == pos and n >= 0
                pos += n
                sizes.append(n)
            assert pos == C and max(sizes) - min(sizes) <= 1


def test_initializestates_mirror(cpm, O):
    cpm.params.cars_per_zone, cpm.params.T = 7, 24
    try:
        st, tr = cpm.initializestates(35)
        st_o, tr_o = O.initializestates(35, 7, 24)
        assert np.array_equal(st, st_o) and np.array_equal(tr, tr_o)
        assert st.dtype == np.int64 and st.flags.f_contiguous and tr.shape == (35, 24, 4)
    finally:
        cpm.params.__init__()


def test_correctparameters_mirror(cpm, O):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.uniform(-0.5, 1.5, size=4)
        assert cpm.correctparameters(*a) == O.correctparameters(*a)


def test_averagedrivingtime_mirror(cpm, O):
    rng = np.random.default_rng(1)
    tr = np.asfortranarray(rng.uniform(0, 3000, size=(50, 24, 4)))
    assert cpm.averagedrivingtime(50, 0.5, tr) == pytest.approx(O.averagedrivingtime(50, 0.5, tr), rel=1e-14)


def test_julia_float_text():
    from carparkingmaps_amd.reference_api import julia_float
    assert julia_float(0.25) == "0.25"
    assert julia_float(1.0) == "1.0"
    assert julia_float(1e-5) == "1.0e-5"
    assert julia_float(2.5e-7) == "2.5e-7"
    assert julia_float(0.0001) == "0.0001"
    assert julia_float(float("nan")) == "NaN"
    assert julia_float(0.1 + 0.2) == "0.30000000000000004"


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, Z, cpz, out_dir):
    import sys
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from carparkingmaps_amd.distributed import allreduce_counts, shard_cars, split_counts
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, C = 24, Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, TABLE_SEED), O.synth_p_dest_dense(Z, T, TABLE_SEED)
    cdf = O.build_cdf(p_dest)
    out = {}
    for deal in ("interleaved", "contiguous"):
        first, stride, n = shard_cars(C, rank, world, deal)
        zone0 = ((first + stride * np.arange(n)) // cpz + 1).astype(np.int64)
        r = O.fast_run(p_drive, cdf, n, SIM_SEED, zone0, car_offset=first, car_stride=stride, nthreads=2)
        # same word layout the device path fills: parking[T][Z] | driving[T][Z] | sum_tt_q16 | status
        counts = torch.from_numpy(np.concatenate([r["parking"].ravel(order="F"), r["driving"].ravel(order="F"),
                                                  [r["sum_tt_q16"], 0]]).astype(np.int64))
        work = allreduce_counts(counts, async_op=(deal == "interleaved"))   # both forms of the collective
        if work is not None:
            work.wait()
        pk, dr, _ = split_counts(counts, Z, T)
        out[deal + "_parking"], out[deal + "_driving"] = pk, dr
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), **out)
    dist.destroy_process_group()


def test_two_rank_count_allreduce_equals_single_run(O, tmp_path):
    """world size 2 over gloo: each rank samples its share of the cars (oracle), the int64 count tensors are all-reduced; the sum
    equals the single run for the interleaved deal (car g -> rank g mod N) and for the contiguous one."""
    import torch.multiprocessing as mp
    Z, cpz, world = 19, 33, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, Z, cpz, str(tmp_path)), nprocs=world, join=True)
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, 24, TABLE_SEED), O.synth_p_dest_dense(Z, 24, TABLE_SEED)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, np.arange(C) // cpz + 1)
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        for deal in ("interleaved", "contiguous"):
            assert np.array_equal(got[deal + "_parking"], ref["parking"]), deal
            assert np.array_equal(got[deal + "_driving"], ref["driving"]), deal


def test_shard_cars_tile_the_fleet():
    from carparkingmaps_amd.distributed import shard_cars
    for C in (0, 1, 7, 64, 1000, 1001):
        for world in (1, 2, 3, 8):
            for deal in ("interleaved", "contiguous"):
                seen = np.zeros(C, dtype=np.int64)
                for rank in range(world):
                    first, stride, count = shard_cars(C, rank, world, deal)
                    cars = first + stride * np.arange(count)
                    assert count == 0 or cars[-1] < C
                    seen[cars] += 1
                assert (seen == 1).all(), (C, world, deal)


def test_host_array_stamp_sees_in_place_edits(cpm):
    """What decides whether a host table is uploaded again: address, shape and a pass over the WHOLE buffer -- an edit of any
    single element changes it (the sampled checksum it replaces missed most of them); params.trust_unchanged skips the pass."""
    from carparkingmaps_amd import reference_api as R
    rng = np.random.default_rng(2)
    a = np.asfortranarray(rng.random((37, 37, 24)))
    s0 = R._stamp(a)
    assert R._stamp(a) == s0
    for idx in [(0, 0, 0), (36, 36, 23), (5, 17, 11), (1, 0, 0)]:
        b = a[idx]
        a[idx] = np.nextafter(b, 2.0)                    # one ulp in one element
        assert R._stamp(a) != s0, idx
        a[idx] = b
    assert R._stamp(a) == s0
    # permutations in place keep every word (a plain sum of the words would not change): two zones' rows swapped, a slab sorted
    a[[3, 7]] = a[[7, 3]]
    assert R._stamp(a) != s0
    a[[3, 7]] = a[[7, 3]]
    assert R._stamp(a) == s0
    keep = a[:, :, 5].copy()
    a[:, :, 5] = np.sort(keep, axis=0)
    assert R._stamp(a) != s0
    a[:, :, 5] = keep
    assert R._stamp(a) == s0
    big = np.asfortranarray(rng.random((3, 1 << 20)))      # several chunks of the fingerprint: words swapped across a chunk border
    b0 = R._stamp(big)
    flat = big.reshape(-1, order="A")
    i, j = (1 << 20) - 1, (1 << 20) + 5
    flat[i], flat[j] = flat[j], flat[i]
    assert R._stamp(big) != b0
    flat[i], flat[j] = flat[j], flat[i]
    assert R._stamp(big) == b0
    assert R._stamp(a.copy(order="F"))[0] != s0[0]       # another buffer: another address
    R.params.trust_unchanged = True
    try:
        t0 = R._stamp(a)
        a[3, 3, 3] += 1.0
        assert R._stamp(a) == t0                         # the caller took responsibility
    finally:
        R.params.__init__()


def test_geojson_duplicate_zone_ids_keep_the_old_tail(cpm):
    """src/processgeodata.jl:19-97 assigns the coordinate matrices element by element: a later feature with an id seen before
    overwrites the row from column 1 and leaves the rest of the earlier, longer entry in place."""
    from carparkingmaps_amd.reference_api import geojson_vertex_lists
    def poly(pts):
        return {"type": "Polygon", "coordinates": [[list(p) for p in pts]]}
    long_ring = [(144.0 + 0.01 * k, -37.0 - 0.01 * k) for k in range(6)]
    short_ring = [(145.5, -38.5), (145.6, -38.6)]
    feats = [{"properties": {"MOVEMENT_ID": "1"}, "geometry": poly(long_ring)},
             {"properties": {"MOVEMENT_ID": "2"}, "geometry": poly(short_ring)},
             {"properties": {"MOVEMENT_ID": "1"}, "geometry": poly(short_ring)},
             {"properties": {"MOVEMENT_ID": "3"}, "geometry": poly(short_ring)}]
    Z, zones = geojson_vertex_lists(feats)
    assert Z == 3
    lons, lats = zones[1]
    assert len(lons) == 2 * len(long_ring)               # every [long, lat] pair is stored once per Float64 leaf
    assert lons[:4] == [145.5, 145.5, 145.6, 145.6] and lats[:4] == [-38.5, -38.5, -38.6, -38.6]
    assert lons[4:] == [p[0] for p in long_ring for _ in (0, 1)][4:]
    assert zones[2][0] == [145.5, 145.5, 145.6, 145.6]
