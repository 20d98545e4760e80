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
    from carparkingmaps_amd.distributed import allreduce_counts, shard_range, split_counts
    from oracle import oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    T, C = 24, Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, TABLE_SEED), O.synth_p_dest_dense(Z, T, TABLE_SEED)
    b, n = shard_range(C, rank, world)
    zone0 = (np.arange(b, b + n) // cpz + 1).astype(np.int64)
    r = O.fast_run(p_drive, O.build_cdf(p_dest), n, SIM_SEED, zone0, car_offset=b, nthreads=2)
    # same word layout the device path fills: parking[T][Z] | driving[T][Z] | sum_tt_q16 | status
    counts = torch.from_numpy(np.concatenate([r["parking"].ravel(order="F"), r["driving"].ravel(order="F"),
                                              [r["sum_tt_q16"], 0]]).astype(np.int64))
    allreduce_counts(counts)
    pk, dr, _ = split_counts(counts, Z, T)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), parking=pk, driving=dr)
    dist.destroy_process_group()


def test_two_rank_count_allreduce_equals_single_run(O, tmp_path):
    import torch.multiprocessing as mp
    Z, cpz, world = 19, 33, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, Z, cpz, str(tmp_path)), nprocs=world, join=True)
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, 24, TABLE_SEED), O.synth_p_dest_dense(Z, 24, TABLE_SEED)
    ref = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, np.arange(C) // cpz + 1)
    for rank in range(world):
        got = np.load(tmp_path / f"rank{rank}.npz")
        assert np.array_equal(got["parking"], ref["parking"])
        assert np.array_equal(got["driving"], ref["driving"])
