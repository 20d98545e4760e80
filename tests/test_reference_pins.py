"""What the reference itself holds that can pin this path: two notebook printouts in README.md, transcribed by
tests/golden/make_readme_pins.py (README.md:505-531: `state_matrix` after the initial placement; README.md:259: the first 30
rows of the Melbourne Uber Movement file; README.md:264,310: 2,357 zones, 11,566,494 rows, 91 % sparsity).

They pin INPUTS of the path -- a1 (initializestates) and f2 (the CSV reader and createdatamatrix, with the hod 0 -> 24 remap of
row 11) -- for the oracle (CPU) and for the HIP library (GPU).  The sampler itself stays "parity unpinned": the reference is
unseeded and holds no output of it (Oracle)."""
import json
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _pin_state():
    return json.load(open(os.path.join(G, "readme_initializestates.json")))


def _pin_rows():
    d = json.load(open(os.path.join(G, "readme_rawdata_rows.json")))
    return d, np.array(d["rows"], dtype=np.float64)


def _check_first_column(col0, pin):
    C = pin["shape"][0]
    top = np.array(pin["first_rows"])
    bottom = np.array(pin["last_rows"])
    assert col0.shape == (C,)
    assert np.array_equal(col0[:len(top)], top[:, 0])            # rows 1..13 of the printout: zone 1
    assert np.array_equal(col0[C - len(bottom):], bottom[:, 0])  # the last 12 rows: zone 2357
    assert (top[:, 1:] == 0).all() and (bottom[:, 1:] == 0).all()
    # what the printed corners imply together with the loop they came from (src/initializestates.jl:11-16): blocks of cars_per_zone
    cpz = pin["cars_per_zone"]
    assert col0[cpz - 1] == 1 and col0[cpz] == 2 and col0[C - cpz] == bottom[0, 0] and col0[C - cpz - 1] == bottom[0, 0] - 1


def test_oracle_and_host_mirror_reproduce_the_readme_state_matrix(O, cpm):
    pin = _pin_state()
    C, T = pin["shape"]
    st, _ = O.initializestates(C, pin["cars_per_zone"], T, with_trans=False)
    assert st.shape == (C, T) and st.dtype == np.int64
    _check_first_column(st[:, 0], pin)
    assert not st[:, 1:].any()                                   # columns 2..24 are zero until written
    from carparkingmaps_amd import reference_api as R
    saved = (R.params.cars_per_zone, R.params.T)
    try:
        R.params.cars_per_zone, R.params.T = pin["cars_per_zone"], T
        sm, tm = R.initializestates(C)
        assert sm.shape == (C, T) and tm.shape == (C, T, 4) and sm.dtype == np.int64
        _check_first_column(sm[:, 0], pin)
        assert not sm[:, 1:].any()
    finally:
        R.params.cars_per_zone, R.params.T = saved


@pytest.mark.gpu
def test_hip_initial_placement_equals_the_readme_printout(cpm):
    pin = _pin_state()
    C, T = pin["shape"]
    Z = pin["last_rows"][-1][0]
    with cpm.Sampler(Z, T) as s:
        s.init_states(C, pin["cars_per_zone"])
        _check_first_column(s.get_state(), pin)
        # a rank's share under both deals holds the same cars' zones
        for first, stride in ((0, 8), (7, 8)):
            n = len(range(first, C, stride))
            s.init_states(C, pin["cars_per_zone"], first, n, car_stride=stride)
            assert np.array_equal(s.get_state(), (first + stride * np.arange(n)) // pin["cars_per_zone"] + 1)


def _expected_cells(rows, Z, T=24):
    """The loop of src/createdatamatrix.jl:9-22 applied by hand to the printed rows: (origin, destination, hour) -> (mean, std)."""
    cells = {}
    for r in rows:
        o, d, h = int(r[0]) or Z, int(r[1]) or Z, int(r[2]) or 24
        cells[(o, d, h)] = (r[3], r[4])
    return cells


def test_csv_reader_and_oracle_on_the_readme_rows(O, cpm):
    from carparkingmaps_amd.sampler import parse_uber_csv
    meta, rows = _pin_rows()
    raw = parse_uber_csv(os.path.join(G, "readme_rawdata_rows.csv"))
    assert raw.shape == (30, 5)
    assert np.array_equal(raw, rows[:, :5])                      # every field is the printed number, bit for bit
    assert raw[10, 2] == 0                                       # row 11: hod = 0 (midnight)
    Z = 96                                                       # any number of zones above the largest printed id (87)
    dm = O.createdatamatrix(raw, Z)
    cells = _expected_cells(rows, Z)
    assert len(cells) == 30 and (18, 35, 24) in cells            # row 11 lands in hour 24 (src/createdatamatrix.jl:15-17)
    for (o, d, h), (m, sd) in cells.items():
        assert dm[o - 1, d - 1, h - 1, 0] == m and dm[o - 1, d - 1, h - 1, 1] == sd
    assert np.count_nonzero(dm) == 60
    # the notebook's own size statements (README.md:264,302-310) are consistent with each other
    Ma = meta["number_zones"] * (meta["number_zones"] - 1) * 24
    assert round(100 * (1 - meta["rows_total"] / Ma)) == meta["sparsity_percent"]


@pytest.mark.gpu
def test_hip_createdatamatrix_on_the_readme_rows(cpm):
    _, rows = _pin_rows()
    Z = 96
    with cpm.Sampler(Z, 24) as s:
        assert s.createdatamatrix_csv(os.path.join(G, "readme_rawdata_rows.csv")) == 30
        dm = s.get_datamatrix()
    cells = _expected_cells(rows, Z)
    for (o, d, h), (m, sd) in cells.items():
        assert dm[o - 1, d - 1, h - 1, 0] == m and dm[o - 1, d - 1, h - 1, 1] == sd
    assert np.count_nonzero(dm) == 60
