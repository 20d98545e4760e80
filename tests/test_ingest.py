"""Data formats either side of the sampler path: the native Uber-CSV reader, the GeoJSON walk and
centroid sums of the host layer, and the oracle restatements they are checked against.  CPU only (the device side is
in tests/test_gpu_parity.py)."""
import json
import os

import numpy as np
import pytest

HEADER = ("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,"
          "geometric_standard_deviation_travel_time\n")


def _write_csv(path, rows, header=HEADER, eol="\n", final_eol=True):
    lines = [",".join(repr(float(v)) if isinstance(v, float) else str(v) for v in r) for r in rows]
    text = header + eol.join(lines) + (eol if final_eol else "")
    with open(path, "w", newline="") as f:
        f.write(text)


def test_csv_reader_matches_numpy_on_the_reference_format(cpm, tmp_path):
    from carparkingmaps_amd.sampler import parse_uber_csv
    rng = np.random.default_rng(3)
    n = 50_000
    rows = [(int(rng.integers(0, 2357)), int(rng.integers(0, 2357)), int(rng.integers(0, 24)), float(np.round(rng.random() * 3000, 2)),
             float(np.round(rng.random() * 400, 2)), float(np.round(rng.random() * 3000, 2)), float(np.round(1 + rng.random(), 2)))
            for _ in range(n)]
    p = tmp_path / "melbourne-2019-1-All-HourlyAggregate.csv"
    _write_csv(p, rows)
    got = parse_uber_csv(str(p))
    want = np.loadtxt(p, delimiter=",", skiprows=1, usecols=range(5))
    assert got.shape == (n, 5) and np.array_equal(got, want)  # bit-equal doubles, rows in file order


@pytest.mark.parametrize("eol,final_eol", [("\n", True), ("\n", False), ("\r\n", True), ("\r\n", False)])
def test_csv_reader_line_ends_blank_lines_and_number_forms(cpm, tmp_path, eol, final_eol):
    from carparkingmaps_amd.sampler import parse_uber_csv
    p = tmp_path / "a.csv"
    body = eol.join(["1,2,0,10.5,1.5,9,1.1", "", "0,3,5,2e1,2,9,1.2", " 7 , 8,9,.5,5.,1,1",
                     "3,4,5,0.1234567890123456789,123456789012345678,1,1", "5,6,7,1e-3,-0.0,1,1"]) + (eol if final_eol else "")
    with open(p, "w", newline="") as f:
        f.write(HEADER.replace("\n", eol) + body)
    got = parse_uber_csv(str(p))
    want = np.array([[1, 2, 0, 10.5, 1.5], [0, 3, 5, 20.0, 2.0], [7, 8, 9, 0.5, 5.0],
                     [3, 4, 5, 0.1234567890123456789, 123456789012345678.0], [5, 6, 7, 1e-3, -0.0]])
    assert np.array_equal(got, want)


def test_csv_reader_file_of_exactly_whole_pages_and_empty_files(cpm, tmp_path):
    from carparkingmaps_amd.sampler import parse_uber_csv
    page = os.sysconf("SC_PAGESIZE")
    line = "12,34,5,678.25,9.5,1,1"
    p = tmp_path / "pages.csv"
    text = "h\n"
    while len(text) + len(line) + 1 + len("1,2,3,4,5") <= 2 * page:
        text += line + "\n"
    text += "1,2,3,4,5"                      # last line without a line end ...
    text += "0" * (2 * page - len(text))     # ... padded with digits to end exactly on a page boundary
    assert len(text) == 2 * page
    p.write_text(text)
    got = parse_uber_csv(str(p))
    assert got[0].tolist() == [12, 34, 5, 678.25, 9.5] and got[-1, :4].tolist() == [1, 2, 3, 4]
    e = tmp_path / "empty.csv"
    e.write_text("")
    assert parse_uber_csv(str(e)).shape == (0, 5)
    h = tmp_path / "header_only.csv"
    h.write_text(HEADER)
    assert parse_uber_csv(str(h)).shape == (0, 5)


@pytest.mark.parametrize("bad", ["1,2,x,3,4,5,6", "1,2,3,4", "1,2,3,,5,6,7", "1;2;3;4;5"])
def test_csv_reader_reports_the_offending_line(cpm, tmp_path, bad):
    from carparkingmaps_amd.sampler import parse_uber_csv
    p = tmp_path / "bad.csv"
    p.write_text(HEADER + "1,2,3,4,5,6,7\n" + bad + "\n1,2,3,4,5,6,7\n")
    with pytest.raises(cpm.CpmError, match="line 3"):
        parse_uber_csv(str(p))
    with pytest.raises(cpm.CpmError):
        parse_uber_csv(str(tmp_path / "missing.csv"))


def test_oracle_createdatamatrix_remaps_and_last_row_wins(O):
    Z, T = 4, 24
    raw = np.array([[1, 2, 0, 10.5, 1.5],      # hod 0 -> hour 24
                    [0, 3, 5, 20.0, 2.0],      # source 0 -> zone Z
                    [2, 0, 7, 30.0, 3.0],      # dest 0 -> zone Z
                    [1, 2, 0, 11.5, 1.25],     # same cell as row 1: overwrites it
                    [4, 4, 24, 99.0, 9.0]])    # hod 24 is in range as it stands
    dm = O.createdatamatrix(raw, Z, T)
    assert dm[0, 1, 23, 0] == 11.5 and dm[0, 1, 23, 1] == 1.25
    assert dm[3, 2, 4, 0] == 20.0 and dm[1, 3, 6, 1] == 3.0 and dm[3, 3, 23, 0] == 99.0
    assert np.count_nonzero(dm) == 8
    for bad in ([5, 1, 1, 1, 1], [1, 1, 25, 1, 1], [1.5, 1, 1, 1, 1], [-1, 1, 1, 1, 1]):
        with pytest.raises(RuntimeError):
            O.createdatamatrix(np.array([bad], dtype=float), Z, T)


def _square(x0, y0, s):
    return [[x0, y0], [x0 + s, y0], [x0 + s, y0 + s], [x0, y0 + s], [x0, y0]]


def _feature(mid, coords, kind="Polygon"):
    return {"type": "Feature", "properties": {"MOVEMENT_ID": str(mid), "DISPLAY_NAME": f"zone {mid}"},
            "geometry": {"type": kind, "coordinates": coords}}


def test_geojson_walk_keeps_the_reference_quirks(cpm):
    from carparkingmaps_amd.reference_api import geojson_vertex_lists
    feats = [_feature(0, [_square(144.5, -37.5, 0.25)]),                                   # id 0 -> zone number_zones
             _feature(1, [[_square(145.5, -38.5, 0.5)], [_square(10.5, 20.5, 1.0)]], "MultiPolygon"),
             _feature(2, [[[145, -37.25], [145.5, -37.25], [145.5, -36.75], [145, -36.75], [145, -37.25]]]),  # integer longitudes
             _feature(3, [[[[[[_square(1.5, 2.5, 1.0)]]]]]])]                              # too deep: never stored
    Z, zones = geojson_vertex_lists(feats)
    assert Z == 4                                    # first id 0 -> last id + 1 (:12-17)
    lons, lats = zones[4]                            # id 0 stored as zone 4 (:22-25)
    assert lons == [144.5, 144.5, 144.75, 144.75, 144.75, 144.75, 144.5, 144.5, 144.5, 144.5]  # every pair twice: once per Float64 leaf
    assert lats == [-37.5, -37.5, -37.5, -37.5, -37.25, -37.25, -37.25, -37.25, -37.5, -37.5]
    assert len(zones[1][0]) == 20                    # both polygons of the MultiPolygon, in order
    assert zones[2][0] == [145, 145.5, 145.5, 145.5, 145.5, 145, 145]  # the integer leaf 145 is skipped, its Float64 sibling stores the pair once
    assert zones[3] == ([], [])


def test_centroids_of_the_host_layer_equal_the_oracle(cpm, O):
    from carparkingmaps_amd.reference_api import geojson_vertex_lists, polygon_centroids
    rng = np.random.default_rng(5)
    feats = []
    for k in range(1, 41):
        m = int(rng.integers(3, 60))
        ang = np.sort(rng.random(m) * 2 * np.pi)
        r = 0.02 + 0.03 * rng.random(m)
        cx, cy = 144.0 + 2.0 * rng.random(), -38.5 + 1.5 * rng.random()
        ring = [[float(cx + r[i] * np.cos(ang[i])), float(cy + r[i] * np.sin(ang[i]))] for i in range(m)]
        ring.append(ring[0])
        feats.append(_feature(k, [ring]))
    feats[7]["geometry"]["coordinates"] = []         # a zone without data: centroid stays (0, 0)
    Z, zones = geojson_vertex_lists(feats)
    clat, clong = polygon_centroids(Z, zones)
    lon_rows = [zones[z][0] for z in range(1, Z + 1)]
    lat_rows = [zones[z][1] for z in range(1, Z + 1)]
    olat, olong, area = O.centroids(lon_rows, lat_rows)
    assert np.array_equal(clat, olat) and np.array_equal(clong, olong)   # same operations in the same order
    assert clat[7] == 0 and clong[7] == 0 and area[7] == 0
    sq = polygon_centroids(1, {1: ([1.0, 1.0, 2.0, 2.0, 2.0, 2.0, 1.0, 1.0], [1.0, 1.0, 1.0, 1.0, 2.0, 2.0, 2.0, 2.0])})
    assert sq[0][0] == 1.5 and sq[1][0] == 1.5       # unit square, open ring: the loop closes it itself (:111-115)


def test_oracle_distance_matrix_formula(O):
    lat = np.array([-37.8, -37.9, -37.7, 0.0])
    lon = np.array([144.9, 145.0, 145.1, 0.0])
    d = O.distance_matrix(lat, lon)
    assert np.array_equal(d, d.T) and np.array_equal(np.diag(d), np.ones(4))
    want = 111.3 * np.sqrt(np.cos((lat[0] + lat[1]) / 2 * 0.01745) ** 2 * (lon[0] - lon[1]) ** 2 + (lat[0] - lat[1]) ** 2)
    assert abs(d[0, 1] - want) <= 1e-12 * want


def test_saveparameters_and_results_directory(cpm, tmp_path):
    from carparkingmaps_amd.reference_api import createresultsdirectory, saveparameters
    root = str(tmp_path) + "/"
    path = createresultsdirectory(root, "Melbourne")
    assert path == root + "Melbourne" and os.path.isdir(path)
    assert createresultsdirectory(root, "Melbourne") == path          # exists already: returned as is
    saveparameters(path, 24, 2357, 1000, 2357000, 0.5, 0.1, 0.9, 2, 0.0625)
    head, row = open(os.path.join(path, "sampling_parameters.csv")).read().splitlines()
    assert head.split(",")[0] == "T (time steps)" and head.split(",")[-1] == "A_drive" and len(head.split(",")) == 9
    assert row == "24.0,2357.0,1000.0,2.357e6,0.5,0.1,0.9,2.0,0.0625"
