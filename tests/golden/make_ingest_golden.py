#!/usr/bin/env python3
"""Generates the committed ingest fixture: a small Uber-Movement-shaped CSV, a GeoJSON with the nesting forms
src/processgeodata.jl walks, and what the reference's createdatamatrix / processgeodata make of them according to the
CPU oracle (oracle/cpm_oracle.c; "parity unpinned" against the Julia program, which cannot run here).

    python tests/golden/make_ingest_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402
from carparkingmaps_amd.reference_api import geojson_vertex_lists  # noqa: E402  (host-side JSON walk, no GPU needed)

Z, T = 12, 24
rng = np.random.default_rng(20261004)
rows = []
for _ in range(400):
    rows.append((int(rng.integers(0, Z)), int(rng.integers(0, Z)), int(rng.integers(0, 24)), round(300 + 2100 * float(rng.random()), 2),
                 round(30 + 300 * float(rng.random()), 2)))
rows += [rows[5][:3] + (1234.5, 67.25), rows[17][:3] + (999.0, 0.0), (0, 0, 0, 600.25, 60.5)]  # repeated keys (last row wins), id 0 / hod 0 remaps
with open(os.path.join(HERE, "ingest_small.csv"), "w") as f:
    f.write("sourceid,dstid,hod,mean_travel_time,standard_deviation_travel_time,geometric_mean_travel_time,"
            "geometric_standard_deviation_travel_time\n")
    for r in rows:
        f.write(f"{r[0]},{r[1]},{r[2]},{r[3]!r},{r[4]!r},{round(r[3] * 0.93, 2)!r},1.31\n")


def ring(cx, cy, m, r0):
    ang = np.sort(rng.random(m) * 2 * np.pi)
    rad = r0 * (0.6 + 0.4 * rng.random(m))
    pts = [[round(float(cx + rad[i] * np.cos(ang[i])), 6), round(float(cy + rad[i] * np.sin(ang[i])), 6)] for i in range(m)]
    return pts + [pts[0]]


feats = []
for k in range(Z):  # MOVEMENT_IDs 0 .. Z-1: id 0 becomes zone Z
    cx, cy = 144.5 + 0.12 * (k % 4), -38.1 + 0.11 * (k // 4)
    if k % 3 == 0:
        geom = {"type": "MultiPolygon", "coordinates": [[ring(cx, cy, 9, 0.03)], [ring(cx + 0.04, cy + 0.04, 5, 0.01)]]}
    else:
        geom = {"type": "Polygon", "coordinates": [ring(cx, cy, 7 + k, 0.035)]}
    feats.append({"type": "Feature", "properties": {"MOVEMENT_ID": str(k), "DISPLAY_NAME": f"Zone {k}"}, "geometry": geom})
with open(os.path.join(HERE, "ingest_small.geojson"), "w") as f:
    json.dump({"type": "FeatureCollection", "features": feats}, f)

raw = np.loadtxt(os.path.join(HERE, "ingest_small.csv"), delimiter=",", skiprows=1, usecols=range(5))
dm = O.createdatamatrix(raw, Z, T)
number_zones, zones = geojson_vertex_lists(feats)
clat, clong, area = O.centroids([zones[z][0] for z in range(1, Z + 1)], [zones[z][1] for z in range(1, Z + 1)])
dist = O.distance_matrix(clat, clong)
nz = np.flatnonzero(dm.reshape(-1, order="F"))
np.savez_compressed(os.path.join(HERE, "ingest_small_expected.npz"), Z=Z, T=T, n_rows=raw.shape[0], dm_index=nz,
                    dm_value=dm.reshape(-1, order="F")[nz], centroid_lat=clat, centroid_long=clong, area=area, dist=dist,
                    vertex_counts=np.array([len(zones[z][0]) for z in range(1, Z + 1)]))
print("wrote ingest_small.csv / .geojson / _expected.npz:", raw.shape[0], "rows,", nz.size, "non-zero datamatrix entries")
