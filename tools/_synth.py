"""Synthetic inputs for the development tools (numpy; NOT the oracle's generators: the tools time kernels, they do not check
results, and nothing outside tests/, bench.py's cpu_baseline leg and __graft_entry__.smoke() touches oracle/)."""
import numpy as np


def datamatrix(Z, T, seed=1, density=0.0868):
    """Melbourne-shaped datamatrix (Z, Z, T, 2) and distance matrix (Z, Z), Fortran order: `density` of the (o, d, t) cells hold
    a mean travel time of 300-2400 s and a standard deviation of 10-40 % of it; distances from random centroids in a 2 x 1.5 degree box."""
    rng = np.random.default_rng(seed)
    dm = np.zeros((Z, Z, T, 2), dtype=np.float64, order="F")
    for t in range(T):
        mask = rng.random((Z, Z)) < density
        np.fill_diagonal(mask, False)
        mean = np.where(mask, 300.0 + 2100.0 * rng.random((Z, Z)), 0.0)
        dm[:, :, t, 0] = mean
        dm[:, :, t, 1] = mean * (0.1 + 0.3 * rng.random((Z, Z)))
    lat = -38.5 + 1.5 * rng.random(Z)
    lon = 144.0 + 2.0 * rng.random(Z)
    c = np.cos((lat[:, None] + lat[None, :]) / 2 * 0.01745)
    dist = 111.3 * np.sqrt(c * c * (lon[:, None] - lon[None, :]) ** 2 + (lat[:, None] - lat[None, :]) ** 2)
    np.fill_diagonal(dist, 1.0)
    return dm, np.asfortranarray(dist)


def dense_tables(Z, T, seed=1):
    """p_drive (Z, T) in [0.1, 0.9] and a dense row-normalised p_dest (Z, Z, T) with a zero diagonal, Fortran order."""
    rng = np.random.default_rng(seed)
    p_drive = np.asfortranarray(0.1 + 0.8 * rng.random((Z, T)))
    p_dest = np.zeros((Z, Z, T), dtype=np.float64, order="F")
    for t in range(T):
        w = rng.random((Z, Z)) ** 2
        np.fill_diagonal(w, 0.0)
        p_dest[:, :, t] = w / w.sum(axis=1, keepdims=True)
    return p_drive, p_dest
