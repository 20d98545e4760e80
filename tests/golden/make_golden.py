#!/usr/bin/env python3
"""Generates the committed golden vectors from the CPU oracle (oracle/cpm_oracle.c).

The reference itself ships no fixtures, no seeds and cannot run here (Julia is absent), so these
vectors pin the BUILD's contract (Philox4x32-10 keyed by global car id, canonical sequential CDF,
deviation D1), not the Julia program's unseeded output: "parity unpinned" in the sense of the
task statement.  They protect against silent drift of the oracle and give the GPU tests vectors
that do not depend on the oracle being rebuilt.

    python tests/golden/make_golden.py            # small vectors (seconds)
    python tests/golden/make_golden.py --big      # + checksums of the headline configs (minutes, ~8 GB)
    python tests/golden/make_golden.py --s8k      # only: checksums of two ranks' shards of the 8-GPU configuration (~26 GB)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402

TABLE_SEED, SIM_SEED, T = 0x5EED7AB1E, 0x5EEDCA125, 24


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def small():
    Z, cpz = 24, 16
    C = Z * cpz
    dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED, density=0.35)
    p_drive = O.createpdrive(dm, dist, Z, T, 0.1, 0.9, 0.5)
    p_dest = O.createpdestin(dm, Z, T, 2)
    st, tr = O.initializestates(C, cpz, T)
    init = O.solveinitialvalueproblem(st, tr, p_drive, p_dest, C, Z, SIM_SEED)
    st, tr = O.initializestates(C, cpz, T)
    st[:, 0] = init
    O.resampling(st, tr, C, Z, p_drive, p_dest, dm, dist, SIM_SEED)
    pk, dr, dens = O.histogram(Z, st, tr)
    np.savez_compressed(os.path.join(HERE, "small_z24.npz"), Z=Z, cpz=cpz, table_seed=TABLE_SEED, sim_seed=SIM_SEED,
                        datamatrix=dm, dist=dist, p_drive=p_drive, p_dest=p_dest, cdf=O.build_cdf(p_dest),
                        initial_state=init, state=st, trans=tr, parking=pk.astype(np.int64), driving=dr.astype(np.int64),
                        density=dens, activity=O.trafficactivity(dr), A_drive=O.averagedrivingtime(C, 0.0, tr),
                        sum_tt_q16=O.sum_travel_time_q16(tr))
    # dense synthetic, a size that is not a multiple of anything
    Z, cpz = 37, 11
    C = Z * cpz
    p_drive, p_dest = O.synth_p_drive(Z, T, TABLE_SEED), O.synth_p_dest_dense(Z, T, TABLE_SEED)
    r = O.fast_run(p_drive, O.build_cdf(p_dest), C, SIM_SEED, np.arange(C) // cpz + 1, want_state=True)
    np.savez_compressed(os.path.join(HERE, "dense_z37.npz"), Z=Z, cpz=cpz, table_seed=TABLE_SEED, sim_seed=SIM_SEED,
                        p_drive=p_drive, p_dest=p_dest, initial_state=r["zone0"], state=r["state"],
                        parking=r["parking"], driving=r["driving"])
    kat = {"philox4x32_10": [
        {"ctr": [0, 0, 0, 0], "key": [0, 0], "out": O.philox4x32_10([0, 0, 0, 0], [0, 0])},
        {"ctr": [0xFFFFFFFF] * 4, "key": [0xFFFFFFFF] * 2, "out": O.philox4x32_10([0xFFFFFFFF] * 4, [0xFFFFFFFF] * 2)},
        {"ctr": [0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], "key": [0xA4093822, 0x299F31D0],
         "out": O.philox4x32_10([0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344], [0xA4093822, 0x299F31D0])}],
        "uniforms": [{"seed": SIM_SEED, "car": c, "step": s, "u": list(O.uniforms(SIM_SEED, c, s, 0))}
                     for c, s in [(0, 0), (1, 23), (4095999, 46), (2 ** 33 + 5, 7)]]}
    json.dump(kat, open(os.path.join(HERE, "rng_kat.json"), "w"), indent=1)


def big():
    out = {}
    for name, Z, cpz in [("s4k_dense_z4096_cpz1000", 4096, 1000), ("dense_z2357_cpz1000", 2357, 1000)]:
        C = Z * cpz
        p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
        p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED)
        cdf = O.build_cdf(p_dest)
        del p_dest
        r = O.fast_run(p_drive, cdf, C, SIM_SEED, np.arange(C, dtype=np.int64) // cpz + 1)
        del cdf
        out[name] = {"Z": Z, "cpz": cpz, "C": C, "table_seed": TABLE_SEED, "sim_seed": SIM_SEED,
                     "initial_state_sha256": sha(r["zone0"]), "parking_sha256": sha(r["parking"].ravel(order="F")),
                     "driving_sha256": sha(r["driving"].ravel(order="F")), "driving_total": int(r["driving"].sum()),
                     "parking_hour24_first8": [int(x) for x in r["parking"][:8, 23]]}
        print(name, out[name], flush=True)
    json.dump(out, open(os.path.join(HERE, "big_checksums.json"), "w"), indent=1)


def s8k_shards():
    """BASELINE.json configs[3]: Z = 8,192, 4,000 cars/zone (C = 32,768,000) dealt over 8 GPUs.  Checksums of what ranks 0 and 7
    compute (IVP + resample of their 4,096,000 cars), for the contiguous deal (shard_range) and for the interleaved one
    (car g -> rank g mod 8).  ~26 GB of host memory, minutes on 8 cores."""
    Z, cpz, world = 8192, 4000, 8
    C = Z * cpz
    p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
    p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED)
    cdf = O.build_cdf(p_dest)
    del p_dest
    out = {"Z": Z, "cpz": cpz, "C": C, "world": world, "table_seed": TABLE_SEED, "sim_seed": SIM_SEED, "shards": {}}
    n = C // world
    for deal in ("contiguous", "interleaved"):
        for rank in (0, world - 1):
            if deal == "contiguous":
                first, stride = rank * n, 1
            else:
                first, stride = rank, world
            cars = first + stride * np.arange(n, dtype=np.int64)
            r = O.fast_run(p_drive, cdf, n, SIM_SEED, cars // cpz + 1, car_offset=first, car_stride=stride)
            out["shards"][f"{deal}_rank{rank}"] = {
                "car_first": first, "car_stride": stride, "car_count": n,
                "initial_state_sha256": sha(r["zone0"]), "parking_sha256": sha(r["parking"].ravel(order="F")),
                "driving_sha256": sha(r["driving"].ravel(order="F")), "driving_total": int(r["driving"].sum()),
                "parking_hour24_first8": [int(x) for x in r["parking"][:8, 23]],
                "parking_max": int(r["parking"].max())}
            print(deal, rank, out["shards"][f"{deal}_rank{rank}"], flush=True)
    json.dump(out, open(os.path.join(HERE, "s8k_shard_checksums.json"), "w"), indent=1)


if __name__ == "__main__":
    if "--s8k" in sys.argv:
        s8k_shards()
        sys.exit(0)
    small()
    if "--big" in sys.argv:
        big()
