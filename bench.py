#!/usr/bin/env python3
"""bench.py -- car-steps/s of the 24-hour resample (src/resampling.jl + the saveresults
histogram) on N MI355X, with the HBM roofline of the path and the CPU restatement timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full T=24 hour resample of every car (tables, CDF and the post-IVP initial
state already resident in HBM), through the zone x hour count tensor -- and, for N > 1, through
the single RCCL all-reduce of that tensor (step k's all-reduce runs under step k+1's kernels; the
timed region ends when the last one has finished).  Workload at N = 1: synthetic dense Z = 4,096 zones,
1,000 cars/zone (BASELINE.json configs[2], the configuration the metric is quoted on).  For
N > 1 `value` keeps the cars per GPU fixed (weak scaling): Z = 4,096, 1,000*N cars/zone, car g on rank g mod N;
the same JSON line then also carries `strong` (the metric's own C = 4,096,000 dealt over the N ranks) and, at N = 8,
`configs3` (BASELINE.json configs[3]: Z = 8,192 x 4,000 cars/zone).

roofline: B = algorithmic HBM bytes of one hourly launch = rows + Z*8 (p_drive) + C_g*8 (4 B id in, 4 B out)
+ 2*Z*8 (counts), with the element size of the rows the kernel really streams.  `achieved` / `frac` are for the dominant
kernel (the hourly sampler launch: B over its hipEvent duration); `whole_resample` is 8(d)'s own figure, T*B over the wall
time of a step; `kernels` lists every hourly kernel of the step with its own bytes, duration and measured traffic.

Records outside the headline (N = 1 only, each a few seconds): `table_build` (the one pass from p_destin to the samplers' row
tables), `per_dataset` (main.jl:79-95 at Melbourne's shape: createpdrive + createpdestin + IVP + resample with travel times),
`skewed` (the headline workload on peaky destination tables), `two_resamples_in_flight`, `full_pipeline`, `cpu_baseline`.
`--zones 2357 --cars-per-zone 100 --melbourne` is BASELINE.json configs[0]'s shape (the reference's own CPU-runnable case):
its cpu_baseline times the faithful restatement on the Melbourne-shaped sparse tables.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

TABLE_SEED = 0x5EED7AB1E
SIM_SEED = 0x5EEDCA125
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
PROFILE_TAG = "round4"  # profiles/<tag>_traffic.json: PMC run of this same command (tools/collect_profiles.sh)
T = 24


def library_identity():
    """what was measured: the loaded library's path and SHA-256, and whether the run can count as the product's
    (CPM_LIB_PATH loads a diagnostic / ablation twin, CPM_BENCH_NOCHECK skips the counts check: either makes the line invalid)"""
    import hashlib
    from carparkingmaps_amd import _lib
    h = hashlib.sha256()
    with open(_lib.LIB_PATH, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            h.update(chunk)
    reasons = [k for k in ("CPM_LIB_PATH", "CPM_BENCH_NOCHECK") if os.environ.get(k)]
    return {"path": os.path.relpath(_lib.LIB_PATH, ROOT), "sha256": h.hexdigest(),
            "env": {k: v for k, v in os.environ.items() if k.startswith("CPM_")}}, reasons


def pack_row_words(Z):
    """words of one row pack of the grouped path (cpm_grouped.h: pack_guide_bits / pack_zq / pack_row_words)"""
    g = 3
    while (1 << g) < Z:
        g += 1
    G = max(3, g - 2)
    return max(256, (1 << G) // 2 + 4 + (Z + 62) // 32 * 32)


def cpu_baseline(sampler, Z, n_cars, time_budget_s, melbourne):
    """The faithful three-pass restatement of src/resampling.jl (oracle, single thread) on a
    bounded sample: the first `n_cars` cars, 24 h, starting from their post-IVP zones; its counts
    are checked against the HIP path on the same cars.  melbourne: on the sparse Melbourne-shaped tables (the oracle's
    createpdrive / createpdestin of its synthetic datamatrix), travel-time pass included (src/resampling.jl:53-78)."""
    import numpy as np
    from oracle import oracle as O

    dm = dist = None
    if melbourne:
        dm, dist = O.synth_datamatrix(Z, T, TABLE_SEED)   # (bit-identical to the device's generator: tests/test_gpu_parity.py)
        p_drive = sampler.get_p_drive()                   # the tables the device built from it (createpdrive agrees with the oracle's to
        p_dest = sampler.build_p_dest(2)                  # 4e-16, not bit for bit: the CPU leg runs on the SAME tables as the GPU)
    else:
        p_drive = O.synth_p_drive(Z, T, TABLE_SEED)
        p_dest = O.synth_p_dest_dense(Z, T, TABLE_SEED)        # reference layout, Z*Z*T*8 bytes on the host
    zones = sampler.get_state()[:n_cars].copy()
    # calibrate the sample to the time budget on a small slice first
    probe = min(256, n_cars)
    st, tr = O.initializestates(probe, 1, T)
    st[:, 0] = zones[:probe]
    t0 = time.perf_counter()
    O.resampling(st, tr, probe, Z, p_drive, p_dest, dm, dist, SIM_SEED, car_offset=0)
    per_car = (time.perf_counter() - t0) / probe
    n = int(max(probe, min(n_cars, time_budget_s / max(per_car, 1e-9))))
    st, tr = O.initializestates(n, 1, T)
    st[:, 0] = zones[:n]
    t0 = time.perf_counter()
    O.resampling(st, tr, n, Z, p_drive, p_dest, dm, dist, SIM_SEED, car_offset=0)
    dt = time.perf_counter() - t0
    pk, dr, _ = O.histogram(Z, st, tr)
    # the same cars through the HIP path
    C_total, cpz, begin, count, stride = sampler.C_total, sampler.cars_per_zone, sampler.car_begin, sampler.car_count, sampler.car_stride
    full = sampler.get_state()
    sampler.init_states(C_total, cpz, 0, n)
    sampler.set_state(zones[:n])
    r = sampler.resample(SIM_SEED, travel=melbourne)
    ok = bool(np.array_equal(r["parking"], pk.astype(np.int64)) and np.array_equal(r["driving"], dr.astype(np.int64)))
    if melbourne:
        ok = ok and r["sum_tt_q16"] == int(O.sum_travel_time_q16(tr))
    sampler.init_states(C_total, cpz, begin, count, car_stride=stride)
    sampler.set_state(full)
    # all-core best effort beside it (oracle fast twin: prebuilt CDF + binary search, OpenMP over cars)
    cdf = O.build_cdf(p_dest)
    nfast = int(min(len(full), 1_000_000))
    t0 = time.perf_counter()
    O.fast_run(p_drive, cdf, nfast, SIM_SEED, full[:nfast], car_offset=0, do_ivp=False)
    dt_fast = time.perf_counter() - t0
    fast = {"value": nfast * T / dt_fast, "unit": "car-steps/s", "cores": O.max_threads(), "kind": "port",
            "sample": f"first {nfast} cars x {T} h, prebuilt CDF + binary search, OpenMP over cars, {dt_fast:.1f} s"}
    tables = "Melbourne-shaped sparse tables (8.68 % of the datamatrix populated), travel-time pass included" if melbourne else "dense synthetic tables"
    return {"value": n * T / dt, "unit": "car-steps/s", "cores": 1, "kind": "port", "all_cores": fast,
            "sample": f"first {n} cars x {T} h of the same workload, {tables} (faithful three-pass restatement of "
                      f"src/resampling.jl, strided p_dest row gather + sum + linear walk), {dt:.1f} s",
            "counts_match_gpu": ok}


def pmc_traffic(Z, cars_per_gpu, skew):
    """HBM bytes per launch of every hourly kernel from the committed PMC run of this same command
    (tools/collect_profiles.sh: FETCH_SIZE and WRITE_SIZE in their own rocprofv3 passes, corrected as
    MI355X_MICROARCH.md prescribes; tools/summarize_profiles.py).  {} when no run matches this workload."""
    if Z != 4096 or cars_per_gpu != 4096000 or skew:
        return {}, None
    path = os.path.join(ROOT, "profiles", PROFILE_TAG + "_traffic.json")
    if not os.path.exists(path):
        return {}, None
    out = {}
    # entries are "<kernel> @grid=<threads>": the side records launch the same kernels at other sizes (Z = 2,357) and on other
    # tables (--skew), so a kernel is taken at THIS workload's grid, and the two-launch kernels only from a run that had no fused hour
    grids = {"hour": None, "sampler": Z * 256, "place": None, "build_rows": ((Z + 63) // 64) * 24 * 256}
    grids["build_rows_full"] = grids["build_rows"]
    for k, v in json.load(open(path)).items():
        if v.get("launches", 0) <= 0:
            continue
        name, _, grid = k.partition(" @grid=")
        grid = int(grid) if grid else None
        targs = [x.strip() for x in name.split("<", 1)[1].split(">", 1)[0].split(",")] if "<" in name else []
        if "k_grouped_hour<" in name and targs[:2] in (["6", "5"], ["4", "5"]) and all(x == "false" for x in targs[2:]):   # the fused hour at Z = 4,096 (six cars per lane, NQ 5, dense packs, zone order)
            key = "hour"
        elif "k_grouped_sample<" in name and len(targs) >= 4 and targs[3] == "true" and targs[4:] in ([], ["false"]):   # (the grouped form on dense packs; the plain form only runs hour 24)
            key = "sampler"
        elif "k_grouped_place" in name:
            key = "place"
        elif "k_build_rows<false, true>" in name or "k_build_rows<0, 1>" in name:
            key = "build_rows"
        elif "k_build_rows<true, true>" in name or "k_build_rows<1, 1>" in name:
            key = "build_rows_full"
        else:
            continue
        if grids[key] is not None and grid is not None and grid != grids[key]:
            continue
        if key not in out or v["launches"] > out[key][1]:
            out[key] = (v["hbm_bytes_per_launch"], v["launches"])
    out = {k: v[0] for k, v in out.items()}
    if "hour" in out:                                        # (then the two-launch kernels in the file ran on the skewed tables)
        out.pop("sampler", None)
        out.pop("place", None)
    return out, "profiles/" + PROFILE_TAG + "_traffic.json"


class Job:
    """One workload on this rank's GPU: tables resident, cars dealt, IVP done; `run_steps` enqueues pipelined resamples."""

    def __init__(self, env, Z, cpz_total, kernel=0, skew=0, melbourne=False, travel=False):
        import carparkingmaps_amd as cpm  # noqa: F401
        from carparkingmaps_amd.distributed import ShardedSampler
        self.env, self.Z, self.cpz, self.C, self.travel = env, Z, cpz_total, Z * cpz_total, travel
        self.ss = ShardedSampler(Z, T, rank=env.rank, world_size=env.world, device=env.local_rank, deal=env.deal)
        self.s = self.ss.s
        self.s.set_kernel(kernel)
        self.kernel = kernel
        if melbourne:
            self.s.synth_datamatrix(TABLE_SEED)
            self.s.build_p_drive(0.1, 0.9, 0.5, want=False)
            self.s.build_p_dest(2, want=False)
        else:
            self.s.synth_tables(TABLE_SEED, skew_q=skew)
        self.first, self.count = self.ss.init_states(self.C, cpz_total)
        self.s.solve_ivp_async(SIM_SEED)          # 23 untimed burn-in steps (main.jl:91-92)
        self.s.sync()
        if env.rehearse and env.world > 1:  # gloo cannot reduce device tensors: bounce through the host
            import torch
            import torch.distributed as dist
            ss, s = self.ss, self.s

            def resample_allreduce_async(seed, travel=False):
                with torch.cuda.stream(ss.stream):
                    s.resample_dev(seed, ss.counts.data_ptr(), travel=travel)
                    host = ss.counts.cpu()
                    dist.all_reduce(host)
                    ss.counts.copy_(host)
                return ss.counts, 0
            ss.resample_allreduce_async = resample_allreduce_async
            ss.wait = lambda ticket: None

    def run_steps(self, k):
        """k pipelined steps: the all-reduce of a step overlaps the kernels of the next one; returns the last count tensor"""
        buf = None
        for _ in range(k):
            buf, _ = self.ss.resample_allreduce_async(SIM_SEED, travel=self.travel)
        self.ss.synchronize()
        return buf

    def settle(self, warmup):
        """Warm-up; the grouped kernels flag a bucket / run that outgrew its region in the status word (summed over ranks by the
        all-reduce): the step is then invalid, the context doubles its regions when it next enqueues.  Returns the kernel in use."""
        counts = self.run_steps(warmup)
        self.env.barrier()
        kernel_used = self.kernel
        for _ in range(6):
            if counts is None or int(counts[-1].item()) == 0:
                break
            counts = self.run_steps(max(warmup, 2))
            self.env.barrier()
        if counts is not None and int(counts[-1].item()) != 0:
            kernel_used = 2
            self.s.set_kernel(kernel_used)
            counts = self.run_steps(max(warmup, 1))
            self.env.barrier()
        if kernel_used == 0:
            kernel_used = {5: 0}.get(self.s.get_info(1), self.s.get_info(1))  # what AUTO resolved to (0 stands for its default, the grouped path)
        self.kernel_used = kernel_used
        return kernel_used

    def timed(self, steps, min_window_s=0.0):
        """wall time of a block of EXACTLY `steps` steps bracketed by barrier + synchronize on both sides, MAX over ranks; the block
        is repeated until the blocks add up to min_window_s (every rank takes the same number of blocks: the decision is rank 0's,
        broadcast): returns (median block, every block, the last count tensor)"""
        import torch
        import torch.distributed as dist
        env = self.env
        blocks = []
        counts = None
        while True:
            env.barrier()
            t0 = time.perf_counter()
            counts = self.run_steps(steps)
            env.barrier()
            dt = time.perf_counter() - t0
            tmax = torch.tensor([dt], dtype=torch.float64, device="cpu" if env.rehearse else "cuda")
            if env.world > 1:
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            blocks.append(float(tmax.item()))
            go = torch.tensor([1 if (sum(blocks) < min_window_s and len(blocks) < 200) else 0], device="cpu" if env.rehearse else "cuda")
            if env.world > 1:
                dist.broadcast(go, 0)
            if int(go.item()) == 0:
                break
        return statistics.median(blocks), blocks, counts

    def check(self, counts):
        import carparkingmaps_amd as cpm
        parking, driving, _ = cpm.distributed.split_counts(counts, self.Z, T)
        if os.environ.get("CPM_BENCH_NOCHECK") != "1":   # (timing-only ablation builds of tools/build_variants.sh produce invalid counts)
            assert (parking.sum(axis=0) == self.C).all(), "every hour must hold all C cars"
        return parking, driving

    def close(self):
        self.ss.close()


class Env:
    pass


def side_record(env, Z, cpz_total, steps, what, **kw):
    """a secondary workload measured like the headline (settle, then `steps` timed steps): ms per resample and car-steps/s"""
    job = Job(env, Z, cpz_total, **kw)
    job.settle(3)
    dt, _, counts = job.timed(steps)
    parking, _ = job.check(counts)
    ms = dt / steps * 1e3
    rec = {"what": what, "zones": Z, "cars": job.C, "cars_per_gpu": job.count, "steps": steps, "ms_per_step": ms,
           "value": job.C * T * steps / dt, "unit": "car-steps/s", "bucket_region_x_mean": job.s.get_info(2),
           "largest_bucket_x_mean": float(parking.max()) / (job.C / Z),
           "kernel": {0: "auto (zone_grouped)", 1: "car", 2: "zone_lds", 5: "zone_grouped"}[job.kernel_used]}
    return rec, job


def table_build_record(s, Z, traffic):
    """The one pass from the resident p_destin (reference layout, f64) to what the samplers read (k_build_rows): row totals,
    running-sum checkpoints, row packs -- and, in the second figure, the canonical f64 CDF rows the grouped path never reads."""
    out = {}
    rw = pack_row_words(Z)
    nck = (Z + 31) // 32
    for name, full in (("packs", False), ("packs_and_f64_cdf", True)):
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            s.refresh_tables(with_f64_cdf=full)          # blocking (the validation flag is read back)
            ts.append(time.perf_counter() - t0)
        ms = statistics.median(ts) * 1e3
        nbytes = Z * Z * T * 8 + T * Z * rw * 4 + T * Z * 8 + T * nck * Z * 8 + (T * Z * ((Z + 15) // 16 * 16) * 8 if full else 0)
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"ms": ms, "algorithmic_bytes": nbytes, "achieved": gbs, "frac": gbs / HBM_PEAK_GBS,
                     "traffic": traffic.get("build_rows_full" if full else "build_rows")}
    s.refresh_tables(with_f64_cdf=False)
    out["what"] = ("p_destin (Z x Z x T f64, reference layout) -> row packs (guide + CDF high words) + row totals + checkpoints in ONE "
                   "launch (k_build_rows; sequential f64 running sum of src/resampling.jl:39); wall time of the blocking call, median of 5; "
                   "round 2 took three launches for this (CDF, high words, guide): 3.77 ms at Z = 4,096")
    return out


def per_dataset_record(env):
    """main.jl:79-95 for one dataset at Melbourne's shape (Z = 2,357, 1,000 cars/zone): the tables from the (synthetic, device-
    generated) datamatrix, the 23-hour IVP, the 24-hour resample with travel times.  Every stage timed as a blocking call."""
    import carparkingmaps_amd as cpm
    Z, cpz = 2357, 1000
    C = Z * cpz
    s = cpm.Sampler(Z, T, env.local_rank)
    s.synth_datamatrix(TABLE_SEED)

    def once():
        s.synth_datamatrix(TABLE_SEED)       # a NEW dataset every time: nothing derived from the last one is reused (createpdrive's
        s.sync()                             # cached Z x Z x T pass, the sparse travel rows); its generation is not timed
        t0 = time.perf_counter()
        s.build_p_drive(0.1, 0.9, 0.5, want=False)
        s.sync()
        t1 = time.perf_counter()
        s.build_p_dest(2, want=False)
        t2 = time.perf_counter()
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        t3 = time.perf_counter()
        r = s.resample(SIM_SEED, travel=True)    # (builds the dataset's sparse travel rows first)
        t4 = time.perf_counter()
        s.resample(SIM_SEED, travel=True)
        t5 = time.perf_counter()
        return (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4), r
    once()
    runs = [once() for _ in range(3)]
    med = [statistics.median(r[0][k] for r in runs) * 1e3 for k in range(5)]
    r = runs[-1][1]
    assert (r["parking"].sum(axis=0) == C).all()
    s.close()
    total = sum(med[:4])
    cells = Z * Z * T * 8
    return {"what": "one NEW dataset of main.jl:79-95 at Melbourne's shape (Z = 2,357 x 1,000 cars/zone, synthetic sparse datamatrix "
                    "generated in HBM before the clock starts): createpdrive (Z x Z x T pass included), createpdestin (weights, row sums, "
                    "row tables), initializestates + 23-hour IVP, 24-hour resample with travel times (the dataset's travel rows are "
                    "built inside the first one) and the count tensor on the host; blocking calls, median of 3",
            "createpdrive_ms": med[0], "createpdestin_ms": med[1], "ivp_ms": med[2], "resample_ms": med[3],
            "resample_again_ms": med[4], "total_ms": total,
            "tables_algorithmic_bytes": int(cells + Z * Z * 8 + 3 * cells + cells + T * Z * pack_row_words(Z) * 4),
            "value": C * (2 * T - 1) / (total * 1e-3), "unit": "car-steps/s"}


def per_rank_emulated_record(env):
    """What ONE rank does at N = 1, 2, 4, 8 -- on the one GPU of this run: an EMULATION of the per-rank work, not a scaling curve.
    Rank 0's share of an N-way interleaved deal (global cars 0, N, 2N, ...; the whole table on every rank, as DESIGN.md 5 has it):
    ms per resample for the metric's own fleet (Z = 4,096, C = 4,096,000) and for BASELINE.json configs[3] (Z = 8,192,
    C = 32,768,000).  t(1) / t(N) is the speed-up car sharding cannot exceed (the all-reduce and any skew between ranks only
    subtract from it)."""
    import torch
    import carparkingmaps_amd as cpm
    out = {"what": "EMULATED on one GPU, not a scaling measurement: rank 0's share of an N-way interleaved deal of the cars, every rank "
                   "streaming the whole table (car sharding, DESIGN.md 5); bound = t(1) / t(N) is what N GPUs cannot exceed"}
    for name, Z, cpz, steps in (("metric_fleet_Z4096_C4096000", 4096, 1000, 30), ("configs3_Z8192_C32768000", 8192, 4000, 6)):
        C = Z * cpz
        st = torch.cuda.Stream(device=env.local_rank)
        s = cpm.Sampler(Z, T, env.local_rank, stream=st)
        s.synth_tables(TABLE_SEED)
        buf = torch.zeros(s.counts_words(), dtype=torch.int64, device=f"cuda:{env.local_rank}")
        rec = {}
        for N in (1, 2, 4, 8):
            s.init_states(C, cpz, 0, C // N, car_stride=N)
            if s.get_info(1) != cpm.CPM_KERNEL_ZONE_GROUPED:   # (Z = 8,192 packs a driver as 24 bits of car id + 8 of destination: 16.7 M cars per GPU)
                rec[str(N)] = {"cars_on_rank": C // N, "ms_per_step": None,
                               "why": "this many cars do not fit the grouped path's packed driver ids on ONE GPU: the configuration needs >= 2"}
                continue
            s.solve_ivp(SIM_SEED, want=False)
            for _ in range(4):                      # (lets the context settle its bucket regions)
                s.resample_dev(SIM_SEED, buf.data_ptr())
            torch.cuda.synchronize()
            assert int(buf[-1].item()) == 0
            t0 = time.perf_counter()
            for _ in range(steps):
                s.resample_dev(SIM_SEED, buf.data_ptr())
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            assert int(buf[:T * Z].sum().item()) == T * (C // N)
            rec[str(N)] = {"cars_on_rank": C // N, "ms_per_step": ms, "hour_form": s.get_info(4)}
        base = "1" if rec["1"]["ms_per_step"] else "2"
        for N in (2, 4, 8):
            if rec[str(N)]["ms_per_step"] and str(N) != base:
                rec[str(N)]["bound_speedup_vs_%s_rank%s" % (base, "" if base == "1" else "s")] = rec[base]["ms_per_step"] / rec[str(N)]["ms_per_step"]
        out[name] = rec
        s.close()
        del buf
        torch.cuda.empty_cache()
    return out


def melbourne_record(env):
    """The Melbourne-shaped configurations (BASELINE.json configs[0], [1]: Z = 2,357, sparse tables built on the device from the
    synthetic datamatrix, 8.68 % of its cells populated): ms per 24-hour resample with and without travel times."""
    import torch
    import carparkingmaps_amd as cpm
    Z = 2357
    st = torch.cuda.Stream(device=env.local_rank)
    s = cpm.Sampler(Z, T, env.local_rank, stream=st)
    s.synth_datamatrix(TABLE_SEED)
    s.build_p_drive(0.1, 0.9, 0.5, want=False)
    s.build_p_dest(2, want=False)
    buf = torch.zeros(s.counts_words(), dtype=torch.int64, device=f"cuda:{env.local_rank}")
    out = {"what": "Z = 2,357 Melbourne-shaped sparse tables (synthetic datamatrix -> createpdrive / createpdestin on the device): ms per 24-hour "
                   "resample from the post-IVP state, pipelined steps, tables resident",
           "sparse_pack_words": s.get_info(6)}
    for cpz, steps in ((1000, 100), (100, 200)):
        C = Z * cpz
        s.init_states(C, cpz)
        s.solve_ivp(SIM_SEED, want=False)
        rec = {"cars": C}
        for travel in (False, True):
            for _ in range(4):
                s.resample_dev(SIM_SEED, buf.data_ptr(), travel=travel)
            torch.cuda.synchronize()
            assert int(buf[-1].item()) == 0
            t0 = time.perf_counter()
            for _ in range(steps):
                s.resample_dev(SIM_SEED, buf.data_ptr(), travel=travel)
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / steps * 1e3
            rec["ms_per_step_travel" if travel else "ms_per_step"] = ms
        rec["value"] = C * T / (rec["ms_per_step"] * 1e-3)
        rec["unit"] = "car-steps/s"
        rec["hour_form"] = s.get_info(4)
        out[f"cars_per_zone_{cpz}"] = rec
    s.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--zones", type=int, default=4096)
    ap.add_argument("--cars-per-zone", type=int, default=1000, help="per GPU (weak scaling)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 car, 2 zone_lds, 5 zone_grouped")
    ap.add_argument("--deal", default="interleaved", choices=["interleaved", "contiguous"], help="how cars are dealt over the ranks")
    ap.add_argument("--skew", type=int, default=0, help="Q > 0: skewed destination popularity 1 / (Q + rank) (secondary figure; 32 ~ the "
                                                        "most popular zone 26x the mean); 0: the flat headline tables")
    ap.add_argument("--melbourne", action="store_true", help="Melbourne-shaped sparse tables (synthetic datamatrix -> createpdrive / "
                                                             "createpdestin on the device) and travel times instead of the dense synthetic tables")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pair", action="store_true", help="skip the two-resamples-in-flight figure (profiling runs: its launches overlap, "
                                                           "which would blur the per-kernel statistics)")
    ap.add_argument("--no-side", action="store_true", help="skip the secondary records (table_build, per_dataset, skewed, strong, configs3)")
    ap.add_argument("--skip-side", default="", help="comma-separated side records to leave out (profiling passes: melbourne, per_rank_emulated, per_dataset, skewed, table_build)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import carparkingmaps_amd as cpm

    env = Env()
    env.rank = rank = int(os.environ.get("RANK", "0"))
    env.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    env.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    env.deal = args.deal
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # CPM_BENCH_REHEARSE=1: rehearsal of the N > 1 code path on a box with fewer GPUs than ranks (every
    # rank on the visible devices round-robin, gloo instead of RCCL).  Never used for reported numbers.
    env.rehearse = rehearse = os.environ.get("CPM_BENCH_REHEARSE") == "1"
    if rehearse:
        env.local_rank = env.local_rank % max(torch.cuda.device_count(), 1)
    local_rank = env.local_rank
    torch.cuda.set_device(local_rank)
    backend = None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        backend = dist.get_backend()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    env.barrier = barrier

    Z = args.zones
    cpz = args.cars_per_zone * world
    job = Job(env, Z, cpz, kernel=args.kernel, skew=args.skew, melbourne=args.melbourne, travel=args.melbourne)
    ss, s, C, count = job.ss, job.s, job.C, job.count
    kernel_used = job.settle(args.warmup)
    # Every 49th hourly sampler launch of the timed region carries a hipEvent pair (49 is coprime to 24, so every hour of the
    # day is sampled): the library hands the pair to the launch itself (hipExtLaunchKernelGGL: begin and end of the dispatch),
    # which sits ~1 us above rocprofv3's kernel duration; hipEventRecord on either side of a launch sat ~3 us above it.
    s.set_profile(True, stride=49, kernel=0)
    dt, blocks, counts = job.timed(args.steps, min_window_s=0.2)
    sampler_ms_all = s.last_kernel_ms()
    s.set_profile(False)
    parking, driving = job.check(counts)
    form = s.get_info(4) if kernel_used in (0, 5) else 0     # 0 two launches per hour, 1 one, 3 placing first, 6 all hours in one launch
    fused = form in (1, 3)
    # the timed launches in launch order: the k-th is sampler launch 49 k of the timed region, i.e. hour (49 k) mod 24 of its step; the
    # dominant kernel is the hourly launch of hours 1 .. 23 (hour 24 is sampled, never applied: its plain form is another kernel)
    launches_per_step = T if form != 6 else 2
    sampler_ms = [m for k, m in enumerate(sampler_ms_all) if form == 6 or (49 * k) % launches_per_step != T - 1] or sampler_ms_all

    # the other hourly kernel of the step, timed the same way outside the headline's timed region (none when the hour is ONE launch)
    place_ms = []
    if kernel_used in (0, 5) and form == 0:
        s.set_profile(True, stride=7, kernel=1)
        job.run_steps(max(8, min(args.steps, 40)))
        place_ms = s.last_kernel_ms()
        s.set_profile(False)

    # secondary figure: the whole main.jl:88-95 sequence, initializestates -> 23-hour
    # IVP -> 24-hour resample, 47 car-steps per car; outside the headline's timed region
    barrier()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        ss.init_states(C, cpz)
        with torch.cuda.stream(ss.stream):
            s.solve_ivp_async(SIM_SEED)
        ss.resample_allreduce(SIM_SEED, travel=job.travel)
    ss.synchronize()
    barrier()
    dt_full = (time.perf_counter() - t0) / reps

    # secondary figure: two independent resamples of the same fleet side by side (a second context on its own stream; what a
    # parameter sweep or a run over several seeds does).  The hours of ONE resample are a serial chain of an issue-bound sampler
    # launch and a latency-bound placing launch; two chains interleave on the chip.  Outside the headline's timed region.
    pair_ms = None
    if world == 1 and kernel_used in (0, 5) and not args.no_pair and not args.melbourne:
        st2 = torch.cuda.Stream(device=local_rank)
        s2 = cpm.Sampler(Z, T, local_rank, stream=st2)
        s2.set_kernel(args.kernel)
        s2.synth_tables(TABLE_SEED, skew_q=args.skew)
        s2.init_states(C, cpz)
        s2.solve_ivp(SIM_SEED, want=False)
        buf2 = torch.zeros(s2.counts_words(), dtype=torch.int64, device=f"cuda:{local_rank}")
        for _ in range(3):
            s2.resample_dev(SIM_SEED, buf2.data_ptr())
        torch.cuda.synchronize()
        pairs = max(4, min(args.steps, 100))
        t0 = time.perf_counter()
        for _ in range(pairs):
            ss.resample_allreduce_async(SIM_SEED)
            s2.resample_dev(SIM_SEED, buf2.data_ptr())
        ss.synchronize()
        torch.cuda.synchronize()
        pair_ms = (time.perf_counter() - t0) / pairs * 1e3
        same = bool((buf2[:2 * T * Z].cpu() == counts[:2 * T * Z].cpu()).all()) and int(buf2[-1].item()) == 0
        assert same, "the second context must reproduce the first one's counts"
        s2.close()

    side = {}
    if not args.no_side and not args.skew and not args.melbourne and Z == 4096 and args.cars_per_zone == 1000:
        if world == 1:
            skip = set(x for x in args.skip_side.split(",") if x)
            traffic0, _ = pmc_traffic(Z, count, args.skew)
            if "table_build" not in skip:
                side["table_build"] = table_build_record(s, Z, traffic0)
            if "per_dataset" not in skip:
                side["per_dataset"] = per_dataset_record(env)
            if "melbourne" not in skip:
                side["melbourne"] = melbourne_record(env)
            if "per_rank_emulated" not in skip:
                side["per_rank_emulated"] = per_rank_emulated_record(env)
            if "skewed" not in skip:
                rec, j2 = side_record(env, Z, cpz, 20, "the headline workload on skewed destination tables (popularity 1 / (32 + rank): the shape of real "
                                      "Uber Movement rows, README.md output_24_0.svg); bucket regions grown by the context as needed", skew=32)
                j2.close()
                side["skewed"] = rec
        else:
            # what BASELINE.json asks of an N-GPU run beside the weak line: the metric's own fleet dealt over the ranks ...
            rec, j2 = side_record(env, Z, 1000, 40, f"strong scaling: the metric's own configuration (Z = 4,096 x 1,000 cars/zone, C = 4,096,000) "
                                  f"dealt over {world} ranks; every rank streams the whole table each hour, so this form is table-bound by design")
            j2.close()
            side["strong"] = rec
            if world == 8:  # ... and configs[3]
                rec, j2 = side_record(env, 8192, 4000, 10, "BASELINE.json configs[3]: Z = 8,192 x 4,000 cars/zone (C = 32,768,000) sharded by car over 8 ranks, "
                                      "one all-reduce of the zone x hour counts per step")
                j2.close()
                side["configs3"] = rec

    if rank == 0:
        car_steps = C * T
        ms_per_step = dt / args.steps * 1e3
        lib, invalid_reasons = library_identity()
        alg_bytes = s.algorithmic_bytes_per_hour()
        avg_ms = sum(sampler_ms) / max(len(sampler_ms), 1)
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(Z, count, args.skew)
        f64_bytes = Z * Z * 8 + Z * 8 + count * 8 + 2 * Z * 8  # SURVEY.md 8(d) with the reference's 8-byte rows
        drivers_per_hour = float(driving.sum()) / T / world       # this GPU's share
        place_bytes = int(drivers_per_hour * 8 + Z * 32 * 4 + Z * 4)  # packed driver in (4 B) + id out (4 B), run lengths, bucket sizes
        place_avg = sum(place_ms) / max(len(place_ms), 1)
        whole = T * alg_bytes / (ms_per_step * 1e-3) / 1e9

        def kernel_entry(name, what, nbytes, ms, n, tr):
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"kernel": name, "what": what, "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": ms, "launches_timed": n,
                    "achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "traffic": tr,
                    "traffic_over_algorithmic": (tr / nbytes) if tr and nbytes else None}
        if fused:
            kernels = [kernel_entry("k_grouped_hour", "the whole hour in one launch: Bernoulli + categorical draw of every car, stayers compacted, drivers into "
                                    "runs, zone x hour counts (sampler workgroups), then the drivers from the runs into next hour's buckets (placing blocks; "
                                    f"their {place_bytes} B of runs and ids are not part of 8(d)'s compulsory bytes and not counted here)",
                                    alg_bytes, avg_ms, len(sampler_ms), traffic.get("hour"))]
        else:
            kernels = [kernel_entry("k_grouped_sample" if kernel_used in (0, 5) else {1: "k_step_car", 2: "k_exact_sample"}[kernel_used],
                                    "Bernoulli + categorical draw of every car, stayers compacted, drivers into runs, zone x hour counts",
                                    alg_bytes, avg_ms, len(sampler_ms), traffic.get("sampler"))]
        if place_ms:
            kernels.append(kernel_entry("k_grouped_place", "drivers from the runs into next hour's buckets (not part of 8(d)'s compulsory bytes: "
                                        "its ids are counted with the sampler's C_g*8)", place_bytes, place_avg, len(place_ms), traffic.get("place")))
        tables = (f"Melbourne-shaped sparse tables (synthetic datamatrix, createpdrive / createpdestin on the device), travel times on" if args.melbourne
                  else f"synthetic {'skewed (destination popularity 1/(%d+rank))' % args.skew if args.skew else 'dense'} p_dest")
        out = {
            "metric": "car-steps/sec at Z=4,096, 1k cars/zone; 1/2/4/8 MI355X + %HBM roofline",
            "value": car_steps * args.steps / dt,
            "unit": "car-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "timed_region_s": sum(blocks), "timed_blocks": len(blocks),   # blocks of EXACTLY `steps` steps, repeated until >= 0.2 s; value = the median block
            "timed_block_ms_min_max": [min(blocks) / args.steps * 1e3, max(blocks) / args.steps * 1e3],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",  # the contract's arithmetic (f64 CDF compare, 53-bit draws); the default kernel evaluates it on 32-bit high words
                             # and 64-bit integers with an exact f64 fallback, bit-identical by construction
            "data": "synthetic" + (" (REHEARSAL: gloo, shared GPU -- not a measurement)" if rehearse else ""),
            "ranks_seen": dist.get_world_size() if world > 1 else 1,
            "backend": backend,
            "config": {"workload": f"{tables}, Z={Z} zones, "
                                   f"{cpz} cars/zone (C={C}), T={T} h resample from the post-IVP state; {args.cars_per_zone} cars/zone per GPU",
                       "zones": Z, "cars": C, "cars_per_gpu": count, "hours": T,
                       "kernel": {0: "auto (zone_grouped)", 1: "car", 2: "zone_lds", 5: "zone_grouped"}[kernel_used]
                       + ("" if kernel_used == args.kernel else " (not the requested one: AUTO's choice or overflow fallback)"),
                       "launches_per_hour": ({0: 2, 1: 1, 3: 1, 6: "all hours in one launch"}[form]) if kernel_used in (0, 5) else None,
                       "hour_form": form, "library": lib,
                       "bucket_region_x_mean": s.get_info(2), "largest_bucket_x_mean": float(parking.max()) / (C / Z),
                       "parallelism": f"cars dealt {args.deal} over {world} rank(s), one RCCL all-reduce of int64[{2 * T * Z + 2}] per step, "
                                      f"double-buffered (overlaps the next step's kernels)",
                       "table_seed": hex(TABLE_SEED), "sim_seed": hex(SIM_SEED),
                       "device": cpm.device_info(local_rank)["name"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get("hour" if fused else "sampler"), "traffic_source": traffic_src,
                         "kernel": ("the hourly launch k_grouped_hour (dominant kernel: sampler workgroups AND the placing blocks of their drivers, one launch per "
                                    "hour): 8(d)'s algorithmic bytes of one hour / its hipEvent duration" if fused else
                                    "hourly sampler launch (dominant kernel): algorithmic bytes of one launch / its hipEvent duration"),
                         "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": avg_ms, "launches_timed": len(sampler_ms),
                         "whole_resample": {"what": "SURVEY.md 8(d): T x algorithmic bytes per launch / wall time of a step (every kernel of the step "
                                                    "in the denominator)", "achieved": whole, "frac": whole / HBM_PEAK_GBS,
                                            "f64_equivalent_frac": T * f64_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         "kernels": kernels,
                         "note": "algorithmic bytes use the element size the kernel streams (SURVEY.md 8d: substitute the variant's true "
                                 "size): on the default grouped path a row is a pack of 4-byte CDF high words plus a 2-byte guide entry per "
                                 "four destinations, not 8-byte f64; f64_equivalent prices the same launch at the reference's 8-byte rows",
                         "f64_equivalent": {"algorithmic_bytes_per_launch": f64_bytes,
                                            "achieved": f64_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
                                            "frac": (f64_bytes / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if avg_ms > 0 else 0.0}},
            "full_pipeline": {"value": C * (2 * T - 1) / dt_full, "unit": "car-steps/s", "ms": dt_full * 1e3,
                              "what": "initializestates + 23-hour IVP + 24-hour resample (main.jl:88-95), 47 car-steps per car"},
        }
        out.update(side)
        if pair_ms is not None:
            out["two_resamples_in_flight"] = {
                "what": "two contexts, two streams, the same fleet and tables: wall time per PAIR of resamples; not the headline (value "
                        "is one resample at a time)", "ms_per_pair": pair_ms, "ms_per_resample": pair_ms / 2,
                "value": 2 * car_steps / (pair_ms * 1e-3), "unit": "car-steps/s",
                "whole_resample_frac": 2 * T * alg_bytes / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        if world == 1 and not args.no_cpu_baseline and not args.skew:
            out["cpu_baseline"] = cpu_baseline(s, Z, min(count, 65536), args.cpu_seconds, args.melbourne)
        if invalid_reasons:  # not the product library, or its counts unchecked: never a measurement of the product
            out = {"invalid": True, "reason": "set in the environment: " + ", ".join(invalid_reasons), "not_a_measurement": {k: v for k, v in out.items() if k != "value"}}
        print(json.dumps(out), flush=True)
    job.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
