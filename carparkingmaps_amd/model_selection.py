"""Model selection on top of the sampler path (BASELINE.json configs[4]).

The reference tunes its four hyper-parameters with three sequential 1-D searches that live only
in the notebook (README.md:947-2435 of the reference): e_drive against the measured circadian
traffic activity (README.md:1134-1314), (p_min, p_max) against the share of life spent driving
A_set = 0.07 (README.md:1436-1676, clamp = src/correctparameters.jl:3-22) and e_dest against the
measured parking densities (README.md:2062-2276).  Every iteration = rebuild one table -> reset
state_matrix[:,1] = initial_state (the IVP is NOT re-run: README.md:1180,1549,2119) -> 24-hour
resample -> histogram -> scalar error.  Here one iteration is one `evaluate()` on the device:
cpm_build_p_drive / cpm_build_p_dest + cpm_resample from the cached post-IVP state.

`grid_sweep` is the build's generalisation named by BASELINE.json: a grid of
(e_drive, p_min, p_max, e_dest) points, dealt round-robin over the ranks of a torch.distributed
job (one process per GPU).  Points are independent, so there is NO collective on the data path;
the per-point scalars are gathered at the end.  Points sharing e_dest share the CDF (the
3 GB table is rebuilt only when e_dest changes), so each rank sorts its points by e_dest.

The reference's measured validation data (traffic_activity_measured.jld2, parking_density_measured.jld2)
are not in the repository; callers pass their own vectors (tests use synthetic ones).
"""
import zlib
from dataclasses import dataclass, field

import numpy as np

from .reference_api import correctparameters


# ------------------------------------------------------------------ objectives (host scalars)
def traffic_activity(driving):
    """min-max normalised column sums of the driving counts (src/saveresults.jl:23-28)."""
    a = np.asarray(driving, dtype=np.float64).sum(axis=0)
    with np.errstate(all="ignore"):
        return (a - a.min()) / (a.max() - a.min())


def traffic_activity_error(activity, measured):
    """mean squared error over the 24 hours (README.md:1279-1283)."""
    d = np.asarray(measured, dtype=np.float64) - np.asarray(activity, dtype=np.float64)
    return float(np.sum(d * d) / d.size)


def a_drive(sum_tt_q16, C, T=24):
    """share of life spent driving (src/averagedrivingtime.jl:10) from the fixed-point time sum."""
    return (sum_tt_q16 / 65536.0) / (C * T * 60 * 60)


def parking_density_error(parking, C, measured):
    """README.md:2219-2244: per-zone min-max normalisation over the day (zones with max == min are
    left as they are), MSE over 24 h for zones that are measured (row sum != 0) and not flat, averaged
    over the validated zones.  (Evaluated hour-major -- the count tensor's own layout, Z contiguous -- so that a point of
    the sweep costs the host tens of microseconds, not half a millisecond: with two contexts in flight the host is what
    a point waits for.)"""
    p = np.asarray(parking).T / float(C)                     # (T, Z); a view of a Julia-ordered (Z, T) array is contiguous this way
    m = np.asarray(measured, dtype=np.float64).T
    lo, hi = p.min(axis=0), p.max(axis=0)
    flat = hi == lo
    valid = (m.sum(axis=0) != 0) & ~flat
    if not valid.any():
        return float("nan")
    if not valid.all():
        p, m, lo, hi = p[:, valid], m[:, valid], lo[valid], hi[valid]
    pn = (p - lo) / (hi - lo)
    d = pn - m
    err = (d * d).sum(axis=0) / p.shape[0]
    return float(err.sum() / err.size)


# ------------------------------------------------------------------ one evaluation
@dataclass
class Point:
    e_drive: float = 0.5
    p_min: float = 0.1
    p_max: float = 0.9
    e_dest: object = 2


@dataclass
class Evaluator:
    """Holds a Sampler whose datamatrix is uploaded and whose car state is the post-IVP state."""
    sampler: object
    C: int
    seed: int
    measured_activity: object = None
    measured_parking: object = None
    travel: bool = True
    fallbacks: int = 0      # pipelined points whose asynchronous step overflowed a bucket region and were evaluated again, blocking
    _e_dest: object = field(default=None, repr=False)
    _pipe: object = field(default=None, repr=False)

    def __post_init__(self):
        if self.measured_parking is not None:   # Julia order, like the count tensors: the error is evaluated hour-major on contiguous rows
            self.measured_parking = np.asfortranarray(np.asarray(self.measured_parking, dtype=np.float64))

    def _install(self, pt):
        s = self.sampler
        s.build_p_drive(pt.p_min, pt.p_max, pt.e_drive, want=False)   # (Z x T work: the Z x Z x T mean is cached by the library)
        key = (type(pt.e_dest).__name__, float(pt.e_dest))
        if key != self._e_dest:              # the CDF and the row packs are rebuilt only when e_dest changes
            s.build_p_dest(pt.e_dest, want=False)
            self._e_dest = key

    def _objectives(self, pt, parking, driving, sum_tt_q16):
        act = traffic_activity(driving)
        out = {"e_drive": pt.e_drive, "p_min": pt.p_min, "p_max": pt.p_max, "e_dest": float(pt.e_dest),
               "A_drive": a_drive(sum_tt_q16, self.C, self.sampler.T), "traffic_activity": act,
               "parking": parking, "driving": driving}
        if self.measured_activity is not None:
            out["activity_error"] = traffic_activity_error(act, self.measured_activity)
        if self.measured_parking is not None:
            out["parking_error"] = parking_density_error(parking, self.C, self.measured_parking)
        return out

    def evaluate(self, pt):
        """One point, blocking."""
        self._install(pt)
        r = self.sampler.resample(self.seed, travel=self.travel)
        return self._objectives(pt, r["parking"], r["driving"], r["sum_tt_q16"])

    # -- pipelined form: point k+1 runs on the GPU while the host reduces point k (two count tensors, pinned host twins) --
    def begin(self, pt, slot):
        """Enqueue the table update, the resample and the copy of its counts to the host for `pt`; returns at once."""
        import torch
        s = self.sampler
        if self._pipe is None:
            if s._stream is None:     # (a Sampler that was given its stream at construction keeps it; see Sampler.__init__)
                s.set_stream(torch.cuda.Stream(device=s.device))
            stream = s._stream_obj if hasattr(s._stream_obj, "cuda_stream") else torch.cuda.ExternalStream(s._stream, device=s.device)
            n = s.counts_words()
            self._pipe = dict(stream=stream,
                              dev=[torch.zeros(n, dtype=torch.int64, device=f"cuda:{s.device}") for _ in range(2)],
                              host=[torch.zeros(n, dtype=torch.int64).pin_memory() for _ in range(2)],
                              done=[torch.cuda.Event() for _ in range(2)])
        p = self._pipe
        self._install(pt)
        with torch.cuda.stream(p["stream"]):
            s.resample_dev(self.seed, p["dev"][slot].data_ptr(), travel=self.travel)
            p["host"][slot].copy_(p["dev"][slot], non_blocking=True)
            p["done"][slot].record(p["stream"])

    def finish(self, pt, slot):
        """Wait for `begin(pt, slot)` and reduce its counts.  A step whose status word is set (a bucket region overflowed: the
        asynchronous form cannot repeat itself) is evaluated again through the blocking call, which grows the regions."""
        p = self._pipe
        p["done"][slot].synchronize()
        flat = p["host"][slot].numpy()
        Z, T = self.sampler.Z, self.sampler.T
        zt = Z * T
        if flat[2 * zt + 1] != 0:
            # (evaluate() installs pt's tables again -- the next point's are in place by now -- and the blocking resample grows the
            #  regions; the point already in flight behind this one was enqueued on the old regions and comes back here too)
            self.fallbacks += 1
            out = self.evaluate(pt)
            out["fallback"] = True
            return out
        both = flat[:2 * zt].copy()                           # (the pinned twin is reused two points later)
        parking = both[:zt].reshape((T, Z)).T                 # Julia order (Z, T): views, no second copy
        driving = both[zt:].reshape((T, Z)).T
        return self._objectives(pt, parking, driving, int(flat[2 * zt]))


# ------------------------------------------------------------------ the reference's three searches
def search_exponent(evaluate_error, initial_values, step_size=10.0, max_iter=3, epsilon=0.001, good=0.02):
    """README.md:1100-1314 (e_drive) and 2040-2276 (e_dest): pick the best of a few initial values,
    take one step, then secant-like updates e <- e - step * d(error)/d(e), resetting to the best
    value whenever the error got worse.  Returns (best_value, best_error, history)."""
    hist = [(v, evaluate_error(v)) for v in initial_values]
    best, best_err = min(hist, key=lambda x: x[1])
    e = best + step_size * best_err                                        # Step 3.1
    for _ in range(max_iter):
        err = evaluate_error(e)
        hist.append((e, err))
        error_gradient = err - best_err
        param_gradient = e - best
        if error_gradient < 0:
            best_err, best = err, e
        else:
            e = best
        if param_gradient == 0:
            break
        e = e - step_size * error_gradient / param_gradient                # Step 3.2.6
        if best_err < good or abs(error_gradient) < epsilon:
            break
    return best, best_err, hist


def tune_p_min_max(evaluate_a_drive, A_set=0.07, p_min=0.1, p_max=0.9, max_iter=5):
    """README.md:1436-1676: move p_max (or p_min at the bounds) until A_drive matches A_set to the
    percent; clamp with correctparameters; reset when |dA| grows."""
    pm, px = [p_min], [p_max]
    A = [evaluate_a_drive(p_min, p_max)]
    dA = [A_set - A[0]]
    for n in range(2, max_iter + 1):
        a, b, d, ad = pm[-1], px[-1], dA[-1], A[-1]
        step = (b + a) * d / (n * ad)
        if b == 1:
            if a == b:
                break
            na, nb = a + step, b
        elif a == 0:
            if a == b:
                break
            na, nb = a, b + step
        elif a == b:
            na, nb = a + step, b
        else:
            na, nb = a, b + step
        na, nb = correctparameters(na, nb, a, b)
        an = evaluate_a_drive(na, nb)
        dn = A_set - an
        if int(A_set * 100) == int(an * 100):
            pm.append(na); px.append(nb); A.append(an); dA.append(dn)
            break
        if n > 2 and abs(dn) > abs(d):                                      # worse: reset
            na, nb, an, dn = a, b, ad, A_set - ad
        pm.append(na); px.append(nb); A.append(an); dA.append(dn)
    return pm[-1], px[-1], A[-1], list(zip(pm, px, A))


# ------------------------------------------------------------------ the grid of config 5
def make_grid(e_drive=(0.25, 0.5, 1.0, 2.0), p_min=(0.0, 0.05, 0.1, 0.2), p_max=(0.5, 0.7, 0.9, 1.0),
              e_dest=(0.5, 1, 2, 4)):
    """4 x 4 x 4 x 4 = 256 points by default."""
    return [Point(a, b, c, d) for d in e_dest for a in e_drive for b in p_min for c in p_max]


def points_of_rank(n_points, rank, world_size, order=None):
    """Block deal: rank r takes the r-th of world_size contiguous slices of `order` (default 0 .. n_points-1).  Independent
    points, no data-path collective.  With the points ordered by e_dest a rank's slice spans as few e_dest values as
    possible, so it rebuilds the CDF and the row packs (Z x Z x T work) as rarely as possible."""
    order = list(range(n_points)) if order is None else list(order)
    base, rem = divmod(len(order), int(world_size))
    begin = rank * base + min(rank, rem)
    return order[begin:begin + base + (1 if rank < rem else 0)]


def grid_sweep(evaluator, grid, rank=0, world_size=1, gather=True, checksums=False):
    """Evaluate this rank's share of `grid`; returns a list (on every rank when gather) of dicts with the
    per-point scalars, ordered like `grid` (checksums: also CRC-32 of both count tensors, for the tests).

    `evaluator` may be a list of Evaluators on the same GPU (each with its own Sampler context, hence its own stream): the
    rank's slice is cut into as many contiguous pieces and the pieces advance in turn, so that a point of every context is on
    the GPU at any time.  A resample is a serial chain of an issue-bound sampler launch and a latency-bound placing launch
    per hour; two independent chains interleave on the chip (measured at S4k: 1.41 ms for two resamples side by side
    against 2 x 0.92 one after the other, profiles/round2_notes.md) -- the hours of ONE resample cannot."""
    lanes = list(evaluator) if isinstance(evaluator, (list, tuple)) else [evaluator]
    C = lanes[0].C
    by_e_dest = sorted(range(len(grid)), key=lambda i: (float(grid[i].e_dest), type(grid[i].e_dest).__name__, i))
    mine = points_of_rank(len(grid), rank, world_size, by_e_dest)
    local = {}

    def lane_results(ev, pts):  # point k+1 is enqueued before point k is reduced on the host
        prev = None
        for k, i in enumerate(pts):
            ev.begin(grid[i], k & 1)
            yield (prev[0], ev.finish(grid[prev[0]], prev[1])) if prev is not None else None   # (None: the lane's first point is on its way)
            prev = (i, k & 1)
        if prev is not None:
            yield prev[0], ev.finish(grid[prev[0]], prev[1])

    def results():  # the lanes in turn: each next() enqueues that lane's next point and reduces its previous one
        gens = [lane_results(ev, points_of_rank(len(mine), l, len(lanes), mine)) for l, ev in enumerate(lanes)]
        while gens:
            for g in list(gens):
                try:
                    r = next(g)
                    if r is not None:
                        yield r
                except StopIteration:
                    gens.remove(g)

    for i, r in results():
        local[i] = {k: v for k, v in r.items() if np.isscalar(v)}
        local[i]["fallback"] = bool(r.get("fallback", False))   # evaluated twice: its asynchronous step had overflowed
        local[i]["driving_total"] = int(r["driving"].sum())
        local[i]["hours_hold_all_cars"] = bool((r["parking"].sum(axis=0) == C).all())
        if checksums:
            local[i]["parking_crc32"] = zlib.crc32(np.ascontiguousarray(r["parking"].ravel(order="F")).tobytes())
            local[i]["driving_crc32"] = zlib.crc32(np.ascontiguousarray(r["driving"].ravel(order="F")).tobytes())
    if world_size > 1 and gather:
        import torch.distributed as dist
        parts = [None] * world_size
        dist.all_gather_object(parts, local)
        local = {}
        for p in parts:
            local.update(p)
    return [local.get(i) for i in range(len(grid))]
