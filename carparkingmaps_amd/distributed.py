"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Cars are independent chains (src/resampling.jl:11-83 reads only car i's row), so the C cars are dealt over the WORLD_SIZE
ranks; every rank holds the full tables and samples its cars with Philox keyed by the GLOBAL car id, which makes the summed
histogram identical for every world size and for either deal:

  "interleaved" (default)  car g belongs to rank g mod N.  Every rank starts with its share of EVERY zone
                           (`initializestates` places the cars zone by zone), so its buckets start at their mean size.
  "contiguous"             rank r owns the range shard_range(C, r, N): all cars of 1/N of the zones, i.e. N x the rank's mean
                           bucket in the first IVP hour (the context then has to grow its bucket regions once).

The only exchange on the path is ONE all-reduce(sum) of the integer tensor [parking | driving | travel-time q16 | status]
(2*T*Z+2 int64 words) before normalisation (src/saveresults.jl:20) -- integer addition, so the result is order-free and
bit-exact.  The count tensor is double-buffered: step k's all-reduce runs (on RCCL's stream) under step k+1's kernels.
"""
import torch
import torch.distributed as dist


def shard_range(C_total, rank, world_size):
    """Contiguous car range [begin, begin+count) of `rank`; ranges tile [0, C_total) exactly."""
    base, rem = divmod(int(C_total), int(world_size))
    begin = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return begin, count


def shard_cars(C_total, rank, world_size, deal="interleaved"):
    """(first, stride, count): this rank simulates the global cars first + k * stride, k < count."""
    if deal == "contiguous":
        begin, count = shard_range(C_total, rank, world_size)
        return begin, 1, count
    if deal != "interleaved":
        raise ValueError(f"unknown deal {deal!r}")
    C_total, rank, world_size = int(C_total), int(rank), int(world_size)
    count = (C_total - rank + world_size - 1) // world_size if C_total > rank else 0
    return rank, world_size, count


def allreduce_counts(counts, async_op=False):
    """In-place sum over ranks of the int64 count tensor (device tensor -> RCCL over xGMI; CPU tensor -> gloo, used by the CPU
    tests of the sharding logic).  Runs whenever a process group exists, also a group of one rank (the collective is then RCCL's
    local copy: the same code path and stream ordering as on several GPUs)."""
    if dist.is_available() and dist.is_initialized():
        return dist.all_reduce(counts, op=dist.ReduceOp.SUM, async_op=async_op)
    return None


def split_counts(counts, Z, T):
    """[2*T*Z+2] int64 -> (parking (Z,T), driving (Z,T) as Fortran-ordered views, sum_tt_q16).
    Raises if the status word is set (a bucket region overflowed on some rank: the step must be repeated)."""
    zt = Z * T
    flat = counts.detach().cpu().numpy()
    if flat.size > 2 * zt + 1 and flat[2 * zt + 1] != 0:
        raise RuntimeError("resample status != 0: a bucket or run outgrew its region on some rank; repeat the step "
                           "(the context has grown its regions) or select CPM_KERNEL_ZONE_LDS")
    parking = flat[:zt].reshape((Z, T), order="F")
    driving = flat[zt:2 * zt].reshape((Z, T), order="F")
    return parking, driving, int(flat[2 * zt])


class ShardedSampler:
    """A Sampler bound to this rank's GPU and share of the cars, with the count all-reduce."""

    def __init__(self, number_zones, T=24, rank=None, world_size=None, device=None, deal="interleaved"):
        from .sampler import Sampler  # needs the HIP library
        self.rank = dist.get_rank() if rank is None else rank
        self.world_size = dist.get_world_size() if world_size is None else world_size
        self.device = torch.cuda.current_device() if device is None else device
        self.deal = deal
        # One explicit torch stream carries the kernels; the collective is ordered behind them by torch (it waits for the work
        # enqueued on the current stream) without a host synchronisation.  (torch's default stream has handle 0, which the C ABI
        # reads as "use the context's own stream" -- never rely on it.)  Handed over at construction: the context then never
        # creates a stream of its own (hardware queues are few; Sampler.__init__).
        self.stream = torch.cuda.Stream(device=self.device)
        self.s = Sampler(number_zones, T, self.device, stream=self.stream)
        self.Z, self.T = int(number_zones), int(T)
        dev = f"cuda:{self.device}"
        self._bufs = [torch.zeros(self.s.counts_words(), dtype=torch.int64, device=dev) for _ in range(2)]
        self._work = [None, None]
        self._k = 0
        self.counts = self._bufs[0]
        torch.cuda.synchronize(self.device)  # counts zero-filled before the first enqueue on self.stream

    def init_states(self, C_total, cars_per_zone):
        first, stride, count = shard_cars(C_total, self.rank, self.world_size, self.deal)
        self.s.init_states(C_total, cars_per_zone, first, count, car_stride=stride)
        self.C_total = int(C_total)
        return first, count

    def resample_allreduce_async(self, seed, travel=False):
        """Enqueue the fused resample of this shard and its all-reduce into the next of the two count tensors; returns
        (tensor, ticket).  The collective runs beside whatever is enqueued next on self.stream (the next step's kernels);
        `wait(ticket)` orders self.stream behind it.  A tensor is reused every second call: its previous collective is
        waited for (on the stream, not on the host) before the kernels overwrite it."""
        i = self._k & 1
        self._k += 1
        buf = self._bufs[i]
        with torch.cuda.stream(self.stream):
            if self._work[i] is not None:
                self._work[i].wait()
            self.s.resample_dev(seed, buf.data_ptr(), travel=travel)
            self._work[i] = allreduce_counts(buf, async_op=True)
        self.counts = buf
        return buf, i

    def wait(self, ticket):
        with torch.cuda.stream(self.stream):
            if self._work[ticket] is not None:
                self._work[ticket].wait()
                self._work[ticket] = None

    def resample_allreduce(self, seed, travel=False):
        """Resample + all-reduce, ordered on self.stream when it returns; returns the device tensor."""
        buf, ticket = self.resample_allreduce_async(seed, travel=travel)
        self.wait(ticket)
        return buf

    def synchronize(self):
        for t in (0, 1):
            self.wait(t)
        self.stream.synchronize()

    def close(self):
        self.s.close()
