"""Multi-GPU layer: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Cars are independent chains (src/resampling.jl:11-83 reads only car i's row), so the C cars are
cut into WORLD_SIZE contiguous ranges; every rank holds the full tables and samples its range
with Philox keyed by the GLOBAL car id, which makes the summed histogram identical for every
world size.  The only exchange on the path is ONE all-reduce(sum) of the integer tensor
[parking | driving | travel-time q16 | status] (2*T*Z+2 int64 words) before normalisation
(src/saveresults.jl:20) -- integer addition, so the result is order-free and bit-exact.
"""
import torch
import torch.distributed as dist


def shard_range(C_total, rank, world_size):
    """Contiguous car range [begin, begin+count) of `rank`; ranges tile [0, C_total) exactly."""
    base, rem = divmod(int(C_total), int(world_size))
    begin = rank * base + min(rank, rem)
    count = base + (1 if rank < rem else 0)
    return begin, count


def allreduce_counts(counts):
    """In-place sum over ranks of the int64 count tensor (device tensor -> RCCL over xGMI;
    CPU tensor -> gloo, used by the CPU tests of the sharding logic)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM)
    return counts


def split_counts(counts, Z, T):
    """[2*T*Z+2] int64 -> (parking (Z,T), driving (Z,T) as Fortran-ordered views, sum_tt_q16).
    Raises if the status word is set (fused-kernel rank overflow: repeat with another kernel)."""
    zt = Z * T
    flat = counts.detach().cpu().numpy()
    if flat.size > 2 * zt + 1 and flat[2 * zt + 1] != 0:
        raise RuntimeError("resample status != 0: fused zone kernel overflow, repeat with CPM_KERNEL_ZONE_LDS")
    parking = flat[:zt].reshape((Z, T), order="F")
    driving = flat[zt:2 * zt].reshape((Z, T), order="F")
    return parking, driving, int(flat[2 * zt])


class ShardedSampler:
    """A Sampler bound to this rank's GPU and car range, with the count all-reduce."""

    def __init__(self, number_zones, T=24, rank=None, world_size=None, device=None):
        from .sampler import Sampler  # needs the HIP library
        self.rank = dist.get_rank() if rank is None else rank
        self.world_size = dist.get_world_size() if world_size is None else world_size
        self.device = torch.cuda.current_device() if device is None else device
        self.s = Sampler(number_zones, T, self.device)
        self.Z, self.T = int(number_zones), int(T)
        self.counts = torch.zeros(self.s.counts_words(), dtype=torch.int64, device=f"cuda:{self.device}")
        # One explicit torch stream carries both the kernels and the collective, so the all-reduce is
        # ordered behind the resample without a host synchronisation.  (torch's default stream has
        # handle 0, which the C ABI reads as "use the context's own stream" -- never rely on it.)
        self.stream = torch.cuda.Stream(device=self.device)
        self.s.set_stream(self.stream.cuda_stream)
        torch.cuda.synchronize(self.device)  # counts zero-filled before the first enqueue on self.stream

    def init_states(self, C_total, cars_per_zone):
        begin, count = shard_range(C_total, self.rank, self.world_size)
        self.s.init_states(C_total, cars_per_zone, begin, count)
        self.C_total = int(C_total)
        return begin, count

    def resample_allreduce(self, seed, travel=False):
        """Enqueue the fused resample of this shard and the all-reduce; returns the device tensor."""
        with torch.cuda.stream(self.stream):
            self.s.resample_dev(seed, self.counts.data_ptr(), travel=travel)
            allreduce_counts(self.counts)
        return self.counts

    def synchronize(self):
        self.stream.synchronize()

    def close(self):
        self.s.close()
