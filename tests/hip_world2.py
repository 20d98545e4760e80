"""Run by tests/test_gpu_parity.py::test_two_hip_ranks_on_one_gpu_sum_to_the_single_run, once per rank (the test starts two of these
processes): the N > 1 path of bench.py / ShardedSampler with HIP ranks side by side -- every rank creates its own context on the
one visible GPU, is dealt its share of the fleet (interleaved or contiguous), runs IVP + resample there, and the integer count
tensors are summed over a gloo group (a one-GPU box cannot hold an RCCL group of two ranks: RCCL itself runs with one rank in
tests/rccl_world1.py; the driver's 8-GPU node runs it with eight).  Rank 0 prints one JSON line with the hashes of the sums."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
Z, cpz, table_seed, sim_seed, deal = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6], 0), int(sys.argv[7], 0), sys.argv[8]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = port

import torch
import torch.distributed as dist

dist.init_process_group("gloo", rank=rank, world_size=world)
from carparkingmaps_amd.distributed import ShardedSampler, split_counts  # noqa: E402

T, C = 24, Z * cpz
ss = ShardedSampler(Z, T, device=0, deal=deal)          # rank / world size from the process group
ss.s.synth_tables(table_seed)
first, count = ss.init_states(C, cpz)
ss.s.solve_ivp_async(sim_seed)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "count": count, "steps": []}
for k in range(2):
    with torch.cuda.stream(ss.stream):
        ss.s.resample_dev(sim_seed + k, ss.counts.data_ptr(), travel=False)
        host = ss.counts.cpu()
    own_cars = int(host[:Z].sum())                      # this rank's cars at hour 1
    dist.all_reduce(host)                               # the path's one exchange: integer sum of [parking | driving | time | status]
    pk, dr, tt = split_counts(host, Z, T)
    out["steps"].append({"parking": hashlib.sha256(pk.tobytes(order="F")).hexdigest(), "driving": hashlib.sha256(dr.tobytes(order="F")).hexdigest(),
                         "cars_per_hour_ok": bool((pk.sum(axis=0) == C).all()), "own_cars_ok": own_cars == count})
counts = [None] * world
dist.all_gather_object(counts, count)
out["counts"] = counts
ss.close()
dist.barrier()
dist.destroy_process_group()
if rank == 0:
    print("RESULT " + json.dumps(out), flush=True)
