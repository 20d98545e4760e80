"""Run by tests/test_gpu_parity.py::test_rccl_allreduce_on_the_sampler_stream in a process of its own: a torch.distributed
process group over RCCL (backend "nccl") is created BEFORE anything touches the GPU, then ShardedSampler runs the fused resample
and the device all-reduce of the count tensor -- the same calls bench.py --gpus N makes, with one rank.  Prints one JSON line."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = sys.argv[1]
Z, cpz, table_seed, sim_seed = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4], 0), int(sys.argv[5], 0)

import numpy as np
import torch
import torch.distributed as dist

dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from carparkingmaps_amd.distributed import ShardedSampler, split_counts  # noqa: E402

T, C = 24, Z * cpz
ss = ShardedSampler(Z, T)                       # rank / world size from the process group
ss.s.synth_tables(table_seed)
ss.init_states(C, cpz)
ss.s.solve_ivp_async(sim_seed)
out = {"backend": dist.get_backend(), "world": dist.get_world_size(), "steps": []}
tickets = []
for k in range(4):                              # pipelined: a step's all-reduce runs beside the next step's kernels
    buf, ticket = ss.resample_allreduce_async(sim_seed + (k & 1))
    tickets.append((buf, ticket, k))
    if len(tickets) == 2:                       # a tensor is reused every second step: read it before that
        b, t, kk = tickets.pop(0)
        ss.wait(t)
        with torch.cuda.stream(ss.stream):
            host = b.to("cpu")
        pk, dr, tt = split_counts(host, Z, T)
        out["steps"].append({"k": kk, "parking": hashlib.sha256(pk.tobytes(order="F")).hexdigest(), "driving": hashlib.sha256(dr.tobytes(order="F")).hexdigest(),
                             "cars_per_hour_ok": bool((pk.sum(axis=0) == C).all())})
ss.synchronize()
counts = ss.resample_allreduce(sim_seed)        # the blocking form
with torch.cuda.stream(ss.stream):
    pk, dr, tt = split_counts(counts.to("cpu"), Z, T)
out["sync"] = {"parking": hashlib.sha256(pk.tobytes(order="F")).hexdigest(), "driving": hashlib.sha256(dr.tobytes(order="F")).hexdigest()}
ss.close()
dist.destroy_process_group()
print("RESULT " + json.dumps(out), flush=True)
