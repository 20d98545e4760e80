import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import carparkingmaps_amd as cpm
from carparkingmaps_amd.distributed import ShardedSampler, split_counts
Z, T, cpz, world = 4096, 24, 400, 2
C = Z * cpz
tot = None
for rank in range(world):
    ss = ShardedSampler(Z, T, rank=rank, world_size=world, device=0)
    ss.s.synth_tables(0x5EED7AB1E)
    b, n = ss.init_states(C, cpz)
    ss.s.solve_ivp_async(0x5EEDCA125); ss.s.sync()
    st = ss.s.get_state()
    print("rank", rank, "range", b, n, "state min/max", st.min(), st.max(), flush=True)
    for it in range(3):
        ss.s.resample_dev(0x5EEDCA125, ss.counts.data_ptr())
        torch.cuda.synchronize()
        flat = ss.counts.cpu().numpy()
        pk = flat[:Z*T].reshape((Z, T), order="F")
        print("  iter", it, "hour sums", pk.sum(axis=0)[:4], "status", flat[2*Z*T+1], flush=True)
    tot = flat.copy() if tot is None else tot + flat
    ss.close()
pk = tot[:Z*T].reshape((Z, T), order="F")
print("total hour sums", pk.sum(axis=0)[:6], "C", C)
