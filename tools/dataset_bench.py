#!/usr/bin/env python3
"""Development aid, run on the GPU box: bench.py's per_dataset record alone (main.jl:79-95 for one NEW dataset at Melbourne's shape)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


class Env:
    local_rank = 0


print(json.dumps(bench.per_dataset_record(Env()), indent=1))
