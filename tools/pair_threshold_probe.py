"""Up to which batch should the two interpolations of a cold-sampling step run as ONE stacked forward of 2B rows
(`DYffusion.fuse_interpolator_pair_max_batch`) instead of two forwards that share the encoder pass?  Times a sampling pass at
several batch sizes under both rules (run on the GPU box)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
exp = bench.build_models(dev)
out = {}
for B in [int(a) for a in sys.argv[1:]] or [6, 9, 12, 13, 25]:
    x, f = bench.synthetic_state(0, B, dev)
    row = {}
    for limit in (0, 8, 64):
        exp.model.fuse_interpolator_pair_max_batch = limit
        exp.set_batch_offset(0)
        y = x
        for _ in range(2):
            y = bench.one_pass(exp, y, f)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 4 if B > 8 else 8
        for _ in range(n):
            y = bench.one_pass(exp, y, f)
        torch.cuda.synchronize()
        row["stack_up_to_%d" % limit] = round((time.perf_counter() - t0) / n * 1e3, 2)
    out[str(B)] = row
print(json.dumps(out))
