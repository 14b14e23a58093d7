"""Do two INDEPENDENT small sampling jobs on two HIP streams fill each other's launch gaps?  Two samplers (own networks, own
workspaces), B rows each: K passes each issued alternately on two streams from one host thread, against the same work on one
stream.  Measurement aid for the small-batch question (NOTEBOOK.md); run on the GPU box.
    python tools/two_stream_probe.py [B=3] [K=6]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
exps = [bench.build_models(dev), bench.build_models(dev)]
states = [bench.synthetic_state(0, B, dev), bench.synthetic_state(1, B, dev)]
for e in exps:
    e.set_batch_offset(0)
streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]


def run(two_streams, k):
    xs = [s[0] for s in states]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if two_streams:
        # interleave at forward granularity is not possible from outside the sampler: alternate whole passes, each on its own
        # stream; the host enqueues far ahead of the device, so both streams always have work queued
        for _ in range(k):
            for j in (0, 1):
                with torch.cuda.stream(streams[j]):
                    xs[j] = bench.one_pass(exps[j], xs[j], states[j][1])
    else:
        for _ in range(k):
            for j in (0, 1):
                xs[j] = bench.one_pass(exps[j], xs[j], states[j][1])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3


for _ in range(2):
    run(False, 1)
    run(True, 1)
for rep in range(3):
    a = run(False, K)
    b = run(True, K)
    print("B=%d x 2 jobs: one stream %.2f ms per pass pair, two streams %.2f ms (%.3f x)" % (B, a, b, b / a), flush=True)
