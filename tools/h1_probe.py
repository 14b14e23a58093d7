#!/usr/bin/env python3
"""What the three split-precision passes cost and buy: BASELINE.json configs[1] (interpolator SFNO, 180 x 360, 68 + 2 -> 34
channels, E = 256, 8 blocks, B = 1) under the shipped library and under a single-pass measurement build
(`make -C spherical-dyffusion_amd/csrc O=/tmp/obj_h1 OUT=$PWD/build/variants/libsdy_amd_h1.so EXTRA=-DSDY_H3_PASSES=1`).
One process per library (the binding is chosen at import):

    python tools/h1_probe.py run /tmp/c2_default.pt
    SDY_AMD_LIB=$PWD/build/variants/libsdy_amd_h1.so python tools/h1_probe.py run /tmp/c2_h1.pt
    python tools/h1_probe.py compare /tmp/c2_default.pt /tmp/c2_h1.pt
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def run(path):
    import sdy_amd
    from sdy_amd import synthetic

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    net = synthetic.build_network(68, 34, 2, dropout_mlp=0.1, drop_path_rate=0.1, time_range=(1.0, 5.0), weight_seed=4321)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(1, 68, 180, 360, generator=g).to(dev)
    c = torch.randn(1, 2, 180, 360, generator=g).to(dev)
    t = torch.tensor([3.0], device=dev)
    y = net(x, time=t, condition=c)                     # dropout off: deterministic
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        net(x, time=t, condition=c)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    torch.save({"y": y.cpu(), "ms": ms, "lib": sdy_amd.LIB_PATH}, path)
    print(json.dumps({"lib": sdy_amd.LIB_PATH, "c2_forward_b1_ms": round(ms, 3)}))


def compare(a, b):
    A, B = torch.load(a), torch.load(b)
    ya, yb = A["y"].double(), B["y"].double()
    print(json.dumps({"reference": A["lib"], "other": B["lib"], "rel_l2": float((ya - yb).norm() / ya.norm()),
                      "max_abs": float((ya - yb).abs().max()), "ms_reference": round(A["ms"], 3), "ms_other": round(B["ms"], 3)}))


if __name__ == "__main__":
    if sys.argv[1] == "run":
        run(sys.argv[2])
    else:
        compare(sys.argv[2], sys.argv[3])
