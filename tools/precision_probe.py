"""Error of the loaded library (SDY_AMD_LIB selects a variant build) against the REFERENCE's full-size output
(tests/golden/fx_sfno_full.npz) and its fp16-range behaviour: prints one JSON line.  Run on the GPU box."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import golden_utils as gu
import sdy_amd
from conftest import rel_l2

z = gu.load("fx_sfno_full")
cfg, n_in, n_cond, sd, x, cond, t = gu.seeded_case(z)
net = sdy_amd.SphericalFourierNeuralOperatorNet(
    num_input_channels=n_in, num_output_channels=cfg.out_chans, num_conditional_channels=n_cond,
    spatial_shape_in=(cfg.nlat, cfg.nlon), embed_dim=cfg.embed_dim, num_layers=cfg.num_layers, with_time_emb=True)
net.load_state_dict(sd, strict=True)
net.set_min_max_time(cfg.min_time, cfg.max_time)
ref = torch.from_numpy(z["y"])
out = {"lib": os.path.basename(sdy_amd.LIB_PATH)}
y = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
out["rel_l2_vs_reference"] = rel_l2(y, ref)
out["worst_channel"] = max(rel_l2(y[:, c], ref[:, c]) for c in range(ref.shape[1]))
# range: scale the inputs; the status word tells when a staged value left the fp16 range
for s in (30.0, 300.0, 3000.0, 30000.0):
    sdy_amd.ops.status_flags(reset=True)
    ys = net((x * s).cuda(), time=t.cuda(), condition=(cond * s).cuda())
    torch.cuda.synchronize()
    out[f"flags_at_input_scale_{int(s)}"] = int(sdy_amd.ops.status_flags(reset=True))
    out[f"finite_at_input_scale_{int(s)}"] = bool(torch.isfinite(ys).all())
# timing of one forward (B = 1), 10 reps
torch.cuda.synchronize(); import time; t0 = time.perf_counter()
for _ in range(10): net(x.cuda(), time=t.cuda(), condition=cond.cuda())
torch.cuda.synchronize(); out["fwd_ms_b1"] = round((time.perf_counter() - t0) * 100, 3)
print(json.dumps(out))
