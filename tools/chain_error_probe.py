"""How much of the difference between the HIP path and the fp32 oracle over a whole horizon-6 sampling pass (16 chained
SFNO forwards, interpolator dropout on) is either side's rounding?  Runs the chain three times on identical inputs and
identical (Philox-replayed) masks -- oracle in float64 (the yardstick), oracle in float32 (the reference's arithmetic), the
HIP path -- and prints every pairwise relative L2 per lead time, plus the chain's sensitivity (output change / input change for
a 1e-6 relative perturbation of the initial condition, on the device).  Run on the GPU box:
    python tools/chain_error_probe.py [layers=8]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import sdy_amd
from conftest import rel_l2
from helpers import PhiloxMasks, make_pair
from oracle.dyffusion import OracleDYffusion
from oracle.sfno import OracleSFNO, SFNOConfig

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 8
NLAT, NLON, E, HZ, C, F = 180, 360, 256, 6, 63, 2
fcfg = SFNOConfig(in_chans=C + F, out_chans=C, nlat=NLAT, nlon=NLON, embed_dim=E, num_layers=layers, with_time_emb=True,
                  min_time=0.0, max_time=HZ - 1.0)
icfg = SFNOConfig(in_chans=2 * C + F, out_chans=C, nlat=NLAT, nlon=NLON, embed_dim=E, num_layers=layers, with_time_emb=True,
                  dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=HZ - 1.0)
fnet, fora32, fsd = make_pair(fcfg, C, F, seed=4321)
inet, iora32, isd = make_pair(icfg, 2 * C, F, seed=4322, net_seed=1000)
exp = sdy_amd.MultiHorizonForecastingDYffusion(fnet, sdy_amd.InterpolationExperiment(inet, horizon=HZ), horizon=HZ)
g = torch.Generator(device="cpu").manual_seed(1234)
x0 = torch.randn(1, C, NLAT, NLON, generator=g)
forc = torch.randn(1, F, NLAT, NLON, generator=g)


def oracle_chain(fora, iora):
    masks = PhiloxMasks(icfg, seed=1000)
    n = {"i": 0}

    def ora_i(x, time, condition=None, static_condition=None):
        masks.call = n["i"]
        n["i"] += 1
        return iora(x, time=time, condition=condition, static_condition=static_condition, mask_fn=masks)

    o = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(
        x, time=time, condition=condition, static_condition=static_condition), ora_i, timesteps=HZ)
    return o.sample(x0, static_condition=forc)


out = {"layers": layers}
t0 = time.time()
got = {k: v.cpu() for k, v in exp.model.sample(x0.cuda(), static_condition=forc.cuda()).items()}
# sensitivity of the chain itself: the same pass from a perturbed initial condition (same dropout stream)
fnet._call = inet._call = 0
eps = 1e-6
xp = x0 * (1.0 + eps * torch.randn(x0.shape, generator=g))
gotp = {k: v.cpu() for k, v in exp.model.sample(xp.cuda(), static_condition=forc.cuda()).items()}
d_in = rel_l2(xp, x0)
out["sensitivity"] = {k: rel_l2(gotp[k], got[k]) / d_in for k in sorted(got)}
ref32 = oracle_chain(fora32, iora32)
out["oracle32_s"] = round(time.time() - t0, 1)
ref64 = oracle_chain(OracleSFNO(fcfg, fsd, dtype=torch.float64), OracleSFNO(icfg, isd, dtype=torch.float64))
out["total_s"] = round(time.time() - t0, 1)
for name, a, b in (("hip_vs_oracle32", got, ref32), ("hip_vs_oracle64", got, ref64), ("oracle32_vs_oracle64", ref32, ref64)):
    out[name] = {k: rel_l2(a[k], b[k]) for k in sorted(got)}
print(json.dumps(out))
