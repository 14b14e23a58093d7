"""Helper of test_gpu_variants.py::test_drop_path_skip_is_bit_identical: seeded full-width SFNO forwards with a HIGH drop-path
rate (so that every middle block drops some trajectories, and single-trajectory batches drop whole blocks), saved to a file.

Run in a subprocess because SDY_NO_DROP_SKIP is read once per process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.sfno import SFNOConfig  # noqa: E402
from tests.helpers import make_pair  # noqa: E402


def main(out_path: str) -> None:
    outs = {}
    for grid in ("equiangular", "legendre-gauss"):
        cfg = SFNOConfig(in_chans=70, out_chans=34, nlat=180, nlon=360, embed_dim=256, num_layers=5, with_time_emb=True,
                         dropout_mlp=0.1, drop_path_rate=0.5, min_time=1.0, max_time=5.0, data_grid=grid)
        net, _, _ = make_pair(cfg, 68, 2)
        g = torch.Generator(device="cpu").manual_seed(77)
        x = torch.randn(6, 68, cfg.nlat, cfg.nlon, generator=g).cuda()
        cond = torch.randn(6, 2, cfg.nlat, cfg.nlon, generator=g).cuda()
        t = torch.tensor([1.0, 3.0, 4.0, 2.0, 5.0, 1.5]).cuda()
        net.inference_dropout = True
        outs[f"{grid}/b6"] = net(x, time=t, condition=cond).cpu()
        outs[f"{grid}/b6_stacked"] = net(x, time=t, condition=cond, rows_per_call=3).cpu()      # two calls of 3 trajectories
        net.batch_offset = 11
        for k in range(4):                                                                       # B = 1: whole blocks dropped
            outs[f"{grid}/b1_{k}"] = net(x[k:k + 1], time=t[k:k + 1], condition=cond[k:k + 1]).cpu()
        for k in range(6):      # the sampler's stacked pair: one trajectory, two calls (one kept row beside one dropped row)
            outs[f"{grid}/pair_{k}"] = net(x[k:k + 1].repeat(2, 1, 1, 1), time=torch.stack([t[k], t[(k + 1) % 6]]),
                                           condition=cond[k:k + 1].repeat(2, 1, 1, 1), rows_per_call=1).cpu()
    torch.save(outs, out_path)


if __name__ == "__main__":
    main(sys.argv[1])
