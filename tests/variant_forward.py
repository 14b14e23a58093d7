"""Helper of test_gpu_variants.py: one seeded full-width SFNO forward (dropout on) saved to a file.

Run in a subprocess because the kernel-selection switches (SDY_NO_*) are read once per process."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from oracle.sfno import SFNOConfig  # noqa: E402
from tests.helpers import make_pair  # noqa: E402


def main(out_path: str) -> None:
    cfg = SFNOConfig(in_chans=70, out_chans=34, nlat=180, nlon=360, embed_dim=256, num_layers=2, with_time_emb=True,
                     dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=5.0)
    net, _, _ = make_pair(cfg, 68, 2)
    g = torch.Generator(device="cpu").manual_seed(77)
    x = torch.randn(3, 68, cfg.nlat, cfg.nlon, generator=g).cuda()
    cond = torch.randn(3, 2, cfg.nlat, cfg.nlon, generator=g).cuda()
    t = torch.tensor([1.0, 3.0, 4.0]).cuda()
    net.inference_dropout = True
    y = net(x, time=t, condition=cond)
    torch.save(y.cpu(), out_path)


if __name__ == "__main__":
    main(sys.argv[1])
