"""Network-level parity: the native SFNO forward vs the CPU oracle on identical weights and inputs.

Tolerance: north_star's bound, 1e-4 relative L2 for the full network in fp32 (measured errors are ~1e-6).
"""
import pytest
import torch

from conftest import rel_l2
from helpers import PhiloxMasks, make_pair
from oracle.sfno import SFNOConfig

pytestmark = pytest.mark.gpu

TOL_NET = 1e-4     # north_star bound
TOL_TIGHT = 2e-5   # what fp32 MFMA actually achieves, with margin


def _inputs(cfg, n_in, n_cond, B, seed=1234):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn(B, n_in, cfg.nlat, cfg.nlon, generator=g)
    cond = torch.randn(B, n_cond, cfg.nlat, cfg.nlon, generator=g) if n_cond else None
    return x, cond


def test_c1_single_block_32x64_e8():
    """BASELINE.json configs[0]: one SFNO block, 32x64 grid, 8 channels (block 0 == last block: equiangular both ways)."""
    cfg = SFNOConfig(in_chans=8, out_chans=8, nlat=32, nlon=64, embed_dim=8, num_layers=1, with_time_emb=True,
                     min_time=0.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 8, 0)
    x, _ = _inputs(cfg, 8, 0, 2)
    t = torch.tensor([1.0, 4.0])
    ref = ora(x, time=t)
    got = net(x.cuda(), time=t.cuda())
    err = rel_l2(got, ref)
    assert err < TOL_TIGHT, f"C1 rel L2 {err:.3e}"


@pytest.mark.parametrize("data_grid", ["equiangular", "legendre-gauss"])
def test_tiny_sfno_with_condition_and_time(data_grid):
    cfg = SFNOConfig(in_chans=10, out_chans=6, nlat=32, nlon=64, embed_dim=16, num_layers=3, with_time_emb=True,
                     data_grid=data_grid, min_time=0.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 8, 2)
    x, cond = _inputs(cfg, 8, 2, 3)
    t = torch.tensor([0.0, 2.0, 5.0])
    ref = ora(x, time=t, condition=cond)
    got = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    err = rel_l2(got, ref)
    assert err < TOL_TIGHT, f"tiny SFNO ({data_grid}) rel L2 {err:.3e}"
    # static_condition is the other way the reference feeds forcings (stepper_multistep.py:383-384)
    got2 = net(x.cuda(), time=t.cuda(), static_condition=cond.cuda())
    assert torch.equal(got, got2)
    # time embedding tap
    trep, ss = net.time_embedding(t.cuda())
    assert rel_l2(trep, ora.time_repr(t)) < 2e-6


def test_time_embedding_ragged_sizes():
    """The time MLP's dense-layer kernel away from its comfortable sizes: a contraction longer than one 1024-term LDS chunk and
    not a multiple of 64 (time_dim = 12 * 88 = 1056), fewer than 64 outputs per layer matrix (2 E = 24), 11 rows (two row
    blocks, the second with three rows) -- t_repr and the blocks' (scale | shift) against the oracle, then a whole forward."""
    cfg = SFNOConfig(in_chans=5, out_chans=3, nlat=32, nlon=64, embed_dim=12, num_layers=2, with_time_emb=True,
                     time_dim_mult=88, min_time=0.0, max_time=9.0)
    net, ora, _ = make_pair(cfg, 5, 0)
    t = torch.linspace(0.0, 9.0, 11)
    trep, ss = net.time_embedding(t.cuda())
    assert trep.shape == (11, 1056) and rel_l2(trep, ora.time_repr(t)) < 2e-6
    x, _ = _inputs(cfg, 5, 0, 11)
    err = rel_l2(net(x.cuda(), time=t.cuda()), ora(x, time=t))
    assert err < TOL_TIGHT, f"rel L2 {err:.3e}"


def test_time_embedding_of_a_row_does_not_depend_on_the_batch():
    """Member sharding relies on it: a trajectory's (t_repr, scale | shift) are BITWISE the same whatever batch it sits in and
    wherever in it (the dense-layer kernel keeps one explicit FMA chain per row: with the sums left to the compiler's
    contraction, row 7 of every 8-row block came out one ulp away from rows 0 .. 6)."""
    cfg = SFNOConfig(in_chans=10, out_chans=6, nlat=32, nlon=64, embed_dim=256, num_layers=8, with_time_emb=True,
                     min_time=0.0, max_time=5.0)
    net, _, _ = make_pair(cfg, 10, 0)
    a, sa = net.time_embedding(torch.full((25,), 3.0).cuda())
    b, sb = net.time_embedding(torch.tensor([4.0, 3.0]).cuda())
    c, sc = net.time_embedding(torch.tensor([3.0]).cuda())
    for i in range(1, 25):
        assert torch.equal(a[i], a[0]) and torch.equal(sa[i], sa[0]), f"row {i} of 25 equal times differs from row 0"
    assert torch.equal(b[1], a[0]) and torch.equal(c[0], a[0]) and torch.equal(sb[1], sa[0]) and torch.equal(sc[0], sa[0])


def test_tiny_sfno_no_time_no_skip():
    cfg = SFNOConfig(in_chans=4, out_chans=4, nlat=32, nlon=64, embed_dim=8, num_layers=2, with_time_emb=False,
                     big_skip=False, pos_embed=False)
    net, ora, _ = make_pair(cfg, 4, 0)
    x, _ = _inputs(cfg, 4, 0, 2)
    err = rel_l2(net(x.cuda()), ora(x))
    assert err < TOL_TIGHT, f"rel L2 {err:.3e}"


def test_tiny_sfno_dropout_stream_matches_oracle():
    """Interpolator configuration: dropout + drop path ON at inference (dyffusion.py:226-235).  The oracle replays the
    device Philox stream, so the comparison is exact-mask, not statistical."""
    cfg = SFNOConfig(in_chans=18, out_chans=8, nlat=32, nlon=64, embed_dim=16, num_layers=4, with_time_emb=True,
                     dropout_mlp=0.1, drop_path_rate=0.5, min_time=1.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 16, 2, net_seed=777)
    net.batch_offset = 3
    x, cond = _inputs(cfg, 16, 2, 4)
    t = torch.tensor([1.0, 2.0, 3.0, 5.0])
    masks = PhiloxMasks(cfg, seed=777, batch_offset=3)
    net.enable_inference_dropout()
    outs = []
    for call in range(2):   # two calls: the stream must advance
        masks.call = call
        ref = ora(x, time=t, condition=cond, mask_fn=masks)
        got = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
        err = rel_l2(got, ref)
        assert err < TOL_TIGHT, f"dropout call {call}: rel L2 {err:.3e}"
        outs.append(got)
    assert not torch.equal(outs[0], outs[1]), "dropout stream did not advance between calls"
    net.disable_inference_dropout()
    ref_off = ora(x, time=t, condition=cond)
    assert rel_l2(net(x.cuda(), time=t.cuda(), condition=cond.cuda()), ref_off) < TOL_TIGHT


def test_injected_masks():
    cfg = SFNOConfig(in_chans=8, out_chans=8, nlat=32, nlon=64, embed_dim=8, num_layers=2, with_time_emb=True,
                     dropout_mlp=0.2, drop_path_rate=0.3, min_time=1.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 8, 0)
    B = 2
    x, _ = _inputs(cfg, 8, 0, B)
    t = torch.tensor([1.0, 2.0])
    g = torch.Generator(device="cpu").manual_seed(5)
    hid = int(cfg.embed_dim * cfg.mlp_ratio)
    km = []
    for i in range(cfg.num_layers):
        km.append((torch.rand(B, hid, 32, 64, generator=g) > 0.2).float())
        km.append((torch.rand(B, cfg.embed_dim, 32, 64, generator=g) > 0.2).float())
    dpk = torch.tensor([[1.0, 1.0], [0.0, 1.0]])   # [layer][b]

    def mask_fn(kind, layer, shape):
        if kind == "drop_path":
            return dpk[layer].reshape(-1, 1, 1, 1)
        return km[2 * layer + (0 if kind == "mlp_hidden" else 1)]

    ref = ora(x, time=t, mask_fn=mask_fn)
    net.enable_inference_dropout()
    got = net(x.cuda(), time=t.cuda(), keep_masks=km, drop_path_keep=dpk)
    err = rel_l2(got, ref)
    assert err < TOL_TIGHT, f"injected masks rel L2 {err:.3e}"


def test_c2_full_size_interpolator_forward():
    """BASELINE.json configs[1]: interpolator SFNO single step, 180x360, E=256, 8 layers, 68+2 -> 34 channels, B=1."""
    cfg = SFNOConfig(in_chans=70, out_chans=34, nlat=180, nlon=360, embed_dim=256, num_layers=8, with_time_emb=True,
                     dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 68, 2)
    x, cond = _inputs(cfg, 68, 2, 1)
    t = torch.tensor([3.0])
    torch.set_num_threads(max(1, torch.get_num_threads()))
    ref = ora(x, time=t, condition=cond)                       # dropout off: exact comparison
    got = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    err = rel_l2(got, ref)
    assert err < TOL_NET, f"C2 full-size rel L2 {err:.3e} (bound 1e-4)"
    assert err < TOL_TIGHT, f"C2 full-size rel L2 {err:.3e} (fp32-MFMA expectation)"
    # linearity-free, size-independent sanity at full size: batch consistency (B=2 rows equal B=1 results)
    x2 = torch.cat([x, x.flip(0)], 0).cuda()
    got2 = net(x2, time=torch.tensor([3.0, 3.0]).cuda(), condition=torch.cat([cond, cond], 0).cuda())
    assert torch.equal(got2[0], got[0]) and torch.equal(got2[1], got[0])


def test_c2_full_size_interpolator_forward_with_dropout():
    """BASELINE.json configs[1] exactly as SURVEY.md 8(d) writes it: interpolator SFNO, 180x360, E = 256, ALL 8 blocks,
    68 + 2 -> 34 channels, B = 1, time = 3.0, dropout AND drop path ON -- the fused `mlp_h3_kernel<true>` at full depth
    (layers 1-6 with their intermediate drop-path rates, the LG <-> LG blocks, stream ids of every layer) against the oracle
    replaying the Philox stream (reference: src/models/sfno/sfnonet.py:622,789-794, layers.py:76-78, drop_path.py:15-22).
    The dropout seed is chosen (by the oracle's generator, on the host) so that trajectory 0 loses at least one whole MLP
    branch to drop path at an inner layer AND keeps at least one, so both outcomes of the draw are compared."""
    from oracle.philox import drop_path_keep

    cfg = SFNOConfig(in_chans=70, out_chans=34, nlat=180, nlon=360, embed_dim=256, num_layers=8, with_time_emb=True,
                     dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=5.0)
    rates = cfg.drop_path_rates

    def kept(seed):
        return [bool(drop_path_keep(seed, 0, layer, rates[layer], 1)[0]) for layer in range(8)]

    seed = next(sd for sd in range(5000, 5400) if not all(kept(sd)[1:7]) and sum(kept(sd)) >= 5)
    assert kept(seed)[0], "layer 0 has rate 0: always kept (sfnonet.py:252)"
    net, ora, _ = make_pair(cfg, 68, 2, net_seed=seed)
    x, cond = _inputs(cfg, 68, 2, 1)
    t = torch.tensor([3.0])
    masks = PhiloxMasks(cfg, seed=seed)
    masks.call = 0
    ref = ora(x, time=t, condition=cond, mask_fn=masks)
    net.enable_inference_dropout()
    got = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    assert torch.isfinite(got).all()
    err = rel_l2(got, ref)
    assert err < TOL_NET, f"C2 full-size, dropout on: rel L2 {err:.3e} (bound 1e-4)"
    assert err < TOL_TIGHT, f"C2 full-size, dropout on: rel L2 {err:.3e} (fp32-MFMA expectation)"
    # the masks matter at this depth: the next call of the stream gives a visibly different field
    again = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    assert rel_l2(again, ref) > 1e-3


@pytest.mark.parametrize("scale", [1.0e4, 3.0e2])
def test_network_inputs_of_any_magnitude_in_split_fp16_mode(scale):
    """|x| = 1e4 (and 300: where round 3's hidden activations overflowed first) in the default split-fp16 mode, production width (E = 256: the fused encoder / decoder kernels),
    against the oracle: the un-normalised tensors -- the inputs at the encoder, [block output | inputs] at the decoder -- are
    staged with a per-tile scale of their own maximum, everything between is InstanceNorm'ed, so the network no longer
    needs inputs below 4094: parity at the network tolerance and a clean status word.
    (The fixed pre-scale of round 3 raised SdyError here and sent the user to SDY_GEMM_MODE=f32.)"""
    import sdy_amd

    cfg = SFNOConfig(in_chans=10, out_chans=6, nlat=32, nlon=64, embed_dim=256, num_layers=2, with_time_emb=True,
                     min_time=0.0, max_time=5.0)
    net, ora, _ = make_pair(cfg, 8, 2)
    assert net.gemm_mode == "h3"
    x, cond = _inputs(cfg, 8, 2, 2)
    x, cond = x * scale, cond * scale
    t = torch.tensor([1.0, 4.0])
    sdy_amd.ops.status_flags(reset=True)
    got = net(x.cuda(), time=t.cuda(), condition=cond.cuda())
    assert sdy_amd.ops.status_flags(reset=True) == 0
    ref = ora(x, time=t, condition=cond)
    err = rel_l2(got, ref)
    assert err < TOL_TIGHT, f"inputs x {scale:g}: rel L2 {err:.3e}"
