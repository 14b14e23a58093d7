"""Production-size parity of the sampler (BASELINE.json configs[2] and the B = 25 batch of configs[3]): the kernels the
benchmark actually runs - `mlp_h3_kernel<true>` (fused MLP with the Philox dropout), `conv_h3`, `dh_h3`, `leg_par`,
`fft360` - inside the full-width network, against the CPU oracle replaying the same dropout stream.

The small-grid fixtures under tests/golden reach only the generic tile kernels (E <= 16); these tests close that gap.
"""
import pytest
import torch

from conftest import rel_l2
from helpers import PhiloxMasks, make_pair
from oracle.dyffusion import OracleDYffusion
from oracle.sfno import SFNOConfig

pytestmark = pytest.mark.gpu

NLAT, NLON, E, HZ = 180, 360, 256, 6
C_STATE, C_FORC = 63, 2          # the metric's nominal "63 ch": encoder widths 65 (forecaster) / 128 (interpolator)


def _build(layers, seed_i=1000):
    import sdy_amd

    fcfg = SFNOConfig(in_chans=C_STATE + C_FORC, out_chans=C_STATE, nlat=NLAT, nlon=NLON, embed_dim=E, num_layers=layers,
                      with_time_emb=True, min_time=0.0, max_time=HZ - 1.0)
    icfg = SFNOConfig(in_chans=2 * C_STATE + C_FORC, out_chans=C_STATE, nlat=NLAT, nlon=NLON, embed_dim=E,
                      num_layers=layers, with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0,
                      max_time=HZ - 1.0)
    fnet, fora, _ = make_pair(fcfg, C_STATE, C_FORC, seed=4321)
    inet, iora, _ = make_pair(icfg, 2 * C_STATE, C_FORC, seed=4322, net_seed=seed_i)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(fnet, sdy_amd.InterpolationExperiment(inet, horizon=HZ), horizon=HZ)
    return exp, fnet, inet, fora, iora, icfg


def test_production_forward_takes_the_fused_paths():
    """Which kernels a production-shape forward (180 x 360, E = 256, equiangular data grid, dropout on) actually launches, by the
    stage timer -- so that a silent fall-back to a slower path cannot pass as parity: the first and last block's inner skip is
    folded into their dhconv weights (4 of 4 middle blocks launch the skip convolution, the GELU rides on the inverse FFT: no
    pass of its own), the MLP is the fused kernel, encoder and decoder one launch each, every block on dh_h3 / leg_par / fft360."""
    import sdy_amd

    layers = 6
    exp, fnet, inet, _, _, _ = _build(layers)
    g = torch.Generator(device="cpu").manual_seed(3)
    x = torch.randn(2, 2 * C_STATE, NLAT, NLON, generator=g).cuda()
    c = torch.randn(2, C_FORC, NLAT, NLON, generator=g).cuda()
    inet.enable_inference_dropout()
    with sdy_amd.ops.stage_timer() as t:
        y = inet(x, time=torch.tensor([2.0, 4.0]).cuda(), static_condition=c)
    inet.disable_inference_dropout()
    assert torch.isfinite(y).all()
    n = {k: v[0] for k, v in t.stages.items()}
    assert n.get("inner-skip conv") == layers - 2 and "inner skip folded (gelu)" not in n
    assert n.get("mlp fused (dropout)") == layers and "mlp fc1" not in n and "mlp fc2" not in n
    assert n.get("encoder (fused pair)") == 1 and n.get("decoder (fused pair)") == 1
    assert n.get("dhconv") == layers and n.get("rfft (lon)") == layers
    assert n.get("legendre synthesis") == layers + 2 and n.get("irfft (lon)") == layers + 2      # + the two residual resamplings
    assert n.get("legendre analysis") == layers


@pytest.mark.parametrize("layers", [2, 8])
def test_c3_full_size_sampling_pass_with_dropout_vs_oracle(layers):
    """BASELINE.json configs[2]: one horizon-6 DYffusion sampling pass, 180x360, E = 256, B = 1, interpolator dropout and
    drop path ON, against the oracle replaying the same Philox stream.

    layers = 2 (first + last block: both grid changes): the quick version; bound 1e-4 (north_star) and the 2e-5 fp32
    expectation at every lead time.

    layers = 8: the production depth -- sixteen chained 8-block forwards.  The chain itself amplifies any perturbation, by a
    factor that doubles with every lead time (measured on the device below: about 3 at t1, 76 at t6 with these trained-like
    random weights), so two fp32 implementations cannot agree to 2e-5 at t6: against a float64 run of the same chain the
    reference's own fp32 arithmetic (the oracle) is 7.1e-5 off at t6 and the HIP path 5.4e-5 (tools/chain_error_probe.py,
    profiles/r4a/chain_error_probe_8_blocks.json).  What IS asserted at full depth: the north_star bound 1e-4 wherever the
    chain's sensitivity leaves room for it (lead times with amplification <= 30), the 2e-5 expectation at t1, and at EVERY lead
    time an error no larger than 4e-6 x the measured amplification -- i.e. the difference to the oracle stays at the level of
    one forward's rounding (1.1e-6 per unit of amplification measured) all the way through the pass."""
    exp, fnet, inet, fora, iora, icfg = _build(layers=layers)
    masks = PhiloxMasks(icfg, seed=1000)
    if layers > 2:
        masks.device = "cuda"   # 160 masks of 33 M decisions: the torch restatement of the stream, evaluated on the GPU
    n = {"i": 0}

    def ora_i(x, time, condition=None, static_condition=None):
        masks.call = n["i"]
        n["i"] += 1
        return iora(x, time=time, condition=condition, static_condition=static_condition, mask_fn=masks)

    oracle = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(
        x, time=time, condition=condition, static_condition=static_condition), ora_i, timesteps=HZ)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x0 = torch.randn(1, C_STATE, NLAT, NLON, generator=g)
    forc = torch.randn(1, C_FORC, NLAT, NLON, generator=g)
    got = exp.model.sample(x0.cuda(), static_condition=forc.cuda())
    assert (fnet._call, inet._call) == (6, 10)                  # the 16-call trace of tests/golden/fx_trace.json
    ref = oracle.sample(x0, static_condition=forc)
    assert n["i"] == 10
    assert sorted(got) == sorted(ref) == [f"t{k}_preds" for k in range(1, HZ + 1)]
    err = {}
    for k, v in ref.items():
        assert torch.isfinite(got[k]).all()
        err[k] = rel_l2(got[k], v)
    if layers == 2:
        worst = max(err.values())
        assert worst < 1e-4, f"C3 rel L2 {worst:.3e} (bound 1e-4)"
        assert worst < 2e-5, f"C3 rel L2 {worst:.3e} (fp32 expectation)"
    else:
        # the chain's own amplification: the same pass (same dropout stream) from an initial condition perturbed by 1e-6
        fnet._call = inet._call = 0
        xp = x0 * (1.0 + 1e-6 * torch.randn(x0.shape, generator=g))
        pert = exp.model.sample(xp.cuda(), static_condition=forc.cuda())
        d_in = rel_l2(xp, x0)
        amp = {k: rel_l2(pert[k], got[k]) / d_in for k in got}
        assert amp["t1_preds"] < 10 and amp["t6_preds"] > amp["t1_preds"], amp
        assert err["t1_preds"] < 2e-5, err
        for k in sorted(err):
            assert err[k] < 4e-6 * max(amp[k], 1.0), f"{k}: rel L2 {err[k]:.3e} at amplification {amp[k]:.1f}"
            if amp[k] <= 30.0:
                assert err[k] < 1e-4, f"{k}: rel L2 {err[k]:.3e} (bound 1e-4, amplification {amp[k]:.1f})"
    # the masks matter: the same pass with the dropout stream of another seed differs visibly
    inet.seed += 1
    fnet._call = inet._call = 0
    other = exp.model.sample(x0.cuda(), static_condition=forc.cuda())
    assert rel_l2(other["t3_preds"], ref["t3_preds"]) > 1e-3


@pytest.mark.parametrize("weight_seeds", [(4321, 4322), (9001, 9002)])
def test_c3_chain_error_against_a_float64_yardstick(weight_seeds):
    """The north_star tolerance at EVERY lead time of a full-depth pass, and where the rest of the difference comes from.

    One horizon-6 sampling pass at production depth (8 blocks, 180 x 360, E = 256, interpolator dropout and drop path on, the
    Philox stream replayed) three times on identical inputs and masks: the HIP path, the oracle's op sequence in float32 and
    in float64 (the yardstick; both evaluated by torch on the GPU, oracle.sfno.OracleSFNO(device="cuda") -- sixteen chained
    double-precision forwards take ten minutes on the host cores).  Two independent weight draws.  Asserted:
      * relative L2 of the HIP path against the float64 chain < 1e-4 (BASELINE.json north_star) at t1 .. t6, and against the fp32
        oracle < 1e-4 wherever that oracle itself is within 0.6e-4 of the float64 chain (all but t6 of the second draw);
      * the HIP path is no further from the float64 chain than the reference's own fp32 arithmetic is
        (err(HIP, f64) <= 1.1 err(oracle32, f64)): what separates the two fp32 implementations is the chain's amplification of
        one forward's rounding (src/diffusion/dyffusion.py:457-567 chains sixteen forwards), not an error of either."""
    import sdy_amd
    from oracle.sfno import OracleSFNO, make_state_dict

    sf, si = weight_seeds
    fcfg = SFNOConfig(in_chans=C_STATE + C_FORC, out_chans=C_STATE, nlat=NLAT, nlon=NLON, embed_dim=E, num_layers=8,
                      with_time_emb=True, min_time=0.0, max_time=HZ - 1.0)
    icfg = SFNOConfig(in_chans=2 * C_STATE + C_FORC, out_chans=C_STATE, nlat=NLAT, nlon=NLON, embed_dim=E, num_layers=8,
                      with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0, max_time=HZ - 1.0)
    fnet, _, fsd = make_pair(fcfg, C_STATE, C_FORC, seed=sf)
    inet, _, isd = make_pair(icfg, 2 * C_STATE, C_FORC, seed=si, net_seed=1000 + sf)
    exp = sdy_amd.MultiHorizonForecastingDYffusion(fnet, sdy_amd.InterpolationExperiment(inet, horizon=HZ), horizon=HZ)
    g = torch.Generator(device="cpu").manual_seed(sf)
    x0 = torch.randn(1, C_STATE, NLAT, NLON, generator=g)
    forc = torch.randn(1, C_FORC, NLAT, NLON, generator=g)
    got = {k: v.cpu() for k, v in exp.model.sample(x0.cuda(), static_condition=forc.cuda()).items()}

    def oracle_chain(dtype):
        fora, iora = OracleSFNO(fcfg, fsd, dtype=dtype, device="cuda"), OracleSFNO(icfg, isd, dtype=dtype, device="cuda")
        masks = PhiloxMasks(icfg, seed=1000 + sf)
        masks.device = "cuda"
        n = {"i": 0}

        def ora_i(x, time, condition=None, static_condition=None):
            masks.call = n["i"]
            n["i"] += 1
            return iora(x, time=time, condition=condition, static_condition=static_condition, mask_fn=masks)

        o = OracleDYffusion(lambda x, time, condition=None, static_condition=None: fora(
            x, time=time, condition=condition, static_condition=static_condition), ora_i, timesteps=HZ)
        out = {k: v.cpu() for k, v in o.sample(x0.cuda().to(dtype), static_condition=forc.cuda()).items()}
        del fora, iora
        torch.cuda.empty_cache()
        return out

    ref32, ref64 = oracle_chain(torch.float32), oracle_chain(torch.float64)
    keys = [f"t{k}_preds" for k in range(1, HZ + 1)]
    assert sorted(got) == sorted(ref32) == sorted(ref64) == sorted(keys)
    e_hip32 = {k: rel_l2(got[k], ref32[k]) for k in keys}
    e_hip64 = {k: rel_l2(got[k], ref64[k]) for k in keys}
    e_ref = {k: rel_l2(ref32[k], ref64[k]) for k in keys}
    report = {k: (f"{e_hip32[k]:.2e}", f"{e_hip64[k]:.2e}", f"{e_ref[k]:.2e}") for k in keys}   # (HIP|o32, HIP|f64, o32|f64)
    print("chain errors (HIP vs fp32 oracle, HIP vs float64, fp32 oracle vs float64):", report)
    for k in keys:
        assert torch.isfinite(got[k]).all()
        # the HIP path against the exact chain: inside the bound, and no further out than the reference's own arithmetic
        assert e_hip64[k] < 1e-4, f"{k}: HIP path {e_hip64[k]:.3e} from the float64 chain (bound 1e-4); {report}"
        assert e_hip64[k] <= 1.1 * e_ref[k], f"{k}: HIP path {e_hip64[k]:.3e} from the float64 chain, the fp32 oracle {e_ref[k]:.3e}"
        # against the fp32 oracle itself: 1e-4 wherever that oracle is within 0.6e-4 of the exact chain (two fp32 implementations
        # each e from the truth can be 2 e apart; measured: 1.2 e) -- t1 .. t5 of both draws and t6 of the first; beyond that, no
        # further apart than their two distances
        if e_ref[k] <= 0.6e-4:
            assert e_hip32[k] < 1e-4, f"{k}: rel L2 {e_hip32[k]:.3e} against the fp32 oracle (north_star bound 1e-4); {report}"
        assert e_hip32[k] <= 1.05 * (e_hip64[k] + e_ref[k]), report
    assert e_hip32["t5_preds"] < 1e-4 and e_ref["t6_preds"] > 3 * e_ref["t1_preds"], report   # the chain amplifies rounding


def test_b25_full_depth_batch_rows_are_independent_trajectories():
    """The benchmark's workload (25 members, 8 blocks, full width): finite, 16 network calls, and row b of the batch is
    the trajectory a process would compute ALONE with batch_offset = b (what member sharding over GPUs relies on)."""
    exp, fnet, inet, *_ = _build(layers=8)
    B = 25
    g = torch.Generator(device="cpu").manual_seed(99)
    x0 = torch.randn(1, C_STATE, NLAT, NLON, generator=g).expand(B, -1, -1, -1).contiguous().cuda()
    forc = torch.randn(1, C_FORC, NLAT, NLON, generator=g).expand(B, -1, -1, -1).contiguous().cuda()
    exp.set_batch_offset(0)
    full = exp.model.sample(x0, static_condition=forc)
    assert (fnet._call, inet._call) == (6, 10)
    for k in range(1, HZ + 1):
        assert torch.isfinite(full[f"t{k}_preds"]).all()
    t6 = full["t6_preds"]
    assert float((t6[0] - t6[1]).abs().max()) > 1e-3          # members diverge: per-trajectory dropout streams
    for b in (0, 7, 24):
        fnet._call = inet._call = 0
        exp.set_batch_offset(b)
        alone = exp.model.sample(x0[b:b + 1], static_condition=forc[b:b + 1])
        for k in (1, 6):
            a, f = alone[f"t{k}_preds"][0], full[f"t{k}_preds"][b]
            e = rel_l2(a, f)
            assert e < 2e-6, f"row {b}, t{k}: batch row vs lone trajectory rel L2 {e:.3e} (bitwise: {torch.equal(a, f)})"


def test_c4_rollout_25_members_through_the_window_driver():
    """BASELINE.json configs[3] in short: 25 members of one initial condition, full grid / width / depth, two windows of 6
    steps through run_inference -> stepper -> sampler -> SFNO with the device time-mean aggregator (tools/c4_rollout.py runs
    the 100-step version).  No oracle at this size: finite, shapes, the driver's own throughput timer, members diverge."""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c4_rollout

    r = c4_rollout.run(torch.device("cuda", 0), steps=12, members=25)
    assert r["finite"] and r["windows"] == 2
    assert r["member_forecast_steps_per_s_device_time"] >= r["member_forecast_steps_per_s"]      # wall includes the loader
    assert r["prediction_shape"] == (25, 1, 6, NLAT, NLON)          # second window: initial time dropped, members stacked
    assert r["member_forecast_steps_per_s"] > 10.0
    assert r["time_mean_rmse_channel_mean"] > 0.0


def test_c5_shape_4_ics_x_25_members_whole_job_and_8_gpu_shares():
    """BASELINE.json configs[4] in one window: 4 initial conditions x 25 members at full grid / width / depth through
    run_inference -> stepper -> sampler (`src/configs/inference/ckpts_from_huggingface_10years.yaml`; the reference shards
    initial conditions over ranks, `src/ace_inference/core/data_loading/inference.py:110-113`).  Once as the whole job on one
    GPU (B = 100, also as two device batches of 50: `max_batch`), then as the shares `ensemble.shard(4, 25, rank, 8)` gives
    ranks 0, 3 and 7 of an 8-GPU node: 13 rows of IC 0; 13 rows that straddle ICs 1 and 2 (the cut inside an IC's members);
    the last 12 rows of IC 3.  A share's rows must equal the whole job's rows (2e-5: masks are bit-identical, InstanceNorm sums
    depend on the batch), everything finite, aggregator logs present."""
    import os
    import sys

    import sdy_amd
    from sdy_amd import ensemble

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import c4_rollout

    dev = torch.device("cuda", 0)
    n_ics, members, window = 4, 25, 6
    exp, stepper, names, out_names = c4_rollout.build(dev)
    wins = list(c4_rollout.windows(names, 1, window, NLAT, NLON, seed=77, n_ics=n_ics))
    keep = [out_names[0], out_names[-1]]
    w = sdy_amd.metrics.spherical_area_weights(torch.linspace(-89.5, 89.5, NLAT), NLON)

    def run(**kw):
        exp.model.model._call = 0
        exp.model.interpolator.model._call = 0
        got = {}

        class W:
            def append_batch(self, target, prediction, start_timestep, start_sample, batch_times=None):
                got["start"] = start_sample
                got["pred"] = {k: prediction[k].clone() for k in keep}
                got["target_rows"] = next(iter(target.values())).shape[0]
                got["finite"] = all(bool(torch.isfinite(v).all()) for v in prediction.values())

        agg = sdy_amd.metrics.TimeMeanAggregator(w, is_ensemble=True)
        data = wins
        if "unit_range" in kw:      # a rank's loader holds only the initial conditions its share touches
            lo, n = kw.pop("ics")
            data = [type(wins[0])(data={k: v[lo:lo + n] for k, v in wins[0].data.items()}, times=None)]
            kw["trajectory_offset"] = lo
        timers = sdy_amd.run_inference(agg, stepper, data, window, window, n_ensemble_members=members, eval_device=dev,
                                       writer=W(), **kw)
        got["logs"] = agg.get_logs("inference")
        got["timers"] = timers
        return got

    whole = run()
    assert whole["finite"] and whole["pred"][keep[0]].shape == (members, n_ics, window + 1, NLAT, NLON)
    assert whole["timers"]["forecast_steps_per_second"] > 10.0 and whole["timers"]["wall"] >= whole["timers"]["run_on_batch"] * 0.5
    assert len(whole["logs"]) == 2 * len(out_names) + 1 and all(v == v for v in whole["logs"].values())
    v = whole["pred"][keep[0]]
    assert float((v[0, 0, -1] - v[1, 0, -1]).abs().max()) > 1e-3          # members of an IC diverge
    assert float((v[0, 0, -1] - v[0, 1, -1]).abs().max()) > 1e-3          # ICs differ

    halves = run(max_batch=50)                                              # the same job as two device batches
    for k in keep:
        e = rel_l2(halves["pred"][k], whole["pred"][k])
        assert e < 2e-5, f"max_batch=50 vs one batch, {k}: {e:.3e}"
    assert abs(halves["logs"]["inference/rmse/channel_mean"] - whole["logs"]["inference/rmse/channel_mean"]) < 1e-4

    for rank, want in ((0, (0, 13, 0, 1)), (3, (39, 13, 1, 2)), (7, (88, 12, 3, 1))):
        start, cnt, ic_lo, n_ic = ensemble.shard(n_ics, members, rank, 8)
        assert (start, cnt, ic_lo, n_ic) == want
        part = run(unit_range=(start, cnt), ics=(ic_lo, n_ic))
        assert part["finite"] and part["start"] == start and part["target_rows"] == n_ic
        assert all(v == v for v in part["logs"].values()) and len(part["logs"]) == 2 * len(out_names) + 1
        for k in keep:
            assert part["pred"][k].shape == (cnt, window + 1, NLAT, NLON)      # a ragged share: flat rows
            for r in (0, cnt // 2, cnt - 1):
                ic, m = divmod(start + r, members)
                e = rel_l2(part["pred"][k][r], whole["pred"][k][m, ic])
                assert e < 2e-5, f"rank {rank} of 8, {k}, unit {start + r} = (IC {ic}, member {m}): {e:.3e}"


def test_bench_two_ranks_as_the_driver_launches_it():
    """The driver's SCALE step runs `bench.py --gpus N`; on a 1-GPU box the same code path runs with both ranks on GPU 0
    (`--share-gpu`, gloo): a fresh subprocess (the parent must start its children before anything touches HIP), rank 0's JSON
    line.  Five members over two ranks: two resident members each and ONE relayed between the ranks in time slices
    (ensemble.relay_plan, the 25-over-8 schedule in miniature; real send / recv of its state); `--no-relay` gives the static
    3 + 2 split.  Every trajectory's final state must be the one-process job's, whatever the schedule."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(*extra):
        cmd = [sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "0", "--members", "5",
               "--no-cpu-baseline", "--no-extras", "--digests", *extra]
        p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, p.stdout[-2000:]
        return json.loads(lines[0])

    one = run("--gpus", "1")
    two = run("--gpus", "2", "--share-gpu")
    static = run("--gpus", "2", "--share-gpu", "--no-relay")
    assert two["n_gpus"] == 2 and two["config"]["members_per_gpu"] == [2, 2] and two["config"]["relayed_members"] == 1
    assert static["config"]["members_per_gpu"] == [3, 2] and static["config"]["relayed_members"] == 0
    for res in (two, static):
        assert res["scaling"] == "strong" and res["value"] > 0 and res["steps"] == 2
        assert res["config"]["forecast_steps_per_step"] == 5 * 6
        assert sorted(res["trajectory_digests"]) == sorted(one["trajectory_digests"]) == [str(u) for u in range(5)]
        # Three chained full-depth passes amplify a rounding-level difference between batch compositions by ~76 per pass
        # (test_c3_chain_error...), so states cannot be compared value by value here (tests/test_gpu_dyffusion.py::
        # test_relayed_remainder_member... does that on a well-conditioned network, tests/test_distributed_cpu.py bit for bit
        # with real send / recv); the norm of a trajectory's state still identifies its sample: the same member agrees to
        # ~1e-3 across schedules, different members differ by 1e-2 and more.
        for u, (s1, n1) in one["trajectory_digests"].items():
            s2, n2 = res["trajectory_digests"][u]
            assert abs(n2 - n1) <= 2.5e-3 * n1, (u, n1, n2)
    norms = sorted(v[1] for v in one["trajectory_digests"].values())
    assert min(b - a for a, b in zip(norms, norms[1:])) > 5e-3 * norms[0], norms


def test_rccl_path_when_two_devices_are_visible():
    """`bench.py --gpus 2` over RCCL proper (init_process_group("nccl"), barriers, the max-over-ranks all_reduce and the relay's
    send / recv between two GPUs).  The GPU boxes of the development pool have ONE device, where two ranks cannot share a
    communicator (RCCL refuses duplicate devices) and the N > 1 path runs over gloo (`--share-gpu`, above); this test runs
    wherever at least two devices are visible, so that the first multi-GPU run of a round is not the first RCCL run."""
    import json
    import os
    import subprocess
    import sys

    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: RCCL needs a device per rank")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0", "--members", "5",
           "--no-cpu-baseline", "--no-extras", "--digests"]
    p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert res["n_gpus"] == 2 and res["config"]["relayed_members"] == 1 and len(res["trajectory_digests"]) == 5
    assert "TEST MODE" not in res["data"]


def test_c4_rollout_two_and_three_ranks_shared_gpu():
    """`tools/c4_rollout.py --gpus N --share-gpu`: the C4 / C5 runner's own N > 1 path as fresh self-launching subprocesses on a
    small network -- `run_inference(relay=ensemble.relay_plan(...))` with the hand-overs over torch.distributed (gloo here,
    RCCL on N GPUs): 5 members over 2 ranks = 2 resident each + one relayed, 7 over 3 = 2 each + one relayed through three
    hosts; `--no-relay` = `run_inference(unit_range=shard)`, the static 3 + 2 split.  The ranks' time-mean maps, summed over
    ranks with their row counts, reproduce the one-process statistics whatever the schedule."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def run(steps, members, *extra):
        cmd = [sys.executable, os.path.join(root, "tools", "c4_rollout.py"), "--steps", str(steps), "--members", str(members),
               "--layers", "2", "--embed", "16", "--grid", "32", "64", *extra]
        p = subprocess.run(cmd, cwd=root, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        return json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])

    r1 = run(12, 5)
    r2 = run(12, 5, "--gpus", "2", "--share-gpu")
    rs = run(12, 5, "--gpus", "2", "--share-gpu", "--no-relay")
    assert r1["finite"] and r2["finite"] and rs["finite"] and r1["rows"] == 5
    assert r2["n_gpus"] == 2 and r2["rows"] == 2 and r2["relayed_windows"] == 1          # rank 0: 2 members + window 0 of member 4
    assert rs["rows"] == 3 and rs["relayed_windows"] == 0
    for r in (r2, rs):
        assert abs(r1["time_mean_rmse_channel_mean"] - r["time_mean_rmse_channel_mean"]) < 2e-4 * r1["time_mean_rmse_channel_mean"]
        assert r["job_member_forecast_steps_per_s"] > 0
    r7 = run(18, 7)
    r3 = run(18, 7, "--gpus", "3", "--share-gpu")
    assert r3["n_gpus"] == 3 and r3["rows"] == 2 and r3["relayed_windows"] == 1 and r3["finite"]
    assert abs(r7["time_mean_rmse_channel_mean"] - r3["time_mean_rmse_channel_mean"]) < 2e-4 * r7["time_mean_rmse_channel_mean"]
