#!/usr/bin/env python3
"""Benchmark of the hot path: 25-member ensemble DYffusion sampling on the 180x360 grid.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one horizon-6 sampling pass (6 forecaster + 10 interpolator SFNO forwards, interpolator dropout stream
on, cold-sampling updates, autoregressive feedback x0 <- t6) for the 25 ensemble members this rank owns, i.e.
25 x 6 = 150 member-forecast-steps per rank per step.  Ranks own independent initial conditions (the reference
shards ICs over ranks: src/ace_inference/core/data_loading/inference.py:110-113); no collective on the data path.

Prints ONE JSON line on rank 0 (see the keys below).  `roofline` is measured live with HIP events on the stream the
kernels run on; `cpu_baseline` times the CPU oracle (same op sequence as the reference's PyTorch path) on the host
cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

MEMBERS = 25
HORIZON = 6
STATE_CH = 63          # BASELINE.json metric: "180x360, 63ch" (nominal; the shipped YAML has 34 prognostic channels)
FORCING_CH = 2
NLAT, NLON = 180, 360
EMBED, LAYERS = 256, 8
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: BF16/FP16 MFMA dense peak
MLP_TRAFFIC_B25 = 5.656e9   # bytes per launch, PMC passes at B = 25 (profiles/r1f/pmc_traffic_mlp.txt)
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E peak (6.3 TB/s measured with a float4 copy)


def build_models(device, rank):
    import torch

    import sdy_amd
    from helpers import make_pair
    from oracle.sfno import SFNOConfig

    fcfg = SFNOConfig(in_chans=STATE_CH + FORCING_CH, out_chans=STATE_CH, nlat=NLAT, nlon=NLON, embed_dim=EMBED,
                      num_layers=LAYERS, with_time_emb=True, min_time=0.0, max_time=HORIZON - 1.0)
    icfg = SFNOConfig(in_chans=2 * STATE_CH + FORCING_CH, out_chans=STATE_CH, nlat=NLAT, nlon=NLON, embed_dim=EMBED,
                      num_layers=LAYERS, with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0,
                      max_time=HORIZON - 1.0)
    with torch.cuda.device(device):
        fnet, fora, _ = make_pair(fcfg, STATE_CH, FORCING_CH, seed=4321)
        inet, _, _ = make_pair(icfg, 2 * STATE_CH, FORCING_CH, seed=4322, net_seed=1000)
    inet.batch_offset = rank * MEMBERS          # global trajectory index: results do not depend on the sharding
    exp = sdy_amd.MultiHorizonForecastingDYffusion(fnet, sdy_amd.InterpolationExperiment(inet, horizon=HORIZON),
                                                   horizon=HORIZON)
    return exp, fora, fcfg


def one_pass(exp, x0, forcings):
    """The stepper's per-window call pattern (stepper_multistep.py:365-427): horizon 1 runs the sampler, 2..6 read the cache;
    returns the new autoregressive state (t6)."""
    out = None
    for h in range(1, HORIZON + 1):
        batch = {"dynamics": x0, "static_condition": forcings}
        with exp.ema_scope(), exp.inference_dropout_scope():
            out = exp.get_preds_at_t_for_batch(batch, horizon=h, split="predict", is_autoregressive=False,
                                               prepare_inputs=False, ensemble=False, num_predictions=1)
    return out[f"t{HORIZON}_preds_normed"]


def roofline_probe(device, B, reps=5):
    """Dominant kernel, timed with HIP events on the launch stream.
    gemm_mode h3 (default): the fused MLP kernel `mlp_h3_kernel` (fc1 256->512 + GELU + fc2 512->256 in one launch).
    gemm_mode f32: the fp32-MFMA fc1 GEMM.  Algorithmic flops / bytes per launch: DESIGN.md 'Kernels'."""
    import torch

    import sdy_amd

    HW, hid = NLAT * NLON, 2 * EMBED
    x = torch.randn(B, EMBED, NLAT, NLON, device=device)
    w = torch.randn(hid, EMBED, device=device) / 16.0
    bias = torch.randn(hid, device=device) * 0.1
    pa = torch.rand(B, EMBED, device=device) + 0.5
    pd = torch.randn(B, EMBED, device=device) * 0.1
    h3 = os.environ.get("SDY_GEMM_MODE", "h3") == "h3"
    if h3:
        w2 = torch.randn(EMBED, hid, device=device) / 22.0
        b2 = torch.randn(EMBED, device=device) * 0.1
        res = torch.randn(B, EMBED, NLAT, NLON, device=device)
        out = torch.empty_like(x)
        prep = sdy_amd.ops.pack_mlp_h3(w, w2, device)
        run = lambda: sdy_amd.ops.mlp_fused(x, w, bias, w2, b2, pre_affine=(pa, pd), add=res, out=out, prepared=prep)
    else:
        wt = w.t().contiguous()
        out = torch.empty(B, hid, NLAT, NLON, device=device)
        kw = dict(pre_affine=(pa, pd), gelu=True, kernel_tag=1, out=out, wt_prepared=wt)
        run = lambda: sdy_amd.ops.conv1x1(x, w, bias, **kw)
    run()
    torch.cuda.synchronize(device)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    if not h3:
        flops = 2.0 * EMBED * hid * HW * B
        achieved = flops / (ms * 1e-3) / 1e12
        return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": None,
                "kernel": "gemm_f32_kernel<2,2,false,false,1>, MLP fc1 256->512, B=%d" % B,
                "ms_per_launch": round(ms, 4), "flops_per_launch": flops}
    # Fused MLP: algorithmic flops 2 * 2*E*hid per pixel, algorithmic bytes 3 tensors of B*E*HW fp32 (x, residual, out):
    # 170 flop/B.  Split precision issues 3 f16 MFMA passes, so the machine balance is (2500/3 TF) / 8 TB/s = 104 flop/B:
    # the kernel is matrix-bound.  `achieved` counts the algorithmic flops ONCE against the dense f16 peak (the 3-pass
    # ceiling is peak / 3).  `traffic`: FETCH_SIZE*2 + WRITE_SIZE from separate rocprofv3 PMC passes at B = 25
    # (profiles/r1f/pmc_traffic_mlp.txt).
    flops = 4.0 * EMBED * hid * HW * B
    alg_bytes = 3.0 * B * EMBED * HW * 4
    achieved = flops / (ms * 1e-3) / 1e12
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F16_MFMA_TFLOPS, 4), "traffic": MLP_TRAFFIC_B25 if B == 25 else None,
            "kernel": "mlp_h3_kernel<false> (fused MLP 256->512->256, 3-pass split-f16 MFMA), B=%d" % B,
            "ms_per_launch": round(ms, 4), "flops_per_launch": flops, "algorithmic_bytes_per_launch": alg_bytes,
            "three_pass_ceiling_frac": round(3.0 * achieved / PEAK_F16_MFMA_TFLOPS, 4),
            "hbm_view": {"achieved_gbs": round(alg_bytes / (ms * 1e-3) / 1e9, 1), "peak_gbs": PEAK_HBM_GBS}}


def cpu_baseline(fora, fcfg):
    """CPU oracle (same torch op sequence as the reference's CPU PyTorch path) on the host cores: ONE forecaster forward at
    B=1, scaled by 16 forwards per 6 forecast steps."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(1, STATE_CH, NLAT, NLON, generator=g)
    c = torch.randn(1, FORCING_CH, NLAT, NLON, generator=g)
    t = torch.tensor([2.0])
    t0 = time.perf_counter()
    fora(x, time=t, condition=c)
    dt = time.perf_counter() - t0
    steps_per_s = HORIZON / (16.0 * dt)
    return {"value": round(steps_per_s, 5), "unit": "member-forecast-steps/s", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "1 oracle SFNO forward (B=1, 180x360, E=256, 8 layers, fp32): %.2f s; "
            "16 forwards per 6 forecast steps" % dt}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--members", type=int, default=MEMBERS, help="ensemble members per GPU (default 25)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE={world})")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)

    B = args.members
    exp, fora, fcfg = build_models(device, rank)
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)      # one initial condition per rank
    x_ic = torch.randn(1, STATE_CH, NLAT, NLON, generator=g)
    x0 = x_ic.expand(B, -1, -1, -1).contiguous().to(device)          # 25 members start from the same IC
    forc = torch.randn(1, FORCING_CH, NLAT, NLON, generator=g).expand(B, -1, -1, -1).contiguous().to(device)

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    x = x0
    for _ in range(args.warmup):
        x = one_pass(exp, x, forc)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        x = one_pass(exp, x, forc)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert torch.isfinite(x).all(), "non-finite state after the rollout"

    if rank == 0:
        total_steps = world * B * HORIZON * args.steps
        res = {
            "metric": "forecast-steps/sec (180x360, 63ch), 25-member ensemble DYffusion sampling, whole job",
            "value": round(total_steps / dt, 3),
            "unit": "member-forecast-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32 (1x1-conv GEMMs as 3-pass split-f16 MFMA with f32 accumulation)" if os.environ.get("SDY_GEMM_MODE", "h3") == "h3" else "f32",
            "data": "synthetic",
            "config": {
                "workload": "25-member ensemble x horizon-6 DYffusion sampling pass (6 forecaster + 10 interpolator SFNO "
                            "forwards, dropout stream on, AR feedback), 180x360, %d state + %d forcing channels, "
                            "embed 256, 8 blocks; one initial condition per GPU" % (STATE_CH, FORCING_CH),
                "members_per_gpu": B, "horizon": HORIZON, "forwards_per_step": 16,
                "per_gpu_forecast_steps_per_s": round(total_steps / dt / world, 3),
                "parallelism": "ICs sharded over GPUs, no data-path collective",
            },
        }
        res["roofline"] = roofline_probe(device, B)
        if world == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(fora, fcfg)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
