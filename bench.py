#!/usr/bin/env python3
"""Benchmark of the hot path: 25-member ensemble DYffusion sampling on the 180x360 grid (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W [--scaling strong|weak]

works by itself for any N (the parent starts one child process per GPU BEFORE anything touches HIP and relays rank 0's
JSON line), and also under an external launcher:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one horizon-6 sampling pass (6 forecaster + 10 interpolator SFNO forwards, interpolator dropout stream
on, cold-sampling updates, autoregressive feedback x0 <- t6) of the whole job.

  --scaling strong (default; the BASELINE metric "25-member ensemble at 1/2/4/8 GPU"): ONE initial condition, 25 members,
      every member keyed by its global index (`batch_offset`) and every pass by its window (dropout call counters), 150
      member-forecast-steps per step in total.  Each rank keeps 25 // N resident members; the 25 % N remainder members are
      RELAYED (`ensemble.relay_plan`): each advances as a batch of one, hosted by one rank per slice of the timed windows and
      handed to the next with a send / recv of its state (N = 8: 3 members per GPU + member 24 visiting every GPU for 1/8 of
      the windows, instead of one GPU with 4 members setting the pace).  `--no-relay`: the static split of
      `ensemble.partition` (N = 8: 4,3,3,3,3,3,3,3).  At N > 1 a weak-scaling leg (25 members of its own
      initial condition on every rank, the reference's IC sharding: src/ace_inference/core/data_loading/inference.py:110-113)
      is timed as well and reported under config.weak_scaling.
  --scaling weak: only that leg; value = N * 150 member-forecast-steps per step.
There is no collective on the data path; RCCL serves the barriers, the max-over-ranks time and the relay's hand-overs.

Rank 0 prints ONE JSON line.  Beyond the contract keys:
  roofline      dominant kernel (`mlp_h3_kernel<true>`, the fused MLP with the Philox dropout of the interpolator), timed
                IN THE NETWORK with HIP events on the launch stream (sdy_profile_*), plus `kernels`: the same measurement
                for every stage of the forward with its algorithmic flops / bytes and both roofline fractions.
  cpu_baseline  the CPU oracle (same torch op sequence as the reference's CPU path) on the host cores, bounded sample.
  latency       B = 1: one interpolator forward (BASELINE config C2) and one horizon-6 pass (C3).
  f32_fallback  the same 25-member pass with gemm_mode="f32" (the escape hatch when the fp16-range flag fires).
  c5_extrapolation  wall time of the 10-year job (4 ICs x 25 members x 14600 steps) at the measured rate.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MEMBERS = 25
HORIZON = 6
STATE_CH = 63          # BASELINE.json metric: "180x360, 63ch" (nominal; the shipped YAML has 34 prognostic channels)
FORCING_CH = 2
NLAT, NLON = 180, 360
EMBED, LAYERS = 256, 8
HIDDEN = 2 * EMBED
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: BF16/FP16 MFMA dense peak
PEAK_HBM_GBS = 8000.0          # same guide: HBM3E peak (6.3 TB/s measured with a float4 copy)
POLAR_LIVE = 0.77              # share of (order, latitude) pairs the polar cut-off keeps
NZ_PAIRS, ALL_PAIRS = 16290, 32580   # (l, m) pairs with m <= l / dense


# ---- self-launch ---------------------------------------------------------------------------------------------------------
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_children(n: int) -> int:
    """One child per GPU, started before this process imports torch or touches HIP.  Children inherit stdout (only rank 0
    prints).  Returns the worst exit code."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("SDY_BENCH_CU_SPLIT"):     # EXPERIMENT (--share-gpu): disjoint CU sets for the ranks of one GPU
            n_cu, per = 256, 256 // n
            env["HSA_CU_MASK"] = "0:%d-%d" % (r * per, (r + 1) * per - 1)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for p in list(pending):
            r = p.poll()
            if r is None:
                continue
            pending.remove(p)
            if r != 0:
                rc = rc or r
                for q in pending:       # a rank failed: the others would wait in a barrier forever
                    q.terminate()
        time.sleep(0.05)
    return rc


# ---- models / one step -------------------------------------------------------------------------------------------------
def build_models(device):
    """Forecaster + interpolator + sampler with trained-like random weights (`sdy_amd.synthetic`: no checkpoint can be
    fetched here).  Product code only: nothing under `oracle/` or `tests/` is imported for the timed path."""
    from sdy_amd import synthetic

    exp, _, _ = synthetic.build_sampler(device, state_chans=STATE_CH, forcing_chans=FORCING_CH, nlat=NLAT, nlon=NLON,
                                        embed=EMBED, layers=LAYERS, horizon=HORIZON)
    return exp


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


def synthetic_state(ic_index, B, device):
    """B members starting from initial condition `ic_index` (fields and forcings N(0, 1): the data are standardised)."""
    import torch

    g = torch.Generator(device="cpu").manual_seed(1234 + ic_index)
    x = torch.randn(1, STATE_CH, NLAT, NLON, generator=g).expand(B, -1, -1, -1).contiguous().to(device)
    f = torch.randn(1, FORCING_CH, NLAT, NLON, generator=g).expand(B, -1, -1, -1).contiguous().to(device)
    return x, f


# ---- roofline ------------------------------------------------------------------------------------------------------------
def profile_round_key(path):
    """Order of a directory under profiles/: (round number, letters) of its name -- "r10a" follows "r9z" ("r10a" < "r5d" as
    strings), and file times say nothing after a checkout."""
    import re

    m = re.match(r"r(\d+)([a-z]*)$", os.path.basename(os.path.dirname(path)))
    return (int(m.group(1)), m.group(2)) if m else (-1, "")


def measured_traffic():
    """HBM bytes per launch of the dominant kernel, `mlp_h3_kernel<true>` at 25 rows, from the NEWEST counter summary under
    profiles/ (profiles/<round>/pmc_summary.txt: separate rocprofv3 --pmc passes over an interpolator forward with every launch
    at 25 rows, FETCH_SIZE x 2 + WRITE_SIZE with the guide's gfx950 correction; tools/profile_round.sh, tools/pmc_summary.py).
    Counters cannot be collected inside a timed run, so the bench line cites the file it read -- and refuses one that is OLDER
    than the newest kernel_stats.csv (a round that re-profiled its kernels but forgot the counter passes must not keep citing the
    previous round's traffic).  Returns (bytes, path, None) or (None, None, reason)."""
    import glob
    import re

    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_summary.txt")), key=profile_round_key)
    stats = sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "kernel_stats.csv")), key=profile_round_key)
    if not pmc:
        return None, None, "no profiles/*/pmc_summary.txt"
    path = pmc[-1]
    if stats and profile_round_key(stats[-1]) > profile_round_key(path):
        return None, None, ("stale: the newest counter summary (%s) is older than the newest kernel trace (%s)"
                            % (os.path.relpath(path, ROOT), os.path.relpath(stats[-1], ROOT)))
    for line in open(path):
        if "mlp_h3_kernel<true, false>" in line:
            m = re.search(r"\s([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+[0-9.]+%", line)
            if m:
                return float(m.group(3)) * 1e9, os.path.relpath(path, ROOT), None
    return None, None, "no mlp_h3_kernel<true, false> row in %s" % os.path.relpath(path, ROOT)


def stage_work(B):
    """Algorithmic work per LAUNCH of every stage of one SFNO forward at batch B: (flops, HBM bytes, bound).
    Flops are the useful count (m > l zeros skipped, no credit for the three split-f16 passes or the parity fold);
    bytes are the tensors a launch must read and write once (fp32), weights once per launch."""
    HW, E, Hd = NLAT * NLON, EMBED, HIDDEN
    act = 4.0 * E * HW * B                                  # one (B, 256, 180, 360) fp32 tensor
    xf = POLAR_LIVE * 4.0 * 2 * E * NLAT * NLAT * B         # grid-frequency tensor, m < 180, polar cut-off
    cs = 4.0 * 2 * E * NZ_PAIRS * B                         # triangular coefficient tensor
    leg_f = 4.0 * NZ_PAIRS * NLAT * E * B
    cin_f, cin_i = STATE_CH + FORCING_CH, 2 * STATE_CH + FORCING_CH
    w = {
        "mlp fused (dropout)": (4.0 * E * Hd * HW * B, 3 * act, "mfma"),
        "mlp fused": (4.0 * E * Hd * HW * B, 3 * act, "mfma"),
        "inner-skip conv": (2.0 * E * E * HW * B, 3 * act, "hbm"),
        # first / last block: the skip's matrix is folded into the dhconv weights, what remains is GELU + statistics over y
        "inner skip folded (gelu)": (0.0, 2 * act, "hbm"),
        "legendre analysis": (leg_f, xf + cs, "hbm"),
        "legendre synthesis": (leg_f, xf + cs, "hbm"),
        "rfft (lon)": (0.0, act + xf, "hbm"),
        "irfft (lon)": (0.0, act + xf, "hbm"),
        # (PMC, profiles/r4b/pmc_summary.txt: matrix pipe 52.8 % busy, 41k cycles per tile against 24.6k of MFMA work and 8k of
        #  weight stream from L2 -- issue / matrix bound, not HBM bound)
        "dhconv": (8.0 * E * E * NZ_PAIRS * B, 2 * cs + 4.0 * 2 * E * E * NLAT, "mfma"),
        "encoder.2 conv": (2.0 * E * E * HW * B, 2 * act + 4.0 * E * HW, "hbm"),
        # the two networks differ in their input width: per-launch average over the 6 + 10 forwards of a pass
        "encoder.0 conv": (2.0 * E * HW * B * (6 * cin_f + 10 * cin_i) / 16, act + 4.0 * HW * B * (6 * cin_f + 10 * cin_i) / 16,
                           "hbm"),
        "decoder.0 conv": (2.0 * E * HW * B * (E + (6 * cin_f + 10 * cin_i) / 16),
                           2 * act + 4.0 * HW * B * (6 * cin_f + 10 * cin_i) / 16, "hbm"),
        "decoder.2 conv": (2.0 * E * STATE_CH * HW * B, act + 4.0 * STATE_CH * HW * B, "hbm"),
        # encoder / decoder as one launch each: the 256-channel hidden activation is neither written nor read, which puts
        # both above the three-pass machine balance (2500 / 3 TFLOP/s over 8 TB/s = 104 flop/B): 130 flop/B
        "encoder (fused pair)": (2.0 * E * HW * B * (E + (6 * cin_f + 10 * cin_i) / 16),
                                 act + 4.0 * E * HW + 4.0 * HW * B * (6 * cin_f + 10 * cin_i) / 16, "mfma"),
        "decoder (fused pair)": (2.0 * E * HW * B * (E + STATE_CH + (6 * cin_f + 10 * cin_i) / 16),
                                 act + 4.0 * HW * B * (STATE_CH + (6 * cin_f + 10 * cin_i) / 16), "mfma"),
    }
    return w


def profile_pass(exp, x, forc, B):
    """One sampling pass under the stage timer (HIP events on the launch stream around every kernel launch of the 16
    forwards).  Returns (per-stage list sorted by time, total kernel ms)."""
    import torch

    import sdy_amd

    torch.cuda.synchronize()
    with sdy_amd.ops.stage_timer() as t:
        one_pass(exp, x, forc)
        torch.cuda.synchronize()
    work = stage_work(B)
    total = sum(ms for _, ms in t.stages.values())
    rows = []
    for name, (cnt, ms) in sorted(t.stages.items(), key=lambda kv: -kv[1][1]):
        avg = ms / cnt
        fl, by, bound = work.get(name, (0.0, 0.0, "hbm"))
        # drop-path skip: a block's kernels run on the trajectories its DropPath draw keeps, so a stage's average launch covers
        # fewer than B rows; its algorithmic work is scaled to the rows it actually processed (weights are a rounding error)
        avg_rows = t.rows.get(name, cnt * B) / cnt
        fl, by = fl * avg_rows / B, by * avg_rows / B
        rows.append({"name": name, "launches_per_step": cnt, "avg_rows": round(avg_rows, 2), "ms": round(avg, 4), "share": round(ms / total, 4),
                     "gflop": round(fl / 1e9, 2), "gbytes": round(by / 1e9, 3), "bound": bound,
                     "frac_mfma": round(fl / (avg * 1e-3) / 1e12 / PEAK_F16_MFMA_TFLOPS, 4) if fl else None,
                     "frac_hbm": round(by / (avg * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if by else None})
    return rows, total


def roofline_from(rows, total_ms, B, h3):
    dom = next((r for r in rows if r["name"] == "mlp fused (dropout)"), rows[0])
    fl, by = dom["gflop"] * 1e9, dom["gbytes"] * 1e9
    ms = dom["ms"]
    hbm_ms = sum(r["ms"] * r["launches_per_step"] for r in rows if r["bound"] == "hbm")
    hbm_bytes = sum(r["gbytes"] * 1e9 * r["launches_per_step"] for r in rows if r["bound"] == "hbm")
    peak = PEAK_F16_MFMA_TFLOPS if h3 else PEAK_F32_MFMA_TFLOPS
    achieved = fl / (ms * 1e-3) / 1e12
    traffic25, traffic_src, traffic_why = measured_traffic()
    return {
        "bound": "mfma", "achieved": round(achieved, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4),
        # counter traffic of a 25-row launch, scaled to the rows an average launch of this run covered (drop-path skip)
        "traffic": round(traffic25 * dom.get("avg_rows", B) / 25.0) if traffic25 else None,
        "traffic_source": ("%s (mlp_h3_kernel<true> at 25 rows: %.3f GB), x %.2f / 25 rows per launch here"
                           % (traffic_src, traffic25 / 1e9, dom.get("avg_rows", B))) if traffic25 else traffic_why,
        "kernel": "mlp_h3_kernel<true> = stage '%s' (fused MLP 256->512->256 + GELU + Philox dropout + residual, 3-pass "
                  "split-f16 MFMA), B=%d, timed in the network" % (dom["name"], B),
        "ms_per_launch": ms, "launches_per_step": dom["launches_per_step"],
        "flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
        "three_pass_ceiling_frac": round(3.0 * achieved / peak, 4) if h3 else None,
        "timing": "HIP events on the launch stream around every launch of one sampling pass (sdy_profile_*)",
        "kernel_ms_per_step": round(total_ms, 2),
        "hbm_bound_kernels": {"share_of_step": round(hbm_ms / total_ms, 4),
                              "achieved_gbs": round(hbm_bytes / (hbm_ms * 1e-3) / 1e9, 1), "peak_gbs": PEAK_HBM_GBS,
                              "frac": round(hbm_bytes / (hbm_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)},
        "kernels": rows,
    }


# ---- extras ----------------------------------------------------------------------------------------------------------------
def latency_b1(exp, device):
    """BASELINE C2: one interpolator forward at B = 1 (dropout stream on); C3: one horizon-6 sampling pass at B = 1."""
    import torch

    x, f = synthetic_state(0, 1, device)
    ip = exp.model.interpolator
    inp = torch.cat([x, x], dim=1)
    t = torch.full((1,), 3.0, device=device)

    def fwd():
        with ip.inference_dropout_scope(condition=True):
            return ip.predict_packed(inp, time=t, static_condition=f)["preds"]

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize(device)
        return (time.perf_counter() - t0) / reps * 1e3

    c2 = timed(fwd, 10)
    c3 = timed(lambda: one_pass(exp, x, f), 3)
    # how dense is the stream at B = 1?  kernel time of one pass (HIP events around every launch) against its wall time:
    # a ratio near 1 means there are no launch gaps for a HIP graph to remove
    _, kernel_ms = profile_pass(exp, x, f, 1)
    return {"c2_interpolator_forward_b1_ms": round(c2, 3), "c3_horizon6_pass_b1_ms": round(c3, 2),
            "c3_forecast_steps_per_s_b1": round(HORIZON / (c3 * 1e-3), 2),
            "c3_kernel_ms_b1": round(kernel_ms, 2), "c3_stream_density_b1": round(kernel_ms / c3, 3)}


def cpu_baseline(threads):
    """CPU oracle (same torch op sequence as the reference's CPU PyTorch path) on the host cores: one forecaster and one
    interpolator forward at B = 1 (about 20-30 s of CPU work), scaled to a horizon-6 pass = 6 + 10 forwards.  The only
    place `oracle/` is imported, on rank 0 at N = 1 only; the oracle's weight generator produces the benchmark networks'
    weights (tests/test_host_logic.py holds the two generators together)."""
    import torch

    from oracle.sfno import OracleSFNO, SFNOConfig, make_state_dict

    fcfg = SFNOConfig(in_chans=STATE_CH + FORCING_CH, out_chans=STATE_CH, nlat=NLAT, nlon=NLON, embed_dim=EMBED,
                      num_layers=LAYERS, with_time_emb=True, min_time=0.0, max_time=HORIZON - 1.0)
    icfg = SFNOConfig(in_chans=2 * STATE_CH + FORCING_CH, out_chans=STATE_CH, nlat=NLAT, nlon=NLON, embed_dim=EMBED,
                      num_layers=LAYERS, with_time_emb=True, dropout_mlp=0.1, drop_path_rate=0.1, min_time=1.0,
                      max_time=HORIZON - 1.0)
    fora, iora = OracleSFNO(fcfg, make_state_dict(fcfg, seed=4321)), OracleSFNO(icfg, make_state_dict(icfg, seed=4322))
    torch.set_num_threads(threads)
    g = torch.Generator(device="cpu").manual_seed(1234)
    x = torch.randn(1, STATE_CH, NLAT, NLON, generator=g)
    c = torch.randn(1, FORCING_CH, NLAT, NLON, generator=g)
    def timed(fn):      # one untimed forward first (thread pool start-up, first-touch of 0.8 GB of weights and tables), then two
        fn()
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return min(ts), max(ts)

    tf, tf_hi = timed(lambda: fora(x, time=torch.tensor([2.0]), condition=c))
    ti, ti_hi = timed(lambda: iora(torch.cat([x, x], 1), time=torch.tensor([3.0]), condition=c))
    steps_per_s = HORIZON / (6 * tf + 10 * ti)
    return {"value": round(steps_per_s, 5), "unit": "member-forecast-steps/s", "cores": threads, "kind": "port",
            "spread": round((6 * tf_hi + 10 * ti_hi) / (6 * tf + 10 * ti) - 1.0, 3),
            "sample": "oracle SFNO forwards at B=1 (180x360, E=256, 8 layers, fp32, torch.set_num_threads(%d) of %d "
                      "host CPUs), one warm-up forward each, then the faster of two: forecaster %.2f s (slower: %.2f), "
                      "interpolator %.2f s (%.2f); a horizon-6 pass = 6 + 10 forwards"
                      % (threads, os.cpu_count() or 0, tf, tf_hi, ti, ti_hi)}


def f32_fallback(device, B, steps=1):
    """The documented escape hatch of the split-fp16 path (`SDY_GEMM_MODE=f32`, DESIGN.md section 5: fp32-input MFMA tile
    GEMMs, no fp16 range limit, none of the fused fragment-stream kernels): the same sampling pass in that mode, so the
    fallback has a measured price."""
    import torch

    from sdy_amd import synthetic

    exp, _, _ = synthetic.build_sampler(device, state_chans=STATE_CH, forcing_chans=FORCING_CH, nlat=NLAT, nlon=NLON,
                                        embed=EMBED, layers=LAYERS, horizon=HORIZON, gemm_mode="f32")
    exp.set_batch_offset(0)
    x, forc = synthetic_state(0, B, device)
    x = one_pass(exp, x, forc)                       # warm-up: weight upload, workspace
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(steps):
        x = one_pass(exp, x, forc)
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    assert torch.isfinite(x).all()
    return {"gemm_mode": "f32", "members": B, "ms_per_step": round(dt * 1e3, 2),
            "value": round(B * HORIZON / dt, 3), "unit": "member-forecast-steps/s",
            "note": "SDY_GEMM_MODE=f32: v_mfma_f32_32x32x2_f32 tile GEMMs, hidden activation through HBM; timed passes: %d" % steps}


# ---- main ------------------------------------------------------------------------------------------------------------------
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10,
                    help="timed sampling passes (6 forecast steps each); use >= 100 (600 steps) for a C5-style long run")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scaling", choices=("strong", "weak"), default="strong")
    ap.add_argument("--members", type=int, default=MEMBERS, help="ensemble members of the job (strong) / per GPU (weak)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: min(32, host CPUs))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the stage profile, B=1 latencies and the weak leg")
    ap.add_argument("--no-relay", action="store_true",
                    help="strong scaling with the static split only (25 over 8 = 4,3,3,...), no relayed remainder members")
    ap.add_argument("--digests", action="store_true", help="add (sum, norm) of every trajectory's final state to the JSON line")
    ap.add_argument("--share-gpu", action="store_true",
                    help="TEST ONLY: all ranks on GPU 0 with the gloo backend (exercises the N>1 path on a 1-GPU box; "
                         "the numbers mean nothing)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Stand-alone multi-GPU run: become the launcher.  Nothing in this process has touched HIP (torch is not even
        # imported), and the children are fresh processes, not re-executions of this one.
        sys.exit(launch_children(args.gpus))

    import torch
    import torch.distributed as dist

    import sdy_amd
    from sdy_amd import ensemble

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus}, or run "
                         f"`python bench.py --gpus {args.gpus}` without a launcher")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)      # "nccl" is RCCL on ROCm
    red_dev = torch.device("cpu") if args.share_gpu else device

    exp = build_models(device)
    h3 = sdy_amd._lib.default_gemm_mode() == "h3"

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    digests = {}
    relay_info = {}

    def digest(first_unit, x):
        """(sum, L2 norm) in float64 of every trajectory's final state: what `--digests` prints, so that runs with different
        world sizes / schedules can be compared trajectory by trajectory (tests/test_gpu_fullsize.py)."""
        for r in range(x.shape[0]):
            v = x[r].double()
            digests[first_unit + r] = [float(v.sum()), float(v.norm())]

    def timed_leg(ic_index, first_unit, B):
        """W warm-up + K timed sampling passes of this rank's B trajectories; returns the max-over-ranks seconds."""
        exp.set_batch_offset(first_unit)
        if B > 0:
            x, forc = synthetic_state(ic_index, B, device)
        for _ in range(args.warmup):
            if B > 0:
                x = one_pass(exp, x, forc)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            if B > 0:
                x = one_pass(exp, x, forc)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if B > 0:
            assert torch.isfinite(x).all(), "non-finite state after the rollout"
            if ic_index == 0 and args.scaling == "strong":
                digest(first_unit, x)
        return dt

    def relay_leg(M):
        """Strong scaling with the remainder members RELAYED (ensemble.relay_plan): every rank keeps M // world resident
        members; the M % world others advance as batches of one, each rank hosting a slice of the timed windows and handing
        the state to the next (torch.distributed send / recv: the only data-path message, 16 MB per hand-over).  Every pass
        sets the trajectory offset and the dropout call counters of its (trajectory, window), so the samples are those of
        the one-GPU job."""
        plan = ensemble.relay_plan(M, world, args.steps, rank)
        q = plan.count
        x1, forc1 = synthetic_state(0, 1, device)
        xr, forc_r = synthetic_state(0, q, device) if q else (None, None)
        c0 = exp.dropout_calls()
        one_pass(exp, x1, forc1)                       # B = 1 warm-up (workspace, kernels); also: calls per pass
        per_pass = tuple(b - a for a, b in zip(c0, exp.dropout_calls()))

        def run(x, forc, first_unit, w_abs):
            exp.set_batch_offset(first_unit)
            exp.set_dropout_calls((per_pass[0] * w_abs, per_pass[1] * w_abs))
            return one_pass(exp, x, forc)

        for w in range(args.warmup):
            if q:
                xr = run(xr, forc_r, plan.start, w)
        # hand-overs: torch.distributed send / recv announced through the store (ensemble.RelayComm); the ring's pairs are
        # connected outside the timed region with the same unbatched send / recv the hand-overs use
        comm = ensemble.RelayComm(device=device) if world > 1 else None
        if comm is not None:
            comm.warm_up()
        state = {"res": xr}

        def resident_step(w):
            state["res"] = run(state["res"], forc_r, plan.start, args.warmup + w)

        def relay_step(task, w, x):
            return run(x, forc1, task.unit, args.warmup + w)

        def initial_state(unit):
            x = x1.clone()
            for w in range(args.warmup):               # (a relay trajectory's warm-up windows: untimed, on its first host)
                x = run(x, forc1, unit, w)
            return x

        firsts = {t.unit: initial_state(t.unit) for t in plan.tasks if t.src is None}
        barrier()
        t0 = time.perf_counter()
        # the product's schedule (ensemble.RelayRunner, the one loop.run_inference(relay=) drives): a host advances a relay
        # trajectory through the windows its resident batch has passed, as soon as the state is there
        finals = ensemble.run_relay(plan, args.steps, resident_step, relay_step, lambda u: firsts[u], comm, like=lambda t: x1)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt, comm.recv_wait_s], device=red_dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt[0].item())
            relay_info["recv_wait_s_max_over_ranks"] = round(float(tt[1].item()), 4)
        if q:
            assert torch.isfinite(state["res"]).all(), "non-finite state after the rollout"
            digest(plan.start, state["res"])
        for u, x in finals.items():
            assert torch.isfinite(x).all(), "non-finite relay state after the rollout"
            digest(u, x)
        return dt

    M = args.members
    parts = ensemble.partition(M, world)                   # strong: members of ONE initial condition over the ranks
    relay = world > 1 and M % world != 0 and M >= world and not args.no_relay
    legs = {}
    if args.scaling == "strong":
        start, cnt = parts[rank]
        legs["strong"] = (relay_leg(M) if relay else timed_leg(0, start, cnt), M)
        if world > 1 and not args.no_extras:
            legs["weak"] = (timed_leg(rank, rank * M, M), world * M)
    else:
        legs["weak"] = (timed_leg(rank, rank * M, M), world * M)
    dt, n_traj = legs[args.scaling]
    value = n_traj * HORIZON * args.steps / dt
    all_digests = None
    if args.digests:
        if world > 1:
            gathered = [None] * world
            dist.all_gather_object(gathered, digests)
            all_digests = {str(k): v for d in gathered for k, v in d.items()}
        else:
            all_digests = {str(k): v for k, v in digests.items()}

    if rank == 0:
        # the batch rank 0 actually ran: its resident block when the remainder is relayed (25 over 8: 3, not the 4 of a static split)
        B0 = (M // world if relay else parts[0][1]) if args.scaling == "strong" else M
        res = {
            "metric": "forecast-steps/sec (180x360, 63ch), 25-member ensemble DYffusion sampling, whole job",
            "value": round(value, 3),
            "unit": "member-forecast-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": ("f32 tensors; every GEMM-class kernel (1x1 convs, fused MLP, Legendre analysis/synthesis, dhconv) as "
                      "3-pass split-f16 MFMA with f32 accumulation; FFT / norms / pointwise in f32 (statistics f64)")
            if h3 else "f32 (fp32-input MFMA)",
            "data": "synthetic" + (" (TEST MODE --share-gpu: all ranks on one GPU, numbers meaningless)"
                                   if args.share_gpu else ""),
            "config": {
                "workload": "%d-member ensemble x horizon-6 DYffusion sampling pass (6 forecaster + 10 interpolator SFNO "
                            "forwards, dropout stream on, AR feedback), 180x360, %d state + %d forcing channels, embed 256, "
                            "8 blocks" % (M, STATE_CH, FORCING_CH),
                "horizon": HORIZON, "forwards_per_step": 16,
                "members_per_gpu": ([M // world] * world if relay else [c for _, c in parts]) if args.scaling == "strong"
                else [M] * world,
                "relayed_members": (M % world) if (relay and args.scaling == "strong") else 0,
                **({"relay": relay_info} if relay_info else {}),
                "forecast_steps_per_step": n_traj * HORIZON,
                "per_gpu_forecast_steps_per_s": round(value / world, 3),
                "ensemble_steps_per_s": round(value / n_traj, 4),
                "parallelism": ("members of one initial condition sharded over GPUs" if args.scaling == "strong" else
                                "one initial condition (x %d members) per GPU" % M) +
                               (", the %d remainder member(s) relayed between GPUs in time slices as batches of one (one 16 MB "
                                "send / recv per hand-over; no collective)" % (M % world) if relay and args.scaling == "strong"
                                else ", no data-path collective"),
            },
        }
        if "weak" in legs and args.scaling == "strong":
            wdt, wn = legs["weak"]
            res["config"]["weak_scaling"] = {"value": round(wn * HORIZON * args.steps / wdt, 3),
                                             "ms_per_step": round(wdt / args.steps * 1e3, 2), "members_per_gpu": M,
                                             "note": "one initial condition x %d members per GPU" % M}
        # C5 (10 years = 14600 steps, 4 ICs x 25 members) at the measured whole-job rate
        res["c5_extrapolation"] = {"trajectories": 100, "steps": 14600, "measured_steps": HORIZON * args.steps,
                                   "hours_at_this_rate": round(100 * 14600 / value / 3600.0, 2), "n_gpus": world,
                                   "note": "extrapolated from the measured rate; run --steps >= 100 for a 600-step sample"}
        if all_digests is not None:
            res["trajectory_digests"] = dict(sorted(all_digests.items(), key=lambda kv: int(kv[0])))

        def extra(key, fn):
            """An optional measurement must not cost the headline: its failure is recorded under its key."""
            try:
                res[key] = fn()
            except Exception as e:      # noqa: BLE001
                res[key] = {"error": "%s: %s" % (type(e).__name__, e)}

        if not args.no_extras:
            def roof():
                x, forc = synthetic_state(0, B0, device)
                exp.set_batch_offset(0)
                rows, total = profile_pass(exp, x, forc, B0)
                return roofline_from(rows, total, B0, h3)

            extra("roofline", roof)
            if world == 1:
                extra("latency", lambda: latency_b1(exp, device))

                def f32():
                    r = f32_fallback(device, B0)
                    r["slowdown_vs_h3"] = round(value / r["value"], 2)
                    return r

                extra("f32_fallback", f32)
        if world == 1 and not args.no_cpu_baseline:
            extra("cpu_baseline", lambda: cpu_baseline(args.cpu_threads or min(32, os.cpu_count() or 1)))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
