"""The N = 2 / 4 / 8 projection of DESIGN.md section 6 from a one-GPU small-batch profile (tools/small_batch_profile.py):
a rank's window costs t(q) for the static split's largest rank, t(q) + t(1) r / N with the remainder members relayed
(ensemble.relay_plan).  A PROJECTION: it assumes the ranks do not interact.
    python tools/relay_projection.py profiles/r5c/single_gpu_small_batch.json [members=25]"""
import json
import sys

prof = json.load(open(sys.argv[1]))
M = int(sys.argv[2]) if len(sys.argv) > 2 else 25
t = {int(B): o["pass_ms"] for B, o in prof.items()}
HORIZON = 6


def pass_ms(B):
    if B in t:
        return t[B]
    lo = max(b for b in t if b < B)
    hi = min(b for b in t if b > B)
    return t[lo] + (t[hi] - t[lo]) * (B - lo) / (hi - lo)


one = M * HORIZON / (pass_ms(M) * 1e-3)
print(f"N=1: {M} members, {one:.1f} member-forecast-steps/s")
for N in (2, 4, 8):
    q, r = divmod(M, N)
    static = M * HORIZON / (pass_ms(q + (1 if r else 0)) * 1e-3)
    relay = M * HORIZON / ((pass_ms(q) + (pass_ms(1) * r / N if r else 0.0)) * 1e-3)
    print(f"N={N}: static {q + (1 if r else 0)} x {r} + {q} x {N - r}: {static:.0f}   relayed {q} x {N} + {r} relayed: {relay:.0f} "
          f"({100 * relay / (N * one):.0f} % of N x the one-GPU rate)")
