"""The N = 2 / 4 / 8 projection of DESIGN.md section 6 from a one-GPU small-batch profile (tools/small_batch_profile.py).
A static split runs at the pace of its largest share, t(q + 1) per window.  With the remainder members relayed
(`ensemble.relay_plan`) the schedule of `ensemble.RelayRunner` -- a host advances a relay trajectory through the windows its
resident batch has already passed, as soon as the state has arrived -- is SIMULATED rank by rank with the measured pass times
(`simulate`): the job takes what the most loaded rank's own work takes, no rank waits before the end.
A PROJECTION: it assumes the ranks do not interact and that a hand-over costs nothing.
    python tools/relay_projection.py profiles/r6b/single_gpu_small_batch.json [members=25] [windows=20]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import sdy_amd  # noqa: E402,F401
from sdy_amd import ensemble  # noqa: E402

HORIZON = 6


def simulate(n_units, world, n_windows, t):
    """Event-driven replay of RelayRunner on every rank.  `t[B]` = time of one window of a batch of B.  Returns (makespan,
    per-rank finish times) in t's unit."""
    plans = [ensemble.relay_plan(n_units, world, n_windows, r) for r in range(world)]
    now = [0.0] * world
    seen = [0] * world                                           # windows the rank's resident loop has passed
    todo = {r: [[tk, tk.w_begin] for tk in plans[r].tasks] for r in range(world)}
    arrival, done, finish = {}, [False] * world, [0.0] * world
    while not all(done):
        r = min((i for i in range(world) if not done[i]), key=lambda i: now[i])
        moved = False
        if seen[r] < n_windows:
            now[r] += t[plans[r].count] if plans[r].count else 0.0
            seen[r] += 1
            moved = True
        for item in todo[r]:
            tk, nxt = item
            if nxt >= tk.w_end:
                continue
            avail = 0.0 if tk.src is None else arrival.get((tk.unit, tk.w_begin))
            if avail is None:
                continue                                         # (its sender has not finished yet)
            if avail > now[r]:
                if seen[r] < n_windows:
                    continue                                     # not here yet: the resident batch goes on
                now[r] = avail                                   # end of the job: the only place a rank waits
            while item[1] < min(seen[r], tk.w_end):
                now[r] += t[1]
                item[1] += 1
                moved = True
            if item[1] >= tk.w_end and tk.dst is not None:
                arrival[(tk.unit, tk.w_end)] = now[r]
        if seen[r] >= n_windows and all(c >= tk.w_end for tk, c in todo[r]):
            done[r], finish[r] = True, now[r]
        elif not moved:
            now[r] += 1e-6 * max(t.values())                     # waiting for a state nobody has sent yet
    return max(finish), finish


def main():
    prof = json.load(open(sys.argv[1]))
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    W = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    meas = {int(B): o["pass_ms"] for B, o in prof.items()}

    def pass_ms(B):
        if B in meas:
            return meas[B]
        lo = max(b for b in meas if b < B)
        hi = min(b for b in meas if b > B)
        return meas[lo] + (meas[hi] - meas[lo]) * (B - lo) / (hi - lo)

    one = M * HORIZON / (pass_ms(M) * 1e-3)
    print(f"N=1: {M} members, {one:.1f} member-forecast-steps/s")
    for N in (2, 4, 8):
        q, r = divmod(M, N)
        static = M * HORIZON / (pass_ms(q + (1 if r else 0)) * 1e-3)
        t = {1: pass_ms(1), q: pass_ms(q)}
        makespan, _ = simulate(M, N, W, t)
        relay = M * HORIZON * W / (makespan * 1e-3)
        print(f"N={N}: static {q + (1 if r else 0)} x {r} + {q} x {N - r}: {static:.0f}   relayed {q} x {N} + {r} relayed over {W} "
              f"windows: {relay:.0f} ({100 * relay / (N * one):.0f} % of N x the one-GPU rate)")


if __name__ == "__main__":
    main()
