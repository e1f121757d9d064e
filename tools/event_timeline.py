#!/usr/bin/env python3
"""Timeline of steady-state ADMM iterations WITHOUT a profiler: an event behind every launch of the
iteration on its stream (ggl_trace_start / ggl_trace_read), the host's own marks beside them.  rocprofv3's kernel trace
serialises the cross-stream waits of the two concurrent parts (0.87 ms per profiled iteration against 0.73 ms); events
cost ~1 us of host time per launch and nothing on the device.

Per traced iteration: for every stream ("lane") the completion time of each launch, hence each launch's duration if it started
when its predecessor on that lane finished, and the idle time where it could not have; the host marks in the same clock
(offset: the base event's completion as seen by the host, a few microseconds).

    python tools/event_timeline.py [workload] [iterations to trace] [opt=val ...]     workloads: bench.py's names"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the workload table)
from gglasso_amd import synth, solver, _lib  # noqa: E402
from gglasso_amd._lib import ptr, check  # noqa: E402

TAGS = {1: "copy", 2: "form_W", 3: "bound_rows", 4: "cw_final", 10: "product", 11: "pair", 20: "Theta", 21: "reduce",
        100: "host: step entered", 101: "host: Theta+reduce queued", 102: "host: early part queued",
        103: "host: residuals seen", 104: "host: rest of next chain queued, return"}


def main():
    args = [a for a in sys.argv[1:] if "=" not in a]
    opts = {a.split("=")[0]: float(a.split("=")[1]) for a in sys.argv[1:] if "=" in a}
    name = args[0] if args else "ggl_K32_p500"
    n_trace = int(args[1]) if len(args) > 1 else 3
    reg, K, p, latent, l1, l2, seed = bench.WORKLOADS[name][:7]
    assert not latent
    S, _ = synth.make_problem(reg, K, p, N=2 * p, seed=seed)
    Om0 = np.repeat(np.eye(p)[None], K, axis=0)
    eng = solver.HipEngine(S, Om0, Om0, np.zeros_like(S), options=opts)
    lib = _lib.load()
    nk = np.ones(K)
    rho = 1.0
    dim = K * (p * p + p) / 2

    def step():
        nonlocal rho
        sq = eng.step(rho, l1, l2, reg, False, None, nk)
        r, s, _, _ = solver.residuals_from_norms(sq, rho, 1e-20, 1e-20, dim)
        rn = solver.next_rho(rho, r, s)
        if rn != rho:
            eng.scale_X(rho / rn)
        rho = rn

    for _ in range(40):                       # into the steady state (7 products, pipelining on)
        step()
    check(lib.ggl_trace_start(eng.h, 4096))
    import time
    t0 = time.perf_counter()
    for _ in range(n_trace + 2):
        step()
    wall = (time.perf_counter() - t0) / (n_trace + 2)
    out = np.zeros((8192, 4))
    n = check(lib.ggl_trace_read(eng.h, ptr(out), 8192))
    rows = out[:n]
    eng.close()
    dev = rows[rows[:, 0] == 0]
    host = rows[rows[:, 0] == 1]
    # iterations by the host's "step entered" marks; the chain of iteration t+1 is queued DURING step t (early part, rest)
    entered = host[host[:, 2] == 100][:, 3]
    print(f"{name} {opts or ''}: {wall * 1e6:.1f} us per iteration while tracing (host clock, {n_trace + 2} iterations); "
          f"{len(dev)} device events, {len(host)} host marks")
    red = dev[dev[:, 2] == 21][:, 3]          # completion of the norm reductions: an iteration's device work ends there
    what = "norm reduction"
    if len(red) < 3:
        # GGL_OPT_REDUCE_RIDER: the reduction rides in the next A' launch and has no completion event of its own
        red = dev[dev[:, 2] == 20][:, 3]
        what = "Theta kernel (the norm reduction rides in the next launch)"
    for it in range(1, min(n_trace + 1, len(red))):
        a, b = red[it - 1], red[it]
        print(f"\n--- iteration {it}: from the end of the previous {what} (t = 0) to the end of this one ({b - a:.1f} us) ---")
        ev = [r for r in rows if a - 1e-9 < r[3] <= b + 40 and not (r[0] == 0 and r[3] <= a)]
        ev.sort(key=lambda r: r[3])
        last_end = {}
        for r in ev:
            t = r[3] - a
            if r[0] == 1:
                print(f"{t:9.1f}            {TAGS.get(int(r[2]), int(r[2]))}")
                continue
            lane = int(r[1])
            prev = last_end.get(lane, None)
            span = f"{r[3] - prev:7.1f} us since the previous completion on lane {lane}" if prev is not None else "  (first on its lane)"
            last_end[lane] = r[3]
            print(f"{t:9.1f}  lane {lane}  {TAGS.get(int(r[2]), int(r[2])):10s} done   {span}")
    # summary: device-idle estimate between the last product of an iteration and the Theta-step's completion, etc.
    th = dev[dev[:, 2] == 20][:, 3]
    prod = dev[(dev[:, 2] == 10) | (dev[:, 2] == 11)]
    if len(th) > 2 and len(red) > 2:
        per = np.diff(red)
        print(f"\nsteady state: {per.mean():.1f} us between the ends of two iterations' {what} (min {per.min():.1f}, max {per.max():.1f})")
        gaps = []
        for t_th in th[1:]:
            before = prod[prod[:, 3] < t_th][:, 3]
            if len(before):
                gaps.append(t_th - before.max())
        print(f"last product done -> Theta done: {np.mean(gaps):.1f} us (the Theta kernel itself + the join of the part streams before it)")


if __name__ == "__main__":
    main()
