#!/usr/bin/env python3
"""Does a dependent launch of a DIFFERENT kernel cost more than one of the same kernel?  (rocprofv3 traces of the ADMM iteration
show 0.0 us between consecutive launches of one product kernel and ~6 us at every change of kernel.)  Device-bound streams of
elementwise torch kernels, same kernel N times against two kernels alternating, no profiler: microseconds per launch."""
import time
import torch

dev = torch.device("cuda:0")
n = 1 << 22                                   # 32 MB of doubles: ~10-15 us per kernel, the host stays ahead
a = torch.ones(n, dtype=torch.float64, device=dev)
b = torch.ones(n, dtype=torch.float64, device=dev)


def run(ops, reps=2000):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        ops[i % len(ops)]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e6


add = lambda: a.add_(1.0)
mul = lambda: a.mul_(1.0000001)
sub = lambda: a.sub_(b)                        # two inputs: another kernel again
for _ in range(2):
    run([add], 500)
print("us per launch, 2000 dependent launches each (same stream, same tensor):")
for name, ops in (("add only", [add]), ("mul only", [mul]), ("sub only", [sub]), ("add / mul alternating", [add, mul]),
                  ("add / mul / sub", [add, mul, sub]), ("add only again", [add])):
    r = [run(ops) for _ in range(3)]
    print(f"  {name:24s} {min(r):7.2f}  (three runs: {', '.join(f'{x:.2f}' for x in r)})")


# the same question for the kernels of the ADMM iteration (development library: ggl_dev_switch_bench): a product launch of the
# direct-to-LDS kernel (32 KB of LDS per workgroup) followed by an LDS-free kernel
import ctypes  # noqa: E402
import os  # noqa: E402
import sys  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gglasso_amd import _lib  # noqa: E402
lib = _lib.load_dev()
out = ctypes.c_double()
names = {0: "product only", 1: "product, one-thread kernel", 2: "one-thread kernel only", 3: "product, elementwise kernel over its output"}
for K, p in ((4, 500), (20, 200), (16, 500)):
    print(f"K = {K}, p = {p}: us per repetition (400 repetitions, default stream)")
    for mode in (0, 1, 2, 3, 0):
        r = []
        for _ in range(3):
            rc = lib.ggl_dev_switch_bench(K, p, -1, 400, mode, ctypes.byref(out))
            assert rc == 0, rc
            r.append(out.value * 1e3)
        print(f"  {names[mode]:46s} {min(r):7.2f}  ({', '.join(f'{x:.2f}' for x in r)})")
