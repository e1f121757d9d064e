#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (the cases of tests/fuzz_checks.py, as many as asked for): ADMM_MGL / ADMM_SGL through the C
ABI against the CPU oracle on shapes and parameters the fixed tests do not visit.
    python tools/fuzz_parity.py [cases] [seed] [solver|batch|block|ext|ops|stats|grid|isolate|mgrid|kgrid|egrid|bigbatch]     -- prints every case that is off by more than 1e-9 (inputs -> gpurun_out/fuzz/)
                                                      and a summary line"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_checks  # noqa: E402

if os.environ.get("GGL_DEBUG_POISON", "") not in ("", "0"):
    # every ctx of this process starts from buffers filled with that byte instead of zeros (1 = 0xFF: NaN; 127 / 71: finite garbage)
    from gglasso_amd import _lib
    _lib.load().ggl_debug_poison(int(os.environ["GGL_DEBUG_POISON"]))
    print("ctx buffers pre-filled with byte", int(os.environ["GGL_DEBUG_POISON"]))
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
kind = sys.argv[3] if len(sys.argv) > 3 else "solver"
t0 = time.time()
bad, notes, mx = fuzz_checks.run_cases(cases, seed, out=lambda m: print(m, flush=True), dump_dir=os.path.join(ROOT, "gpurun_out", "fuzz"),
                                       kind=kind)
if fuzz_checks.GROUPED:
    print("grouped Omega-steps per case (forced, steps):", fuzz_checks.GROUPED)
print(f"{cases} {kind} cases (seed {seed}): {bad} off by more than {fuzz_checks.TOL:g} (relative to max(1, |ref|_max)) or with another status; "
      f"{notes} last-bit stopping notes; largest deviation of the rest {mx:.2e}; {time.time() - t0:.0f} s")
