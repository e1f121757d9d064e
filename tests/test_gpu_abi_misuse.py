"""The C ABI under arguments a careless binding could pass (include/ggl_hip.h: "every function returns 0 on success, <0 on
error"): tools/abi_misuse.py in a child process -- K / p of 0 or negative, a device that does not exist, NULL handles and
buffers, indices out of range, unknown options / penalties, non-positive rho, periods that do not divide K -- every probe an
error CODE with a message, no crash, and the same ctx still steps afterwards.  Round 6 found two things with it: a failed
hipSetDevice left its code as the thread's last error and the NEXT valid call reported it; a period larger than K was taken
as "all K"; later an unknown eigensolver selector and a non-positive beta of the log-det prox were accepted.  (The reference's counterpart: the asserts at the top of its solvers, solver/admm_solver.py:118-140.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_misuse_of_the_c_abi_is_an_error_code_never_a_crash():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "abi_misuse.py")], capture_output=True, text=True, timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and lines and lines[-1] == "ok", r.stdout[-3000:] + r.stderr[-2000:]
    assert not any(ln.startswith("BAD") for ln in lines)
    assert sum(ln.startswith("ok ") for ln in lines) >= 60
