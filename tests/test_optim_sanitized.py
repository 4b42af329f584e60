"""The host-side L-BFGS state machine (conditional-ude_amd/csrc/cude_optim.h: header-only, no HIP) under
AddressSanitizer + UndefinedBehaviorSanitizer.  GPU sanitizers are not available on the pool; this is the part of the
library a sanitizer can see: history ring indexing, the reducer path, the non-finite line-search branch and degenerate
inputs (tests/cpp/lbfgs_sanitized.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_lbfgs_state_machine_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "lbfgs_sanitized")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
           "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "conditional-ude_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "lbfgs_sanitized.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0, run.stdout + run.stderr
    out = {l.split()[0]: l.split()[1:] for l in run.stdout.strip().splitlines()}
    it, calls, conv, f = out["rosenbrock12"]
    assert conv == "1" and float(f) < 1e-14 and 20 < int(it) < 500 and int(calls) >= int(it)
    it3, _, conv3, f3 = out["rosenbrock12_sharded_m3"]
    assert conv3 == "1" and float(f3) < 1e-12 and out["reducer_calls"] == ["1"]
    assert out["nonfinite_walls"][2] == "1" and float(out["nonfinite_walls"][3]) < 3.0 + 1e-9     # min of sum cosh + 0.1 x
    assert out["maxiters0"][0] == "0"
    assert out["stationary_start"][2] == "1" and float(out["stationary_start"][3]) == 0.0
    assert out["nan_start"][2] == "0"
