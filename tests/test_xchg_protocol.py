"""The protocol of the peer-write exchange (conditional-ude_amd/csrc/cude_xchg.h) with host threads as ranks: the very
function the kernels run (xchg_combine over a memory policy), here with std::atomic accesses, for 1, 2, 3 and 8 ranks --
bit-identical rank-ordered sums on every rank whatever the arrival order, a rank one round ahead of a slow reader (the
two-parity argument), max, and the bounded wait.  Built twice: plain, and under ThreadSanitizer (the data hand-off is
through relaxed atomics only; everything else a rank touches is its own).  tests/cpp/xchg_protocol.cpp.
The GPU side (two processes on one GPU through HIP IPC, captured graphs) is tests/test_gpu_xchg.py."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "cpp", "xchg_protocol.cpp")
INC = os.path.join(ROOT, "conditional-ude_amd", "csrc")


def _run(exe, env=None):
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert run.returncode == 0, run.stdout + run.stderr
    out = dict(l.split() for l in run.stdout.strip().splitlines())
    expected = {f"{k}_ranks{n}" for n in (1, 2, 3, 8) for k in ("sum", "max", "ahead")} | \
               {f"late_rank{n}" for n in (1, 2, 3, 8)} | {"timeout", "recovered"}
    assert set(out) == expected
    assert all(v == "1" for v in out.values()), out


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_exchange_protocol_with_threads_as_ranks(tmp_path):
    exe = str(tmp_path / "xchg_protocol")
    build = subprocess.run(["g++", "-std=c++17", "-O2", "-pthread", "-I", INC, SRC, "-o", exe], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr
    _run(exe)


@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_exchange_protocol_is_clean_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "xchg_protocol_tsan")
    build = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-pthread", "-fsanitize=thread", "-I", INC, SRC, "-o", exe],
                           capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr:
        pytest.skip("thread sanitizer runtime not installed")
    assert build.returncode == 0, build.stderr
    _run(exe, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
