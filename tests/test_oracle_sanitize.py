"""The scalar C oracle under AddressSanitizer and UndefinedBehaviorSanitizer: every entry point on
small seeded inputs (oracle/sanitize_driver.c), degenerate boxes, NaN scores and empty candidate
sets included.  Host code only; no GPU involved."""
import os
import shutil
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.mark.skipif(shutil.which("gcc") is None and shutil.which("cc") is None, reason="no C compiler")
def test_c_oracle_is_clean_under_asan_and_ubsan():
    build = subprocess.run(["make", "-C", ORACLE, "-B", "sanitize_driver"], capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="halt_on_error=1")
    env.pop("LD_PRELOAD", None)          # the sanitizer runtime must come first in the process
    run = subprocess.run([os.path.join(ORACLE, "sanitize_driver")], capture_output=True, text=True, env=env,
                         timeout=300)
    assert run.returncode == 0, run.stdout[-1000:] + run.stderr[-3000:]
    assert "sanitize ok" in run.stdout and "runtime error" not in run.stderr
