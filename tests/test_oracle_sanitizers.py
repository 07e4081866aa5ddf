"""CPU tier: the oracle under AddressSanitizer + UndefinedBehaviorSanitizer (the oracle defines the semantics, so it
must itself be free of out-of-bounds reads and undefined behaviour).  The golden scenarios are replayed in a subprocess that loads
oracle/_san/libygz_oracle_san.so."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan():
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(asan) or not os.path.exists(asan):
        pytest.skip("libasan not available")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "san"])
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1",
               YGZ_ORACLE_LIB=os.path.join(ROOT, "oracle", "_san", "libygz_oracle_san.so"))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-p", "no:cacheprovider", "tests/test_oracle_paths_golden.py",
                          "tests/test_oracle_golden.py", "tests/test_oracle_dso.py", "tests/test_oracle_stereo.py", "tests/test_oracle_direct.py",
                          "tests/test_oracle_matcher_align.py"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    log = out.stdout + out.stderr
    assert "AddressSanitizer" not in log and "runtime error:" not in log, log[-4000:]
    assert out.returncode == 0, log[-4000:]
