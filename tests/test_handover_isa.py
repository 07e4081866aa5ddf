"""The matcher's fence-free hand-over (csrc/match_kernels.hip: st_dev / ld_dev + s_waitcnt) rests on what the compiler makes of two builtins on
gfx950; orb_ygz_slam_amd/build.py checks the device assembly on every build (check_handover_isa) and refuses to produce a library otherwise.
Here the check itself is exercised (hipcc cross-compiles without a GPU), and it must be able to fail."""
import shutil

import pytest

from orb_ygz_slam_amd import build as B


def _have_hipcc():
    try:
        B.hipcc()
        return True
    except Exception:
        return False


@pytest.mark.skipif(not _have_hipcc(), reason="hipcc not installed")
def test_handover_lowering_is_what_the_kernel_relies_on():
    got = B.check_handover_isa()
    assert got["probe_loads_sc1"] == 1 and got["probe_stores_sc1"] == 1 and got["probe_plain"] == 0
    assert got["match_stores_sc1"] >= 22 and got["match_loads_sc1"] >= 22
    assert got["match_wbl2"] >= 1 and got["match_inv"] >= 1          # the fenced alternative is compiled in as well


@pytest.mark.skipif(not _have_hipcc(), reason="hipcc not installed")
def test_the_check_can_fail(monkeypatch):
    """with the scoped accesses compiled as ordinary ones (what a target / compiler that ignores the scope would emit) the build stops"""
    flags = list(B.FLAGS)
    monkeypatch.setattr(B, "FLAGS", flags + ["-D__hip_atomic_store(p,v,o,s)=(*(p)=(v))", "-D__hip_atomic_load(p,o,s)=(*(p))"])
    with pytest.raises(RuntimeError, match="check_handover_isa"):
        B.check_handover_isa()
