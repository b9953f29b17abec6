"""The LDS-DMA helper of the weights-stationary and weight-gradient kernels (csrc/igemm.hip: lds_dma16_buf) writes M0 and
does not restore it.  That holds only while nothing else in those kernels touches M0, and clang ignores an "m0" clobber, so
the property is checked on the compiler's OUTPUT: scripts/check_m0.py compiles igemm.hip to gfx950 assembly (no GPU needed)
and fails on any M0 access that is not one of the helpers' own."""
import os, shutil, subprocess, sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_stray_m0_access_in_kernels_that_do_not_restore_it():
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_m0.py")], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if "other M0 accesses" in l]
    assert len(lines) >= 20, r.stdout          # igemm4 (7), igemm4c (5), wgrad3 (4), wgrad3s2 (3), igemm3s2 (9)
    assert all(l.rstrip().endswith("other M0 accesses: 0") for l in lines)
