"""Manual same-box A/B of whole-step time between builds of the library (not collected by pytest):
VG_LIB_EXP=<tag> python tests/gpu_bench_lib.py <bench.py arguments>   ->  bench.py on vae_gan_amd/libvg_hip_<tag>.so"""
import os
import runpy
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vae_gan_amd import lib as L
_v = os.environ.get("VG_LIB_EXP")
if _v:
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), f"libvg_hip_{'exp' if _v == '1' else _v}.so")
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
