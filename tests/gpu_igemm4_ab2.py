import os, sys
from vae_gan_amd import lib as L
if sys.argv[1] == "dma":
    L.LIB_PATH = os.path.join(os.path.dirname(L.LIB_PATH), "libvg_hip_ablate.so")
import gpu_igemm4_ab as A
A.main()
