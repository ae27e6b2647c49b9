import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fusion_probe as F
from colmap_amd._lib import lib
for wide in ("1", "0"):
    lib().colmap_amd_set_switch(b"COLMAP_AMD_FUSION_WIDE", wide.encode())
    print("WIDE", wide, flush=True)
    F.run(8, 1280, 960)
    F.run(8, 2560, 1920)
