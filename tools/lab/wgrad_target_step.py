"""GPU lab: bench.py's step for several grid targets of the grouped LoRA weight-gradient launch (kernels.WGRAD_GRID_BLOCKS).
   python tools/lab/wgrad_target_step.py TARGET <bench args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import kernels as K

K.WGRAD_GRID_BLOCKS = int(sys.argv[1])
sys.argv = [sys.argv[0]] + sys.argv[2:]
import bench

bench.main()
