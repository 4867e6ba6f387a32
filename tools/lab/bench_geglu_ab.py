"""GPU lab: same-box A/B of the GEGLU product formed in the FF projection's epilogue (round 6) against the stand-alone geglu_fwd pass
(rounds 1-5): bench.py's own main() with ops.linear_geglu replaced by the two-node form.   python tools/lab/bench_geglu_ab.py <bench args>"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from hcp_diffusion_amd import ops

ops.linear_geglu = lambda x, host, lora=None: (ops.linear(x, host, lora), None)       # geglu_linear(gact=None) runs hcp_geglu_fwd itself
import bench

bench.main()
