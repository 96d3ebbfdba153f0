"""K1 (multi-hot gather) as a physical HBM measurement: bench.py's k1_past_llc (2 GB table, permutation-slice
tokens, three disjoint slices rotated between launches) printed alone.  Run under
`rocprofv3 --pmc FETCH_SIZE` for the per-dispatch counter rows of exactly these launches."""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'a-recsys_amd'))
import torch

import bench

print(json.dumps(bench.k1_past_llc(torch.device('cuda', 0))))
