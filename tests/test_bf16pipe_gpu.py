"""The scorer on the bf16 matrix pipe (csrc/gemm_bx6.hip; EXPERIMENT, off by default): the environment
switches are read once per process, so the parity tests of the paths they change are re-run here in a child
process with ARX_GEMM_BX6=1 ARX_MW_GEMM_FUSE=1 -- the fused 'mw' forward (hinge GEMM: act bits + row sums), the
two bit-operand backward products, the plain logits GEMM, and whole training steps (id-only and HET models,
small and BASELINE-sized) against the oracle at the same 1e-4."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ, ARX_GEMM_BX6="1", ARX_MW_GEMM_FUSE="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    return out


def test_bf16pipe_kernels(dev):
    out = _run(["tests/test_kernels_gpu.py", "-k", "mw_gemm_fused or gemm_bits or gemm_nt_bx6"], 600)
    assert " passed" in out and "failed" not in out


def test_bf16pipe_training_steps_match_oracle(dev):
    out = _run(["tests/test_hmf_gpu.py", "-k", "hinge_epilogue or steps_match_oracle or dropout_replayed"], 900)
    assert " passed" in out and "failed" not in out


def test_bf16pipe_sharded_step_matches_oracle(dev):
    out = _run(["tests/test_dist_gpu.py", "-k", "sharded_hip_backend_world1 or sharded_c5_shape"], 900)
    assert " passed" in out and "failed" not in out


def test_bf16pipe_fullsize_steps_match_oracle(dev):
    out = _run(["tests/test_fullsize_gpu.py", "-k", "not lstm"], 1200)
    assert " passed" in out and "failed" not in out
