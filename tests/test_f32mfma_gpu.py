"""The A/B reference of the scorer: the f32-input MFMA kernels (ARX_SCORER_F32=1; logits GEMM, wave-per-row loss
kernel, two f32 backward GEMMs).  The default path -- the scorer on the bf16 matrix pipe, f32-exact, every other
`-m gpu` test of this suite runs it -- is selected once per process, so the parity tests of what the switch
changes are re-run here in a child process with the switch set: the kernels, whole training steps (id-only and
HET models, small and BASELINE-sized) and the sharded step, against the oracle at the same 1e-4.  bench.py
reports this path as sub.*_f32mfma beside the headline."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ, ARX_SCORER_F32="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    return out


def test_f32mfma_kernels(dev):
    out = _run(["tests/test_kernels_gpu.py", "-k", "gemm or loss_mw"], 600)
    assert " passed" in out and "failed" not in out


def test_f32mfma_training_steps_match_oracle(dev):
    out = _run(["tests/test_hmf_gpu.py", "-k", "steps_match_oracle or dropout_replayed"], 900)
    assert " passed" in out and "failed" not in out


def test_f32mfma_sharded_step_matches_oracle(dev):
    out = _run(["tests/test_dist_gpu.py", "-k", "sharded_hip_backend_world1"], 900)
    assert " passed" in out and "failed" not in out


def test_f32mfma_fullsize_steps_match_oracle(dev):
    out = _run(["tests/test_fullsize_gpu.py", "-k", "hmf_matches_embedding_space_oracle or bit_reproducible"], 1200)
    assert " passed" in out and "failed" not in out
