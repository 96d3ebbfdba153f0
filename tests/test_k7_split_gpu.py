"""The opt-in form of the rider's K7 apply (ARX_K7_RIDER_SPLIT=1, round 5; include/arx.h, "Round 5, opt-in"): run
records in sorted order for the one-hot list too, phase 7 = the entity table's runs alone, phase 8 = ONE launch with
the token runs and the other one-hot tables' runs.  The switch is read once per process, so the tests of what it
changes -- whole HET / MIX training steps (small and BASELINE-sized, bit-reproducibility included) and the sequence
model with multi-hot items -- are re-run here in a child process with the switch set, against the oracle at the same
1e-4.  (The default path, window + finish + token apply, is what every other `-m gpu` test runs.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, timeout):
    env = dict(os.environ, ARX_K7_RIDER_SPLIT="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    return out


def test_k7_split_training_steps_match_oracle(dev):
    out = _run(["tests/test_hmf_gpu.py", "tests/test_lstm_gpu.py", "-k", "steps_match_oracle"], 900)
    assert " passed" in out and "failed" not in out


def test_k7_split_fullsize_steps_match_oracle(dev):
    out = _run(["tests/test_fullsize_gpu.py", "-k", "hmf_matches_embedding_space_oracle or bit_reproducible"], 1200)
    assert " passed" in out and "failed" not in out
