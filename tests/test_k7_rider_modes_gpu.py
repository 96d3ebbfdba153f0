"""The form of the rider's K7 apply that is not the process default (ARX_K7_RIDER = win | split, round 5;
include/arx.h): `split` = run records in sorted order for the one-hot list too, phase 7 = the entity table's runs
alone, phase 8 = ONE launch with the token runs and the other one-hot tables' runs (`win`, the default = the window +
finish launches of rounds 3-4, is what every other `-m gpu` test runs).  The
switch is read once per process, so the tests of what it changes -- whole HET / MIX training steps (small and
BASELINE-sized, bit-reproducibility included) and the sequence model with multi-hot items -- are re-run here in a
child process per mode, against the oracle at the same 1e-4."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


MODES = ["split"]


def _run(args, timeout, mode):
    env = dict(os.environ, ARX_K7_RIDER=mode)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] + args, cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-3000:]
    return out


@pytest.mark.parametrize("mode", MODES)
def test_k7_rider_modes_training_steps_match_oracle(dev, mode):
    out = _run(["tests/test_hmf_gpu.py", "tests/test_lstm_gpu.py", "-k", "steps_match_oracle"], 900, mode)
    assert " passed" in out and "failed" not in out


@pytest.mark.parametrize("mode", MODES)
def test_k7_rider_modes_fullsize_steps_match_oracle(dev, mode):
    out = _run(["tests/test_fullsize_gpu.py", "-k", "hmf_matches_embedding_space_oracle or bit_reproducible"], 1200, mode)
    assert " passed" in out and "failed" not in out
