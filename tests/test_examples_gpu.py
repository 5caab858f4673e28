"""The reference's four scripts as retyped on this library (examples/): each runs a short training on the GPU, its
loss falls, and it samples.  One child process at a time (the scripts are programs, not modules)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args], cwd=ROOT,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return out.stdout


def _losses(stdout, key):
    return [float(v) for v in re.findall(key + r":? (-?[0-9.]+(?:e-?[0-9]+)?)", stdout)]


def test_real_nvp_script_trains_and_samples():
    """src/real_nvp.py:286-315: 8 split-halves layers, SGD on batches of 128 (one backward launch per step)."""
    out = _run("real_nvp.py", "2001")
    nll = _losses(out, "NLL")
    assert len(nll) == 3 and nll[-1] < nll[0] - 1.0, out
    assert "samples (1000000, 2)" in out


@pytest.mark.parametrize("script,args,key", [
    ("real_nvp_toy_densities.py", ["--training-steps", "1001", "--eval-frequency", "500"], "validation loss"),
    ("maf_toy_densities.py", ["--training-steps", "401", "--eval-frequency", "200", "--density", "gaussian_blobs"],
     "validation loss"),
    ("maf_mnist.py", ["--training-steps", "41", "--eval-frequency", "20"], "validation loss"),
])
def test_distrax_style_scripts_train_and_sample(script, args, key):
    out = _run(script, *args)
    loss = _losses(out, key)
    assert len(loss) >= 3 and loss[-1] < loss[0], out
    assert "samples (" in out
