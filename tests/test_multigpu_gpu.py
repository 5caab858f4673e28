"""World-size-2 run of the RCCL path on two GPUs of one node: the sharded NLL (one all-reduce of one float64) and a
training step (gradient all-reduce) against the same work done by one process on the whole batch.  Skipped where
fewer than two GPUs are visible (the one-GPU boxes this repository is developed on: the RCCL path has run there with
one rank only, and RCCL refuses two ranks on one device)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def nfa(nf_lib):
    from normalizing_flows_amd import flow
    return flow


def test_rccl_world2_nll_and_train_step(nfa):
    from normalizing_flows_amd import _lib, training
    from oracle import flows_np as O
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs")
    script = os.path.join(ROOT, "tests", "_rccl_worker.py")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29651", PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, script, str(r), "2"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600)[0])
    except subprocess.TimeoutExpired:
        # a rank hung (one-sided failure inside a collective): end BOTH children, so that no process is left
        # holding a GPU, and fail with what they printed
        for p in procs:
            p.kill()
        tails = [p.communicate()[0] for p in procs]
        pytest.fail(f"RCCL world-2 workers timed out; output so far: {tails}")
    assert all(p.returncode == 0 for p in procs), outs
    fields = [o.strip().splitlines()[-1].split() for o in outs]
    nlls = [float(f[3]) for f in fields]
    sums = [float(f[5]) for f in fields]
    assert nlls[0] == nlls[1] and sums[0] == sums[1], outs   # both ranks hold the all-reduced values
    # one process, whole batch
    spec = O.make_affine_spec(4, final_std=0.02)
    cfg = nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=4, hidden=(64, 64))
    f = nfa.CouplingFlow(cfg, spec.layers)
    x = O.make_inputs("banana", 4096)
    assert abs(nlls[0] - O.nll(spec, x)) <= 1e-5 * max(1.0, abs(O.nll(spec, x)))
    assert abs(nlls[0] - f.nll(x)) <= 1e-9 * max(1.0, abs(nlls[0]))
    f.train_step(x, training.adam(1e-3))
    w = np.concatenate([np.concatenate([W.ravel(), b.ravel()]) for layer in f.get_weights() for W, b in layer])
    assert abs(sums[0] - float(np.abs(w).sum())) <= 1e-5 * float(np.abs(w).sum())
