"""Fresh process for tests/test_api_gpu.py::test_two_threads_two_handles_first_launches: two threads, each with its
own handles, make the FIRST launch of the large-LDS kernels of this process at the same moment (a barrier in front of
the call; ctypes drops the GIL inside it).  include/nf_hip.h promises distinct handles on distinct threads; the
kernels' per-device set-up (160 KB of dynamic LDS) must be finished before either thread launches."""
import json
import os
import sys
import threading

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from normalizing_flows_amd import _lib, flow as nfa   # noqa: E402
from oracle import flows_np as O                      # noqa: E402  (tests/ may use the oracle as the checker)

_lib.load()
_lib.require_device()
aff = O.make_affine_spec(8, final_std=0.03)
rqs = O.make_rqs_spec(4, num_bins=12, final_std=0.05)
x = {0: O.make_inputs("banana", 4096, seed=1), 1: O.make_inputs("energy2", 70001, seed=2)}
ref = {(t, k): O.log_prob(spec, x[t]) for t in (0, 1) for k, spec in (("aff", aff), ("rqs", rqs))}
barrier = threading.Barrier(2)
out, errors = {}, []


def work(t):
    try:
        _lib.check(_lib.load().nf_set_device(0))
        fa = nfa.CouplingFlow(nfa.FlowConfig(kind=_lib.KIND_AFFINE_MASKED, event_dim=2, num_layers=8, hidden=(64, 64)),
                              aff.layers)
        fr = nfa.CouplingFlow(nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=2, num_layers=4, hidden=(64, 64),
                                             rqs_bins=12, rqs_range_min=-8.0, rqs_range_max=8.0), rqs.layers)
        for rnd in range(3):
            barrier.wait(timeout=60)
            out[(t, "aff", rnd)] = fa.log_prob(x[t])     # thread 0: msplit kernel (4096 rows); thread 1: chain_kernel
            barrier.wait(timeout=60)
            out[(t, "rqs", rnd)] = fr.log_prob(x[t])
    except Exception as exc:  # noqa: BLE001
        errors.append(f"thread {t}: {type(exc).__name__}: {exc}")
        barrier.abort()


threads = [threading.Thread(target=work, args=(t,)) for t in (0, 1)]
for th in threads:
    th.start()
for th in threads:
    th.join()
res = {"errors": errors}
if not errors:
    for (t, k, rnd), got in out.items():
        r = ref[(t, k)]
        res[f"{k}_t{t}_r{rnd}"] = float((np.abs(got.astype(np.float64) - r) / np.maximum(1.0, np.abs(r))).max())
print(json.dumps(res))
