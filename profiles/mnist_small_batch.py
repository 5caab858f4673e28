"""log_prob latency of the MNIST-shaped flow (784-D, MLP 512x3, K = 8, 8 layers: src/maf_mnist.py) at the batch sizes the
reference script uses (128) and a few larger ones, device-resident rows.   python profiles/mnist_small_batch.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _synthetic as O
from normalizing_flows_amd import _lib, flow as nfa

spec = O.make_rqs_spec(8, event_size=784, hidden=(512, 512, 512), num_bins=8, range_min=0.0, range_max=1.0,
                       base="uniform01", final_std=0.05)
cfg = nfa.FlowConfig(kind=_lib.KIND_RQS_MASKED, event_dim=784, num_layers=8, hidden=(512, 512, 512), rqs_bins=8,
                     rqs_range_min=0.0, rqs_range_max=1.0, base=_lib.BASE_UNIFORM01)
f = nfa.CouplingFlow(cfg, spec.layers)
for rows in [int(v) for v in os.environ.get("NF_MNIST_ROWS", "128,512,2048,8192").split(",")]:
    x = nfa.DeviceArray.from_numpy(O.make_inputs("mnist_like", rows))
    out = nfa.DeviceArray((rows,))
    for _ in range(3):
        f.log_prob(x, out)
    f.synchronize()
    import time
    t0 = time.perf_counter()
    f.timer_start()
    for _ in range(20):
        f.log_prob(x, out)
    t_enq = (time.perf_counter() - t0) / 20      # host time to ENQUEUE a call (nothing waited for yet)
    ms = f.timer_stop() / 20
    print(f"rows {rows:6d}: {ms * 1e3:8.1f} us per log_prob call ({rows / ms / 1e3:7.3f} M evals/s); host enqueue "
          f"{t_enq * 1e6:7.1f} us per call", flush=True)
