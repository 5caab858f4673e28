"""Stub rank process for the CPU test of bench.py's self-spawning parent (tests/test_abi_and_host.py):
records what the parent handed it and behaves as NF_STUB_MODE says.  Touches no GPU and no library."""
import json
import os
import sys
import time

rank = int(os.environ["RANK"])
record = {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                                         "HSA_ENABLE_IPC_MODE_LEGACY")}
record.update(argv=sys.argv[1:], pid=os.getpid(), sid=os.getsid(0))
with open(os.path.join(os.environ["NF_STUB_DIR"], f"rank{rank}.json"), "w") as fh:
    json.dump(record, fh)
mode = os.environ.get("NF_STUB_MODE", "ok")
if mode == "fail" and rank == 1:
    time.sleep(0.3)
    sys.exit(3)
if mode == "hang" or (mode == "fail" and rank != 1):
    print("partial output that must not be relayed", flush=True)
    time.sleep(120)
if rank == 0:
    print(json.dumps({"metric": "stub", "n_gpus": int(os.environ["WORLD_SIZE"])}), flush=True)
else:
    print(f"rank {rank} chatter", flush=True)   # other ranks' stdout goes to the parent's stderr, never into the line
