"""ONE bounded diagnostic (run once, in a child process of its own, under `timeout`): does a communicator whose
grouped ncclSend / ncclRecv ran on the legacy NULL stream tear down?  Round 2's `pytest -m gpu` hung in the teardown of
exactly such a communicator (gpurun_out/r02/s3); processes that ran the same exchange on a created stream never did.
The library no longer lets RCCL run on the NULL stream; SDFR_COMM_ALLOW_NULL_STREAM=1 restores the old behaviour for
this one question.  sdfr_comm_close is bounded (SDFR_COMM_CLOSE_TIMEOUT_S), so the answer comes back either way.

    python tools/diag_comm_null_stream.py null|created"""
import os
import sys
import time

mode = sys.argv[1] if len(sys.argv) > 1 else "null"
if mode == "null":
    os.environ["SDFR_COMM_ALLOW_NULL_STREAM"] = "1"
os.environ.setdefault("SDFR_COMM_CLOSE_TIMEOUT_S", "20")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import sdf_playground_amd as sp

print("librccl:", sp.Comm.library_info(), flush=True)
c = sp.Comm(sp.Comm.unique_id(), 0, 1, 0)
st = torch.cuda.Stream()
c.selftest(1 << 20, None if mode == "null" else st.cuda_stream)
c.selftest(12345, None if mode == "null" else st.cuda_stream)
# what the hanging process also did before its teardown: renders, gathers at world 1, handles created and destroyed
r = sp.SDFRenderer(0)
r.initShader("labyrinth")
out = torch.empty((139, 200, 4), dtype=torch.float16, device="cuda")
for _ in range(3):
    r.renderGather(c, 200, 139, out=out, fmt=sp.RGBA16F)
r.sync()
r.close()
t = time.time()
try:
    c.close()
    print("RESULT %s stream: sdfr_comm_close returned after %.3f s" % (mode, time.time() - t), flush=True)
    code = 0
except sp.SdfrError as e:
    print("RESULT %s stream: sdfr_comm_close FAILED after %.3f s: %s" % (mode, time.time() - t, e), flush=True)
    code = 4
os._exit(code)  # a teardown that stuck may have left a thread inside RCCL: do not run destructors over it
