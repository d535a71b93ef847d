"""How many host synchronisations does an eval-mode forward cost?  (VERDICT r02 item 7 / SURVEY 8b:
"no host sync inside the op".)  Run under `rocprofv3 --hip-trace --stats`: after two validating
warm-up calls, N forwards are enqueued and the stream is synchronised once at the end; the HIP API
stats then show hipStreamSynchronize / hipMemcpy counts for the whole run.

    rocprofv3 --hip-trace --stats --output-format csv -d gpurun_out/sync -- python3 tools/sync_probe.py 640
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from explainn_amd import ExplaiNN  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = torch.device("cuda", 0)
torch.manual_seed(0)
m = ExplaiNN(300, 19, 200, 1).to(dev).eval()
idx = torch.randint(0, 4, (1024, 200))
x = torch.zeros(1024, 4, 200).scatter_(1, idx[:, None, :], 1.0).to(dev)
with torch.no_grad(), m.eval_cache():
    m(x); m(x)                         # the two validating calls (one flag read each)
    torch.cuda.synchronize()
    print("SYNC_PROBE_BEGIN %d forwards" % N, flush=True)
    for _ in range(N):
        out = m(x)                     # enqueued; the sticky flag is read every 64th call
    m.check_input()                    # one read at the end
print("SYNC_PROBE_END checksum %.6f" % float(out.sum().item()), flush=True)
