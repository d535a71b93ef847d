"""world_size-2 gloo test of the data-parallel glue (CPU; the GPU kernels are not involved)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(r, ws, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from explainn_amd import ExplaiNN
    from explainn_amd.parallel import (GradAllReduce, broadcast_parameters, shard_batch,
                                       shard_bounds)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=r, world_size=ws)
    torch.manual_seed(100 + r)                       # ranks start from DIFFERENT parameters
    m = ExplaiNN(4, 5, 26, 2)
    broadcast_parameters(m)
    flat = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    ref = [torch.zeros_like(flat) for _ in range(ws)]
    dist.all_gather(ref, flat)
    same = all(torch.equal(ref[0], t) for t in ref)
    g = torch.full((10,), float(r + 1))
    GradAllReduce(g)()
    # the two-part reduction StepEngine uses (tail first, asynchronously; head after the step)
    g2 = torch.arange(10.) * (r + 1)
    sync = GradAllReduce(g2, split=3)
    work = sync.start_tail()
    assert work is not None
    g2[:3] += 100.0 * (r + 1)             # "the rest of the backward" writes the head meanwhile
    sync.finish(work)
    g3 = torch.full((4,), float(r))
    s3 = GradAllReduce(g3, split=0)       # nothing to split off: one plain all-reduce
    assert s3.start_tail() is None
    s3.finish(None)
    x = torch.arange(7 * 4 * 26, dtype=torch.float32).reshape(7, 4, 26); y = torch.arange(7.)[:, None]
    xs, ys = shard_batch(x, y)
    q.put((r, same, g.tolist(), ys.flatten().tolist(), shard_bounds(7, ws, r), g2.tolist(),
           g3.tolist()))
    dist.barrier(); dist.destroy_process_group()


def test_broadcast_allreduce_and_sharding_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue(); port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs: p.join(timeout=60)
    assert all(p.exitcode == 0 for p in procs)
    assert res[0][1] and res[1][1], "parameters equal on all ranks after broadcast"
    assert res[0][2] == [1.5] * 10 and res[1][2] == [1.5] * 10, "gradient average"
    assert res[0][3] == [0., 1., 2., 3.] and res[1][3] == [4., 5., 6.]
    assert res[0][4] == (0, 4) and res[1][4] == (4, 7)
    want = [1.5 * i + (150.0 if i < 3 else 0.0) for i in range(10)]
    assert res[0][5] == want and res[1][5] == want, "two-part gradient average"
    assert res[0][6] == [0.5] * 4 and res[1][6] == [0.5] * 4
