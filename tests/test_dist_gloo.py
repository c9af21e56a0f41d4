"""world_size-2 gloo test of the data-parallel gradient exchange (bucket spans, stage hooks, 1/world scaling)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.spec_encoder = torch.nn.Linear(7, 5)
        self.spat_encoder = torch.nn.Linear(5, 3)
        self.decoder = torch.nn.Linear(3, 9)
        self._hook = None

    def set_backward_stage_hook(self, fn):
        self._hook = fn


def _worker(rank, world, port, q):
    os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "RANK": str(rank), "WORLD_SIZE": str(world),
                       "LOCAL_RANK": str(rank)})
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import sarssl_boot  # noqa: F401
    from sar_ssl_amd import dist as sdist, runtime
    r, w, _ = sdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and sdist.world_size() == 2
    torch.manual_seed(100 + rank)                      # different initial parameters per rank
    net = _Net()
    flat = runtime.FlatParams(net)
    sdist.broadcast_parameters(flat, src=0)            # -> identical everywhere
    ref = flat.flat.clone()
    dist.broadcast(ref, src=0)
    ok_bcast = torch.equal(ref, flat.flat)
    red = sdist.FlatGradAllReduce(net, flat)
    spans = red.spans
    flat.grad.fill_(float(rank + 1))
    for stage in ("decoder", "spat_encoder", "spec_encoder"):      # backward order
        net._hook(stage)
    scale = red.finish()
    q.put((rank, ok_bcast, scale, float(flat.grad.min()), float(flat.grad.max()), sorted(spans.items())))
    dist.destroy_process_group()


def test_flat_grad_allreduce_two_ranks():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_bcast, scale, gmin, gmax, spans in res:
        assert ok_bcast and scale == 0.5
        assert gmin == gmax == 3.0                         # 1 + 2 summed on every element, incl. alignment padding
    spans = dict(res[0][5])
    assert set(spans) == {"spec_encoder", "spat_encoder", "decoder"}
    assert spans["spec_encoder"][0] == 0 and spans["spec_encoder"][1] == spans["spat_encoder"][0]
    assert spans["spat_encoder"][1] == spans["decoder"][0]
