"""world_size-2 gloo tests of the N>1 path: utterance sharding, ragged gather of embedding rows,
trial-range sharded scoring (the scorer is a stand-in here: the HIP scorer needs a GPU and is
covered by -m gpu tests; what is tested is the exchange logic)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from deeplip_amd import dist as ddist


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_utt, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        table = torch.arange(n_utt * 4, dtype=torch.float32).view(n_utt, 4)
        lo, hi = ddist.shard_range(n_utt)
        full = ddist.gather_rows(table[lo:hi].clone(), n_utt)
        ia = torch.arange(0, 11, dtype=torch.int32) % n_utt
        ib = (torch.arange(0, 11, dtype=torch.int32) * 3) % n_utt
        scores = ddist.score_trials_sharded(lambda a, b: (full[a.long()] * full[b.long()]).sum(1), ia, ib)
        m = ddist.allreduce_metrics([1.0, float(rank)], "cpu")
        q.put((rank, (lo, hi), full.numpy(), scores.numpy(), m))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_utt", [7, 8])
def test_shard_gather_score_world2(n_utt):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_utt, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    table = np.arange(n_utt * 4, dtype=np.float32).reshape(n_utt, 4)
    ia = np.arange(11) % n_utt; ib = (np.arange(11) * 3) % n_utt
    ref = (table[ia] * table[ib]).sum(1)
    ranges = sorted(r[1] for r in res)
    assert ranges[0][0] == 0 and ranges[-1][1] == n_utt and ranges[0][1] == ranges[1][0]
    for rank, _, full, scores, m in res:
        assert np.array_equal(full, table)
        assert np.array_equal(scores, ref)
        assert m == [2.0, 1.0]


def test_shard_range_partition():
    for n in (0, 1, 7, 64, 25834):
        for w in (1, 2, 3, 8):
            r = [ddist.shard_range(n, i, w) for i in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in r) - min(h - l for l, h in r) <= 1


def _grad_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ps = [torch.nn.Parameter(torch.zeros(3, 5)), torch.nn.Parameter(torch.zeros(7)), torch.nn.Parameter(torch.zeros(2))]
        ps[0].grad = torch.full((3, 5), float(rank + 1))
        ps[1].grad = torch.arange(7, dtype=torch.float32) * (rank + 1)      # ps[2] has no gradient (frozen)
        n = ddist.allreduce_grads(ps)
        q.put((rank, n, ps[0].grad.numpy(), ps[1].grad.numpy(), ps[2].grad is None))
    finally:
        dist.destroy_process_group()


def test_allreduce_grads_world2():
    """The data-parallel gradient exchange of train_fusion.py / train_video.py: flat all-reduce, mean over ranks,
    parameters without a gradient skipped."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    for rank, n, g0, g1, none2 in res:
        assert n == 22 and none2
        assert np.allclose(g0, 1.5) and np.allclose(g1, np.arange(7) * 1.5)
    assert ddist.allreduce_grads([torch.nn.Parameter(torch.zeros(2))]) == 0      # not distributed: no-op


def _bucket_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)                                   # identical replicas
        net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                                  torch.nn.Linear(16, 3))
        unused = torch.nn.Parameter(torch.ones(5))             # never touched by the loss: must stay consistent
        params = list(net.parameters()) + [unused]
        buckets = ddist.GradBuckets(params, bucket_bytes=600)  # several small buckets
        g = torch.Generator().manual_seed(100)                 # the GLOBAL batch; each rank takes its half
        x = torch.randn(8, 6, generator=g); y = torch.randn(8, 3, generator=g)
        lo, hi = ddist.shard_range(8)
        out = []
        for step in range(2):                                  # two steps: the views must survive zeroing
            buckets.zero()
            loss = ((net(x[lo:hi]) - y[lo:hi]) ** 2).mean()
            loss.backward()
            n = buckets.finish()
            out.append([p.grad.clone().numpy() for p in params])
        q.put((rank, n, len(buckets.buckets), out))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_world2_equal_full_batch_gradient():
    """deeplip_amd.dist.GradBuckets (the DP exchange of train_audio.py / train_video.py): gradients accumulate into
    bucket views, bucket all-reduces are launched from the accumulation hooks, and after finish() every rank holds
    the mean over ranks == the gradient of the full batch; a parameter no rank touched keeps a zero gradient."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 16), torch.nn.Tanh(),
                              torch.nn.Linear(16, 3))
    g = torch.Generator().manual_seed(100)
    x = torch.randn(8, 6, generator=g); y = torch.randn(8, 3, generator=g)
    ((net(x) - y) ** 2).mean().backward()
    ref = [p.grad.numpy() for p in net.parameters()]
    for rank, n, nb, out in res:
        assert nb >= 3 and n == sum(p.numel() for p in net.parameters()) + 5
        for step in out:
            for a, b in zip(step[:-1], ref):
                assert np.allclose(a, b, atol=1e-6)
            assert np.array_equal(step[-1], np.zeros(5, dtype=np.float32))


def test_grad_buckets_detects_detached_views():
    p = torch.nn.Parameter(torch.zeros(4))
    b = ddist.GradBuckets([p])
    p.grad = None                                              # what optimizer.zero_grad() (set_to_none=True) does
    with pytest.raises(RuntimeError, match="set_to_none=False"):
        b.finish()


def _bench_exchange_worker(rank, world, port, q):
    """bench.py's step() exchange and train_fusion's epoch reduction with stand-in encoders (their arithmetic needs a
    GPU; what runs here is exactly the N > 1 control flow: shard -> local rows -> all-gather -> every rank holds all)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world), RANK=str(rank))
    assert ddist.init_from_env(torch.device("cpu")) == (rank, world)
    try:
        import bench
        B = 4
        fused = torch.arange(B * 8, dtype=torch.float32).view(B, 8) + 1000 * rank      # this rank's [B, D] rows
        full = bench.exchange(fused, world)
        stats = ddist.allreduce_metrics([2.5 * (rank + 1), float(B)], "cpu")
        q.put((rank, full.numpy(), stats))
    finally:
        dist.destroy_process_group()


def test_bench_exchange_and_metric_reduction_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_exchange_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=180) for _ in procs]
    [p.join(60) for p in procs]
    base = np.arange(32, dtype=np.float32).reshape(4, 8)
    want = np.concatenate([base, base + 1000], 0)
    for rank, full, stats in res:
        assert full.shape == (8, 8) and np.array_equal(full, want)
        assert stats == [7.5, 8.0]


def _branchy_bucket_worker(rank, world, port, q):
    """One rank's graph skips a parameter that sits in a MIDDLE bucket (a length-dependent branch): its bucket never
    completes there during backward.  Launches must still pair up across ranks (index order), not "whatever is ready"."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        a, mid, b = torch.nn.Linear(6, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 3)
        params = list(a.parameters()) + list(mid.parameters()) + list(b.parameters())
        buckets = ddist.GradBuckets(params, bucket_bytes=300)        # b | mid.bias | mid.weight | a ...: different sizes
        sizes = [int(t.numel()) for t in buckets.buckets]
        g = torch.Generator().manual_seed(7)
        x = torch.randn(8, 6, generator=g); y = torch.randn(8, 3, generator=g)
        lo, hi = ddist.shard_range(8)
        order = []
        real_launch = buckets._launch
        buckets._launch = lambda bi: (order.append(bi), real_launch(bi))[1]
        buckets.zero()
        h = torch.tanh(a(x[lo:hi]))
        if rank == 0:
            h = h + torch.tanh(mid(h))                               # rank 1 never touches `mid`
        ((b(h) - y[lo:hi]) ** 2).mean().backward()
        buckets.finish()
        q.put((rank, sizes, order, [p.grad.clone().numpy() for p in params]))
    finally:
        dist.destroy_process_group()


def test_grad_buckets_launch_in_index_order_when_a_rank_skips_a_middle_bucket():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_branchy_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda r: r[0])
    [p.join(60) for p in procs]
    torch.manual_seed(0)
    a, mid, b = torch.nn.Linear(6, 16), torch.nn.Linear(16, 16), torch.nn.Linear(16, 3)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(8, 6, generator=g); y = torch.randn(8, 3, generator=g)
    params = list(a.parameters()) + list(mid.parameters()) + list(b.parameters())
    # reference: mean over ranks of the per-rank gradients (rank 1's graph has no `mid`)
    per_rank = []
    for rank, (lo, hi) in enumerate(((0, 4), (4, 8))):
        for p in params:
            p.grad = None
        h = torch.tanh(a(x[lo:hi]))
        if rank == 0:
            h = h + torch.tanh(mid(h))
        ((b(h) - y[lo:hi]) ** 2).mean().backward()
        per_rank.append([torch.zeros_like(p) if p.grad is None else p.grad.clone() for p in params])
    ref = [(u + v).numpy() / 2 for u, v in zip(*per_rank)]
    sizes = res[0][1]
    assert len(sizes) >= 3 and len(set(sizes)) > 1                  # distinct sizes: a mis-paired launch could not pass
    for rank, _, order, grads in res:
        assert order == list(range(len(sizes))), (rank, order)     # every rank: the same, index, order
        for got, want in zip(grads, ref):
            assert np.allclose(got, want, atol=1e-6)


def test_grad_buckets_second_backward_before_finish_raises():
    p = torch.nn.Parameter(torch.ones(4))
    b = ddist.GradBuckets([p])
    (p * 2).sum().backward()
    with pytest.raises(RuntimeError, match="second gradient"):
        (p * 3).sum().backward()
    b.finish()                                                      # single process: nothing in flight, state resets
    (p * 2).sum().backward()


def _capture_group_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = ddist.capture_group()
        assert g is not None and g is ddist.capture_group() and g is not dist.group.WORLD      # one per job, not the default group
        t = torch.full((4,), float(rank + 1))
        dist.all_reduce(t, group=g)                          # the group works like the default one (same ranks)
        verdict = torch.tensor([1.0 if rank == 0 else 0.0])
        dist.all_reduce(verdict, op=dist.ReduceOp.MIN)       # TrainStepGraph._agree: one rank's doubt is everybody's
        q.put((rank, t.tolist(), float(verdict)))
    finally:
        dist.destroy_process_group()


def test_capture_group_world2():
    """The process group recorded training steps send their collectives to (deeplip_amd.dist.capture_group): created collectively,
    once per job, over the same ranks; and the MIN all-reduce by which the ranks agree on keeping or dropping a recorded step."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_capture_group_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = [q.get(timeout=120) for _ in procs]
    [p.join(60) for p in procs]
    for rank, t, verdict in res:
        assert t == [3.0] * 4 and verdict == 0.0
    assert ddist.capture_group() is None                     # outside a job
