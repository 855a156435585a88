"""World-size-2 gloo test of the N > 1 path (CPU): per-rank minibatches differ, replicated selector
state is identical, and the flat gradient bucket (student grads | log_temperatures) averages correctly."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
    from types import SimpleNamespace

    from basd_amd import ddp, synth
    from basd_amd.losses import BASDLoss
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        shape = synth.LossShape("ddp", 4, 16, 32, 12, 4, 48, 1, 1, False, 10)
        torch.manual_seed(42)                       # same module seed on every rank: replicated projections
        mod = BASDLoss(torch.nn.CrossEntropyLoss(), shape.d_s, shape.d_t, shape.depth, shape.n_s,
                       config=SimpleNamespace(num_extraction_points=4), teacher_has_cls_token=False)
        inp = synth.make_inputs(shape, ddp.rank_seed(1234))
        # replicated state identical, data different
        proj = mod.layer_selector.proj_t.clone()
        gathered = [torch.empty_like(proj) for _ in range(world)]
        dist.all_gather(gathered, proj)
        same_proj = all(torch.equal(g, gathered[0]) for g in gathered)
        first = inp.student[0][0, 0, :4].clone()
        firsts = [torch.empty_like(first) for _ in range(world)]
        dist.all_gather(firsts, first)
        different_data = not torch.equal(firsts[0], firsts[1])
        # gradient bucket: student part + the 4 temperatures
        bucket = ddp.FlatGradBucket(100, list(mod.parameters()), "cpu")
        bucket.student_view.fill_(float(rank + 1))
        mod.layer_selector.log_temperatures.grad = torch.full((4,), 10.0 * (rank + 1))
        bucket.pack_loss_grads()
        bucket.all_reduce_mean()
        bucket.unpack_loss_grads()
        expect = sum(range(1, world + 1)) / world
        ok = (torch.allclose(bucket.student_view, torch.full((100,), expect))
              and torch.allclose(mod.layer_selector.log_temperatures.grad, torch.full((4,), 10.0 * expect))
              and bucket.buffer.numel() == 104)
        out[rank] = (same_proj, different_data, ok)
    finally:
        dist.destroy_process_group()


def test_two_rank_gradient_bucket():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        for r in range(world):
            assert out[r] == (True, True, True), out[r]


def test_bench_self_launch_on_cpu():
    """`python bench.py --gpus 2` outside torchrun starts its own two workers (torch.distributed.run as a child
    process); with --launch-check they form a gloo group, all-reduce the real gradient bucket and exit 0, and no
    process -- the parent included -- initialises HIP."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch_check"] is True and line["n_gpus"] == 2
    assert line["grad_allreduce_bytes"] == 4 * (22_050_664 + 4)        # DeiT-S parameters + the 4 temperatures
    assert line["hip_initialised"] is False


def test_async_ring_bucket_two_ranks():
    """The double-buffered bucket: the all-reduce of one slot stays queued while the next slot is filled."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        procs = [ctx.Process(target=_ring_worker, args=(r, world, port, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        assert all(out[r] for r in range(world)), dict(out)


def _ring_worker(rank, world, port, out):
    sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))
    from basd_amd import ddp
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        p = torch.nn.Parameter(torch.zeros(4))
        bucket = ddp.FlatGradBucket(8, [p], "cpu", slots=2)
        ok = True
        for step in range(5):
            bucket.next_slot()
            bucket.student_view.fill_(float((rank + 1) * (step + 1)))
            p.grad = torch.full((4,), float(rank + step))
            bucket.pack_loss_grads()
            bucket.all_reduce_mean(async_op=True)
        bucket.wait(all_slots=True)
        # the two slots hold steps 3 and 4
        for step, buf in ((3, bucket._slots[0]), (4, bucket._slots[1])):
            ok = ok and torch.allclose(buf[:8], torch.full((8,), 1.5 * (step + 1)))
            ok = ok and torch.allclose(buf[8:], torch.full((4,), 0.5 + step))
        out[rank] = bool(ok)
    finally:
        dist.destroy_process_group()
