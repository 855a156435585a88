"""BASD distillation inner-loop benchmark (BASELINE.json metric) on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg2] [--breakdown]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the second form itself, as a
child process, before this process has made any GPU call (a process that has touched the GPU is never re-exec'd).
`--launch-check` rehearses that launch path on the CPU: the workers form a gloo group, all-reduce the real
gradient bucket once and exit without initialising HIP.

One step = the loss hot path on one synthetic minibatch per GPU: BASDLoss forward (selector ranks /
subspaces / principal angles, attention-weighted Procrustes loss, CE, UW-SO) + backward to the student
tokens and logits + (N > 1) RCCL all-reduce of a DeiT-S-sized fp32 gradient buffer (22.05 M parameters
+ the 4 selector temperatures) that a stand-in per-feature head fills from the token gradients.
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import gc
import json
import os
import socket
import statistics
import subprocess
import sys
import time
from types import SimpleNamespace

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-inductive-bias-distillation_amd"))

# Hardware queues per device of the HIP runtime (default 4), BEFORE the runtime loads: the loss uses four streams and the
# communicator brings its own -- with four queues they alias, and launches of unrelated streams wait for each other
# (measured at world size 1 with the process group initialised: 2.7 ms per step against 1.6)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import torch
import torch.distributed as dist

from basd_amd import _lib, chain, ddp, ops, synth
from basd_amd.losses import BASDLoss

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TF = 157.3      # MI355X_MICROARCH.md: fp32-input MFMA (v_mfma_f32_32x32x2_f32), dense
MFMA_BF16_PEAK_TF = 2500.0    # MI355X_MICROARCH.md: bf16 MFMA, dense (the split-operand Gram / GEMM kernels run on it)
STUDENT_PARAMS = {"cfg1": 5_700_000, "cfg2": 22_050_664, "cfg4": 86_600_000, "cfg5": 86_600_000}
LABEL_SMOOTHING = {"cfg1": 0.01, "cfg2": 0.001, "cfg4": 0.001, "cfg5": 0.001}


def algorithmic_bytes(shape: synth.LossShape, batch: int, elem: int = 4) -> dict:
    """every input element read once in forward and once in backward, every student-token
    gradient written once; attention counted as the rows actually consumed."""
    e = shape.points
    student = e * batch * shape.n_s * shape.d_s * elem
    teacher = shape.layers_t * batch * shape.n_t * shape.d_t * elem
    a = shape.n_t + (1 if shape.has_cls else 0)
    attn = shape.layers_t * batch * shape.heads * (shape.n_t if shape.has_cls else a * a) * elem
    fwd = student + teacher + attn
    grad = e * batch * shape.n_s * shape.d_s * 4
    return {"fwd": fwd, "bwd": fwd + grad, "step": 2 * fwd + grad, "student": student, "student_grad": grad}


TRAFFIC_FILE = os.path.join("profiles", "r03_traffic.json")


def measured_traffic(cfg: str) -> dict:
    """HBM bytes per launch from the rocprofv3 PMC passes of this same command (profiles/r03_traffic.json, written by
    tools/pmc_traffic.py: FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950 + WRITE_SIZE); {} when not
    collected.  These are numbers of an EARLIER profiled run of this command, copied into the line: `traffic_source`
    says so."""
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), TRAFFIC_FILE)
    try:
        with open(path) as f:
            return json.load(f).get(cfg, {})
    except (OSError, ValueError):
        return {}


def build(shape: synth.LossShape, cfg: str, device):
    torch.manual_seed(42)                                            # reference configs/config.yaml:8
    crit = torch.nn.CrossEntropyLoss(label_smoothing=LABEL_SMOOTHING[cfg])
    return BASDLoss(crit, shape.d_s, shape.d_t, shape.depth, shape.n_s,
                    config=SimpleNamespace(num_extraction_points=shape.points),
                    teacher_has_cls_token=shape.has_cls).to(device)


def one_step(mod, inp, leaves, logits, bucket):
    for v in leaves.values():
        v.grad = None
    logits.grad = None
    loss = mod(logits, inp.targets, leaves, inp.teacher, inp.attn)
    loss.backward()
    # stand-in student head: per-feature bias gradients (token means of the first image) of every extraction layer fill
    # the front of the student part of the bucket (the rest keeps its DeiT-S size so that the all-reduce volume is
    # real): ONE small column-mean launch of the library straight into the bucket.  (A real student backward would
    # consume all of the token gradients -- and take tens of milliseconds; round 2's stand-in re-read all 308 MB of them,
    # which SURVEY 8(d)'s byte count does not contain and which sat in front of the next step's selector chain.)  The
    # selector temperatures' gradient IS its slice of the bucket (``attach_grads``: the view the trainer uses,
    # nothing to pack), which the reference forgets to reduce.
    grads = [leaves[l].grad[:1] for l in mod.token_layers]
    bucket.wait()                       # the previous step's all-reduce of this buffer (queued async) is joined first
    ops.column_means(grads, out=bucket.student_view[: len(grads) * grads[0].shape[-1]])
    bucket.reattach_missing()
    # RCCL over xGMI on the communicator's own stream: it runs underneath the next step's forward (whose teacher /
    # selector side does not depend on the optimizer step); no-op at world size 1
    bucket.all_reduce_mean(async_op=True)
    if _SHADOW is not None:
        # experiment (BASD_BENCH_SHADOW_ALLREDUCE=1, one GPU): a stand-in for that all-reduce -- a FIFTH stream that waits
        # for the caller's and is joined by it one step later, with the 88 MB bucket copied twice on it (2 x (88 MB read +
        # 88 MB written): about a ring all-reduce's HBM traffic per rank); "2": the two joins alone, no copies.
        # Measured: 1.63 -> 2.1-2.2 ms per step either way -- it is the fifth stream with a cross-
        # stream dependency that costs, not its work (the same wait issued by one of the loss's own four streams: 1.65).
        side, tmp = _SHADOW
        cur = torch.cuda.current_stream()
        cur.wait_stream(side)
        side.wait_stream(cur)
        if os.environ.get("BASD_BENCH_SHADOW_ALLREDUCE") == "1":
            with torch.cuda.stream(side):
                tmp[0].copy_(bucket.buffer)
                tmp[1].copy_(tmp[0])
    return loss


_SHADOW = None


def cpu_baseline(cfg: str, shape: synth.LossShape, sample_batch: int, repeats: int = 3) -> dict:
    """The CPU oracle (restatement of the reference, pinned by tests/golden) on a bounded sample of the same
    workload: one warm-up step, then the median of `repeats` forward+backward steps."""
    from oracle import basd_oracle as O
    # LAPACK's SVD does not scale past a few cores (128 threads are SLOWER than 16 on the GPU box's host):
    # use the one-GPU CPU share
    cores = min(16, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    torch.manual_seed(42)
    state = O.SelectorState.create(shape.points, shape.d_s, shape.d_t)
    crit = torch.nn.CrossEntropyLoss(label_smoothing=LABEL_SMOOTHING[cfg])
    inp = synth.make_inputs(shape, 1234, batch=sample_batch)
    layers = O.extraction_layers(shape.depth, shape.points)
    times = []
    for it in range(repeats + 1):
        for v in inp.student.values():
            v.grad = None
            v.requires_grad_(True)
        inp.logits.grad = None
        inp.logits.requires_grad_(True)
        t0 = time.perf_counter()
        loss, _ = O.basd_forward(state, crit, layers, shape.n_s, shape.has_cls, inp.logits, inp.targets, inp.student,
                                 inp.teacher, inp.attn)
        loss.backward()
        if it > 0:
            times.append(time.perf_counter() - t0)
    dt = statistics.median(times)
    return {"value": sample_batch / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"CPU oracle, {cfg} shapes at batch {sample_batch}: 1 warm-up + median of {repeats} "
                      f"forward+backward steps = {dt:.2f} s/step ({', '.join(f'{t:.2f}' for t in times)})"}


def _free_port() -> int:
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def launch_workers(n: int, argv: list[str]) -> int:
    """`python bench.py --gpus N` outside torchrun: start N ranks of this file under torch.distributed.run as a
    CHILD process and hand its exit code back.  Nothing in this (parent) process has initialised HIP: importing
    torch and this repo's modules does not, and the library is only dlopen'ed by the first kernel call."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC for RCCL across processes on this host
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def launch_check(args, rank: int, world: int) -> None:
    """CPU rehearsal of the N > 1 launch path (no GPU call anywhere): gloo group, the real-size gradient bucket
    (student parameters + the selector temperatures) all-reduced once, max-over-ranks time, one JSON line."""
    dist.init_process_group("gloo")
    shape = synth.CONFIGS[args.config]
    mod = build(shape, args.config, "cpu")
    bucket = ddp.FlatGradBucket(STUDENT_PARAMS[args.config], list(mod.parameters()), "cpu")
    bucket.attach_grads([])
    bucket.student_view.fill_(float(rank + 1))
    mod.layer_selector.log_temperatures.grad.fill_(10.0 * (rank + 1))
    dist.barrier()
    t0 = time.perf_counter()
    bucket.all_reduce_mean(async_op=True)
    bucket.wait(all_slots=True)
    dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    expect = sum(range(1, world + 1)) / world
    ok = bool(torch.allclose(bucket.student_view[:16], torch.full((16,), expect))
              and torch.allclose(bucket.buffer[-shape.points:], torch.full((shape.points,), 10.0 * expect)))
    if rank == 0:
        print(json.dumps({"launch_check": ok, "n_gpus": world, "backend": "gloo",
                          "grad_allreduce_bytes": int(bucket.buffer.numel() * 4),
                          "allreduce_ms_max_over_ranks": 1e3 * float(t.item()),
                          "hip_initialised": bool(torch.cuda.is_initialized())}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(1)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="cfg2", choices=sorted(synth.CONFIGS))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the config's)")
    ap.add_argument("--contiguous", action="store_true", help="contiguous inputs instead of the callers' strided views")
    ap.add_argument("--breakdown", action="store_true", help="print a per-entry-point time table to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=32)
    ap.add_argument("--cpu-repeats", type=int, default=3)
    ap.add_argument("--cpu-baseline-only", action="store_true",
                    help="time only the CPU oracle (e.g. --cpu-sample-batch 256 --cpu-repeats 1 for the full headline batch) "
                         "and print its record; no GPU call")
    ap.add_argument("--teacher-rank", type=int, default=0,
                    help="signal rank of the synthetic teacher features (default: the config's 48); e.g. 160 ~ what a "
                         "random-init ResNet-50 feeds the selector: a secondary workload, named in config.workload")
    ap.add_argument("--launch-check", action="store_true",
                    help="CPU rehearsal of the N > 1 launch: gloo group + one all-reduce of the gradient bucket, no GPU call")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_workers(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.launch_check:
        launch_check(args, rank, world)
        return
    if args.cpu_baseline_only:
        shape0 = synth.CONFIGS[args.config]
        print(json.dumps({"cpu_baseline": cpu_baseline(args.config, shape0, min(args.cpu_sample_batch, shape0.batch),
                                                       repeats=args.cpu_repeats)}))
        return
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    # BASD_FORCE_ALLREDUCE=1: test hook -- initialise the communicator and run the per-step all-reduce at world size 1 too
    # (one GPU: what the N > 1 plumbing costs a step, without a second rank)
    use_comm = world > 1 or os.environ.get("BASD_FORCE_ALLREDUCE") == "1"

    shape = synth.CONFIGS[args.config]
    if args.teacher_rank > 0:
        import dataclasses
        shape = dataclasses.replace(shape, r_t=args.teacher_rank,
                                    name=f"{shape.name} (synthetic teacher signal rank {args.teacher_rank})")
    batch = args.batch or shape.batch
    mod = build(shape, args.config, device)
    global _SHADOW
    # per-rank minibatch (weak scaling): seed 1234 + rank, generated once, resident in HBM
    # cfg-5 (BASELINE.json configs[4]) hands bf16 features over; the kernels widen them and compute in fp32
    in_dtype = torch.bfloat16 if args.config == "cfg5" else torch.float32
    inp = synth.make_inputs(shape, 1234 + rank, batch=batch, device=device, dtype=in_dtype,
                            strided=not args.contiguous, attn_on_device=shape.layers_t > 1)
    leaves = {k: v.detach().requires_grad_(True) for k, v in inp.student.items()}
    logits = inp.logits.detach().requires_grad_(True)
    if use_comm:
        # The loss's private streams FIRST, the communicator SECOND: two bare steps create and use the loss's streams (and
        # their hardware queues), then RCCL is initialised.  The other way round -- process group first, as round 2 did --
        # the same step takes 2.7 ms instead of 1.6 on one GPU with NOTHING being reduced (the communicator's streams take
        # the queues the loss's streams then have to share); with this order and 8 hardware queues the per-step all-reduce
        # itself costs nothing measurable at world size 1 (1.61 ms).
        for _ in range(2):
            for v in leaves.values():
                v.grad = None
            logits.grad = None
            mod(logits, inp.targets, leaves, inp.teacher, inp.attn).backward()
        mod.layer_selector.finish_pending()
        torch.cuda.synchronize()
        dist.init_process_group("nccl", device_id=device)
    bucket = ddp.FlatGradBucket(STUDENT_PARAMS[args.config], list(mod.parameters()), device)
    if os.environ.get("BASD_BENCH_SHADOW_ALLREDUCE", "0") in ("1", "2") and world == 1:
        _SHADOW = (torch.cuda.Stream(device=device), [torch.empty_like(bucket.buffer) for _ in range(2)])
    bucket.attach_grads([])             # the loss parameters' gradients live in the bucket (views, as in the trainer)
    multi_layer = shape.layers_t > 1

    def step():
        return one_step(mod, inp, leaves, logits, bucket)

    for _ in range(args.warmup):
        loss = step()
    # Python's cyclic collector: a generation-2 pass over the ~10^6 objects that importing torch leaves behind costs tens
    # of milliseconds and lands in one step out of ~25 (measured: one 45 ms step in the first 40).  Everything alive now is
    # set-up state; freeze it (the collector stays ON for what the steps allocate) -- what a long-running trainer does
    # too, see INTEGRATION.md.
    gc.collect()
    gc.freeze()
    mod.layer_selector.finish_pending()       # deferred selector tail of the last warm-up step: outside the timing
    bucket.wait(all_slots=True)
    # Kernels reported with a roofline object, timed live with HIP events recorded on the stream they are queued on
    # (each of these entry points is one kernel, plus a small fold for the Gram):
    #   basd_tridiag_ranked / basd_tridiag   Householder tridiagonalisation (two kernels: shared stage + tail stage)
    #   basd_syrk_multi                      symmetric Gram matrices on the fp32 MFMA
    #   basd_colmean_multi                   column sums: the purest HBM stream of the step
    timed = {"basd_tridiag", "basd_tridiag_ranked", "basd_syrk_multi", "basd_colmean_multi", "basd_jacobi_onesided"}
    _lib.timing = {}
    _lib.timed_names = None if args.breakdown else timed
    # single-teacher steps queue the selector inside ONE library call (basd_selector_chain): there the same launches are
    # bracketed by timed events the call records itself, on the stream each launch is queued on
    chain.TIMING = []
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    mod.layer_selector.finish_pending()       # ... and that of the last timed step: inside
    bucket.wait(all_slots=True)               # the all-reduces still queued on the communicator's stream
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timing, _lib.timing = _lib.timing, None
    marks, chain.TIMING = chain.TIMING, None
    own_ms = 1e3 * elapsed / args.steps
    ms_min = ms_max = own_ms
    if world > 1:
        t = torch.tensor([elapsed, -elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, ms_min, ms_max = float(t[0].item()), 1e3 * float(-t[1].item()) / args.steps, 1e3 * float(t[0].item()) / args.steps

    per_call = {k: [a.elapsed_time(b) for a, b in v] for k, v in timing.items()}      # ms

    def span(a: str, b: str) -> list:
        """per-step duration (ms) between two marks of the selector chain"""
        out = []
        for m in marks:
            try:
                out.append(1e-3 * chain.elapsed_us(m[a], m[b]))
            except (KeyError, RuntimeError):
                pass
        return out
    if marks:
        # same names as the entry points the kernel-by-kernel layout times; the factorisation the host waits for is the
        # teacher side's (its span contains the wait for a free CU and the rank kernel: what the step pays)
        per_call["basd_tridiag_ranked"] = span("tm_tri0", "ranks")
        per_call["basd_syrk_multi"] = span("tm_scol1", "tm_sgram")
        per_call["basd_colmean_multi"] = span("tm_scol0", "tm_scol1")      # (not the stand-in head's small launch)
    if args.breakdown and rank == 0:
        tot = sum(sum(v) for v in per_call.values())
        print(f"{'entry point':32s} {'calls/step':>10s} {'ms/step':>9s} {'share':>6s}", file=sys.stderr)
        for k, v in sorted(per_call.items(), key=lambda kv: -sum(kv[1])):
            print(f"{k:32s} {len(v) / args.steps:10.1f} {sum(v) / args.steps:9.3f} {100 * sum(v) / tot:5.1f}%",
                  file=sys.stderr)
        print(f"{'sum of kernel spans':32s} {'':10s} {tot / args.steps:9.3f}   (wall {1e3 * elapsed / args.steps:.3f} ms/step)",
              file=sys.stderr)

    if rank == 0:
        ms_step = 1e3 * elapsed / args.steps
        ab = algorithmic_bytes(shape, batch)
        d_s, E, L = shape.d_s, shape.points, shape.layers_t
        traffic = measured_traffic(args.config)
        traffic_source = (f"{TRAFFIC_FILE}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of an earlier run of this "
                          "command (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction), not collected in this run"
                          if traffic else None)

        def mean_ms(names):
            """mean duration (ms) of the calls of these entry points, and calls per step"""
            calls = [c for n in names for c in per_call.get(n, [])]
            return (sum(calls) / len(calls), len(calls) / args.steps) if calls else (0.0, 0)

        # --- dominant kernel by GPU time: the tridiagonalisation of the selector's Gram matrices
        chain_step = bool(marks)
        if ops.EIG_SOLVER == "tridiag":
            if chain_step:
                eig_ms, eig_calls = mean_ms(["basd_tridiag_ranked"])
                n_mats = 2 * L if mod.chain_mode in (1, 2, 3) else 2 * L + E
            else:
                # kernel-by-kernel layout: every launch of the two entry points, all their matrices
                eig_ms, eig_calls = mean_ms(["basd_tridiag", "basd_tridiag_ranked"])
                n_mats = (2 * L + E) / max(eig_calls, 1)
            packed = 256 < d_s <= 384
            kernel_name = ("tridiag_packed_kernel (Householder tridiagonalisation, whole factorisation of one matrix in one "
                           "CU's registers: upper triangle as packed row pairs; "
                           + ("teacher-side launch, the one the host waits for)" if mod.chain_mode in (1, 2, 3) else
                              "one launch over the teacher and student matrices)") if packed else
                           "tridiag_kernel + tridiag_tail2_kernel (Householder tridiagonalisation of the selector's Gram "
                           "matrices: shared stage + register-resident tail stage)")
            note = ("latency-bound, neither an HBM stream nor MFMA work: n - 1 dependent Householder steps, each one pass over "
                    "the register-resident upper triangle + one wave's scalar chain + two workgroup barriers"
                    if packed else
                    "latency-bound, neither an HBM stream nor MFMA work: n - 1 dependent Householder steps (first stage: "
                    "matrix shared by several workgroups through L2; last 256 steps: matrix in one CU's registers)") + \
                   (".  `bound` is kept to the contract's vocabulary; the HBM fraction says how far from a stream this "
                    "kernel is by construction.  The roofline-bound kernels of the path are under roofline_mfma / "
                    "roofline_hbm_stream; the duration is the launch's span on its stream, inside the step")
        else:
            eig_ms, eig_calls = mean_ms(["basd_jacobi_onesided"])
            n_mats = 2 * L + E
            kernel_name = "jacobi_block_round_kernel (block one-sided Jacobi, symmetric eigen-solves of the selector)"
            note = "latency-bound chain of dependent pair-steps, not an HBM stream (DESIGN.md section 5)"
        launch_bytes = int(n_mats * 2 * d_s * d_s * 4)      # every matrix read once, its reflectors written once
        achieved = launch_bytes / (eig_ms * 1e-3) / 1e9 if eig_ms > 0 else 0.0
        # --- the MFMA-bound kernel: symmetric Gram launches (lower 128x128 tile pairs only).  The flops are those of the
        # launches that are TIMED: the student launch (E matrices of order d_s over B n_s rows) in chain steps; in the
        # kernel-by-kernel layout every syrk launch of the step -- student, and the teacher side's (2L Grams of the
        # projected tokens, or L Grams in the teacher's own space where the teacher is about as wide as the student)
        def pairs(n):
            t = (n + 127) // 128
            return t * (t + 1) // 2
        student_flops = E * pairs(d_s) * 128 * 128 * 2.0 * batch * shape.n_s
        syrk_ms, syrk_calls = mean_ms(["basd_syrk_multi"])
        if chain_step:
            syrk_flops = student_flops
        else:
            m_t = batch * shape.n_t
            own_space = m_t >= d_s and shape.d_t <= 1.5 * d_s and mod.layer_selector.teacher_space_gram
            teacher_flops = (L * pairs(shape.d_t) if own_space else 2 * L * pairs(d_s)) * 128 * 128 * 2.0 * m_t \
                if m_t >= d_s else 0.0
            syrk_flops = (student_flops + teacher_flops) / max(syrk_calls, 1)      # per launch, like syrk_ms
        syrk_tf = syrk_flops / (syrk_ms * 1e-3) / 1e12 if syrk_ms > 0 else 0.0
        # --- the HBM stream: column sums of the E student token tensors (every element read once, nothing written)
        col_calls = [c for c in per_call.get("basd_colmean_multi", []) if c > 0]
        if not chain_step and col_calls:
            # the kernel-by-kernel layout times every launch of the entry point, in call order: the student layers' is
            # the first of a step (then the teacher layers' where their Gram is formed in their own space, then the
            # stand-in head's small one)
            col_calls = col_calls[0::max(1, round(len(col_calls) / args.steps))]
        col_ms = sum(col_calls) / len(col_calls) if col_calls else 0.0
        col_bytes = ab["student"]
        col_gbs = col_bytes / (col_ms * 1e-3) / 1e9 if col_ms > 0 else 0.0
        line = {
            "metric": "distillation images/sec (BASD loss fwd+bwd+grad all-reduce), DeiT-S<-ResNet-50 @ bs256/GPU"
            if args.config == "cfg2" else f"distillation images/sec (BASD loss fwd+bwd+grad all-reduce), {shape.name}",
            "value": world * batch * args.steps / elapsed,
            "unit": "images/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_step,
            "ms_per_step_min_max_over_ranks": [ms_min, ms_max],
            "rccl_world_size": dist.get_world_size() if world > 1 else 1,
            "collective_backend": dist.get_backend() if world > 1 else None,
            "steps_per_s": 1e3 / ms_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "loss": float(loss.item()),
            "config": {
                "workload": shape.name, "per_gpu_batch": batch, "global_batch": world * batch,
                "student_tokens": [E, batch, shape.n_s, d_s],
                "teacher_tokens": [L, batch, shape.n_t, shape.d_t],
                "input_dtype": str(inp.student[mod.token_layers[0]].dtype).replace("torch.", ""),
                "layout": "contiguous" if args.contiguous else "strided (CLS-sliced / channel-major views)",
                "backward": True,
                "grad_allreduce_bytes": int(bucket.buffer.numel() * 4),
                "grad_allreduce": "RCCL all-reduce (mean) per step on the communicator's stream" if world > 1
                else "none at world size 1 (bucket packed, nothing to exchange)",
                "rank_readback": mod.rank_readback, "readback_deferred_steps": mod.readback_deferred_steps,
                "selector": (f"basd_selector_chain mode {mod.chain_mode}" if marks else "kernel by kernel"),
                "parallelism": f"dp{world}",
            },
            "path_hbm": {
                "algorithmic_bytes_per_step": ab["step"],
                "achieved_GBps": ab["step"] / (ms_step * 1e-3) / 1e9,
                "frac_of_8TBps": ab["step"] / (ms_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "roofline": {
                "kernel": kernel_name,
                "bound": "hbm", "regime": "latency", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get("tridiag"),
                "traffic_source": traffic_source if traffic.get("tridiag") else None,
                "launch_ms": eig_ms, "launches_per_step": eig_calls, "matrices_per_launch": n_mats,
                "algorithmic_bytes_per_launch": launch_bytes,
                "note": note,
            },
            "roofline_mfma": ({
                # split operands: every fp32-grade multiply-add is SIX bf16 MFMA multiply-adds (hi hi, hi mid, mid hi, mid mid,
                # hi lo, lo hi of the three-way exact bf16 split); `achieved` counts those, against the dense bf16 peak
                "kernel": ("syrk_tn_split_kernel (+ syrk_reduce_kernel): centred Gram matrices of the E student layers" if marks else
                           "syrk_tn_split_kernel (+ syrk_reduce_kernel): the step's symmetric Gram launches (student layers; teacher layers)"),
                "bound": "mfma", "achieved": 6 * syrk_tf, "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                "frac": 6 * syrk_tf / MFMA_BF16_PEAK_TF, "traffic": traffic.get("syrk_tn_kernel"),
                "traffic_source": traffic_source if traffic.get("syrk_tn_kernel") else None,
                "launch_ms": syrk_ms, "launches_per_step": syrk_calls, "executed_flops_per_launch": 6 * syrk_flops,
                "fp32_grade_tflops": syrk_tf, "fp32_mfma_peak_tflops": MFMA_F32_PEAK_TF,
                "note": "bf16 MFMA (v_mfma_f32_32x32x16_bf16) on three-way split fp32 operands, fp32 accumulation: fp32-grade "
                        "results at six bf16 multiply-adds each (fp32_grade_tflops = what an fp32 MFMA kernel would have to "
                        "sustain, against its 157 TF peak); flops counted are the lower-triangular 128x128 tile pairs actually "
                        "executed; timed inside the step, i.e. while the other streams' kernels share the chip",
            } if _lib.query("basd_gemm_tuning_get") else {
                "kernel": "syrk_tn_kernel (+ syrk_reduce_kernel): centred Gram matrices of the E student layers" if marks else
                          "syrk_tn_kernel (+ syrk_reduce_kernel): the step's symmetric Gram launches (student layers; teacher layers)",
                "bound": "mfma", "achieved": syrk_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                "frac": syrk_tf / MFMA_F32_PEAK_TF, "traffic": traffic.get("syrk_tn_kernel"),
                "traffic_source": traffic_source if traffic.get("syrk_tn_kernel") else None,
                "launch_ms": syrk_ms, "launches_per_step": syrk_calls, "executed_flops_per_launch": syrk_flops,
                "note": "fp32 MFMA (v_mfma_f32_32x32x2_f32); flops counted are the lower-triangular 128x128 tile "
                        "pairs actually executed; timed inside the step, i.e. while the other streams' kernels "
                        "share the chip",
            }),
            "roofline_hbm_stream": {
                "kernel": "colsum_partial_vec_kernel (+ colsum_final_kernel): column means of the E student layers",
                "bound": "hbm", "achieved": col_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": col_gbs / HBM_PEAK_GBS, "traffic": traffic.get("colsum_partial_vec_kernel"),
                "traffic_source": traffic_source if traffic.get("colsum_partial_vec_kernel") else None,
                "launch_ms": col_ms, "launch_ms_min_max": [min(col_calls), max(col_calls)] if col_calls else None,
                "launches_per_step": len(col_calls) / args.steps,
                "algorithmic_bytes_per_launch": col_bytes,
                "note": "every element of E (B, N, D) token tensors read once, nothing written back; the selector's launch "
                        "over the strided CLS-sliced views, inside the step (beside the teacher side's factorisation and "
                        "the Procrustes kernels of the caller's stream)",
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.config, shape, min(args.cpu_sample_batch, batch),
                                                repeats=args.cpu_repeats)
        print(json.dumps(line))
    if use_comm:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
