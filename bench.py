#!/usr/bin/env python
"""bench.py — GGN-vector products/s (and posterior samples/s) of the MI355X-native engine.

Workload (BASELINE.json configs[3]): CIFAR-CNN `ResNet1M` (D = 1 084 586,
162 366 720 MACs/example), synthetic inputs X ~ U[0,1]^(n x 32 x 32 x 3) with n = 50 inducing points
per GPU, seeded random-init weights + BN statistics, P = 256 Rademacher probes, alpha = 0.005,
full_set_size = 49 000.  One *step* = one block matvec  V (P, D) -> (GGN + alpha I) V  over the WHOLE data set
of 50 * N examples: every rank sweeps its own 50-example slice for the same P probes, then ONE all-reduce of the
(P, D) block adds the partial sums (the data sum shards across ranks; weak scaling: per-GPU work is fixed).
`value` = P * steps / time — GGN-vector products per second over the whole 50 * N-example set (one
GGN-vp = one product over ALL examples of the config), inputs resident in HBM; the per-shard and the
per-(example x probe) rates are secondary fields.

`--gpus N` with no WORLD_SIZE in the environment starts the N ranks itself (one child process per GPU,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) BEFORE anything touches the GPU; under
`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` the ranks already exist and the flag is
only checked against WORLD_SIZE.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import lip_amd  # noqa: E402,F401
from lip_amd import _native as nv  # noqa: E402
from lip_amd import krylov  # noqa: E402
from lip_amd.dist import ShardedDataSum  # noqa: E402
from lip_amd.engine import LinearizedNet  # noqa: E402
from lip_amd.scalemodels import ResNet1M  # noqa: E402
from lip_amd.toymodels import create_state  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak


def cpu_baseline(net, n, seconds_budget=12.0):
    """The CPU restatement (oracle, PyTorch fp32) timed on this box's host cores on a bounded sample of the
    same workload: (a) literal per-example jvp -> H -> vjp loop of src/ggn.py:133-144, (b) the same
    arithmetic with the example loop batched.  `value` is the faster of the two."""
    from oracle.ggn import compute_ggn_vp as vp_literal, compute_ggn_vp_batched as vp_batched
    st = create_state(net, seed=1231231234, dtype=torch.float32)
    g = torch.Generator().manual_seed(7)
    D = sum(t.numel() for t in _leaves(st.params["params"]))
    v = torch.randn(D, generator=g)

    def rate(factory, n_s, budget=seconds_budget):
        Z = torch.rand((n_s,) + tuple(net.input_shape_raw), generator=g)
        vp = factory(st, Z, "classifier", full_set_size=49000)
        vp(v)                                             # warm-up
        t0, reps = time.perf_counter(), 0
        while reps < 1 or (time.perf_counter() - t0 < budget and reps < 5):
            vp(v)
            reps += 1
        dt = (time.perf_counter() - t0) / reps
        return 1.0 / (dt / n_s * n), reps                 # GGN-vp/s over n examples (linear in examples)

    all_threads = torch.get_num_threads()
    lit_nt = min(all_threads, 8)                          # the per-example loop is small-GEMM work: 8 threads beat 128
    torch.set_num_threads(lit_nt)
    try:
        lit, reps_a = rate(vp_literal, min(n, 4), budget=seconds_budget / 2)
    finally:
        torch.set_num_threads(all_threads)
    by_threads = {}
    for nt in sorted({all_threads, min(all_threads, 32), min(all_threads, 16), min(all_threads, 8)}, reverse=True):
        torch.set_num_threads(nt)                         # more threads is not faster here: report the best count
        try:
            by_threads[nt], reps_b = rate(vp_batched, n, budget=seconds_budget / 2)
        finally:
            torch.set_num_threads(all_threads)
    best_nt = max(by_threads, key=by_threads.get)
    bat = by_threads[best_nt]
    # posterior samples/s of the reference's algorithm on these cores, from a bounded sample: src/sample.py:55-156 costs
    # d x (one W sweep + one W^T sweep) for the Gram (src/ggn.py:207-219) and 2 W + 2 W^T sweeps per draw; a W^T sweep is
    # the forward half and a W sweep the reverse half of one example-batched GGN-vp, so a (W, W^T) pair costs one
    # product: seconds(S draws) = (d + 2 S) / rate  (the d x d Lanczos / solves are negligible next to that)
    d_small, S_ref = n * 10, 200
    cpu_samples = S_ref / ((d_small + 2 * S_ref) / max(lit, bat))
    return dict(value=max(lit, bat), unit="GGN-vp/s", cores=best_nt if bat >= lit else all_threads, kind="port",
                posterior_samples_per_s_extrapolated=cpu_samples,
                posterior_samples_model=f"S / ((d + 2 S) / GGN-vp rate) at d = {d_small}, S = {S_ref}: Gram build d sweeps pairs + "
                                        "2 pairs per draw, small-space algebra not counted",
                literal_per_example=lit, example_batched=bat,
                example_batched_by_threads={str(k): v for k, v in by_threads.items()}, host_threads=all_threads,
                sample=f"CPU restatement in PyTorch fp32 (not reference JAX): (a) literal per-example loop at {lit_nt} "
                       f"threads, {reps_a} x (1 probe x {min(n, 4)} of {n} examples) extrapolated linearly; (b) example-"
                       f"batched, (1 probe x all {n} examples) repeated within {seconds_budget / 2:.0f} s per thread count "
                       f"{sorted(by_threads)}; value = the best")


def _leaves(tree):
    if isinstance(tree, dict):
        for v in tree.values():
            yield from _leaves(v)
    else:
        yield tree


def spawn_ranks(n_ranks: int, argv) -> int:
    """`bench.py --gpus N` outside a launcher: start N copies of this script, one rank per GPU, and wait.  Nothing in
    this (parent) process has initialised HIP — `torch.cuda.device_count()` does not on this image — and the children
    are ordinary child processes (no exec of a GPU-initialised process).  A rank that fails takes the others down
    and its exit code becomes ours.  Fewer GPUs than ranks is refused unless LIP_DIST_BACKEND=gloo asks for the
    rehearsal of the N-rank path on the cards that exist (ranks share GPUs round-robin, collectives over gloo)."""
    ndev = torch.cuda.device_count()
    rehearsal = os.environ.get("LIP_DIST_BACKEND", "nccl") != "nccl"
    if ndev < n_ranks and not rehearsal:
        print(f"bench.py: --gpus {n_ranks} but only {ndev} GPU(s) visible; refusing to run fewer ranks than asked "
              "(LIP_DIST_BACKEND=gloo rehearses the N-rank path on the GPUs present)", file=sys.stderr)
        return 2
    import signal
    procs = []

    def _stop_children(grace=5.0):
        """terminate, then kill, exactly the Popen children that are still alive (never a pattern)"""
        alive = [pr for pr in procs if pr.poll() is None]
        for pr in alive:
            pr.terminate()
        t_end = time.time() + grace
        for pr in alive:
            try:
                pr.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()

    def _on_signal(signum, frame):               # a killed parent must not leave ranks holding the GPUs
        raise KeyboardInterrupt(f"signal {signum}")

    old_handlers = {sg: signal.signal(sg, _on_signal) for sg in (signal.SIGTERM, signal.SIGINT)}
    rc = 0
    try:
        # the rendezvous port: keep the probing socket OPEN (SO_REUSEADDR) until the children exist, so no other
        # process can take the port between "found free" and rank 0's bind
        sk = socket.socket()
        sk.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        for r in range(n_ranks):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                          stdin=subprocess.DEVNULL, stdout=None if r == 0 else subprocess.DEVNULL))
        sk.close()
        live = list(procs)
        while live:
            time.sleep(0.2)
            for pr in list(live):
                code = pr.poll()
                if code is None:
                    continue
                live.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in live:               # exact PIDs of our own children, never a pattern
                        other.terminate()
    except BaseException:
        rc = rc or 130
        raise
    finally:
        _stop_children()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--probes", type=int, default=256)
    ap.add_argument("--n", type=int, default=50, help="examples (inducing points) per GPU")
    ap.add_argument("--samples", type=int, default=200, help="posterior samples for the samples/s line (0 = skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-resnet50", action="store_true", help="skip the full-resolution ResNet-50 slice (configs[4])")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: --n examples per GPU (the data set grows with N); strong: ONE data set of --n-total examples "
                         "sharded over the ranks (configs[3] as written: a data batch sharded 8 ways)")
    ap.add_argument("--n-total", type=int, default=2048, help="examples of the whole data set under --scaling strong")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(1, torch.cuda.device_count())
    dev = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("LIP_DIST_BACKEND", "nccl")      # "gloo": rehearsal of N ranks on one card
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    net = ResNet1M(10)
    state = create_state(net, seed=1231231234, dtype=torch.float32)           # config/scale/resnet1_cifar10.yml:5
    g = torch.Generator().manual_seed(280300 + rank)                          # ip.seed (+rank: own data slice)
    P, alpha, full = args.probes, 0.005, 49000
    strong = args.scaling == "strong"
    if strong:                                   # one data set of n_total examples, this rank's contiguous slice of it
        args.samples, args.no_cpu_baseline = 0, True        # the secondary legs are defined on the 50-example weak config
        from lip_amd.dist import shard_bounds
        n_total = args.n_total
        lo, hi = shard_bounds(n_total, world, rank)
        n = hi - lo
    else:
        n = args.n
        n_total = n * world
    Z = torch.rand(n, 32, 32, 3, generator=g)
    scale = full / n_total
    # this rank's binding for the headline bookkeeping (per-kernel profile, secondary legs): at most 50 examples
    eng = LinearizedNet(state, Z[:min(n, 50)].to(dev), "classifier", device=dev, workspace_bytes=24 << 30, max_chunk=P)
    if n > 64:
        # a slice larger than one binding: example chunks of 50 behind one probe workspace (the data sum is associative)
        from lip_amd.ggn import ExampleChunkedGGN
        chunked = ExampleChunkedGGN(state.to(device=dev, dtype=torch.float32), Z.to(dev), "classifier", full_set_size=full,
                                    example_chunk=50, workspace_bytes=24 << 30, max_probes=P)
        chunked.scale = scale
        local = lambda V, out=None: chunked(V, alpha / world, out=out)
    else:
        local = lambda V, out=None: eng.ggn_vp(V, scale, alpha / world, out=out)
    # alpha/world per rank sums to alpha*V in the all-reduce (no extra pass over the block); N > 1: the block is
    # swept as 3/4 + 1/4 of the probes, the all-reduce of the first part hides behind the sweep of the second
    op = ShardedDataSum(local, 0.0, chunk=((0.75, 0.25) if world > 1 and P >= 64 else None), profile=world > 1)
    V = krylov.fill_rademacher(P, eng.D, 1234, dev)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        op(V)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        Y = op(V)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = 1e3 * dt / args.steps
    value = P * args.steps / dt                    # products over the WHOLE n_total-example set per second
    per_shard = value * world                      # products counted per 50-example shard (round-1 unit)
    coll = None
    if world > 1:                                  # what the timed steps (+ warm-up) handed to the collective, and what it cost
        cs = op.read_stats()
        coll = dict(allreduce_bytes_per_step=cs["allreduce_bytes"] / max(1, cs["calls"]),
                    exposed_collective_ms_per_step=cs["exposed_wait_ms"] / max(1, cs["calls"]),
                    ring_bytes_per_link_per_step=2.0 * (world - 1) / world * cs["allreduce_bytes"] / max(1, cs["calls"]),
                    note="one all-reduce of the (P, D) float32 block per matvec, issued as 3/4 + 1/4 of the probes; the "
                         "exposed time is what the compute stream waits for after its last launch (events around the waits), "
                         "rank 0's view")

    # N > 1, secondary figure: the OTHER sharding north_star names — Hutchinson probes sharded, no data-path collective:
    # every rank sweeps its own block of P probes over a full 50-example set (the trace estimator then all-reduces P
    # scalars per rank, dist.sharded_hutchinson).  N * P products per step, each over a 50-example set.
    probe_sharded = None
    if world > 1:
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            eng.ggn_vp(V, full / eng.n, alpha)
        barrier()
        t = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        probe_sharded = P * world * args.steps / float(t.item())

    # ---- live per-kernel figure: HIP events on the launch stream, instrumented extra steps -------------
    eng.profile(True)
    prof_steps = max(1, min(3, args.steps))
    for _ in range(prof_steps):
        eng.ggn_vp(V, full / eng.n, alpha)
    prof = eng.profile_read()
    eng.profile(False)
    # one-off primal forward of the binding (reported separately; 2 MACs_fwd FLOP per example)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(5):
        nv.check(eng.lib.lip_engine_primal(eng.h, nv.stream_ptr()), "lip_engine_primal")
    torch.cuda.synchronize()
    primal_ms = 1e3 * (time.perf_counter() - t1) / 5
    flops = eng.flops_per_probe()
    xflops = eng.executed_flops_per_probe()        # Winograd ops multiply 4/9 as often as their algorithmic count says
    kinds = {nv.OP_IGEMM: "igemm_kernel", nv.OP_WGRAD: "wgrad_kernel"}
    per_kernel = {}
    for k, name in kinds.items():
        ms, cnt = prof.get(k, (0.0, 0))
        if cnt:
            per_kernel[name] = dict(ms_per_step=ms / prof_steps, launches_per_step=cnt // prof_steps,
                                    avg_launch_ms=ms / cnt,
                                    tflops=flops[k] * P * prof_steps / (ms * 1e-3) / 1e12,
                                    executed_tflops=xflops[k] * P * prof_steps / (ms * 1e-3) / 1e12,
                                    mfma_pipe_utilisation=xflops[k] * P * prof_steps / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS)
    other_ms = sum(ms for k, (ms, c) in prof.items() if k not in kinds) / prof_steps
    dom = "igemm_kernel"
    achieved = per_kernel[dom]["tflops"]
    # HBM bytes per launch are NOT measured inside this run (the PMC passes need rocprofv3 around the process): null here;
    # the committed counter passes of this same command are profiles/r3_traffic.json (scripts/profile.sh)
    traffic = None
    winograd_on = bool(eng.lib.lip_get_winograd() != 0 and eng.lib.lip_get_precision() == 0)
    roofline = dict(bound="mfma", kernel=dom + " (tangent-forward + data-gradient implicit GEMMs, f32 MFMA"
                                         + ("; the 3x3 stride-1 layers as Winograd F(2x2,3x3)" if winograd_on else "") + ")",
                    achieved=achieved, peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=achieved / PEAK_F32_MFMA_TFLOPS,
                    winograd=winograd_on,
                    executed=dict(tflops=per_kernel[dom]["executed_tflops"], frac=per_kernel[dom]["mfma_pipe_utilisation"],
                                  whole_sweep_tflops=sum(xflops.values()) * P / (ms_per_step * 1e-3) / 1e12,
                                  note="`achieved` / `frac` count ALGORITHMIC FLOPs (SURVEY 8d: 8 MACs_fwd per example-probe) "
                                       "per second, so with the Winograd route on they can exceed the pipe's peak: those ops "
                                       "issue 4/9 of their algorithmic multiplications.  `executed` counts the FLOPs the MFMA "
                                       "instructions actually perform — its `frac` is the utilisation of the f32 matrix pipe"),
                    traffic=traffic, per_kernel=per_kernel, other_kernels_ms_per_step=other_ms,
                    primal_pass_ms=primal_ms, primal_pass_tflops=2 * net.macs_per_example() * n / (primal_ms * 1e-3) / 1e12,
                    whole_sweep_tflops=sum(flops.values()) * P / (ms_per_step * 1e-3) / 1e12,
                    flop_model="algorithmic FLOPs from the op tapes: conv segment 2*R*N*Ktot, data-gradient segment "
                               "2*MACs of its conv, WGRAD 2*R*N*M (= 8*MACs_fwd per example-probe minus the input "
                               "layer's two absent terms, SURVEY 8d)")

    # ---- posterior samples/s (reference algorithm src/sample.py:55-156).  N > 1: the inducing set is small, every
    # rank holds it and draws its own args.samples with its own seed (replicas, no collective); whole-job rate =
    # N * samples / slowest rank -----------
    samples_line = None
    if args.samples > 0:
        from lip_amd.sample import sample
        st_dev = state.to(device=dev, dtype=torch.float32)
        Zd = Z.to(dev)
        # warm-up, like --warmup for the headline: the same call on OTHER data (a different (state, Z) binding, so
        # nothing of the timed call is cached) — the first rocSOLVER eigh / hipBLASLt GEMM of each shape loads code
        # objects from disk (0.2-1 s on a fresh box), a property of the process, not of the sampler
        Zw = torch.rand(Zd.shape, generator=torch.Generator().manual_seed(99)).to(dev)
        sample(st_dev, Zw, eng.D, alpha, 1, "classifier", num_samples=args.samples, full_set_size=full)
        sample(st_dev, Zw, eng.D, alpha, 2, "classifier", num_samples=2000, full_set_size=full)
        del Zw
        from lip_amd.ggn import clear_engine_cache
        from lip_amd import sample as _smod
        _smod._PARTS_CACHE.clear()
        clear_engine_cache()
        import gc
        gc.collect()                      # the warm-up binding's factor / Gram blocks go back to the allocator NOW, not at
        barrier()                         # some later collection inside a timed region
        t1 = time.perf_counter()
        S = sample(st_dev, Zd, eng.D, alpha, 1392 + rank, "classifier", num_samples=args.samples, full_set_size=full)
        torch.cuda.synchronize()
        ts = time.perf_counter() - t1
        # steady state on the bound sampler: one untimed pass first (its 8.7 GB result block may need a fresh hipMalloc,
        # ~0.25 s, depending on what the caching allocator still holds), then the timed one
        S = sample(st_dev, Zd, eng.D, alpha, 2392 + rank, "classifier", num_samples=2000, full_set_size=full)
        del S
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        S = sample(st_dev, Zd, eng.D, alpha, 2393 + rank, "classifier", num_samples=2000, full_set_size=full)
        torch.cuda.synchronize()
        ts2 = time.perf_counter() - t1
        if world > 1:
            tt = torch.tensor([ts, ts2], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            ts, ts2 = float(tt[0].item()), float(tt[1].item())
        parts = next(iter(_smod._PARTS_CACHE.values()))
        r_dirs, n_stiff, blk = int(parts.Qm.shape[0]), int(parts.n_stiff), 256
        # per block of 256 draws on the bound sampler: two GEMM passes over the orthonormalised factor Qm (r, D) —
        # <q_k, eps> and eps * a + T Qm — plus the float64-accumulated pass over the n_stiff stiffest rows
        blk_s = ts2 * blk / 2000
        s_flops = 4.0 * blk * r_dirs * eng.D + 2.0 * blk * n_stiff * eng.D
        s_bytes = 4.0 * eng.D * (2 * r_dirs + 2 * blk)           # 2 passes over the factor + read eps + write the draw
        samples_roofline = dict(
            block=blk, rows_of_factor=r_dirs, stiff_rows_f64=n_stiff, ms_per_block=1e3 * blk_s,
            tflops=s_flops / blk_s / 1e12, frac_of_f32_mfma_peak=s_flops / blk_s / 1e12 / PEAK_F32_MFMA_TFLOPS,
            algorithmic_GBps=s_bytes / blk_s / 1e9, frac_of_hbm_peak=s_bytes / blk_s / 1e9 / 8000.0,
            bound="mfma (arithmetic intensity %.0f FLOP/B against a machine balance of 19.7)" % (s_flops / s_bytes),
            kernels="pass 1 <q_k, eps> on lip_gemm_nt (split-K MFMA, both operands along D: 2.8 ms against hipBLASLt's "
                    "6.4 ms), its stiffest rows again on lip_dot_nt_f64 (float64 accumulation), pass 2 eps a + T Qm as one "
                    "library GEMM with the addend in its epilogue (hipBLASLt through torch.addmm, 107 TFLOP/s on that "
                    "shape); factor rows from lip_vjp_rows, eps from the Philox normal fill",
            model="FLOPs 4 S r D + 2 S n_stiff D; bytes 4 D (2 r + 2 S) = two passes over the factor + read eps + write out")
        samples_line = dict(value=world * args.samples / ts, unit="posterior samples/s", num_samples=world * args.samples,
                            seconds=ts, at_2000_samples_same_binding=world * 2000 / ts2, ranks=world, roofline=samples_roofline,
                            includes="fresh (state, Z) binding: engine build + primal pass + factor rows (one per-example "
                                     "backward sweep of K probes) + float64 Gram + exact small-space f(A), then W^T / W "
                                     "as GEMMs; the second figure reuses the binding (2000 draws)",
                            finite=bool(torch.isfinite(S).all().item()))

    # ---- opt-in split-precision MFMA mode (bf16x3): same workload, implicit-GEMM kernels on bf16 matrix cores ------
    split_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.engine import set_precision
        Y32 = eng.ggn_vp(V, scale, alpha)
        try:
            set_precision("bf16x3")
            Ys0 = eng.ggn_vp(V, scale, alpha).clone()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(3):
                Ys = eng.ggn_vp(V, scale, alpha)
            torch.cuda.synchronize()
            t_s = (time.perf_counter() - t1) / 3
        finally:
            set_precision("f32")
        split_line = dict(value=P / t_s, unit="GGN-vp/s", ms_per_step=1e3 * t_s, dtype="bf16x3",
                          rel_diff_vs_f32=float(((Ys - Y32).abs().max() / Y32.abs().max()).item()),
                          run_to_run=float(((Ys - Ys0).abs().max() / Y32.abs().max()).item()),
                          note="opt-in lip_set_precision(1): operands split hi+lo in bf16, 3 bf16 MFMAs per product, "
                               "f32 accumulate (implicit-GEMM and weight-gradient kernels); NOT the headline: the default is exact f32")

    # ---- opt-in materialised-factor mode (same results, two plain GEMMs; valid while d*D*4 B fits HBM) -------
    factor_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.ggn import compute_ggn_vp
        st_dev = state.to(device=dev, dtype=torch.float32)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fop = compute_ggn_vp(st_dev, Z.to(dev), "classifier", full_set_size=full, mode="factor")
        torch.cuda.synchronize()
        t_build = time.perf_counter() - t1
        Yf = fop(V)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            Yf = fop(V)
        torch.cuda.synchronize()
        t_f = (time.perf_counter() - t1) / 5
        Ym = eng.ggn_vp(V, scale, 0.0)
        factor_line = dict(value=P / t_f, unit="GGN-vp/s", build_seconds=t_build, ms_per_block=1e3 * t_f,
                           rel_diff_vs_matrix_free=float(((Yf - Ym).abs().max() / Ym.abs().max()).item()),
                           note="mode='factor': GGN = Wm^T Wm with Wm (d=500, D) materialised once; not the headline "
                                "(the matrix-free kernel is), reported because it is what an inducing-point user "
                                "should call")

    # ---- few-probe call shape (the reference applies the operator to one vector at a time) -------------------------
    single_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        single_line = {}
        for Ps in (1, 8):
            Vs = V[:Ps].contiguous()
            fn = lambda: eng.ggn_vp(Vs, scale, alpha)
            fn(); fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            dt1 = time.perf_counter() - t1
            single_line[f"P={Ps}"] = dict(ms_per_product_block=1e3 * dt1 / 50, ggn_vp_per_s=Ps * 50 / dt1)
        single_line["note"] = ("one (GGN + alpha I) v over the 50-example set per call, as a single-right-hand-side CG / Lanczos "
                               "issues it: ~65 dependent launches of 20-60 us at less than one block per CU.  Replaying them "
                               "from a captured HIP graph was measured at +-0 (2.29 vs 2.30 ms at P = 1): the product is bound "
                               "by the kernels' own durations, not by their submission")

    # ---- BASELINE configs[4] slice: full-resolution ResNet-50 (25.6 M parameters, K = 1000), 8 images x 64 probes ---
    r50_line = None
    if args.samples > 0 and not args.no_resnet50 and rank == 0 and world == 1:
        from lip_amd.scalemodels import ResNet50
        from lip_amd.toymodels import create_state as _cs50
        n50, P50 = 8, 64
        st50 = _cs50(ResNet50(1000), seed=1, dtype=torch.float32)
        e50 = LinearizedNet(st50, torch.rand(n50, 224, 224, 3, generator=torch.Generator().manual_seed(3)).to(dev),
                            "classifier", workspace_bytes=64 << 30, max_chunk=P50)
        V50 = krylov.fill_rademacher(P50, e50.D, 11, dev)
        e50.ggn_vp(V50, 1.0, 0.0)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            Y50 = e50.ggn_vp(V50, 1.0, 0.0)
        torch.cuda.synchronize()
        t50 = (time.perf_counter() - t1) / 3
        f50 = sum(e50.flops_per_probe().values())
        r50_line = dict(value=P50 / t50, unit="GGN-vp/s over 8 images", ms_per_block=1e3 * t50, D=e50.D,
                        tflops=f50 * P50 / t50 / 1e12, frac_of_f32_mfma_peak=f50 * P50 / t50 / 1e12 / PEAK_F32_MFMA_TFLOPS,
                        images=n50, probes=P50, finite=bool(torch.isfinite(Y50).all().item()),
                        note="224x224x3 inputs, torchvision-style bottleneck ResNet-50, random init; a slice of configs[4] "
                             "(1024 probes x 10k images over 8 GPUs = 160 such blocks x 16 probe chunks per GPU)")
        del e50, V50, Y50
        torch.cuda.empty_cache()
        # the data sum over MANY images: 256 images as 8 chunks of 32 behind one shared probe workspace
        # (ExampleChunkedGGN — the single-GPU twin of the 8-GPU shard of configs[4]), 16 probes.  (32 chunks of 8 images
        # until late in round 3: 84 TFLOP/s — 16 probes over 8 images leave the deep layers 392 rows per launch.)
        from lip_amd.ggn import ExampleChunkedGGN
        n256, P256 = 256, 16
        Z256 = torch.rand(n256, 224, 224, 3, generator=torch.Generator().manual_seed(4)).to(dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ch = ExampleChunkedGGN(st50.to(device=dev, dtype=torch.float32), Z256, "classifier", full_set_size=10000,
                               example_chunk=32, workspace_bytes=72 << 30, max_probes=P256)
        torch.cuda.synchronize()
        t_bind = time.perf_counter() - t1
        V256 = krylov.fill_rademacher(P256, ch.D, 12, dev)
        t1 = time.perf_counter()
        Y256 = ch(V256, alpha=1.0)
        torch.cuda.synchronize()
        t256 = time.perf_counter() - t1
        r50_line["chunked_256_images"] = dict(
            images=n256, probes=P256, chunks=len(ch.engines), bind_seconds=t_bind, seconds_per_block=t256,
            value=P256 / t256, unit="GGN-vp/s over 256 images", tflops=f50 / n50 * n256 * P256 / t256 / 1e12,
            finite=bool(torch.isfinite(Y256).all().item()),
            note="10k images x 1024 probes over 8 GPUs = 40 such 256-image sums x 64 probe blocks per GPU")
        del ch, V256, Y256, Z256
        torch.cuda.empty_cache()
        # log-marginal-likelihood optimisation in alpha at this scale (configs[4]; src/train_alpha.py:13-59): two
        # inducing images, d = 2 x 1000 — the factor (d x D x 4 B = 204 GB) is NOT materialised, the Gram comes
        # matrix-free from d x (W then W^T) sweeps; 100 Adam steps on log alpha reuse its spectrum
        from lip_amd.train_alpha import fit_alpha
        st50d = st50.to(device=dev, dtype=torch.float32)
        Zl = torch.rand(2, 224, 224, 3, generator=torch.Generator().manual_seed(6)).to(dev)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        a_fit, hist = fit_alpha(Zl, st50d, "classifier", full_set_size=10000, alpha0=1.0, alpha_lr=5e-2, steps=100)
        torch.cuda.synchronize()
        r50_line["lml_alpha_fit"] = dict(seconds=time.perf_counter() - t1, inducing_images=2, d=2000, steps=100,
                                         alpha_start=hist[0][0], alpha_end=a_fit, lml_start=hist[0][1], lml_end=hist[-1][1],
                                         increased=bool(hist[-1][1] > hist[0][1]),
                                         note="Gram W^T W (2000 x 2000) assembled matrix-free: 2000 backward + 2000 tangent "
                                              "sweeps over 2 images at 224 x 224, then O(d) per Adam step")
        from lip_amd.ggn import clear_engine_cache
        clear_engine_cache()
        # one gradient step of the inducing-point objective at this scale (configs[4]; src/train_inducing.py:195-232):
        # the reference's stochastic Hutch++ / SLQ estimate differentiated matrix-free (stochastic_grad.py) — a factor
        # of the data batch would be 8 x 1000 rows of 25.6 M floats (819 GB); here the batch enters through products
        # with the data precision only
        from lip_amd.train_inducing import variational_grad_stochastic
        Xg50 = torch.rand(8, 224, 224, 3, generator=torch.Generator().manual_seed(7)).to(dev)
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        t1 = time.perf_counter()
        v50, g50, i50 = variational_grad_stochastic(Zl, Xg50, st50d, 1.0, key=3, model_type="classifier", full_set_size=10000,
                                                    st_samples=48, slq_samples=2, slq_num_matvecs=4, return_terms=True)
        torch.cuda.synchronize()
        r50_line["inducing_gradient_step"] = dict(
            seconds=time.perf_counter() - t1, inducing_images=2, data_images=8, K=1000, d=2000, st_samples=48, slq_samples=2,
            slq_num_matvecs=4, rank_one_directions=i50["directions"], stage_seconds=i50["stage_seconds"], value=v50, grad_norm=float(g50.norm().item()),
            grad_finite=bool(torch.isfinite(g50).all().item()), peak_memory_GiB=torch.cuda.max_memory_allocated() / 2 ** 30,
            note="value and exact gradient of alternative_objective_scalable's estimate on fixed probes (what "
                 "jax.value_and_grad returns, src/train_inducing.py:196): adjoint of Hutch++ (QR included) and of the "
                 "Golub-Kahan SLQ, 2 (2 s1 + s2) products with the data precision, a matrix-free 2000 x 2000 Gram, one "
                 "shared-direction second-order pass; includes binding both engines")
        clear_engine_cache()
        del Zl, st50d, Xg50, g50
        torch.cuda.empty_cache()

    # ---- the north star's Krylov route: D-space Lanczos on the matrix-free GGN + alpha I (36 matvecs, full re-orth.) ----
    lanczos_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.sample import sample_lanczos
        st_l = state.to(device=dev, dtype=torch.float32)
        Zl = Z.to(dev)
        S_l, k_l = 256, 36
        sample_lanczos(st_l, Zl, eng.D, alpha, 5, "classifier", num_samples=8, full_set_size=full, num_matvecs=4)   # warm-up
        # the 40 GB basis (256 x 36 x D floats) comes out of torch's caching allocator in the timed call: a first hipMalloc of
        # that size takes 1 - 3 s on these boxes (measured: 0.97 s, 3.4 s after a free of the same size; 32 GB: 0.2 ms),
        # which is the allocator's time, not the sampler's
        _pre = torch.empty(S_l * k_l * ((eng.D + 3) // 4 * 4), device=dev, dtype=torch.float32)
        del _pre
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        SL = sample_lanczos(st_l, Zl, eng.D, alpha, 6, "classifier", num_samples=S_l, full_set_size=full, num_matvecs=k_l)
        torch.cuda.synchronize()
        t_l = time.perf_counter() - t1
        lanczos_line = dict(value=S_l / t_l, unit="posterior samples/s", num_samples=S_l, num_matvecs=k_l, seconds=t_l,
                            matvec_share=k_l * (S_l / per_shard) / t_l, finite=bool(torch.isfinite(SL).all().item()),
                            note="(GGN + alpha I)^(-1/2) eps by k-step Lanczos with CGS2 re-orthogonalisation on the "
                                 "matrix-free product, basis storage pre-allocated (block of 256 recurrences: the rate is bound by the products, headline / k = "
                                 f"{per_shard / k_l:.0f} samples/s at most); matvec_share = k * block sweep time at the "
                                 "headline rate / total: the rest is the HBM-bound Krylov kernels and the small eigh")
        del SL

    # ---- CG on (GGN + alpha I) x = b, 16 right-hand sides on the CIFAR binding (the solve inside hutchpp_inv_mvp,
    # src/stochtrace.py:138-148, and any posterior-mean computation): plain float32 CG and CG with range(W) deflated
    # (krylov.RangeDeflation), at the reference's CIFAR experiments' alpha = 10 and at the config's alpha = 0.005 ----
    cg_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.sample import range_deflation
        Bc = krylov.fill_normal(16, eng.D, 77, dev)
        cg_line = dict(rhs=16, tol=3e-3, maxiter=200, stall_guard_deflated=3,
                       note="iterations of the float32 recurrence (JAX's stopping rule, per right-hand side), the TRUE relative "
                            "residual ||A x - b|| / ||b|| afterwards (evaluated in the invariant subspaces range(W) / complement: "
                            "through the raw float32 product it is swamped by eps ||A|| ||x||), wall seconds; deflated = "
                            "range(W) solved exactly in the sampler's eigenbasis, CG on the complement (at alpha = 0.005 the "
                            "plain recurrence cannot converge in float32: cond(A) = 3e9); tol 3e-3 instead of JAX's 1e-5: three times the noise floor of "
                            "the deflated float32 product (8e-4 .. 1.6e-3; AT the floor the iteration count and the error vary from run "
                            "to run with the order of the kernels' float atomics — tests/test_sampler_fullsize.py); a float32-stored x bounds the residual "
                            "from below by ~eps cond (180 at alpha = 0.005), so the forward error against the closed form in the "
                            "invariant subspaces is reported beside it")
        for a_cg in (10.0, 0.005):
            Acg = lambda Vb, a=a_cg: eng.ggn_vp(Vb.contiguous(), scale, a)
            defl = range_deflation(st_l, Zl, eng.D, a_cg, "classifier", full)
            for tag, solver in (("plain", lambda: krylov.cg(Acg, Bc, tol=3e-3, maxiter=200, check_every=10)),
                                ("deflated", lambda: krylov.cg_deflated(Acg, Bc, defl, tol=3e-3, maxiter=200, stall=3))):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                Xc, info_c = solver()
                torch.cuda.synchronize()
                t_c = time.perf_counter() - t1
                res_c = float(defl.relative_residual(Acg, Xc, Bc).max().item())
                Xref_c = defl.closed_form(Bc, lambda lam: 1.0 / lam, a_cg)
                err_c = float(((Xc - Xref_c).norm(dim=1) / Xref_c.norm(dim=1)).max().item())
                cg_line[f"alpha={a_cg:g}_{tag}"] = dict(iterations=int(info_c["iterations"]), true_relative_residual=res_c,
                                                          forward_error_vs_closed_form=err_c, seconds=t_c)
        del Bc, Xc

    # ---- one evaluation batch (scale_experiments/evaluate.py:98-154 times its passes): MC predictive of 256 test images ----
    eval_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.lla import predict_lla_scalable
        st_e = state.to(device=dev, dtype=torch.float32)
        Ze = Z.to(dev)
        Xw = torch.rand(256, 32, 32, 3, generator=torch.Generator().manual_seed(41)).to(dev)
        Xe = torch.rand(256, 32, 32, 3, generator=torch.Generator().manual_seed(42)).to(dev)
        predict_lla_scalable(st_e, Xw, Ze, "classifier", alpha, key=3, full_set_size=full, num_samples=args.samples)   # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        LS = predict_lla_scalable(st_e, Xe, Ze, "classifier", alpha, key=4, full_set_size=full, num_samples=args.samples)
        torch.cuda.synchronize()
        t_e = time.perf_counter() - t1
        eval_line = dict(value=256 / t_e, unit="test images/s", batch=256, mc_samples=args.samples, seconds=t_e,
                         tflops=4 * net.macs_per_example() * 256 * args.samples / t_e / 1e12,
                         note="predict_lla_scalable (src/lla.py:133-156): the S draws (cached sampler parts) pushed through "
                              "one tangent-forward block on an engine bound to the test batch; FLOPs = 4 MACs per "
                              "(image, draw)", finite=bool(torch.isfinite(LS).all().item()))
        del LS, Xw, Xe

    # ---- one exact inducing-point gradient step (src/train_inducing.py:195-232) at this config: 50 inducing images,
    # a data batch of 256, factors + Gram algebra + the second-order pass (reverse over the tangent tape) on the engine --
    ipgrad_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd.train_inducing import variational_grad_scalable, _input_grad_of_pairing
        st_g = state.to(device=dev, dtype=torch.float32)
        Zg = Z.to(dev)
        Xg = torch.rand(256, 32, 32, 3, generator=torch.Generator().manual_seed(77)).to(dev)
        variational_grad_scalable(Zg, Xg, st_g, alpha, model_type="classifier", full_set_size=full, x_chunk=128)    # warm-up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loss_g, gZ = variational_grad_scalable(Zg, Xg, st_g, alpha, model_type="classifier", full_set_size=full, x_chunk=128)
        torch.cuda.synchronize()
        t_g = time.perf_counter() - t1
        Mg = torch.randn(n * 10, eng.D, device=dev)
        t1 = time.perf_counter()
        _input_grad_of_pairing(st_g, Zg, Mg, "classifier")
        torch.cuda.synchronize()
        t_so = time.perf_counter() - t1
        # the reference's own (stochastic) quantity at its defaults: 256 probes (s1 = 240, s2 = 16), 2 x 40-step SLQ
        t1 = time.perf_counter()
        loss_s, gZs, info_s = variational_grad_scalable(Zg, Xg, st_g, alpha, key=5, model_type="classifier", full_set_size=full,
                                                        method="stochastic", st_samples=256, slq_samples=2, return_terms=True)
        torch.cuda.synchronize()
        t_gs = time.perf_counter() - t1
        ipgrad_line = dict(seconds_per_step=t_g, second_order_pass_seconds=t_so, loss=loss_g,
                           stochastic=dict(seconds_per_step=t_gs, st_samples=256, slq_samples=2, slq_num_matvecs=int(n * 0.8),
                                           rank_one_directions=info_s["directions"], stage_seconds=info_s["stage_seconds"], value=loss_s,
                                           grad_finite=bool(torch.isfinite(gZs).all().item()),
                                           cosine_to_exact_gradient=float((gZs.double() * gZ.double()).sum() /
                                                                          (gZs.double().norm() * gZ.double().norm())),
                                           note="method='stochastic': value and exact gradient of the Hutch++ / SLQ estimate on "
                                                "fixed probes (src/train_inducing.py:196), matrix-free"),
                           grad_finite=bool(torch.isfinite(gZ).all().item()), inducing_images=n, data_batch=256,
                           note="value and gradient of the exact KL objective the reference's Hutch++ / SLQ estimators target; "
                                "no torch.func: the input derivative of the parameter-JVP pairing is reverse mode over the "
                                "tangent tape, op by op through lip_engine_run_op (second_order.py)")
        del Mg, Xg, gZ

    # ---- trace-estimator legs, end to end (probe fill + block products + quadratic forms) ----------------------
    trace_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        from lip_amd import stochtrace
        from lip_amd.ggn import BlockOperator
        from lip_amd.scalemodels import LargeClassifier
        trace_line = {}

        def timed(fn, reps=3):
            fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(reps):
                out = fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / reps, float(out)

        # configs[2]: MNIST-MLP (784-1024-512-256-128-10 tanh, D = 1 494 154), 50 synthetic inducing images,
        # 64-probe Girard-Hutchinson trace of GGN + alpha I  (src/stochtrace.py:22-34)
        mlp = LargeClassifier((28, 28, 1), [1024, 512, 256, 128], 4, 10)
        st_m = create_state(mlp, seed=12345, dtype=torch.float32)
        Zm = torch.rand(50, 28, 28, 1, generator=torch.Generator().manual_seed(5)).to(dev)
        em = LinearizedNet(st_m, Zm, "classifier", device=dev, workspace_bytes=2 << 30, max_chunk=64)
        op_m = BlockOperator(lambda B: em.ggn_vp(B, 1.0, 1e-3), (em.D,), (em.D,), em, "mnist")
        t_m, tr_m = timed(lambda: stochtrace.stochastic_trace_estimator_mvp(op_m, em.D, 11, num_samples=64, device=dev))
        fl_m = sum(em.flops_per_probe().values()) * 64
        by_m = 4.0 * em.D * 64 * 4                      # write eps, read eps (tangent weights), write Y, read eps . Y
        trace_line["hutchinson_mnist_mlp_64"] = dict(
            seconds=t_m, trace=tr_m, D=em.D, probes=64, examples=50, tflops=fl_m / t_m / 1e12,
            algorithmic_GBps=by_m / t_m / 1e9, frac_of_hbm_peak=by_m / t_m / 1e9 / 8000.0,
            frac_of_f32_mfma_peak=fl_m / t_m / 1e12 / PEAK_F32_MFMA_TFLOPS,
            note="BASELINE configs[2] (synthetic 28x28 inputs: the MNIST blob is absent): 64 Rademacher probes, one block "
                 "product, 64 dots; per probe the sweep reads its D tangent weights and writes its D cotangents, so this "
                 "small net is HBM-bound: bytes = 4 passes over the (64, D) block")
        # the same estimate from the operator's quadratic forms (what compute_ggn_vp / compute_curvature_approx hand to the
        # estimators: eps^T GGN eps = sum_i ||L_i^T J_i eps||^2 — the tangent sweep alone, no (P, D) output block)
        from lip_amd.ggn import attach_quadratic_forms
        op_mq = attach_quadratic_forms(BlockOperator(op_m.rows, (em.D,), (em.D,), em, "mnist"), em, 1.0, 1e-3)
        t_mq, tr_mq = timed(lambda: stochtrace.stochastic_trace_estimator_mvp(op_mq, em.D, 11, num_samples=64, device=dev))
        trace_line["hutchinson_mnist_mlp_64"]["quadratic_forms"] = dict(
            seconds=t_mq, trace=tr_mq, rel_diff=abs(tr_mq - tr_m) / abs(tr_m), algorithmic_GBps=2.0 * em.D * 64 * 4 / t_mq / 1e9,
            note="tangent sweep + K-vector head only (lip_jvp, LIP_HEAD_LT): bytes = write eps, read eps")
        del em, op_m, op_mq, Zm
        # configs[3]: Hutch++ (s1 = 20, s2 = 16 as src/train_inducing.py:139-146 at st_samples = 36) on the CIFAR
        # binding of the headline: 2 s1 + s2 = 56 products as three blocks + the tall-skinny orthonormalisation
        op_c = BlockOperator(lambda B: eng.ggn_vp(B, scale, alpha), (eng.D,), (eng.D,), eng, "cifar")
        probes36 = krylov.fill_rademacher(36, eng.D, 21, dev)
        t_h, tr_h = timed(lambda: stochtrace.hutchpp_v2(op_c, lambda _: probes36, s1=20, s2=16))
        Y20 = op_c.rows(probes36[:20])
        t_q, _ = timed(lambda: krylov.gram_orthonormalize(Y20).sum(), reps=5)
        t_mv, _ = timed(lambda: op_c.rows(probes36[:20]).sum() + op_c.rows(probes36[:20]).sum() + op_c.rows(probes36[20:]).sum(), reps=2)
        qr_bytes = 2 * 12.0 * eng.D * 20                # two passes of (read for the Gram, read + write for the combination)
        trace_line["hutchpp_v2_cifar_s1_20_s2_16"] = dict(
            seconds=t_h, trace=tr_h, products=56, products_seconds=t_mv, orthonormalisation_seconds=t_q,
            orthonormalisation_GBps=qr_bytes / t_q / 1e9, orthonormalisation_frac_of_hbm_peak=qr_bytes / t_q / 1e9 / 8000.0,
            orthonormalisation_note="host-driven: 2 x (Gram kernel + s x s factorisation through rocSOLVER + combination "
                                    "kernel); the two kernels alone are in `krylov` (dot_nt_f64: float64-FMA-bound at "
                                    "s = 20 — gfx950's float64 peak equals 11 us for this Gram, the HBM time 15 us; "
                                    "rows_combine: HBM-bound)",
            note="BASELINE configs[3]: hutchpp_v2 on GGN + alpha I over the 50 inducing images; the (D x 20) QR of "
                 "src/stochtrace.py:128 runs as a Gram orthonormalisation on lip_dot_nt_f64 + lip_rows_combine, twice: "
                 "bytes 2 x 12 D s (SURVEY 8d: >= 12 D s per pass); Hutchinson with 256 probes on the same binding is the "
                 "headline step plus 256 dots")
        t_256, tr_256 = timed(lambda: stochtrace.stochastic_trace_estimator_mvp(op_c, eng.D, 13, num_samples=256, device=dev), reps=2)
        op_cq = attach_quadratic_forms(BlockOperator(op_c.rows, (eng.D,), (eng.D,), eng, "cifar"), eng, scale, alpha)
        t_256q, tr_256q = timed(lambda: stochtrace.stochastic_trace_estimator_mvp(op_cq, eng.D, 13, num_samples=256, device=dev), reps=2)
        t_hq, tr_hq = timed(lambda: stochtrace.hutchpp_v2(op_cq, lambda _: probes36, s1=20, s2=16))
        trace_line["hutchinson_cifar_256"] = dict(seconds=t_256, trace=tr_256, probes=256,
                                                  note="fill 256 Rademacher probes + one block product + 256 dots",
                                                  quadratic_forms=dict(seconds=t_256q, trace=tr_256q, rel_diff=abs(tr_256q - tr_256) / abs(tr_256),
                                                                       note="fill + one tangent sweep + 256 squared norms in output space"))
        trace_line["hutchpp_v2_cifar_s1_20_s2_16"]["quadratic_forms"] = dict(
            seconds=t_hq, trace=tr_hq, rel_diff=abs(tr_hq - tr_h) / abs(tr_h),
            note="20 products for the sketch, the two quadratic forms (20 + 16 rows) from tangent sweeps")

    krylov_line = None
    if args.samples > 0 and rank == 0 and world == 1:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import krylov_bench
        del eng.work                                   # release the 6.7 GB probe workspace first
        torch.cuda.empty_cache()
        kb = krylov_bench.run(D=1084586, P=256, k=36)
        lz = kb["lanczos_step_cgs2"]["j=35"]
        krylov_line = dict(bound="hbm", peak=8000.0, unit="GB/s",
                           cg_step=kb["cg_step"]["GBps"], lanczos_step_j35=lz["GBps"],
                           lanczos_step_j35_frac_traffic=lz["GBps"] / 8000.0,
                           lanczos_step_j35_frac_survey_model=lz["GBps_survey_model"] / 8000.0,
                           lanczos_note="two fractions: against the bytes the two-phase algorithm must move (each of the "
                                        "two Gram-Schmidt passes reads the basis twice: the projection Q^T w is a full-"
                                        "length reduction that has to finish before w - Q c can start, and the basis — "
                                        "k x 4.3 MB per probe — does not fit on chip), and against SURVEY 8d's "
                                        "4 D (7 + 2 j), which counts one read of the basis per pass",
                           dot_nt_f64=kb["dot_nt_f64"], rows_combine=kb["rows_combine"],
                           hutchinson_dot=kb["bdot"]["GBps"], axpby=kb["axpby"]["GBps"],
                           fill_rademacher=kb["fill_rademacher"]["GBps"],
                           frac_cg_step=kb["cg_step"]["GBps"] / 8000.0,
                           bytes_model="CG step 44*D B, Lanczos step j with CGS2 4*D*(4*(j+2)+4) B (traffic) / 4*D*(7+2j) B (SURVEY), "
                                       "dot 8*D B per probe")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(net, n)

    if rank == 0:
        line = dict(metric="GGN-vector products/sec", value=value, example_probe_products_per_s=value * n_total,
                    per_shard_products_per_s=per_shard, probe_sharded_products_per_s=probe_sharded,
                    unit="GGN-vp/s", n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=ms_per_step, higher_is_better=True, scaling=args.scaling, collective=coll,
                    scaling_note=("strong scaling: ONE data set of %d examples, each rank sweeps its contiguous slice (chunks of "
                                  "50 examples behind one probe workspace), one all-reduce per matvec; value = products over "
                                  "the whole set per second" % n_total) if strong else "weak scaling in the DATA sum: per-GPU work is fixed (50 examples x P probes), the data set "
                                 "grows with N, so `value` (products over the whole set per second) stays flat when scaling "
                                 "is perfect — it is step time that should stay constant; the throughput that grows with N "
                                 "is example_probe_products_per_s (= value x 50 N); probe_sharded_products_per_s is the "
                                 "other sharding of north_star, measured in the same run when N > 1: every rank sweeps its own P "
                                 "probes over a 50-example set, no data-path collective (N P products per step)",
                    vs_baseline=None, dtype="f32", data="synthetic",
                    config=dict(workload="CIFAR-CNN ResNet1M GGN-vp (BASELINE configs[3]): D=1084586, "
                                         f"n={n} examples/GPU, P={P} Rademacher probes/block, alpha=0.005, "
                                         "full_set_size=49000; data sum sharded over ranks, one all-reduce per matvec",
                                examples_per_gpu=n, examples_total=n_total, probes=P, D=eng.D, probe_chunk=eng.chunk,
                                backend=(os.environ.get("LIP_DIST_BACKEND", "nccl") if world > 1 else None),
                                physical_gpus=ndev,
                                unit_note=f"one GGN-vp = one product over the WHOLE data set of {n_total} examples "
                                          f"({n} per GPU, weak scaling: the set grows with N); value = P / step time; "
                                          "per_shard_products_per_s = value * N counts a product per 50-example shard; "
                                          "example_probe_products_per_s = value * examples_total is the figure that "
                                          "grows with N under weak scaling",
                                parallelism=f"data-shard x{world}"),
                    roofline=roofline, cpu_baseline=cpu, posterior_samples=samples_line, factor_mode=factor_line, split_precision=split_line, resnet50=r50_line, lanczos_sampler=lanczos_line, cg_solve=cg_line, eval_batch=eval_line, inducing_gradient_step=ipgrad_line, trace_estimators=trace_line, few_probes=single_line, krylov=krylov_line,
                    checksum=float(Y.double().abs().mean().item()))
        print(json.dumps(line))
    if world > 1:
        dist.barrier()                 # rank 0 did the per-kernel profile steps after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
