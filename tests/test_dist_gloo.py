"""world_size-2 gloo test (CPU) of the data-sum sharding: the all-reduced partial GGN products of two
ranks equal the single-process product over the whole data set."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, ret):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import lip_amd  # noqa: F401
    from lip_amd.dist import ShardedDataSum, gather_rows, shard_bounds, sharded_hutchinson
    from lip_amd.toymodels import SimpleClassifier, create_state
    from oracle.ggn import compute_ggn_vp, compute_W_vps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    net = SimpleClassifier(6, 2, 3)
    st = create_state(net, 5, dtype=torch.float64)
    g = torch.Generator().manual_seed(0)
    Z = torch.randn(7, 2, dtype=torch.float64, generator=g)          # ragged split: 4 + 3
    V = torch.randn(3, 81, dtype=torch.float64, generator=g)          # D = 2*6+6 + 6*6+6 + 6*3+3
    lo, hi = shard_bounds(7, world, rank)
    alpha, N = 0.3, 21
    # local partial sum carries the GLOBAL recalibration N / n_total: oracle's factor is N/M_local -> rescale
    vp_loc = compute_ggn_vp(st, Z[lo:hi], "classifier", full_set_size=N)
    fix = (hi - lo) / 7.0
    def local(B, out=None):
        r = torch.stack([vp_loc(v) for v in B]) * fix
        if out is not None:
            out.copy_(r)
            return out
        return r
    op = ShardedDataSum(local, alpha)
    Y = op(V)
    Yc = ShardedDataSum(local, alpha, chunk=2)(V)          # chunked, async all-reduce per chunk
    assert torch.allclose(Y, Yc, rtol=1e-13, atol=1e-13)
    Yf = ShardedDataSum(local, alpha, chunk=(0.75, 0.25))(V)   # uneven schedule: large first, small last chunk
    assert torch.allclose(Y, Yf, rtol=1e-13, atol=1e-13)
    vp_full = compute_ggn_vp(st, Z, "classifier", full_set_size=N)
    ref = torch.stack([vp_full(v) + alpha * v for v in V])
    err = (Y - ref).abs().max().item()
    # W^T slices gather to the full (M, K) block: equal slices (8 = 4 + 4) and the ragged ones shard_bounds
    # hands out (7 = 4 + 3)
    err2 = 0.0
    for m in (8, 7):
        Zm = torch.randn(m, 2, dtype=torch.float64, generator=g)
        lom, him = shard_bounds(m, world, rank)
        _, WT_loc = compute_W_vps(st, Zm[lom:him], "classifier")
        _, WT_full = compute_W_vps(st, Zm, "classifier")
        U = gather_rows(WT_loc(V[0])[None])[0]
        assert tuple(U.shape) == (m, 3)
        err2 += (U - WT_full(V[0])).abs().max().item()
    # probes sharded over ranks: every rank applies the full operator to its slice, one scalar all-reduce
    probes = torch.sign(torch.randn(5, 81, dtype=torch.float64, generator=g))
    full_op = lambda B: torch.stack([vp_full(v) for v in B])
    tr = sharded_hutchinson(full_op, probes)
    tr_ref = (probes * full_op(probes)).sum() / 5
    err3 = abs(float(tr) - float(tr_ref))
    # a single probe: rank 1's slice is empty — it must still reach the scalar all-reduce (no deadlock)
    tr1 = sharded_hutchinson(full_op, probes[:1])
    err3 += abs(float(tr1) - float((probes[:1] * full_op(probes[:1])).sum()))
    # an operator that all-reduces over the SAME ranks would mix different probes: refused
    try:
        sharded_hutchinson(op, probes)
        err3 += 1.0
    except ValueError:
        pass
    # Krylov consumers driven by the sharded data sum (one all-reduce per matvec inside every iteration): Lanczos
    # f(A) b (src/sample.py:113-126) and Hutch++ (src/stochtrace.py:118-135) — equal to the single-process answer and
    # identical on every rank (all ranks must issue the same sequence of collectives)
    from oracle.matfree import dense_funm_sym_eigh, funm_lanczos_sym, tridiag_sym
    from oracle.stochtrace import hutchpp_v2
    prof = ShardedDataSum(local, alpha, profile=True)
    A_sh = lambda v: prof(v[None])[0]
    A_full = lambda v: vp_full(v) + alpha * v
    est = funm_lanczos_sym(dense_funm_sym_eigh(lambda x: x ** -0.5), tridiag_sym(12))
    b = torch.randn(81, dtype=torch.float64, generator=g)
    x_sh, x_full = est(A_sh, b), est(A_full, b)
    err4 = (x_sh - x_full).abs().max().item() / x_full.abs().max().item()
    hp = torch.sign(torch.randn(24, 81, dtype=torch.float64, generator=g))
    t_sh = hutchpp_v2(A_sh, lambda _: hp, s1=8, s2=16)
    t_full = hutchpp_v2(A_full, lambda _: hp, s1=8, s2=16)
    err4 += abs(float(t_sh) - float(t_full)) / abs(float(t_full))
    mine = torch.cat([x_sh, t_sh.reshape(1)])
    both = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(both, mine)
    err4 += max((o - mine).abs().max().item() for o in both)                 # rank-identical, bit for bit
    st_ = prof.read_stats()
    ok_stats = st_["calls"] == 12 + 2 * 8 + 16 and st_["allreduce_bytes"] == st_["calls"] * 81 * 8 and st_["exposed_wait_ms"] >= 0.0
    ret[rank] = (err, err2 + err3, err4, ok_stats)
    dist.destroy_process_group()


def test_sharded_data_sum_world2():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        err, err2, err4, ok_stats = ret[r]
        assert err < 1e-10, f"rank {r}: sharded GGN-vp differs by {err}"
        assert err2 < 1e-12, f"rank {r}: gathered W^T differs by {err2}"
        assert err4 < 1e-9, f"rank {r}: Lanczos / Hutch++ over the sharded data sum differ by {err4}"
        assert ok_stats, f"rank {r}: collective statistics (one all-reduce per matvec) are off"


def test_shard_bounds_cover():
    from lip_amd.dist import shard_bounds
    for n in (1, 7, 50, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
