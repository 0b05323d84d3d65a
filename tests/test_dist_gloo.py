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
    ret[rank] = (err, err2 + err3)
    dist.destroy_process_group()


def test_sharded_data_sum_world2():
    world = 2
    port = 29500 + (os.getpid() % 2000)
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        err, err2 = ret[r]
        assert err < 1e-10, f"rank {r}: sharded GGN-vp differs by {err}"
        assert err2 < 1e-12, f"rank {r}: gathered W^T differs by {err2}"


def test_shard_bounds_cover():
    from lip_amd.dist import shard_bounds
    for n in (1, 7, 50, 256):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
