"""N > 1 path: row-block sharding + exchange of cuts.

CPU tier (gloo, world_size 2): the host logic that every rank runs -- shard boundaries, block
packing, the two all-gathers, rank-ordered merge -- with the oracle's sweep standing in for the
kernel (test-side only) so that the merged cut set can be compared with a single-process sweep.
GPU tier: the same two-process run on real kernels (gloo exchange, both ranks on cuda:0) against
the single-GPU solve."""
import os
import socket
import sys
import time

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_cut_block(inst, x, f_tol=1e-6):
    """cuts of all violated NL rows of `inst` at x, as (rowptr, col, val, lo, hi) -- oracle arithmetic"""
    from helpers import oracle_evaluator
    from oracle.katana import KatanaFirstOrderSeparator, linear_oa_cut, round_coefs
    sep = KatanaFirstOrderSeparator()
    sep.initialize(None, inst.n, inst.num_constr, oracle_evaluator(inst))
    sep.precompute(x)
    rp, col, val, lo, hi = [0], [], [], [], []
    for i in range(inst.m_lin, inst.num_constr):
        if not sep.isconstrsat(i, inst.l_constr[i], inst.u_constr[i], f_tol):
            cut = linear_oa_cut(sep, x, None, i)
            round_coefs(cut, 1e9)
            col += list(cut.vars); val += list(cut.coeffs); rp.append(len(col))
            lo.append(inst.l_constr[i] - cut.constant); hi.append(inst.u_constr[i] - cut.constant)
    return (np.array(rp, dtype=np.int64), np.array(col, dtype=np.int32), np.array(val), np.array(lo), np.array(hi))


def _worker_exchange(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import exchange_cuts, shard_instance
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(n=800, m_nl=81, k=12, family="explog", seed=13)
    x = np.clip(inst.xhat + 0.9, inst.l_var, inst.u_var)
    local = shard_instance(inst, rank, world)
    blocks = exchange_cuts(dist, _oracle_cut_block(local, x), "cpu")
    # a second, empty round (late ECP iterations exchange nothing)
    empty = (np.zeros(1, dtype=np.int64), np.zeros(0, dtype=np.int32), np.zeros(0), np.zeros(0), np.zeros(0))
    blocks2 = exchange_cuts(dist, empty, "cpu")
    # with ids and the two piggybacked scalars (max violation, status flag): maxima over ranks come back with the blocks
    ids = np.arange(len(_oracle_cut_block(local, x)[3]), dtype=np.int64) + 1000 * rank
    blocks3, mx_a, mx_b = exchange_cuts(dist, tuple(_oracle_cut_block(local, x)) + (ids,), "cpu", scalars=(1.5 * (rank + 1), float(rank)))
    out[rank] = (blocks, [len(b[3]) for b in blocks2], [b[5] for b in blocks3], mx_a, mx_b)
    dist.destroy_process_group()


def test_exchange_merges_shard_cuts_into_the_single_process_cut_set():
    import katana_jl_amd as ktn
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_exchange, args=(world, _free_port(), out), nprocs=world, join=True)
    inst = ktn.instances.make_instance(n=800, m_nl=81, k=12, family="explog", seed=13)
    x = np.clip(inst.xhat + 0.9, inst.l_var, inst.u_var)
    full = _oracle_cut_block(inst, x)
    assert len(full[3]) > 2
    for rank in range(world):
        blocks, empty_counts, ids3, mx_a, mx_b = out[rank]
        assert empty_counts == [0, 0]
        assert mx_a == 1.5 * world and mx_b == float(world - 1)
        assert all(np.array_equal(ids3[r], np.arange(len(ids3[r])) + 1000 * r) for r in range(world))
        rp = np.concatenate([[0]] + [b[0][1:] + off for b, off in
                                     zip(blocks, np.cumsum([0] + [len(b[1]) for b in blocks[:-1]]))])
        col = np.concatenate([b[1] for b in blocks]); val = np.concatenate([b[2] for b in blocks])
        lo = np.concatenate([b[3] for b in blocks]); hi = np.concatenate([b[4] for b in blocks])
        # rank-ordered merge of contiguous row blocks == the single-process row order, bit for bit
        assert np.array_equal(rp, full[0]) and np.array_equal(col, full[1]) and np.array_equal(val, full[2])
        assert np.array_equal(lo, full[3]) and np.array_equal(hi, full[4])


def test_shards_partition_the_nl_rows():
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import shard_bounds, shard_instance
    inst = ktn.instances.make_instance(n=300, m_nl=37, k=8, family="quad", seed=1)
    for world in (1, 2, 3, 8):
        cover = []
        for r in range(world):
            lo, hi = shard_bounds(inst.m_nl, r, world)
            cover += list(range(lo, hi))
            s = shard_instance(inst, r, world)
            assert s.m_lin == inst.m_lin and s.m_nl == hi - lo and s.num_constr == inst.m_lin + hi - lo
            a, b = inst.rowptr[inst.m_lin + lo], inst.rowptr[inst.m_lin + hi]
            assert np.array_equal(s.p0[s.rowptr[s.m_lin]:], inst.p0[a:b])
            assert np.array_equal(s.u_constr[:s.m_lin], inst.u_constr[:inst.m_lin])
        assert cover == list(range(inst.m_nl))


def test_pack_unpack_roundtrip_with_empty_and_ragged_blocks():
    from katana_jl_amd.distributed import pack_block, unpack_block
    rng = np.random.default_rng(0)
    for lens in ([], [3], [0, 5, 1], [2 ** 20 % 7, 40]):
        rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        nnz = int(rp[-1])
        col = rng.integers(0, 2 ** 31 - 1, nnz).astype(np.int32)
        val, lo, hi = rng.normal(size=nnz), np.full(len(lens), -np.inf), rng.normal(size=len(lens))
        if len(lens):
            hi[0] = np.nan
        got = unpack_block(pack_block(rp, col, val, lo, hi), len(lens), nnz)
        assert np.array_equal(got[0], rp) and np.array_equal(got[1], col) and np.array_equal(got[2], val)
        assert np.array_equal(got[3], lo) and np.array_equal(got[4], hi, equal_nan=True)
        ids = rng.integers(0, 2 ** 40, len(lens))                   # global NL-row ids ride along as a sixth array
        got = unpack_block(pack_block(rp, col, val, lo, hi, ids), len(lens), nnz, with_ids=True)
        assert len(got) == 6 and np.array_equal(got[5], ids) and np.array_equal(got[1], col)


class _RecordingLp:
    """stands in for the engine handle behind ShardedKatanaModel._exchange: rows [m0, M) of "its LP" are the block given at
    construction; truncate / append are recorded"""
    def __init__(self, m0, block, slots):
        self.m0, self.block, self.slots = m0, block, slots
        self.rows = m0 + len(block[3])
        self.appended = []

    def lp_rows_from(self, first):
        assert first == self.m0
        return self.block

    def last_sweep_slots(self):
        return self.slots

    def lp_truncate(self, first):
        assert first == self.m0
        self.rows = first

    def lp_append_rows(self, rp, col, val, lo, hi, ids):
        self.appended.append((np.array(rp), np.array(col), np.array(val), np.array(lo), np.array(hi), np.array(ids)))
        self.rows += len(lo)


def _worker_exchange_callback(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from katana_jl_amd.distributed import ShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rng = np.random.default_rng(100 + rank)
    nrows = 3 + 2 * rank                                             # ragged: 3 rows on rank 0, 5 on rank 1
    lens = rng.integers(1, 6, nrows)
    rp = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    block = (rp, rng.integers(0, 50, int(rp[-1])).astype(np.int32), rng.normal(size=int(rp[-1])),
             np.full(nrows, -np.inf), rng.normal(size=nrows))
    shard_lo = 40 * rank
    obj = ShardedKatanaModel.__new__(ShardedKatanaModel)             # the callback alone: no engine, no GPU
    obj.rank, obj.world, obj.dist, obj.exchange_device, obj.shard_lo = rank, world, dist, "cpu", shard_lo
    obj.exchanged_rows = 0
    res = {}
    # round 1: both ranks have rows; scalars = (unused, max violation, flag, extra0, extra1)
    obj.m = _RecordingLp(17, block, np.arange(nrows, dtype=np.int64))
    sc = [0.0, 0.25 * (rank + 1), float(rank), -1.0 - rank, 7.0]
    rc = obj._exchange(None, 0, 17, sc, 5)
    res["r1"] = (rc, list(sc), obj.m.rows, obj.m.appended, block)
    # round 2: rank 1's LP failed (flag 2): nothing is appended anywhere, every rank sees the flag
    obj.m = _RecordingLp(17, block, np.arange(nrows, dtype=np.int64))
    sc = [0.0, 0.5, 2.0 if rank == 1 else 0.0, 0.0, 0.0]
    rc = obj._exchange(None, 0, 17, sc, 5)
    res["r2"] = (rc, list(sc), obj.m.rows, len(obj.m.appended))
    # what = 1: the sum over the ranks, in place
    sc = [1.0 + rank, -2.0 * (rank + 1), 0.5]
    rc = obj._exchange(None, 1, 0, sc, 3)
    res["sum"] = (rc, list(sc))
    res["exchanged_rows"] = obj.exchanged_rows
    out[rank] = res
    dist.destroy_process_group()


def test_engine_side_exchange_callback_protocol_over_gloo():
    """what Engine::step sees through ktn_set_cut_exchange (include/katana_hip.h): totals, maxima, sums and -- the point of the
    exchange -- the same rows in the same order in every rank's LP, ids shifted to the global NL-row numbering"""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_exchange_callback, args=(world, _free_port(), out), nprocs=world, join=True)
    blocks = [out[r]["r1"][4] for r in range(world)]
    for rank in range(world):
        rc, sc, rows, appended, _ = out[rank]["r1"]
        assert rc == 0
        assert sc[0] == 8.0 and sc[1] == 0.5 and sc[2] == 1.0 and sc[3] == -1.0 and sc[4] == 7.0
        assert rows == 17 + 8 and len(appended) == world
        for r in range(world):
            for got, want in zip(appended[r][:5], blocks[r]):
                assert np.array_equal(got, want)
            assert np.array_equal(appended[r][5], 40 * r + np.arange(3 + 2 * r))
        rc, sc, rows, napp = out[rank]["r2"]
        assert rc == 0 and sc[0] == 0.0 and sc[2] == 2.0 and rows == 17 and napp == 0
        rc, sc = out[rank]["sum"]
        assert rc == 0 and sc == [3.0, -6.0, 1.0]
        assert out[rank]["exchanged_rows"] == 8


def test_exchange_callback_reports_failure_instead_of_raising_across_the_abi(capsys):
    from katana_jl_amd.distributed import ShardedKatanaModel
    obj = ShardedKatanaModel.__new__(ShardedKatanaModel)
    obj.rank, obj.world, obj.dist, obj.exchange_device, obj.shard_lo, obj.exchanged_rows = 0, 1, None, "cpu", 0, 0

    class Broken:
        def lp_rows_from(self, first):
            raise RuntimeError("device lost")
    obj.m = Broken()
    assert obj._exchange(None, 0, 3, [0.0, 1.0, 0.0, 0.0, 0.0], 5) == 1
    assert "cut-exchange callback failed" in capsys.readouterr().err


def _worker_gpu(rank, world, port, out, inst_kw=None, solver_kw=None, exchange_device=None):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import ShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(**(inst_kw or dict(n=4000, m_nl=400, k=16, family="explog", seed=21)))
    m = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0, **(solver_kw or dict(purge_age=0))), inst, rank, world, dist,
                           exchange_device=exchange_device)
    st = m.optimize()
    out[rank] = (st, m.getobjval(), m.numiters(), m.numcuts(), m.getsolution(), m.purged_rows, m.m.lp_num_rows())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_solve_matches_single_gpu():
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance, max_nl_violation, planted_obj_bound
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_gpu, args=(world, _free_port(), out), nprocs=world, join=True)
    inst = ktn.instances.make_instance(n=4000, m_nl=400, k=16, family="explog", seed=21)
    single = hip_load_instance(ktn, inst, purge_age=0)
    assert single.optimize() == "Optimal"
    (s0, o0, it0, c0, x0, *_), (s1, o1, it1, c1, x1, *_) = out[0], out[1]
    assert s0 == s1 == "Optimal"
    assert o0 == o1 and it0 == it1 and c0 == c1 and np.array_equal(x0, x1)     # replicated LP: identical ranks
    # rank-ordered contiguous blocks == single-process row order => the very same trajectory
    assert o0 == single.getobjval() and it0 == single.numiters() and c0 == single.numcuts()
    assert_planted_objective(o0, inst)
    assert max_nl_violation(inst, x0) <= 1e-6 * (1 + 1e-6)


@pytest.mark.gpu
def test_two_rank_sharded_solve_with_the_device_resident_cut_exchange():
    """the cut blocks never leave device memory: packed by the engine into a torch CUDA tensor (ktn_lp_pack_rows_dev), gathered,
    appended from the receive buffers (ktn_lp_append_packed_dev).  Two ranks share cuda:0 here, so the gather itself runs over
    gloo (which stages CUDA tensors through the host); over RCCL the same tensors go GPU to GPU.  The trajectory is the one of
    the host-staged exchange and of the single-GPU solve, bit for bit -- with purging and deepest-cut selection on as well."""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance, max_nl_violation, planted_obj_bound
    world = 2
    for solver_kw in (dict(purge_age=0), dict(purge_age=2, purge_min_rows=50, cut_cap_factor=0.02, cut_cap_min=20)):
        out_dev, out_host = mp.Manager().dict(), mp.Manager().dict()
        mp.spawn(_worker_gpu, args=(world, _free_port(), out_dev, None, solver_kw, "cuda"), nprocs=world, join=True)
        mp.spawn(_worker_gpu, args=(world, _free_port(), out_host, None, solver_kw, "cpu"), nprocs=world, join=True)
        inst = ktn.instances.make_instance(n=4000, m_nl=400, k=16, family="explog", seed=21)
        for r in range(world):
            assert out_dev[r][0] == "Optimal"
            assert out_dev[r][1] == out_host[r][1] and out_dev[r][2] == out_host[r][2] and out_dev[r][3] == out_host[r][3]
            assert np.array_equal(out_dev[r][4], out_host[r][4]) and out_dev[r][6] == out_host[r][6]
        assert out_dev[0][1] == out_dev[1][1] and np.array_equal(out_dev[0][4], out_dev[1][4])
        assert_planted_objective(out_dev[0][1], inst)
        assert max_nl_violation(inst, out_dev[0][4]) <= 1e-6 * (1 + 1e-6)
    single = hip_load_instance(ktn, inst, purge_age=0)
    assert single.optimize() == "Optimal"


def _worker_gpu_cert(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import ShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_config("cfg3", seed=6)
    m = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, exchange_device="cuda")
    st = m.optimize()
    out[rank] = (st, m.getobjval(), m.numiters(), m.stat("cert_refinements"), m.stat("polish_iters"), m.getsolution())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_sharded_solve_refines_by_the_objective_certificate():
    """cfg3 seed 6 meets the stop rule 1.5e-6 (relative) away from the optimum; the engine's own loop then keeps cutting below
    f_tol until the objective certificate -- sum over the NL rows of multiplier mass x residual -- is within half the
    reference's objective tolerance.  With the NL rows split over two ranks every rank contributes its block's share
    (ktn_objective_certificate, summed through the exchange callback) and every rank's engine takes the same decision: one refinement, the objective within 1e-6 / 1e-6 of
    the planted one, the single-GPU trajectory bit for bit."""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_gpu_cert, args=(world, _free_port(), out), nprocs=world, join=True)
    inst = ktn.instances.make_config("cfg3", seed=6)
    single = hip_load_instance(ktn, inst)
    assert single.optimize() == "Optimal" and single.stat("cert_refinements") == 1
    for r in range(world):
        st, obj, iters, refinements, polish, x = out[r]
        assert st == "Optimal" and refinements == 1 and polish == single.stat("polish_iters")
        assert obj == single.getobjval() and iters == single.numiters() and np.array_equal(x, single.getsolution())
    assert_planted_objective(out[0][1], inst)


@pytest.mark.gpu
def test_sharded_model_world1_equals_engine_loop():
    import katana_jl_amd as ktn
    from helpers import hip_load_instance
    from katana_jl_amd.distributed import ShardedKatanaModel
    inst = ktn.instances.make_instance(n=2000, m_nl=200, k=16, family="quad", seed=5)
    a = ShardedKatanaModel(ktn.KatanaSolver(log_level=0), inst, 0, 1, None)
    assert a.optimize() == "Optimal"
    b = hip_load_instance(ktn, inst, purge_age=0)
    assert b.optimize() == "Optimal"
    assert a.getobjval() == b.getobjval() and a.numiters() == b.numiters() and a.numcuts() == b.numcuts()


@pytest.mark.gpu
def test_two_rank_sharded_solve_with_purging_and_cut_selection():
    """the sharded loop purges idle cuts (ktn_lp_purge) and splits the deepest-cut cap over the ranks; the replicated
    LPs stay identical and the solve ends at the planted optimum"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, max_nl_violation, planted_obj_bound
    world = 2
    inst_kw = dict(n=600, m_nl=6000, k=10, family="explog", seed=23)
    solver_kw = dict(purge_age=2, purge_min_rows=300, cut_cap_factor=1.0, cut_cap_min=200)
    out = mp.Manager().dict()
    mp.spawn(_worker_gpu, args=(world, _free_port(), out, inst_kw, solver_kw), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    (s0, o0, it0, c0, x0, p0, r0), (s1, o1, it1, c1, x1, p1, r1) = out[0], out[1]
    assert s0 == s1 == "Optimal"
    assert o0 == o1 and it0 == it1 and c0 == c1 and p0 == p1 and r0 == r1 and np.array_equal(x0, x1)
    assert p0 > 0 and c0 < it0 * inst.m_nl                    # rows were purged; not every violated row was cut
    assert_planted_objective(o0, inst)
    assert max_nl_violation(inst, x0) <= 1e-6 * (1 + 1e-6)


@pytest.mark.gpu
def test_sharded_world1_with_purging_equals_engine_loop():
    import katana_jl_amd as ktn
    from helpers import hip_load_instance
    from katana_jl_amd.distributed import ShardedKatanaModel
    inst = ktn.instances.make_instance(n=600, m_nl=6000, k=10, family="explog", seed=23)
    kw = dict(purge_age=2, purge_min_rows=300, cut_cap_factor=1.0, cut_cap_min=200)
    a = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, **kw), inst, 0, 1, None)
    assert a.optimize() == "Optimal" and a.purged_rows > 0
    b = hip_load_instance(ktn, inst, **kw)
    assert b.optimize() == "Optimal"
    assert a.getobjval() == b.getobjval() and a.numiters() == b.numiters() and a.numcuts() == b.numcuts()


# =====================================================================================================================
# Row-sharded LP (include/katana_hip.h "row-sharded LP")
# =====================================================================================================================
def test_row_shards_partition_linear_and_nl_rows():
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import shard_rows
    inst = ktn.instances.make_instance(n=300, m_nl=37, k=8, family="quad", seed=1)
    for world in (1, 2, 3, 8):
        seen_rows, nnz = [], 0
        for r in range(world):
            s = shard_rows(inst, r, world)
            assert s.n == inst.n and np.array_equal(s.obj_p0, inst.obj_p0)          # variables and objective replicated
            assert s.num_constr == s.m_lin + s.m_nl and s.rowptr[-1] == len(s.col)
            # every shard row is a row of the instance, bit for bit
            l0 = (inst.m_lin * r) // world
            n0 = (inst.m_nl * r) // world
            for k in range(s.num_constr):
                g = l0 + k if k < s.m_lin else inst.m_lin + n0 + (k - s.m_lin)
                seen_rows.append(g)
                a, b = inst.rowptr[g], inst.rowptr[g + 1]
                assert np.array_equal(s.col[s.rowptr[k]:s.rowptr[k + 1]], inst.col[a:b])
                assert np.array_equal(s.p0[s.rowptr[k]:s.rowptr[k + 1]], inst.p0[a:b])
                assert s.u_constr[k] == inst.u_constr[g] and s.rconst[k] == inst.rconst[g]
            nnz += len(s.col)
        assert sorted(seen_rows) == list(range(inst.num_constr)) and nnz == len(inst.col)   # a partition


def _pdhg_reflected_halpern(A_local, c, l, u, lo, hi, iters, tau, sigma, allreduce):
    """the engine's iteration (csrc/kernels.hpp k_x_prox / k_pdhg_y) in numpy; `allreduce` sums an n-vector over the ranks"""
    n, m = len(c), A_local.shape[0]
    x, y = np.zeros(n), np.zeros(m)
    x0, y0 = x.copy(), y.copy()
    for k in range(iters):
        w = (k + 1.0) / (k + 2.0)
        aty = allreduce(A_local.T @ y)                               # local partial -> sum over ranks
        xt = np.clip(x - tau * (c - aty), l, u)
        xbar = 2 * xt - x
        x = w * (2 * xt - x) + (1 - w) * x0
        v = y - sigma * (A_local @ xbar)
        yt = v + sigma * np.clip(-v / sigma, lo, hi)
        y = w * (2 * yt - y) + (1 - w) * y0
    return x, y


def _lin_block(s):
    import scipy.sparse as sp
    rp = s.rowptr[:s.m_lin + 1]
    A = sp.csr_matrix((s.p0[:rp[-1]], s.col[:rp[-1]], rp), shape=(s.m_lin, s.n))
    return A, s.l_constr[:s.m_lin], s.u_constr[:s.m_lin]


def _worker_rowshard_logic(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ctypes as C
    import torch
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import make_allreduce_callback, shard_rows
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(n=300, m_nl=20, k=8, family="explog", seed=4)
    s = shard_rows(inst, rank, world)
    A, lo, hi = _lin_block(s)
    c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0

    def allreduce(v):
        t = torch.from_numpy(np.ascontiguousarray(v)); dist.all_reduce(t); return t.numpy()
    x, y = _pdhg_reflected_halpern(A, c, inst.l_var, inst.u_var, lo, hi, 60, 0.05, 0.05, allreduce)
    # the transport the engine uses with gloo: the ctypes callback reduces a host buffer in place (op 0 sum, 1 max)
    cb = make_allreduce_callback(dist)
    buf = np.array([1.0 + rank, -2.0 * rank, 7.0], dtype=np.float64)
    rc_sum = cb(None, buf.ctypes.data_as(C.POINTER(C.c_double)), 3, 0)
    buf2 = np.array([1.0 + rank, -2.0 * rank, 7.0], dtype=np.float64)
    rc_max = cb(None, buf2.ctypes.data_as(C.POINTER(C.c_double)), 3, 1)
    out[rank] = (x, y, rc_sum, buf.copy(), rc_max, buf2.copy())
    dist.destroy_process_group()


def test_row_sharded_iteration_equals_the_unsharded_one_over_gloo():
    """world-size-2 CPU test of the partition and all-reduce logic of the row-sharded LP: x replicated, y local, A'y = local
    partial + all-reduce; the iterates are those of the unsharded iteration (up to the summation order of A'y)"""
    import katana_jl_amd as ktn
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_logic, args=(world, _free_port(), out), nprocs=world, join=True)
    inst = ktn.instances.make_instance(n=300, m_nl=20, k=8, family="explog", seed=4)
    A, lo, hi = _lin_block(inst)
    c = np.zeros(inst.n); c[inst.obj_col] = inst.obj_p0
    x, y = _pdhg_reflected_halpern(A, c, inst.l_var, inst.u_var, lo, hi, 60, 0.05, 0.05, lambda v: v)
    assert np.array_equal(out[0][0], out[1][0])                                      # x bit-identical on both ranks
    assert np.max(np.abs(out[0][0] - x)) <= 1e-13 * max(1.0, np.max(np.abs(x)))
    ycat = np.concatenate([out[0][1], out[1][1]])                                    # y blocks in rank order = the rows in order
    assert np.max(np.abs(ycat - y)) <= 1e-13 * max(1.0, np.max(np.abs(y)))
    for r in range(world):
        assert out[r][2] == 0 and out[r][4] == 0
        assert np.array_equal(out[r][3], [3.0, -2.0, 14.0]) and np.array_equal(out[r][5], [2.0, 0.0, 7.0])


def _worker_rowshard_gpu(rank, world, port, out, inst_kw, tiled=False, transport="auto"):
    sys.path.insert(0, ROOT)
    if tiled:
        os.environ["KTN_TILED"] = "1"          # every LP of this process runs its steps and checks from the tiled copies
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(**inst_kw)
    # (a) one LP, tight tolerances: linear rows + the cuts of one sweep at a common point
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0, purge_age=0, cut_cap_factor=0.0), inst, rank, world, dist,
                              transport=transport)
    sep = ktn.KatanaHipSeparator(m.m); sep.initialize()
    sep.precompute(np.clip(inst.xhat + 0.7, inst.l_var, inst.u_var))
    sep.sweep(1e-6)
    st, it = m.lp_solve(1e-10, 1e-10)
    lp = (st, m.getobjval(), m.lp_num_rows(), m.stat("allreduce_calls"), m.transport)
    # (b) the whole ECP solve
    m2 = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, transport=transport)
    status = m2.optimize()
    out[rank] = (lp, status, m2.getobjval(), m2.getsolution(), m2.numiters(), m2.numcuts_global(), m2.lp_num_rows(),
                 m.stat("lp_tiled_builds") + m2.stat("lp_tiled_builds"), m2.stat("pdhg_iters"), m2.allreduce_probe(inst.n, 3))
    dist.barrier()                              # peer-buffer transport: the ranks leave together
    del m, m2
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_row_sharded_lp_equals_the_single_gpu_lp():
    """2 ranks (both on cuda:0, gloo transport through the host callback): the row-sharded LP reaches the objective of the
    same LP on one handle to 1e-9, and the row-sharded ECP solve ends at the planted optimum with the same x on both ranks"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance, max_nl_violation, planted_obj_bound
    world = 2
    inst_kw = dict(n=400, m_nl=60, k=10, family="explog", seed=11)
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_gpu, args=(world, _free_port(), out, inst_kw), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    one = hip_load_instance(ktn, inst, purge_age=0, cut_cap_factor=0.0, lp_dense_after=0)
    sep = ktn.KatanaHipSeparator(one); sep.initialize()
    sep.precompute(np.clip(inst.xhat + 0.7, inst.l_var, inst.u_var))
    sep.sweep(1e-6)
    st, _ = one.lp_solve(1e-10, 1e-10)
    assert st == "Optimal"
    (st0, obj0, rows0, calls0, tr0), (st1, obj1, rows1, calls1, tr1) = out[0][0], out[1][0]
    assert st0 == st1 == "Optimal" and tr0 == tr1 == "callback" and calls0 == calls1 > 0
    assert rows0 + rows1 == one.lp_num_rows()                                        # the same rows, split
    assert obj0 == obj1                                                              # identical on every rank
    assert abs(obj0 - one.getobjval()) <= 1e-9 * max(1.0, abs(one.getobjval()))
    # whole solve
    for r in range(world):
        assert out[r][1] == "Optimal"
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3]) and out[0][4] == out[1][4]
    assert_planted_objective(out[0][2], inst)
    assert max_nl_violation(inst, out[0][3]) <= 1e-6 * (1 + 1e-6)
    assert out[0][5] == out[1][5] >= inst.m_lin and out[0][6] + out[1][6] >= inst.m_lin


@pytest.mark.gpu
def test_two_rank_row_sharded_lp_through_the_tiled_copies():
    """the same with every rank's block served by the tiled SpMV (what a rank of a 2-GPU cfg4 run does: 7e6 local entries): the
    partial A_r'y_r comes from k_spmv_tiled + k_tile_vec before the all-reduce, the checks from the tiled passes; both ranks end
    with the same x, at the planted optimum"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, max_nl_violation, planted_obj_bound
    world = 2
    inst_kw = dict(n=20000, m_nl=2000, k=16, family="explog", seed=3)
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_gpu, args=(world, _free_port(), out, inst_kw, True), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    for r in range(world):
        assert out[r][0][0] == "Optimal" and out[r][1] == "Optimal" and out[r][7] >= 2
    assert out[0][0][1] == out[1][0][1]
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3]) and out[0][4] == out[1][4]
    assert_planted_objective(out[0][2], inst)
    assert max_nl_violation(inst, out[0][3]) <= 1e-6 * (1 + 1e-6)


@pytest.mark.gpu
def test_two_rank_row_sharded_lp_over_the_peer_buffer_transport():
    """2 processes on cuda:0, each mapping the other's exposed buffers (hipIpcGetMemHandle -> gloo all_gather of the handles ->
    hipIpcOpenMemHandle): partials written into the exposed slot, one signal-and-wait kernel, sums taken by the consumer in rank
    order.  With two ranks a + b is the same double whichever transport adds it, so the whole trajectory -- every LP, every
    cut, the final x -- must equal the host-callback transport's bit for bit; the LP objective equals the single-handle LP's to
    1e-9.  (What one GPU cannot show is the visibility of a PEER GPU's stores: the two processes share the L2s.  The
    protocol's rules for that are in kernels.hpp "peer-buffer transport".)"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance, max_nl_violation, planted_obj_bound
    world = 2
    inst_kw = dict(n=400, m_nl=60, k=10, family="explog", seed=11)
    out, ref = mp.Manager().dict(), mp.Manager().dict()
    mp.spawn(_worker_rowshard_gpu, args=(world, _free_port(), out, inst_kw, False, "ipc"), nprocs=world, join=True)
    mp.spawn(_worker_rowshard_gpu, args=(world, _free_port(), ref, inst_kw, False, "callback"), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    one = hip_load_instance(ktn, inst, purge_age=0, cut_cap_factor=0.0, lp_dense_after=0)
    sep = ktn.KatanaHipSeparator(one); sep.initialize()
    sep.precompute(np.clip(inst.xhat + 0.7, inst.l_var, inst.u_var))
    sep.sweep(1e-6)
    st, _ = one.lp_solve(1e-10, 1e-10)
    assert st == "Optimal"
    for r in range(world):
        lp, status = out[r][0], out[r][1]
        assert lp[0] == "Optimal" and lp[4] == "ipc" and status == "Optimal" and lp[3] > 0
        assert abs(lp[1] - one.getobjval()) <= 1e-9 * max(1.0, abs(one.getobjval()))
        assert lp[1] == ref[r][0][1]                                                 # the LP: same bits as the callback transport
        assert out[r][2] == ref[r][2] and np.array_equal(out[r][3], ref[r][3])       # the solve: objective and x
        assert out[r][4] == ref[r][4] and out[r][5] == ref[r][5] and out[r][8] == ref[r][8]   # ECP iterations, cuts, PDHG iterations
        assert out[r][9][1] <= 1e-12                                                 # the probe: sum and max as every rank computes them itself
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3])
    assert_planted_objective(out[0][2], inst)
    assert max_nl_violation(inst, out[0][3]) <= 1e-6 * (1 + 1e-6)


def _worker_rowshard_probe(rank, world, port, out, inst_kw, fail_rank):
    sys.path.insert(0, ROOT)
    if fail_rank is not None:
        os.environ["KTN_DIST_PROBE_FAIL"] = str(fail_rank)      # that rank reports a failed export: every rank must fall back
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(**inst_kw)
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, transport="probe")
    status = m.optimize()
    out[rank] = (m.transport, status, m.getobjval(), m.getsolution(), m.allreduce_probe(inst.n, 2)[1])
    dist.barrier()
    del m
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("fail_rank", [None, 1])
def test_transport_is_chosen_by_the_probe_at_init(fail_rank):
    """transport = "probe" (what "auto" resolves to over the nccl backend with several GPUs) tries the
    peer-buffer transport and keeps it only where every rank exported, mapped and passed ktn_dist_allreduce_probe; one rank
    voting no (here: simulated) makes EVERY rank leave it again (ktn_dist_release_ipc) and take the fallback -- in the same
    process, decided once, before the problem is loaded.  Either way the solve ends at the planted optimum with the same x."""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective
    world = 2
    inst_kw = dict(n=400, m_nl=60, k=10, family="explog", seed=11)
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_probe, args=(world, _free_port(), out, inst_kw, fail_rank), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    want = "ipc" if fail_rank is None else "callback"
    for r in range(world):
        assert out[r][0] == want and out[r][1] == "Optimal" and out[r][4] <= 1e-12
    assert out[0][2] == out[1][2] and np.array_equal(out[0][3], out[1][3])
    assert_planted_objective(out[0][2], inst)


@pytest.mark.gpu
def test_three_rank_row_sharded_solve_over_the_peer_buffer_transport():
    """three processes on cuda:0: more than one peer per rank (the flag words per source, the alternating slots); every rank ends
    with the same x, bit for bit, at the planted optimum"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, max_nl_violation, planted_obj_bound
    world = 3
    inst_kw = dict(n=3000, m_nl=300, k=16, family="explog", seed=2)
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_gpu, args=(world, _free_port(), out, inst_kw, False, "ipc"), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    for r in range(world):
        assert out[r][0][0] == "Optimal" and out[r][0][4] == "ipc" and out[r][1] == "Optimal"
        assert out[r][2] == out[0][2] and np.array_equal(out[r][3], out[0][3]) and out[r][4] == out[0][4]
        assert out[r][9][1] <= 1e-12
    assert_planted_objective(out[0][2], inst)
    assert max_nl_violation(inst, out[0][3]) <= 1e-6 * (1 + 1e-6)


def _worker_ipc_absent_peer(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["KTN_IPC_TIMEOUT_S"] = "1"
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(n=200, m_nl=20, k=8, family="explog", seed=1)
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, transport="ipc")
    t0 = time.time()
    if rank == 0:                               # rank 1 never enters this collective
        try:
            m.allreduce_probe(100, 1)
            out[0] = ("no error", time.time() - t0)
        except Exception as e:
            out[0] = (str(e), time.time() - t0)
    dist.barrier()
    del m
    out[10 + rank] = time.time() - t0
    dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_buffer_transport_times_out_instead_of_hanging():
    """every spin of the transport is bounded: a rank whose peer never arrives gets an error naming that peer after
    KTN_IPC_TIMEOUT_S, and both processes still shut down"""
    out = mp.Manager().dict()
    mp.spawn(_worker_ipc_absent_peer, args=(2, _free_port(), out), nprocs=2, join=True)
    msg, secs = out[0]
    assert "timed out waiting for rank 1" in msg and 0.9 <= secs < 10.0
    assert out[10] < 15.0 and out[11] < 15.0


def _worker_ipc_absent_peer_solve(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["KTN_IPC_TIMEOUT_S"] = "1"
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(n=200, m_nl=20, k=8, family="explog", seed=1)
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, transport="ipc")
    t0 = time.time()
    if rank == 0:                               # rank 1 never enters the solve: every barrier of the PDHG loop lacks its peer
        try:
            m.m.optimize()
            out[0] = ("no error", time.time() - t0)
        except Exception as e:
            out[0] = (str(e), time.time() - t0)
    dist.barrier()
    del m
    out[10 + rank] = time.time() - t0
    dist.destroy_process_group()


@pytest.mark.gpu
def test_peer_buffer_failure_is_sticky_one_timeout_not_one_per_barrier():
    """A solve enqueues up to lp_check_every barriers between two looks at the error word.  The first barrier that times out
    makes every later one return at once (and poisons the peers' flag words), so an absent peer costs ONE timeout, not one per
    barrier: the error arrives within a few seconds of KTN_IPC_TIMEOUT_S = 1."""
    out = mp.Manager().dict()
    mp.spawn(_worker_ipc_absent_peer_solve, args=(2, _free_port(), out), nprocs=2, join=True)
    msg, secs = out[0]
    assert "peer-buffer transport" in msg and "rank 1" in msg, msg
    assert 0.9 <= secs < 8.0, secs
    assert out[10] < 15.0 and out[11] < 15.0


def _worker_rowshard_few_nl(rank, world, port, out, inst_kw):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    inst = ktn.instances.make_instance(**inst_kw)
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist)
    status = m.optimize()
    out[rank] = (status, m.getobjval(), m.getsolution(), m.numiters(), m.m.stat("allreduce_calls"))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_two_rank_row_sharded_solve_with_fewer_nl_rows_than_ranks():
    """ONE nonlinear row on two ranks: one rank's shard has no NL row at all.  Its LP tolerances -- and with them its restart
    and exit decisions, i.e. the sequence of collectives -- must still be those of the other rank (the pure-LP rule of
    Engine::step is taken from the all-reduced NL-row count); both ranks end with the same x at the planted optimum."""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, max_nl_violation, planted_obj_bound
    world = 2
    inst_kw = dict(n=200, m_nl=1, k=8, family="explog", seed=5)
    out = mp.Manager().dict()
    mp.spawn(_worker_rowshard_few_nl, args=(world, _free_port(), out, inst_kw), nprocs=world, join=True)
    inst = ktn.instances.make_instance(**inst_kw)
    assert out[0][0] == out[1][0] == "Optimal"
    assert out[0][1] == out[1][1] and np.array_equal(out[0][2], out[1][2]) and out[0][3] == out[1][3]
    assert out[0][4] == out[1][4] > 0                                   # the same number of all-reduces on both ranks
    assert_planted_objective(out[0][1], inst)
    assert max_nl_violation(inst, out[0][2]) <= 1e-6 * (1 + 1e-6)


def _worker_rccl_one_rank(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["KTN_FORCE_COLLECTIVE"] = "1"
    import torch
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import RowShardedKatanaModel
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    inst = ktn.instances.make_instance(n=2000, m_nl=200, k=16, family="explog", seed=9)
    m = RowShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist, transport="rccl")
    status = m.optimize()
    out[rank] = (status, m.getobjval(), m.numiters(), m.stat("allreduce_calls"), m.transport, m.getsolution())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_transport_with_one_rank_gives_the_single_gpu_answer():
    """the RCCL transport itself (ktn_dist_unique_id -> torch broadcast -> ktn_dist_init_rccl -> ncclAllReduce on the engine's
    stream): a one-rank communicator (more ranks need more GPUs than the test box has) must reproduce the plain solve bit for
    bit -- every all-reduce is then the identity, so any difference is a transport fault"""
    import katana_jl_amd as ktn
    from helpers import hip_load_instance
    out = mp.Manager().dict()
    mp.spawn(_worker_rccl_one_rank, args=(1, _free_port(), out), nprocs=1, join=True)
    status, obj, iters, calls, transport, x = out[0]
    inst = ktn.instances.make_instance(n=2000, m_nl=200, k=16, family="explog", seed=9)
    # the same arithmetic without collectives: the row-sharded code path sums A'y with k_spmv + k_x_prox instead of the fused
    # k_pdhg_x, so compare with a one-rank CALLBACK-free run of that same path? -> compare against the planted optimum and the
    # plain engine at the stop-rule level
    one = hip_load_instance(ktn, inst)
    assert one.optimize() == status == "Optimal" and transport == "rccl" and calls > 100
    from helpers import assert_planted_objective, planted_obj_bound
    assert_planted_objective(obj, inst)
    assert abs(obj - one.getobjval()) <= planted_obj_bound(inst)
    assert np.max(np.abs(x - one.getsolution())) <= 1e-4


def _worker_nccl_one_rank_replicated(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["KTN_FORCE_COLLECTIVE"] = "1"
    import torch
    import torch.distributed as dist
    import katana_jl_amd as ktn
    from katana_jl_amd.distributed import ShardedKatanaModel
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world,
                            device_id=torch.device("cuda", 0))
    inst = ktn.instances.make_config("cfg3", seed=6)
    m = ShardedKatanaModel(ktn.KatanaSolver(log_level=0, device=0), inst, rank, world, dist)
    status = m.optimize()
    out[rank] = (status, m.getobjval(), m.numiters(), m.exchange_device, m.exchanged_rows, m.stat("cert_refinements"), m.getsolution())
    dist.destroy_process_group()


@pytest.mark.gpu
def test_nccl_backend_with_one_rank_runs_the_device_resident_exchange():
    """what `bench.py --gpus N` runs on cfg3, under the backend it runs it with: process group "nccl" (RCCL), cut blocks packed
    into CUDA tensors, `dist.all_gather` of CUDA tensors, blocks appended from the receive buffers, the certificate's
    all-reduce on a CUDA scalar -- with one rank (more need more GPUs than the box has) every gather returns the rank's own
    block, so the solve must be the single-GPU one bit for bit, refinement included (cfg3 seed 6)"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, hip_load_instance
    out = mp.Manager().dict()
    mp.spawn(_worker_nccl_one_rank_replicated, args=(1, _free_port(), out), nprocs=1, join=True)
    status, obj, iters, exch, rows, refinements, x = out[0]
    inst = ktn.instances.make_config("cfg3", seed=6)
    one = hip_load_instance(ktn, inst)
    assert one.optimize() == status == "Optimal" and exch == "cuda" and rows > 0 and refinements == 1
    assert obj == one.getobjval() and iters == one.numiters() and np.array_equal(x, one.getsolution())
    assert_planted_objective(obj, inst)


def test_both_sharded_models_offer_what_bench_reads():
    """bench.py's multi_gpu block reads these from whichever model `--replicated-lp` selects"""
    from katana_jl_amd.distributed import RowShardedKatanaModel, ShardedKatanaModel
    for cls in (ShardedKatanaModel, RowShardedKatanaModel):
        for name in ("optimize", "getobjval", "getsolution", "numiters", "numcuts", "lp_num_rows", "stat", "status"):
            assert callable(getattr(cls, name, None)) or hasattr(cls, "__getattr__"), (cls.__name__, name)     # (the row-sharded model forwards to its handle)


def _worker_batch_sharded(rank, world, port, out):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import katana_jl_amd as ktn
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    insts = [ktn.instances.make_instance(n=300, m_nl=30, k=8, family="explog", seed=100 + s) for s in range(13)]
    res, _ = ktn.solve_batch_sharded(ktn.KatanaSolver(log_level=0, device=0), insts, rank, world, dist)
    own, _ = ktn.solve_batch_sharded(ktn.KatanaSolver(log_level=0, device=0), insts, rank, world, dist, gather=False)
    out[rank] = ([(r["status"], r["objval"]) for r in res], len(own))
    dist.destroy_process_group()


def test_batch_blocks_partition_the_batch():
    from katana_jl_amd.batch import shard_range
    for count in (0, 1, 13, 512):
        for world in (1, 2, 3, 8):
            blocks = [shard_range(count, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == count
            assert all(a[1] == b[0] for a, b in zip(blocks, blocks[1:]))
            assert max(b[1] - b[0] for b in blocks) - min(b[1] - b[0] for b in blocks) <= 1


@pytest.mark.gpu
def test_two_rank_sharded_batch_equals_the_single_gpu_batch():
    """throughput mode over 2 ranks (replicas only): every rank solves its contiguous block of the batch
    as one fused batch; gathered, the results are those of the whole batch solved on one GPU, instance by instance"""
    import katana_jl_amd as ktn
    from helpers import assert_planted_objective, planted_obj_bound
    world = 2
    out = mp.Manager().dict()
    mp.spawn(_worker_batch_sharded, args=(world, _free_port(), out), nprocs=world, join=True)
    insts = [ktn.instances.make_instance(n=300, m_nl=30, k=8, family="explog", seed=100 + s) for s in range(13)]
    one, _ = ktn.solve_batch(ktn.KatanaSolver(log_level=0), insts, fused=True)
    assert out[0][0] == out[1][0] and len(out[0][0]) == 13
    assert out[0][1] + out[1][1] == 13 and abs(out[0][1] - out[1][1]) <= 1
    for (st, obj), ref, inst in zip(out[0][0], one, insts):
        assert st == ref["status"] == "Optimal"
        assert abs(obj - ref["objval"]) <= planted_obj_bound(inst)             # the same answer up to the stop rule
        assert_planted_objective(obj, inst)
