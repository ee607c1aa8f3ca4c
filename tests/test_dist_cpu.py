"""world_size-2 gloo test (CPU) of the sharded-encoder path: two processes each compute half
of the bins (with the reference's 112 kb input halo) and all-gather; the result must equal
the single-process encoding.  The per-rank encode function here is the CPU oracle, so what
is tested is orca_amd.dist (partition, padding, all-gather, reassembly)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_range(sd):
    from oracle import orca_oracle as O

    def enc(x, lo, hi):
        L = x.shape[2]
        a = max(0, lo * 4000 - 112000)
        total = L // 4000
        b = L if hi == total else min(L, hi * 4000 + 112000)
        y = O.encoder_forward(sd, x[:, :, a:b])
        k = lo - a // 4000
        return y[:, :, k: k + (hi - lo)].contiguous()
    return enc


def _worker(rank, world, port, L, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from orca_amd import dist as D
    from orca_amd import synth
    from tests.util import synth_sd
    try:
        r, w, dev = D.init_from_env("gloo")
        assert (r, w) == (rank, world) and dev.type == "cpu"
        x = torch.from_numpy(synth.synth_sequence(L, seed=5)).transpose(1, 2)
        out = D.sharded_encode(_oracle_range(synth_sd("Encoder", 0)), x, L // 4000)
        t = D.max_over_ranks(float(rank + 1), dev)
        # strand-parallel 256 Mb tail: rank parity picks the strand, ONE all-gather of the maps, merge on every rank.  The device work
        # (strand_tail_256m, engine.strand_merge) is replaced by stand-ins: what is tested is the exchange.
        from orca_amd import engine
        seen = []
        D.strand_tail_256m = lambda model, enc0, strand, *a: (seen.append(strand), torch.full((4, 1, 6, 6), float(10 + strand)) + torch.arange(36.).view(6, 6))[1]
        engine.strand_merge = lambda f, r: 0.5 * f + 0.5 * torch.flip(r, [0, 1])
        maps = D.strand_parallel_cascade_256m(None, torch.zeros(2, 128, 8), 0, 0, 0, {})
        assert seen == [rank & 1] and len(maps) == 4 and maps[0].shape == (1, 6, 6)
        want = 0.5 * (10 + torch.arange(36.).view(6, 6)) + 0.5 * torch.flip(11 + torch.arange(36.).view(6, 6), [0, 1])
        assert all(torch.equal(m[0], want) for m in maps)
        # strand x bin sharding of ONE 32 Mb window (strong scaling): rank parity = strand, then bin shards; one all-gather of the encodings,
        # the strands' tails on ranks 0 / 1, one all-gather of the maps.  Stand-ins for the device work: the encoder returns
        # f(strand, bin, channel), the tail returns maps that depend on every entry of its strand's assembled encoding.
        from orca_amd import orca_predict

        class _Net0:
            def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0):
                b = torch.arange(bin_lo, bin_hi, dtype=torch.float32)
                return (b[None, None, :] + 1000.0 * float(reverse) + 0.001 * torch.arange(128.)[None, :, None]).expand(codes.shape[0], -1, -1).contiguous()

        class _Model:
            net0 = _Net0()
            denets = {32: type("D", (), {"num_2d": 1})()}

        def _tail(model, enc0, mpos, wpos, flags, de=None, with_1m=True):
            assert with_1m                                                # (2 ranks: the tails keep the `+ denet_1_pt` term)
            v = (enc0 * torch.arange(1, enc0.shape[2] + 1.)).sum()       # position-weighted: any misplaced bin changes it
            return [torch.full((enc0.shape[0], 1, 250, 250), float(v) * (j + 1)) + (1.0 if flags[0] else 0.0) for j in range(6)], None
        orca_predict.cascade_32m_from_enc = _tail
        codes = torch.zeros((1, 4000 * 75), dtype=torch.uint8)
        assert D.strand_bin_plan(75, rank, 2) == [(rank, 0, 75)] and D.strand_bin_plan(75, 3, 4) == [(1, 38, 75)]
        maps32 = D.strand_bin_sharded_32m(_Model(), codes, 0, 0)
        ref = [_Net0().forward_codes(codes, bool(st), 0, 75) for st in range(2)]
        wf, wr = _tail(None, ref[0], 0, 0, [False])[0], _tail(None, ref[1], 0, 0, [True])[0]
        assert len(maps32) == 6 and all(torch.equal(m[0], 0.5 * wf[j][0, 0] + 0.5 * torch.flip(wr[j][0, 0], [0, 1])) for j, m in enumerate(maps32))
        q.put((rank, out.numpy(), t))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:  # surface the failure instead of letting the parent time out
        q.put((rank, repr(e), -1.0))


def test_sharded_encoder_allgather_gloo_world2():
    from oracle import orca_oracle as O
    from orca_amd import dist as D
    from orca_amd import synth
    from tests.util import synth_sd
    L = 4000 * 75  # 75 bins: uneven split 38 / 37
    assert D.bin_range(75, 0, 2) == (0, 38) and D.bin_range(75, 1, 2) == (38, 75)
    assert D.shard_indices(5, 1, 2) == [1, 3]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, L, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict()
    for _ in range(2):
        rank, arr, t = q.get(timeout=300)
        assert not isinstance(arr, str), arr
        res[rank] = arr
        assert t == 2.0
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    x = torch.from_numpy(synth.synth_sequence(L, seed=5)).transpose(1, 2)
    ref = O.encoder_forward(synth_sd("Encoder", 0), x).numpy()
    assert res[0].shape == ref.shape == (1, 128, 75)
    assert np.abs(res[0] - ref).max() < 1e-5 and np.abs(res[1] - ref).max() < 1e-5
    assert np.array_equal(res[0], res[1])


def _worker4(rank, world, port, q):
    """world = 4 over gloo: what changes beyond two ranks - two bin shards per strand reassembled after the all-gather, ranks 2 / 3
    contributing placeholders to the map gathers (32 Mb and 256 Mb tails) and receiving the result.  Device work replaced by stand-ins."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from orca_amd import dist as D
    from orca_amd import engine, orca_predict
    try:
        D.init_from_env("gloo")
        engine.strand_merge = lambda f, r: 0.5 * f + 0.5 * torch.flip(r, [0, 1])
        calls = []

        class _Net0:
            def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0):
                calls.append((bool(reverse), bin_lo, bin_hi))
                b = torch.arange(bin_lo, bin_hi, dtype=torch.float32)
                return (b[None, None, :] + 1000.0 * float(reverse) + 0.001 * torch.arange(128.)[None, :, None]).expand(codes.shape[0], -1, -1).contiguous()

        class _Model:
            net0 = _Net0()
            denets = {32: type("D", (), {"num_2d": 1})(), 256: type("D", (), {"num_2d": 1})()}

        tails = []

        ones = []

        def _tail(model, enc0, mpos, wpos, flags, de=None, with_1m=True):
            tails.append(bool(flags[0]))
            v = (enc0 * torch.arange(1, enc0.shape[2] + 1.)).sum()
            maps = [torch.full((enc0.shape[0], 1, 250, 250), float(v) * (j + 1)) + (1.0 if flags[0] else 0.0) for j in range(6)]
            if with_1m:
                maps[5] = maps[5] + _one_m(model, enc0, mpos, wpos, flags, record=False)
            return maps, None

        def _one_m(model, enc0, mpos, wpos, flags, record=True):
            if record:
                ones.append(bool(flags[0]))
            return torch.full((enc0.shape[0], 1, 250, 250), 0.25 * float(enc0.sum()) + (7.0 if flags[0] else 3.0))
        orca_predict.cascade_32m_from_enc = _tail
        orca_predict.denet1m_32m_from_enc = _one_m
        codes = torch.zeros((1, 4000 * 75), dtype=torch.uint8)
        maps32 = D.strand_bin_sharded_32m(_Model(), codes, 0, 0)
        lo, hi = D.bin_range(75, rank // 2, 2)
        assert calls == [(bool(rank % 2), lo, hi)]                      # one strand, one bin shard per rank
        assert tails == ([bool(rank)] if rank < 2 else [])              # only ranks 0 / 1 run a tail
        assert ones == ([] if rank < 2 else [bool(rank - 2)])          # ... and ranks 2 / 3 the `+ denet_1_pt` term of strand 0 / 1
        ref = [_Net0().forward_codes(codes, bool(st), 0, 75) for st in range(2)]
        wf, wr = _tail(None, ref[0], 0, 0, [False])[0], _tail(None, ref[1], 0, 0, [True])[0]
        ok32 = all(torch.equal(m[0], 0.5 * wf[j][0, 0] + 0.5 * torch.flip(wr[j][0, 0], [0, 1])) for j, m in enumerate(maps32))
        seen = []
        D.strand_tail_256m = lambda model, enc0, strand, *a: (seen.append(strand), torch.full((4, 1, 250, 250), float(10 + strand)))[1]
        maps256 = D.strand_parallel_cascade_256m(_Model(), torch.zeros(2, 128, 8), 0, 0, 0, None)
        ok256 = seen == ([rank] if rank < 2 else []) and all(torch.equal(m, torch.full((1, 250, 250), 10.5)) for m in maps256)
        q.put((rank, ok32, ok256))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e), False))


def test_strand_bin_sharding_gloo_world4():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker4, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    for _ in range(4):
        rank, ok32, ok256 = q.get(timeout=300)
        assert ok32 is True and ok256 is True, (rank, ok32, ok256)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def _worker4_models(rank, world, port, q):
    """world = 4 over gloo: the TWO-model job (units = model x strand: four independent tails, one per rank) against two one-model runs,
    and a rank that fails in its local work: nobody may stay blocked in a collective, every rank raises."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from orca_amd import dist as D
    from orca_amd import engine, orca_predict
    try:
        D.init_from_env("gloo")
        engine.strand_merge = lambda f, r: 0.5 * f + 0.5 * torch.flip(r, [0, 1])
        calls, tails = [], []

        class _Net0:
            def __init__(self, tag):
                self.tag = tag

            def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0):
                calls.append((self.tag, bool(reverse), bin_lo, bin_hi))
                b = torch.arange(bin_lo, bin_hi, dtype=torch.float32)
                return (b[None, None, :] * (1 + self.tag) + 1000.0 * float(reverse) + 0.001 * torch.arange(128.)[None, :, None]).expand(codes.shape[0], -1, -1).contiguous()

        class _Model:
            def __init__(self, tag):
                self.tag, self.net0 = tag, _Net0(tag)
                self.denets = {32: type("D", (), {"num_2d": 1})()}

        def _tail(model, enc0, mpos, wpos, flags, de=None, with_1m=True):
            tails.append((model.tag, [bool(f) for f in flags]))
            if model.tag == 1 and getattr(_tail, "fail", False) and flags[0]:
                raise ValueError("injected failure in one rank's tail")
            v = (enc0 * torch.arange(1, enc0.shape[2] + 1.)).sum(dim=(1, 2))
            maps = [v.view(-1, 1, 1, 1) * (j + 1) * (1 + model.tag) + torch.tensor([1.0 if f else 0.0 for f in flags]).repeat_interleave(enc0.shape[0] // len(flags)).view(-1, 1, 1, 1)
                    + torch.zeros(enc0.shape[0], 1, 250, 250) for j in range(6)]
            if with_1m:
                maps[5] = maps[5] + _one_m(model, enc0, mpos, wpos, flags)
            return maps, None

        def _one_m(model, enc0, mpos, wpos, flags):
            per = enc0.shape[0] // len(flags)
            return torch.cat([torch.full((per, 1, 250, 250), 0.5 * float(enc0[k * per:(k + 1) * per].sum()) + (7.0 if f else 3.0) + model.tag) for k, f in enumerate(flags)])
        orca_predict.cascade_32m_from_enc = _tail
        orca_predict.denet1m_32m_from_enc = _one_m
        codes = torch.zeros((1, 4000 * 75), dtype=torch.uint8)
        models = [_Model(0), _Model(1)]
        assert D.unit_plan(4, 75, rank, 4) == ([(rank, 0, 75)], [rank], []) and D.unit_plan(4, 75, 5, 8) == ([(1, 38, 75)], [], [1])
        assert D.unit_plan(4, 75, 1, 2) == ([(1, 0, 75), (3, 0, 75)], [1, 3], [])
        two = D.units_sharded_32m(models, codes, 0, 0)
        ok_calls = calls == [(rank // 2, bool(rank & 1), 0, 75)] and tails == [(rank // 2, [bool(rank & 1)])]
        calls.clear(); tails.clear()
        ones = [D.strand_bin_sharded_32m(m, codes, 0, 0) for m in models]      # the one-model job, twice (bins sharded 2 x 2, tails on ranks 0 / 1)
        ok_equal = all(torch.equal(two[m][j], ones[m][j]) for m in range(2) for j in range(6))
        # failure on ONE rank (rank 3 = model 1, reverse strand): every rank raises, none hangs
        _tail.fail = True
        raised = ""
        try:
            D.units_sharded_32m(models, codes, 0, 0)
        except Exception as e:
            raised = type(e).__name__
        q.put((rank, ok_calls, ok_equal, raised))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e), False, ""))


def test_two_model_job_and_failing_rank_gloo_world4():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker4_models, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    for _ in range(4):
        rank, ok_calls, ok_equal, raised = q.get(timeout=300)
        assert ok_calls is True and ok_equal is True, (rank, ok_calls, ok_equal)
        assert raised == ("ValueError" if rank == 3 else "RuntimeError"), (rank, raised)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def test_overflow_guard_policy_is_per_thread_and_immediate_around_a_sharded_encoder():
    """(a) the deferred-guard scope of one thread must not leak into another thread; (b) a ShardedEncoder runs
    its rank-local encoder with the IMMEDIATE guard even inside a deferred scope, so that an fp16-range retry happens
    before the all-gather and never re-enters the collective on one rank only."""
    import threading

    from orca_amd import engine
    from orca_amd.dist import ShardedEncoder

    seen = {}

    class Probe(torch.nn.Module):
        def forward(self, x, bin_lo=0, bin_hi=0):
            seen["defer_inside_local_encode"] = engine._guard["defer"]
            n = engine.encoder_num_bins(x.shape[2]) if bin_hi <= 0 else bin_hi
            return torch.zeros(x.shape[0], 128, n - bin_lo)

    x = torch.zeros(1, 4, 8000)
    with engine.defer_overflow_guard():
        assert engine._guard["defer"] is True
        t = threading.Thread(target=lambda: seen.setdefault("defer_in_other_thread", engine._guard["defer"]))
        t.start(); t.join()
        try:
            ShardedEncoder(Probe()).forward(x)
        except Exception as e:     # encoder_num_bins needs the built library; it is present wherever the CPU suite runs
            raise AssertionError(f"ShardedEncoder.forward failed: {e}")
        assert engine._guard["defer"] is True            # restored
    assert engine._guard["defer"] is False
    assert seen["defer_in_other_thread"] is False
    assert seen["defer_inside_local_encode"] is False


def test_unit_plan_covers_every_bin_once_up_to_16_ranks():
    """Pure bookkeeping of the 32 Mb strong-scaling job (dist.unit_plan) at every world size the driver may use and beyond: each (unit, bin)
    is encoded by exactly one rank, each unit's tail runs on exactly one rank, the `+ denet_1_pt` terms - from 2 x units ranks on - on the
    next `units` ranks, and the widest shard is what `bench.py` sizes the gathered slab by."""
    from orca_amd import dist as D
    for n_units in (2, 4):
        for world in (1, 2, 4, 8, 16):
            if world % n_units and n_units % world:
                continue
            cover = {u: np.zeros(8000, dtype=int) for u in range(n_units)}
            tails, ones = [], []
            for r in range(world):
                enc, t, o = D.unit_plan(n_units, 8000, r, world)
                for u, lo, hi in enc:
                    cover[u][lo:hi] += 1
                    assert 0 <= lo < hi <= 8000
                tails += t
                ones += o
            assert all((c == 1).all() for c in cover.values()), (n_units, world)
            assert sorted(tails) == list(range(n_units))
            assert sorted(ones) == (list(range(n_units)) if world >= 2 * n_units else [])
    # the headline window on 8 ranks, one model: 4 bin shards of 2 000 bins per strand - the conv_small.h boundary
    assert [D.unit_plan(2, 8000, r, 8)[0] for r in range(8)] == [[(r % 2, 2000 * (r // 2), 2000 * (r // 2) + 2000)] for r in range(8)]


def _worker8(rank, world, port, q):
    """world = 8 over gloo (what the driver's 8-GPU node will run first): the one-model job = 2 strands x 4 bin shards, tails on ranks 0 / 1,
    `+ denet_1_pt` on ranks 2 / 3, ranks 4-7 encode only; the two-model job = 4 units x 2 bin shards, tails on ranks 0-3, `+ denet_1_pt` on
    ranks 4-7; both against the unsharded toy result; and the 256 Mb tail (ranks 2-7 only receive)."""
    import sys
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from orca_amd import dist as D
    from orca_amd import engine, orca_predict
    try:
        D.init_from_env("gloo")
        engine.strand_merge = lambda f, r: 0.5 * f + 0.5 * torch.flip(r, [0, 1])
        calls, tails, ones = [], [], []
        NB = 80

        class _Net0:
            def __init__(self, tag):
                self.tag = tag

            def forward_codes(self, codes, reverse=False, bin_lo=0, bin_hi=0):
                calls.append((self.tag, bool(reverse), bin_lo, bin_hi))
                b = torch.arange(bin_lo, bin_hi if bin_hi > 0 else NB, dtype=torch.float32)
                return (b[None, None, :] * (1 + self.tag) + 1000.0 * float(reverse) + 0.001 * torch.arange(128.)[None, :, None]).expand(codes.shape[0], -1, -1).contiguous()

        class _Model:
            def __init__(self, tag):
                self.tag, self.net0 = tag, _Net0(tag)
                self.denets = {lv: type("D", (), {"num_2d": 1})() for lv in (32, 256)}

        def _tail(model, enc0, mpos, wpos, flags, de=None, with_1m=True):
            tails.append((model.tag, [bool(f) for f in flags]))
            v = (enc0 * torch.arange(1, enc0.shape[2] + 1.)).sum(dim=(1, 2))
            maps = [v.view(-1, 1, 1, 1) * (j + 1) * (1 + model.tag) + torch.tensor([1.0 if f else 0.0 for f in flags]).repeat_interleave(enc0.shape[0] // len(flags)).view(-1, 1, 1, 1)
                    + torch.zeros(enc0.shape[0], 1, 250, 250) for j in range(6)]
            if with_1m:
                maps[5] = maps[5] + _one_m(model, enc0, mpos, wpos, flags, record=False)
            return maps, None

        def _one_m(model, enc0, mpos, wpos, flags, record=True):
            if record:
                ones.append((model.tag, [bool(f) for f in flags]))
            per = enc0.shape[0] // len(flags)
            return torch.cat([torch.full((per, 1, 250, 250), 0.5 * float(enc0[k * per:(k + 1) * per].sum()) + (7.0 if f else 3.0) + model.tag) for k, f in enumerate(flags)])
        orca_predict.cascade_32m_from_enc = _tail
        orca_predict.denet1m_32m_from_enc = _one_m
        codes = torch.zeros((1, 4000 * NB), dtype=torch.uint8)
        models = [_Model(0), _Model(1)]
        res = {}
        for name, ms in (("one", models[:1]), ("two", models)):
            calls.clear(); tails.clear(); ones.clear()
            got = D.units_sharded_32m(ms, codes, 0, 0)
            U = 2 * len(ms)
            u, sh = rank % U, rank // U
            lo, hi = D.bin_range(NB, sh, 8 // U)
            ok = calls == [(u // 2, bool(u & 1), lo, hi)]
            ok &= tails == ([(rank // 2, [bool(rank & 1)])] if rank < U else [])
            ok &= ones == ([((rank - U) // 2, [bool((rank - U) & 1)])] if U <= rank < 2 * U else [])
            # against the job done by hand on this rank alone
            t_save, o_save = list(tails), list(ones)
            for m_i, m in enumerate(ms):
                full = [m.net0.forward_codes(codes, bool(st), 0, NB) for st in range(2)]
                wf, wr = _tail(m, full[0], 0, 0, [False])[0], _tail(m, full[1], 0, 0, [True])[0]
                ok &= all(torch.allclose(got[m_i][j][0], 0.5 * wf[j][0, 0] + 0.5 * torch.flip(wr[j][0, 0], [0, 1]), rtol=1e-6, atol=0) for j in range(6))
            res[name] = bool(ok)
        seen = []
        D.strand_tail_256m = lambda model, enc0, strand, *a: (seen.append(strand), torch.full((4, 1, 250, 250), float(10 + strand)))[1]
        maps256 = D.strand_parallel_cascade_256m(models[0], torch.zeros(2, 128, 8), 0, 0, 0, None)
        res["256"] = seen == ([rank] if rank < 2 else []) and all(torch.equal(m, torch.full((1, 250, 250), 10.5)) for m in maps256)
        q.put((rank, res))
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    except Exception as e:
        import traceback
        q.put((rank, {"error": repr(e) + traceback.format_exc()[-600:]}))


def test_one_and_two_model_jobs_gloo_world8():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 35500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker8, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    for _ in range(8):
        rank, res = q.get(timeout=600)
        assert res == {"one": True, "two": True, "256": True}, (rank, res)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0


def test_results_cached_inside_a_deferred_range_check_do_not_survive_a_fired_check(monkeypatch):
    """`run_with_overflow_retry` checks the fp16 range ONCE, behind the whole chain; what the chain put where later calls find it
    before that check (`engine.tentative`: a shared Encoder output of genomepredict_256Mb, an auto-built chromosome encoding of the SV
    drivers) is taken out again when the check fires - before the range-safe retry, which caches its own result - and kept otherwise.
    The device is faked: only the control flow is under test (the GPU suite forces the same through a real 256 Mb driver call)."""
    from orca_amd import engine

    flags = []

    class Ctx:
        device_index = 0

        def take_overflow(self):
            return flags.pop(0)

    monkeypatch.setattr(engine, "get_context", lambda d: Ctx())
    monkeypatch.setattr(engine, "_thread_pools", lambda: {})
    cache, passes = {}, []

    def chain():
        passes.append(engine._guard["force_safe"])
        if "enc" not in cache:
            cache["enc"] = "bf16x3" if engine._guard["force_safe"] else "f16x2"
            engine.tentative(lambda: cache.pop("enc", None))
        return cache["enc"]

    dev = torch.device("cuda:0")
    flags[:] = [False]
    assert engine.run_with_overflow_retry(chain, dev) == "f16x2" and cache == {"enc": "f16x2"} and passes == [False]
    cache.clear(); passes.clear()
    flags[:] = [True]
    with pytest.warns(UserWarning, match="fp16 range"):
        assert engine.run_with_overflow_retry(chain, dev) == "bf16x3"
    assert cache == {"enc": "bf16x3"} and passes == [False, True]          # the overflowed entry was dropped, the retry's kept
    # outside a deferred pass nothing is registered (immediate checks: the module has already retried)
    engine.tentative(lambda: cache.clear())
    assert cache == {"enc": "bf16x3"} and getattr(engine._tls, "tentative", None) is None
