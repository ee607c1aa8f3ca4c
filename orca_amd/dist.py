"""Multi-GPU execution: one process per GPU, torch.distributed over RCCL/xGMI
(backend "nccl" on ROCm; "gloo" in the CPU tests).

The reference's only multi-GPU mechanism is nn.DataParallel's batch split
(orca_models.py:44-50).  What actually shards on this path:

* the Encoder's sequence blocks are independent given a 112 kb input halo
  (orca_modules.py:955-977), so rank r computes a contiguous range of 4 kb bins
  via the C ABI's bin_lo/bin_hi and ONE all-gather per strand assembles the
  [B,128,n_bins] encoding (32.8 MB for 256 Mb on 8 ranks - latency-bound on
  xGMI, ~0.2 % of the encoder time).  Everything after it (Encoder2/Encoder3 +
  decoder cascade, ~1 % of the FLOPs) runs replicated - or, for the 256 Mb model,
  one strand per rank parity with a 1 MB all-gather of the maps
  (`strand_parallel_cascade_256m`).
* independent 32 Mb windows (structural-variant screens, batches) are plain
  replicas: `shard_indices` deals them out, no data-path collective.
"""
import os
import sys

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's env (RANK, WORLD_SIZE,
    LOCAL_RANK, MASTER_ADDR/PORT).  Returns (rank, world, device)."""
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL across processes on this driver: dmabuf IPC only
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if use_gpu else "gloo"
        kw = {"device_id": device} if (backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, device


def bin_range(total_bins, rank, world):
    """Contiguous, balanced partition of [0,total_bins) (first `rem` ranks get one more)."""
    q, rem = divmod(total_bins, world)
    lo = rank * q + min(rank, rem)
    return lo, lo + q + (1 if rank < rem else 0)


def shard_indices(n_items, rank, world):
    """Round-robin deal of independent work items (replica mode)."""
    return list(range(rank, n_items, world))


class AbiComm:
    """RCCL communicator owned by liborca_hip.so (orca_comm_init_rank / orca_allgather, include/orca_hip.h): the
    all-gather is enqueued on the engine context's stream like every other step of the path.  The 128-byte unique id
    is created on rank 0 and handed to the other ranks through torch.distributed's object broadcast (host side, any
    backend).  ``world=1`` needs no process group (single-GPU tests)."""

    def __init__(self, device, group=None, world=None, rank=None):
        import ctypes

        from . import _lib, engine
        self.lib = _lib.load()
        self.ctx = engine.get_context(device)
        if world is None:
            world, rank = dist.get_world_size(group), dist.get_rank(group)
        self.world, self.rank = int(world), int(rank)
        idbuf = ctypes.create_string_buffer(128)
        err = None
        if self.rank == 0:
            try:
                _lib.check(self.lib.orca_comm_unique_id(idbuf), "orca_comm_unique_id")
            except Exception as e:          # every rank must still pass through the broadcast below, then fail TOGETHER
                err = f"{type(e).__name__}: {e}"
        if self.world > 1:
            box = [(err, idbuf.raw) if self.rank == 0 else None]
            dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
            err, raw = box[0]
            idbuf = ctypes.create_string_buffer(raw, 128)
        if err is not None:
            raise RuntimeError(f"AbiComm: rank 0 could not create the RCCL unique id ({err})")
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.orca_comm_init_rank(self.ctx.handle, self.world, self.rank, idbuf, ctypes.byref(self.handle)), "orca_comm_init_rank")

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        # Prefer close() / the context manager.  At interpreter shutdown the HIP context, the stream or the peer ranks may be gone already and
        # ncclCommDestroy can hang or crash there: the handle is then left to the process exit.
        try:
            if sys is None or sys.is_finalizing():
                return
            self.close()
        except Exception:
            pass

    def all_gather(self, slab):
        """slab: contiguous fp32 ROCm tensor -> [world, *slab.shape] (rank-major)."""
        import ctypes

        from . import _lib
        assert slab.is_cuda and slab.dtype == torch.float32 and slab.is_contiguous()
        out = torch.empty((self.world,) + tuple(slab.shape), dtype=torch.float32, device=slab.device)
        self.ctx.sync_stream()
        _lib.check(self.lib.orca_allgather(self.ctx.handle, self.handle, ctypes.c_void_p(slab.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                                           slab.numel()), "orca_allgather")
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.orca_comm_destroy(self.handle)
            self.handle = None


def sharded_encode(encode_range, x, total_bins, group=None, comm=None):
    """encode_range(x, bin_lo, bin_hi) -> [B,128,bin_hi-bin_lo] on this rank's device.
    Every rank computes its bin range, then one all-gather assembles [B,128,total_bins]
    on every rank.  With a single process this is just encode_range(x, 0, total_bins).
    ``comm``: an AbiComm (RCCL through the C ABI); otherwise torch.distributed's all_gather_into_tensor on ``group``
    (RCCL under the "nccl" backend, gloo in the CPU tests)."""
    if comm is not None:
        world, rank = comm.world, comm.rank
    elif dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    if world == 1:
        return encode_range(x, 0, total_bins)
    lo, hi = bin_range(total_bins, rank, world)
    part = encode_range(x, lo, hi)
    B = part.shape[0]
    width = -(-total_bins // world)  # ceil: every rank contributes an equal-size slab
    if hi - lo == width:
        slab = part.contiguous()
    else:
        slab = torch.zeros((B, 128, width), dtype=part.dtype, device=part.device)
        slab[:, :, : hi - lo] = part
    if comm is not None:
        gathered = comm.all_gather(slab)
    else:
        gathered = torch.empty((world * B, 128, width), dtype=part.dtype, device=part.device)
        dist.all_gather_into_tensor(gathered, slab, group=group)  # rank-major along dim 0
    gathered = gathered.view(world, B, 128, width)
    pieces = []
    for r in range(world):
        rlo, rhi = bin_range(total_bins, r, world)
        pieces.append(gathered[r, :, :, : rhi - rlo])
    return torch.cat(pieces, dim=2)


class ShardedEncoder(torch.nn.Module):
    """Drop-in for ``model.net0``: same call signature as Encoder.forward, but the bins
    are computed cooperatively by all ranks of ``group`` (input replicated on every rank)."""

    def __init__(self, encoder, group=None, comm=None):
        super().__init__()
        self.encoder, self.group, self.comm = encoder, group, comm

    def _local(self, fn):
        # the fp16-range retry of the rank's own bins must happen BEFORE the all-gather (a retry decided afterwards, by
        # engine.run_with_overflow_retry, would re-enter the collective on this rank only)
        from . import engine
        with engine.immediate_overflow_guard():
            return fn()

    def forward(self, x):
        from . import engine
        total = engine.encoder_num_bins(x.shape[2])
        return sharded_encode(lambda t, lo, hi: self._local(lambda: self.encoder(t, bin_lo=lo, bin_hi=hi)), x, total, self.group, self.comm)

    def forward_codes(self, codes, reverse=False):
        """Same from packed bases ([B,L] uint8 replicated on every rank, 32 MB per 32 Mb instead of 512 MB)."""
        from . import engine
        total = engine.encoder_num_bins(codes.shape[1])
        return sharded_encode(lambda t, lo, hi: self._local(lambda: self.encoder.forward_codes(t, reverse=reverse, bin_lo=lo, bin_hi=hi)),
                              codes, total, self.group, self.comm)


def strand_tail_256m(model, enc0, strand, mpos, wpos, chrlen, normmat):
    """One strand's share of the 256 Mb tail: rows [strand*B, (strand+1)*B) of ``enc0`` [2B,128,64000] through
    Encoder2 -> Encoder3 -> the four Decoders (orca_predict.cascade_256m; strand 1 = the reverse complement: its backgrounds are
    flipped and its zoom mirrored, orca_predict.py:703, :813-835).  ``normmat``: the 8000 x 8000 background (host array or float64
    ROCm tensor).  Returns [4, C, 250, 250] (batch row 0)."""
    from . import orca_predict
    B = enc0.shape[0] // 2
    preds, _ = orca_predict.cascade_256m(model, enc0[strand * B: (strand + 1) * B], mpos, wpos, chrlen, normmat, reverse_flags=(bool(strand),))
    return torch.stack([p[0] for p in preds]).contiguous()


def strand_parallel_cascade_256m(model, enc0, mpos, wpos, chrlen, normmat, group=None, comm=None, local_only=False):
    """The part of genomepredict_256Mb after the Encoder (orca_predict.py:675-838 + the strand merge :866-877), with the
    two strands' tails on different ranks: the strands are independent until the merge, so rank 0 runs the forward
    strand, rank 1 the reverse strand (the four levels of a strand are one dependent chain: further ranks only receive),
    and ONE all-gather of the [4,C,250,250] maps (1 MB per rank) replaces the second half of the replicated work.  Returns the four merged [C,250,250] maps, identical on every rank.
    With one rank (or no process group) both strands run here, batched, exactly as cascade_256m does."""
    from . import engine, orca_predict
    world, rank = (1, 0) if local_only else _world_rank(group, comm)
    B = enc0.shape[0] // 2
    if world == 1:
        preds, _ = orca_predict.cascade_256m(model, enc0, mpos, wpos, chrlen, normmat)
        fwd, rev = torch.stack([p[0] for p in preds]), torch.stack([p[B] for p in preds])
    else:
        C = getattr(model.denets[256], "num_2d", 1) if hasattr(model, "denets") else 1
        err = None
        if rank < 2:
            try:
                slab = strand_tail_256m(model, enc0, rank, mpos, wpos, chrlen, normmat)
            except Exception as e:      # stay in step with the other ranks (they would block in the all-gather); all raise after it
                err, slab = e, torch.zeros((4, C, 250, 250), dtype=torch.float32, device=enc0.device)
        else:       # the tail is one dependent chain per strand: ranks 2.. have nothing to add and only receive the maps
            slab = torch.zeros((4, C, 250, 250), dtype=torch.float32, device=enc0.device)
        allm, failed = _all_gather_status(slab, err is not None, world, group, comm)
        _raise_failed("strand_parallel_cascade_256m", err, failed)
        fwd, rev = allm[0], allm[1]      # ranks 0 and 1 hold one strand each; the other ranks' slabs are placeholders
    return [torch.stack([engine.strand_merge(fwd[j, c], rev[j, c]) for c in range(fwd.shape[1])]) for j in range(fwd.shape[0])]


def _world_rank(group, comm):
    if comm is not None:
        return comm.world, comm.rank
    if dist.is_initialized():
        return dist.get_world_size(group), dist.get_rank(group)
    return 1, 0


def _all_gather(slab, world, group, comm):
    """[world, *slab.shape] (rank-major) through the C ABI's RCCL communicator or torch.distributed."""
    slab = slab.contiguous()
    if comm is not None:
        return comm.all_gather(slab)
    out = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
    dist.all_gather_into_tensor(out.view(-1), slab.view(-1), group=group)
    return out


def _all_gather_status(slab, failed, world, group, comm, defer=False):
    """_all_gather of ``slab`` with an explicit STATUS word per rank riding behind the payload (four floats, so that the payload rows stay
    16-byte aligned): 1.0 = that rank's rank-local work raised.  ``defer``: return the status vector (device) instead of the failed ranks.  A rank that fails still takes part in the collective (the others would
    block in it) with whatever its slab holds; every rank then learns WHO failed from the status words - not from scanning the payload for
    NaN, which a genuine NaN in a map (bad weights, a bad background) would trigger on every rank.  Returns ([world, *shape]
    view of the payloads, [failed ranks]); the read-back is `world` floats."""
    n = slab.numel()
    assert n % 4 == 0
    flat = torch.empty(n + 4, dtype=torch.float32, device=slab.device)
    flat[:n] = slab.reshape(-1)
    flat[n:] = 1.0 if failed else 0.0
    out = _all_gather(flat, world, group, comm)                    # [world, n + 4]
    payload = out[:, :n].reshape((world,) + tuple(slab.shape))
    if defer:       # the caller reads the words later, together with a following collective's (`_failed_ranks`): no host sync here
        return payload, out[:, n]
    return payload, _failed_ranks(out[:, n])


def _failed_ranks(*status):
    """Ranks whose status word is raised in any of the [world] status vectors: ONE device read for all of them."""
    words = torch.stack([w.reshape(-1) for w in status]).tolist()
    return [[r for r, v in enumerate(row) if v != 0.0] for row in words] if len(status) > 1 else [r for r, v in enumerate(words[0]) if v != 0.0]


def _raise_failed(what, err, failed):
    if err is not None:
        raise err
    if failed:
        raise RuntimeError(f"{what}: rank(s) {failed} failed in their rank-local work (status word of the all-gather)")


def unit_plan(n_units, total_bins, rank, world):
    """Work split of ONE 32 Mb window over `world` ranks (strong scaling).  Units = (model, strand) pairs - independent
    until the strand merge; a unit's Encoder shards further by bins.  world >= n_units: unit = rank % n_units, bin shard rank // n_units of
    world // n_units; world < n_units: rank r takes units r, r + world, .. whole.  Returns ([(unit, bin_lo, bin_hi)] to encode,
    [unit] whose tail - Encoder2 -> six Decoders, one dependent chain - runs here, [unit] whose `+ denet_1_pt` term runs here: from
    2 x n_units ranks on the otherwise idle ranks n_units .. 2 n_units - 1 take it)."""
    if world == 1:
        return [(u, 0, total_bins) for u in range(n_units)], list(range(n_units)), []
    if world >= n_units:
        if world % n_units:
            raise ValueError(f"{n_units} (model, strand) units need a multiple of {n_units} ranks (or a divisor), got {world}")
        lo, hi = bin_range(total_bins, rank // n_units, world // n_units)
        offload = world >= 2 * n_units
        return [(rank % n_units, lo, hi)], ([rank] if rank < n_units else []), ([rank - n_units] if offload and n_units <= rank < 2 * n_units else [])
    if n_units % world:
        raise ValueError(f"{n_units} (model, strand) units need a divisor of {n_units} as rank count (or a multiple), got {world}")
    mine = list(range(rank, n_units, world))
    return [(u, 0, total_bins) for u in mine], mine, []


def units_sharded_32m(models, codes, mpos, wpos, distencs=None, group=None, comm=None, local_only=False, marks=None):
    """`genomepredict`'s device work for ONE 32 Mb window and SEVERAL models (the reference's default call has two,
    orca_predict.py:231 models=["h1esc","hff"]) shared by all ranks: the units (model, strand) are independent until the strand merge
    (`unit_plan`).  ONE all-gather assembles every unit's [B,128,8000] encoding on every rank, the tails run one unit per rank (four
    independent tails for two models: half the serial fraction per model of the one-model job from 4 ranks on), ONE all-gather of the
    [6,C,250,250] maps precedes the strand merges.  ``codes``: [B,L] packed bases (or an engine.CodeWindow holding this rank's share).
    Returns per model the six merged [C,250,250] maps (on every rank).  Replaces nn.DataParallel around the sub-networks
    (orca_models.py:44-50), which only splits batches.

    A rank whose local work raises does NOT leave the others blocked in the collectives: it still takes part, with a STATUS word behind its
    payload (`_all_gather_status`), every rank reads the status words behind that collective and all raise together - the failing rank its own
    error, the others a RuntimeError naming it (ADVICE rounds 3-4; a genuine NaN in a map is no longer mistaken for a failed peer).
    ``local_only``: run the whole job on this rank whatever the process group (the N = 1 time inside an N-rank bench run);
    ``marks``: a list that receives (phase, torch.cuda.Event) after "encode", "gather", "tails", "maps" (bench.py's per-phase times)."""
    from . import engine, orca_predict
    world, rank = (1, 0) if local_only else _world_rank(group, comm)
    M, U = len(models), 2 * len(models)

    def mark(name):
        if marks is not None and codes.device.type == "cuda":
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            marks.append((name, ev))

    B, L = codes.shape
    total = engine.encoder_num_bins(L)
    enc_plan, tail_units, one_m_units = unit_plan(U, total, rank, world)
    shards = max(1, world // U)
    width = -(-total // shards)
    dev = codes.device
    err = None

    def local(u, lo, hi):
        with engine.immediate_overflow_guard():      # a rank's fp16-range retry happens before the collective, never after it on one rank only
            return models[u // 2].net0.forward_codes(codes, reverse=bool(u & 1), bin_lo=lo, bin_hi=hi)

    mark("start")
    if world == 1:
        enc = [local(u, 0, total) for u in range(U)]
        mark("encode")
        mark("gather")
    else:
        slab = torch.zeros((len(enc_plan), B, 128, width), dtype=torch.float32, device=dev)
        try:
            for i, (u, lo, hi) in enumerate(enc_plan):
                slab[i, :, :, : hi - lo] = local(u, lo, hi)
        except Exception as e:                       # stay in step with the other ranks: every rank raises behind the collective
            err = e
        mark("encode")
        # (the status words of THIS collective are read together with the final one's: a read here is a host sync in front of the tails,
        # ADVICE r5 - a rank that failed to encode still runs its tails, on zeros, and every rank raises behind the last collective)
        gathered, enc_status = _all_gather_status(slab, err is not None, world, group, comm, defer=True)          # [world, units per rank, B, 128, width]
        mark("gather")
        enc_err, err = err, None
        enc = []
        for u in range(U):
            if world >= U:
                pieces = []
                for sh in range(shards):
                    rlo, rhi = bin_range(total, sh, shards)
                    pieces.append(gathered[sh * U + u, 0, :, :, : rhi - rlo])
                enc.append(torch.cat(pieces, dim=2))
            else:
                enc.append(gathered[u % world, u // world])
    C = getattr(models[0].denets[32], "num_2d", 1)

    def tail(u, with_1m):
        preds, _ = orca_predict.cascade_32m_from_enc(models[u // 2], enc[u], mpos, wpos, [bool(u & 1)], distencs[u // 2] if isinstance(distencs, (list, tuple)) else distencs,
                                                     with_1m=with_1m)
        return torch.stack([p[0] for p in preds])

    if world == 1:
        maps = []
        for m in range(M):           # both strands of a model as one batch, exactly as cascade_32m does
            preds, _ = orca_predict.cascade_32m_from_enc(models[m], torch.cat([enc[2 * m], enc[2 * m + 1]], dim=0), mpos, wpos, [False, True],
                                                         distencs[m] if isinstance(distencs, (list, tuple)) else distencs)
            maps += [torch.stack([p[0] for p in preds]), torch.stack([p[B] for p in preds])]
        mark("tails")
        mark("maps")
    else:
        offload = world >= 2 * U
        per = max(1, -(-U // world))
        slab = torch.zeros((per, 6, C, 250, 250), dtype=torch.float32, device=dev)
        try:
            for i, u in enumerate(tail_units):
                slab[i] = tail(u, not offload)
            for u in one_m_units:
                slab[0, 5] = orca_predict.denet1m_32m_from_enc(models[u // 2], enc[u], mpos, wpos, [bool(u & 1)])[0]
        except Exception as e:
            err = e
        mark("tails")
        allm, tail_status = _all_gather_status(slab, err is not None, world, group, comm, defer=True)              # [world, per, 6, C, 250, 250]
        mark("maps")
        enc_failed, failed = _failed_ranks(enc_status, tail_status)
        _raise_failed("units_sharded_32m (encode)", enc_err, enc_failed)
        _raise_failed("units_sharded_32m (tails)", err, failed)
        maps = []
        for u in range(U):
            mu = allm[u % world, u // world] if world < U else allm[u, 0]
            if offload:
                mu = mu.clone()
                mu[5] += allm[U + u, 0, 5]
            maps.append(mu)
    return [[torch.stack([engine.strand_merge(maps[2 * m][j, c], maps[2 * m + 1][j, c]) for c in range(maps[2 * m].shape[1])]) for j in range(6)]
            for m in range(M)]


def strand_bin_plan(total_bins, rank, world):
    """`unit_plan` for ONE model (two units = the strands): [(strand, bin_lo, bin_hi)] this rank encodes.  world must be 1 or even."""
    if world > 1 and world % 2:
        raise ValueError("strand x bin sharding needs an even number of ranks")
    return unit_plan(2, total_bins, rank, world)[0]


def strand_bin_sharded_32m(model, codes, mpos, wpos, distencs=None, group=None, comm=None):
    """`units_sharded_32m` for one model: rank r encodes bins `strand_bin_plan` of strand r % 2, ONE all-gather assembles both strands'
    [B,128,8000] encodings on every rank, ranks 0 and 1 run one strand's tail each (the independent `+ denet_1_pt` term of the 4 kb level
    on ranks 2 / 3 from 4 ranks on), ONE all-gather of the [6,C,250,250] maps precedes the strand merge.  Returns the six merged
    [C,250,250] maps (on every rank)."""
    return units_sharded_32m([model], codes, mpos, wpos, distencs, group, comm)[0]


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
