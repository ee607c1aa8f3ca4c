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
        try:
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


def strand_parallel_cascade_256m(model, enc0, mpos, wpos, chrlen, normmat, group=None, comm=None):
    """The part of genomepredict_256Mb after the Encoder (orca_predict.py:675-838 + the strand merge :866-877), with the
    two strands' tails on different ranks: the strands are independent until the merge, so rank 0 runs the forward
    strand, rank 1 the reverse strand (the four levels of a strand are one dependent chain: further ranks only receive),
    and ONE all-gather of the [4,C,250,250] maps (1 MB per rank) replaces the second half of the replicated work.  Returns the four merged [C,250,250] maps, identical on every rank.
    With one rank (or no process group) both strands run here, batched, exactly as cascade_256m does."""
    from . import engine, orca_predict
    if comm is not None:
        world, rank = comm.world, comm.rank
    elif dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    B = enc0.shape[0] // 2
    if world == 1:
        preds, _ = orca_predict.cascade_256m(model, enc0, mpos, wpos, chrlen, normmat)
        fwd, rev = torch.stack([p[0] for p in preds]), torch.stack([p[B] for p in preds])
    else:
        if rank < 2:
            slab = strand_tail_256m(model, enc0, rank, mpos, wpos, chrlen, normmat)
        else:       # the tail is one dependent chain per strand: ranks 2.. have nothing to add and only receive the maps
            C = getattr(model.denets[256], "num_2d", 1) if hasattr(model, "denets") else 1
            slab = torch.zeros((4, C, 250, 250), dtype=torch.float32, device=enc0.device)
        if comm is not None:
            allm = comm.all_gather(slab)
        else:
            allm = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
            dist.all_gather_into_tensor(allm.view(-1), slab.view(-1), group=group)
        fwd, rev = allm[0], allm[1]      # ranks 0 and 1 hold one strand each; the other ranks' slabs are placeholders
    return [torch.stack([engine.strand_merge(fwd[j, c], rev[j, c]) for c in range(fwd.shape[1])]) for j in range(fwd.shape[0])]


def strand_bin_plan(total_bins, rank, world):
    """Work split of ONE 32 Mb window over `world` ranks (strong scaling): the two strands first (rank parity), then
    contiguous bin ranges of the strand's Encoder (world/2 shards per strand; world = 1: both strands, all bins).
    Returns [(strand, bin_lo, bin_hi)] this rank encodes.  world must be 1 or even."""
    if world == 1:
        return [(0, 0, total_bins), (1, 0, total_bins)]
    if world % 2:
        raise ValueError("strand x bin sharding needs an even number of ranks")
    lo, hi = bin_range(total_bins, rank // 2, world // 2)
    return [(rank % 2, lo, hi)]


def strand_bin_sharded_32m(model, codes, mpos, wpos, distencs=None, group=None, comm=None):
    """`genomepredict`'s device work for ONE 32 Mb window ([B,L] packed bases replicated on every rank) shared by all ranks: rank r
    encodes bins `strand_bin_plan` of strand r % 2, ONE all-gather assembles both strands' [B,128,8000] encodings on every rank, ranks
    0 and 1 run one strand's tail each (Encoder2 -> six Decoders: a dependent chain, not shardable further; the independent `+ denet_1_pt`
    term of the 4 kb level runs on ranks 2 / 3 from 4 ranks on, on ranks 0 / 1 otherwise) and ONE
    all-gather of the [6,C,250,250] maps precedes the strand merge.  Returns the six merged [C,250,250] maps (on every rank).
    Replaces nn.DataParallel around the sub-networks (orca_models.py:44-50), which only splits batches."""
    from . import engine, orca_predict
    if comm is not None:
        world, rank = comm.world, comm.rank
    elif dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    else:
        world, rank = 1, 0
    B, L = codes.shape
    total = engine.encoder_num_bins(L)
    plan = strand_bin_plan(total, rank, world)

    def local(strand, lo, hi):
        with engine.immediate_overflow_guard():      # a rank's fp16-range retry happens before the collective, never after it on one rank only
            return model.net0.forward_codes(codes, reverse=bool(strand), bin_lo=lo, bin_hi=hi)

    if world == 1:
        enc0 = torch.cat([local(*plan[0]), local(*plan[1])], dim=0)
    else:
        strand, lo, hi = plan[0]
        part = local(strand, lo, hi)
        width = -(-total // (world // 2))
        slab = torch.zeros((B, 128, width), dtype=part.dtype, device=part.device)
        slab[:, :, : hi - lo] = part
        if comm is not None:
            gathered = comm.all_gather(slab)
        else:
            gathered = torch.empty((world, B, 128, width), dtype=part.dtype, device=part.device)
            dist.all_gather_into_tensor(gathered.view(world * B, 128, width), slab, group=group)
        strands = []
        for st in range(2):
            pieces = []
            for sh in range(world // 2):
                rlo, rhi = bin_range(total, sh, world // 2)
                pieces.append(gathered[2 * sh + st, :, :, : rhi - rlo])
            strands.append(torch.cat(pieces, dim=2))
        enc0 = torch.cat(strands, dim=0)
    if world == 1:
        preds, _ = orca_predict.cascade_32m_from_enc(model, enc0, mpos, wpos, [False, True], distencs)
        fwd, rev = torch.stack([p[0] for p in preds]), torch.stack([p[B] for p in preds])
    else:
        offload_1m = world >= 4      # ranks 2 / 3 are idle during the tails: they take the `+ denet_1_pt` term of strand 0 / 1 (independent of the cascade)
        if rank < 2:
            preds, _ = orca_predict.cascade_32m_from_enc(model, enc0[rank * B: (rank + 1) * B], mpos, wpos, [bool(rank)], distencs, with_1m=not offload_1m)
            slab = torch.stack([p[0] for p in preds]).contiguous()
        else:
            C = getattr(model.denets[32], "num_2d", 1)
            slab = torch.zeros((6, C, 250, 250), dtype=torch.float32, device=enc0.device)
            if offload_1m and rank < 4:
                st = rank - 2
                slab[5] = orca_predict.denet1m_32m_from_enc(model, enc0[st * B: (st + 1) * B], mpos, wpos, [bool(st)])[0]
        if comm is not None:
            allm = comm.all_gather(slab)
        else:
            allm = torch.empty((world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
            dist.all_gather_into_tensor(allm.view(-1), slab.view(-1), group=group)
        fwd, rev = allm[0], allm[1]
        if offload_1m:
            fwd, rev = fwd.clone(), rev.clone()
            fwd[5] += allm[2][5]
            rev[5] += allm[3][5]
    return [torch.stack([engine.strand_merge(fwd[j, c], rev[j, c]) for c in range(fwd.shape[1])]) for j in range(fwd.shape[0])]


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
