"""Multi-GPU: images are independent, so the path shards by image and the only
collective is a gather of the final, fixed-size annotation blocks.

One process per GPU (``torch.distributed.run``), backend ``"nccl"`` (= RCCL over
xGMI on ROCm) for device tensors, ``"gloo"`` for the CPU tests.  Replaces the
reference's single-process ``nn.DataParallel`` gather of the full field tensors to
GPU 0 followed by a host copy (reference ``predictor.py:33-37``,
``decoder/decoder.py:96-100``): ~6.2 MB/image of fields there, <= 70 KB/image of
annotations here (``[max_annotations, 17, 4]`` float32 + counts), one call per batch.
"""
import torch


def shard_bounds(n_items, rank, world_size):
    """Contiguous, balanced split of ``n_items`` images: ``[lo, hi)`` of ``rank``."""
    base, extra = divmod(n_items, world_size)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack(annotations, ids, counts):
    """One fixed-size block per image -- what travels between GPUs: ``int32 [B, 1 + max_ann * (4 K + 2)]`` holding
    the count, the float32 annotation bits and the int64 ids (as two int32 each)."""
    B, M = annotations.shape[0], annotations.shape[1]
    return torch.cat([counts.reshape(B, 1).to(torch.int32),
                      annotations.contiguous().view(torch.int32).reshape(B, -1),
                      ids.contiguous().view(torch.int32).reshape(B, 2 * M)], dim=1).contiguous()


def unpack_block(block, max_ann, n_keypoints):
    """Inverse of :func:`pack`."""
    B = block.shape[0]
    n = max_ann * n_keypoints * 4
    counts = block[:, 0].contiguous()
    annotations = block[:, 1:1 + n].contiguous().view(torch.float32).reshape(B, max_ann, n_keypoints, 4)
    ids = block[:, 1 + n:].contiguous().view(torch.int64).reshape(B, max_ann)
    return annotations, ids, counts


def active(group=None):
    """-> (rank, world size) of an initialised process group with more than one rank, else None."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return None
    world = dist.get_world_size(group)
    return (dist.get_rank(group), world) if world > 1 else None


def gather_annotations(annotations, ids, counts, group=None, force=False):
    """All-gather per-rank decode results in rank order with ONE collective per batch: counts,
    annotations and ids are packed into one fixed-size int32 block per image.  ``force``: run the collective even in a
    group of one rank (the GPU suite drives the RCCL branch that way on a single-GPU box).

    :param annotations: ``[B_local, max_ann, K, 4]`` float32
    :param ids: ``[B_local, max_ann]`` int64, :param counts: ``[B_local]`` int32
    :returns: the same three tensors for the global batch ``[world * B_local, ...]``
              (every rank must contribute the same ``B_local``; pad the last shard).
    """
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return annotations, ids, counts
    world = dist.get_world_size(group)
    block = pack(annotations, ids, counts)
    buf = torch.empty((world,) + tuple(block.shape), dtype=block.dtype, device=block.device)
    if block.is_cuda:
        dist.all_gather_into_tensor(buf, block, group=group)          # RCCL over xGMI
    else:
        dist.all_gather(list(buf.unbind(0)), block, group=group)      # gloo (CPU tests)
    return unpack_block(buf.reshape(world * block.shape[0], block.shape[1]), annotations.shape[1], annotations.shape[2])


def shard_batch(n_items, rank, world_size):
    """The product path's split of a batch of ``n_items`` images (``decoder.CifCaf.batch`` under ``torch.distributed``):
    -> (lo, hi, per): this rank decodes images [lo, hi), every rank contributes ``per`` = ceil(n / world) rows to the gather
    (a shorter shard is padded: the collective wants equal blocks)."""
    lo, hi = shard_bounds(n_items, rank, world_size)
    return lo, hi, -(-n_items // world_size)


def merge_shards(annotations, ids, counts, n_items, world_size):
    """Gathered ``[world * per, ...]`` blocks -> the ``n_items`` rows of the global batch in image order (padding rows dropped)."""
    per = -(-n_items // world_size)
    keep = []
    for r in range(world_size):
        lo, hi = shard_bounds(n_items, r, world_size)
        keep.extend(range(r * per, r * per + (hi - lo)))
    idx = torch.as_tensor(keep, dtype=torch.long, device=annotations.device)
    return annotations.index_select(0, idx), ids.index_select(0, idx), counts.index_select(0, idx)


def unpack(annotations, ids, counts, max_annotations=None):
    """Device/host blocks -> per-image list of ``(ann [n,K,4] numpy, ids [n] numpy)``.

    Raises ``NativeError`` when an image of the (gathered) batch carries ``OPA_COUNT_FAILED`` -- on whatever rank it was
    decoded: a failed decode reports no rows, and handing that on as "nobody in the picture" would be a silent wrong answer
    (the same check ``native.CifCaf.call*`` and ``decoder.CifCaf.batch`` make on their own results)."""
    from . import native
    annotations, ids, counts = annotations.cpu().numpy(), ids.cpu().numpy(), counts.cpu().numpy()
    native.check_counts(counts)
    cap = annotations.shape[1] if max_annotations is None else max_annotations
    out = []
    for b in range(len(counts)):
        n = min(int(counts[b]) & native.COUNT_ROWS_MASK, cap)           # OPA_COUNT_ROWS: the valid rows
        out.append((annotations[b, :n], ids[b, :n]))
    return out


def broadcast_conv_choices(src=0, group=None):
    """Every rank adopts rank ``src``'s table of 1x1-convolution kernel choices (``fused.choices()``).  Each process
    picks gemm / pass+gemm / MIOpen per shape by wall clock on first use and the three round differently, so without
    this the ranks of one job could run -- slightly -- different networks.  Call it after the warm-up step in which
    the choices are made and before the first step that counts; a no-op without an initialised process group.
    ``src`` is a GLOBAL rank (what ``broadcast_object_list`` takes), also for a sub-group.  -> the table now in force."""
    import torch.distributed as dist
    from . import fused
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return fused.choices()
    box = [fused.choices() if dist.get_rank() == src else None]    # global rank, like broadcast_object_list's src
    dist.broadcast_object_list(box, src=src, group=group)
    fused.set_choices(box[0], replace=True)
    return fused.choices()
