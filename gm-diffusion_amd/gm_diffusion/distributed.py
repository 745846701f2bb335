"""
Multi-GPU: one process per GPU, the prompt batch sharded across ranks.

The denoising path has no cross-sample operation (GroupNorm, attention, the guidance-rescale std and
PNDM are all per sample), so ranks never exchange data inside the loop.  The only collectives are
the one-off RCCL broadcasts, over xGMI, of what rank 0 produced for the FULL batch:

  1. the text-encoder hidden states ``cat([negative, positive])`` (the north star's broadcast), and
  2. the initial latents -- drawn for the full batch from one generator exactly like the reference's
     ``randn_tensor`` (stable_diffusion_gm.py:709-710) so results do not depend on the GPU count;

each rank then slices its contiguous rows.  ``gather_outputs`` optionally collects the per-rank
results on rank 0 (ranks may equally write their own slices).  ``torch.distributed`` backend
``"nccl"`` is RCCL on ROCm; the same code runs under ``gloo`` on CPU tensors for the tests.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def shard_range(total, rank=None, world=None):
    """Contiguous [lo, hi) rows of ``total`` for this rank (the first ``total % world`` ranks get one extra)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _active(force=False):
    """Collectives run when there is more than one rank -- or, with ``force``, whenever a process group exists (a
    single-rank RCCL group: the one way to drive the collective path on a one-GPU box)."""
    return is_dist() or (force and dist.is_available() and dist.is_initialized())


def broadcast_from_rank0(t, shape, dtype, device, group=None, force=False):
    """Rank 0 passes the tensor, other ranks pass None and receive a tensor of the agreed shape."""
    if not _active(force):
        return t
    if dist.get_rank() != 0:
        t = torch.empty(shape, dtype=dtype, device=device)
    else:
        t = t.to(device=device, dtype=dtype).contiguous()
    dist.broadcast(t, src=0, group=group)
    return t


def shard_prompt_batch(prompt_embeds, negative_prompt_embeds, latents, total_batch, embed_shape, latent_shape,
                       dtype=torch.float32, device="cpu", group=None, force=False):
    """Broadcast the full-batch conditioning from rank 0 and return this rank's slice
    ``(prompt_embeds, negative_prompt_embeds, latents, (lo, hi))``.

    On rank 0 the three tensors are the full batch; on other ranks they are ignored (may be None)."""
    if not _active(force):
        return prompt_embeds, negative_prompt_embeds, latents, (0, total_batch)
    full_e = (total_batch,) + tuple(embed_shape)
    full_l = (total_batch,) + tuple(latent_shape)
    r0 = dist.get_rank() == 0
    both = torch.cat([negative_prompt_embeds, prompt_embeds]) if r0 else None  # one payload, [2B, L, E]
    both = broadcast_from_rank0(both, (2 * total_batch,) + tuple(embed_shape), dtype, device, group, force)
    latents = broadcast_from_rank0(latents if r0 else None, full_l, torch.float32, device, group, force)
    lo, hi = shard_range(total_batch)
    neg, pos = both[:total_batch], both[total_batch:]
    return pos[lo:hi].contiguous(), neg[lo:hi].contiguous(), latents[lo:hi].contiguous(), (lo, hi)


def gather_outputs(t, total_batch, group=None):
    """Gather per-rank row slices (possibly ragged) of ``t`` on rank 0; returns the full tensor there, None elsewhere."""
    if not is_dist():
        return t
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_range(total_batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    bufs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
    dist.gather(pad, bufs, dst=0, group=group)
    if rank != 0:
        return None
    return torch.cat([b[: hi - lo] for b, (lo, hi) in zip(bufs, sizes)])
