"""Caller-side harness of the sampling path: counterpart of ``SecondStageCondLightningBase.
{setup_conditioning, prepare_batch, sample}`` (lightning_base.py:205-263) without Lightning, plus the
multi-GPU sharding of independent trajectories (one process per GPU, one RCCL gather onto rank 0 at the end).

The frozen stage-1 encoder / decoder are NOT reimplemented here: ``encode`` / ``decode`` are callables
supplied by the caller (the reference's own first-stage model), exactly as the reference composes them.
"""
from __future__ import annotations

from typing import Any, Callable, Dict, Optional, Sequence, Tuple

import torch
import torch.distributed as dist
from torch import Tensor

from .latent_si import LatentSIV3
from .transport import Sampler, Transport, as_transport, device_randn, mix_seed


@torch.no_grad()
def setup_conditioning(latents: Tensor, cond_idx: Sequence[int], mask_cond_mean: bool) -> Tuple[Tensor, Tensor]:
    """lightning_base.py:240-263: frames [c0, c1) are visible; hidden frames see the mean of the visible
    ones (``mask_cond_mean``) or zeros.  Returns (x_cond, x_cond_mask int64)."""
    B, T, L, _ = latents.shape
    c0, c1 = int(cond_idx[0]), int(cond_idx[1])
    mask = torch.zeros(B, T, L, dtype=torch.int64, device=latents.device)
    mask[:, c0:c1] = 1
    visible = mask.unsqueeze(-1).bool()
    if mask_cond_mean:
        fill = latents[:, c0:c1].mean(dim=1).unsqueeze(1)
        x_cond = torch.where(visible, latents, fill)
    else:
        x_cond = torch.where(visible, latents, torch.zeros((), dtype=latents.dtype, device=latents.device))
    return x_cond, mask


def shard_bounds(total: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous split of ``total`` trajectories; the first ``total % world`` ranks get one extra."""
    q, r = divmod(total, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


class SecondStageSampler:
    """``sample(latents)``: conditioning -> noise -> fused sampler -> final latents (-> decode).

    Arguments mirror the reference wrapper's hparams (second_stage/md17.py:20-37): ``cond_idx``,
    ``mask_cond_mean``, ``sampling_method``, ``sampling_kwargs``.
    """

    def __init__(self, backbone: LatentSIV3, transport: Transport, cond_idx=(0, 10), mask_cond_mean: bool = True,
                 sampling_method: str = "ODE", sampling_kwargs: Optional[Dict[str, Any]] = None,
                 encode: Optional[Callable] = None, decode: Optional[Callable] = None, seed: int = 0):
        self.backbone = backbone
        self.si = as_transport(transport)
        self.cond_idx = tuple(cond_idx)
        self.mask_cond_mean = mask_cond_mean
        self.sampling_method = sampling_method
        self.sampling_kwargs = dict(sampling_kwargs or {"sampling_method": "euler", "num_steps": 10})
        self.encode, self.decode = encode, decode
        self.seed = seed
        self.calls = 0  # sampling calls made so far: call k draws its noise from stream mix_seed(seed, k)
        self.last_sampler: Optional[Sampler] = None

    def reseed(self, seed: Optional[int] = None):
        """Restart the noise streams: the next call is call 0 of ``seed`` (default: the current seed) again."""
        if seed is not None:
            self.seed = seed
        self.calls = 0

    def _next_call_seed(self) -> int:
        s = mix_seed(self.seed, self.calls)
        self.calls += 1
        return s

    def forward(self, xt: Tensor, t: Tensor, **model_kwargs) -> Tensor:  # lightning_base.py:173-174
        return self.backbone(x=xt, t=t, **model_kwargs)

    @torch.no_grad()
    def sample_latents(self, latents: Tensor, y: Optional[Tensor] = None, init: Optional[Tensor] = None,
                       first_index: int = 0, noise: Optional[Tensor] = None) -> Tensor:
        """latents: stage-1 latents [B,T,L,C] of the conditioning window (other frames are ignored).
        init: optional explicit initial noise (fixed-noise parity needs the tensor itself).  Without it every call draws fresh noise,
        like the reference's ``torch.randn_like(x_cond)`` per ``sample()`` (lightning_base.py:231) and ``randn`` per Euler-Maruyama step
        (integrators.py:30): call k of this object uses the counter stream ``mix_seed(seed, k)`` for the initial state and for the
        per-step noise, so K consecutive calls give K different samples; ``reseed()`` replays them.
        first_index: global index of ``latents[0]`` when this rank holds rows [first_index, first_index + B) of a sharded batch; the
        streams are indexed by GLOBAL element, so sharded and unsharded runs draw identical noise (every rank must have made the same
        number of calls)."""
        call_seed = self._next_call_seed()
        if latents.shape[0] == 0:  # an empty shard of a sharded run: the call is counted (see sample_sharded), nothing to sample
            return latents.new_zeros(latents.shape)
        x_cond, mask = setup_conditioning(latents, self.cond_idx, self.mask_cond_mean)
        elem_offset = int(first_index) * x_cond[0].numel()
        if init is None:
            init = device_randn(x_cond.shape, x_cond.device, call_seed, elem_offset).to(x_cond.dtype)
        sampler = Sampler(self.si, seed=call_seed)
        sampler.elem_offset = elem_offset
        kw = dict(self.sampling_kwargs)
        if self.sampling_method == "SDE" and noise is not None:
            fn = sampler.sample_sde(**{**{"sampling_method": "Euler", "diffusion_form": "linear", "diffusion_norm": 1.0,
                                          "last_step": "Mean", "last_step_size": 0.04, "num_steps": 250}, **kw}, noise=noise)
        else:
            fn = sampler.get_sample_fn(self.sampling_method, kw)
        mk = {"x_cond": x_cond, "x_cond_mask": mask}
        if y is not None:
            mk["y"] = y
        out = fn(init, self.forward, **mk)[-1]
        self.last_sampler = sampler
        return out

    @torch.no_grad()
    def sample_latents_k(self, latents: Tensor, K: int, y: Optional[Tensor] = None, inits: Optional[Tensor] = None) -> Tensor:
        """K samples per conditioning in ONE fused call.  The reference's test loops re-encode the identical batch and call
        ``sample`` K times in sequence (second_stage/pedestrian.py:193-204, nba.py:205-217, md17.py:157-166); trajectories are
        independent, so the K samples are folded into the batch: [B,...] -> [K*B,...] with the conditioning computed once.
        inits: optional [K,B,T,L,C] initial noises.  Returns [K,B,T,L,C]; sample k equals ``sample_latents(latents, init=inits[k])``
        bit for bit (batch independence)."""
        B = latents.shape[0]
        x_cond, mask = setup_conditioning(latents, self.cond_idx, self.mask_cond_mean)
        xc = x_cond.unsqueeze(0).expand(K, *x_cond.shape).reshape(K * B, *x_cond.shape[1:]).contiguous()
        mk = {"x_cond": xc, "x_cond_mask": mask.unsqueeze(0).expand(K, *mask.shape).reshape(K * B, *mask.shape[1:]).contiguous()}
        if y is not None:
            mk["y"] = y.unsqueeze(0).expand(K, *y.shape).reshape(K * B, *y.shape[1:]).contiguous()
        call_seed = self._next_call_seed()
        if inits is None:  # K * B fresh draws: the K samples of one conditioning differ, and so do consecutive calls
            init = device_randn(xc.shape, xc.device, call_seed).to(xc.dtype)
        else:
            init = inits.reshape(K * B, *inits.shape[2:])
        sampler = Sampler(self.si, seed=call_seed)
        fn = sampler.get_sample_fn(self.sampling_method, dict(self.sampling_kwargs))
        out = fn(init, self.forward, **mk)[-1]
        self.last_sampler = sampler
        return out.reshape(K, B, *out.shape[1:])

    @torch.no_grad()
    def sample(self, batch: Dict[str, Tensor]) -> Dict[str, Tensor]:
        """Full counterpart of lightning_base.py:217-238; needs ``encode`` and ``decode``."""
        if self.encode is None or self.decode is None:
            raise RuntimeError("sample(batch) needs the frozen stage-1 encode/decode callables")
        latents = self.encode(batch)
        final = self.sample_latents(latents, y=batch.get("y"), first_index=int(batch.get("first_index", 0)))
        B = final.shape[0]
        flat = final.reshape(B * final.shape[1], *final.shape[2:])
        ent = batch["entities"].reshape(B * batch["entities"].shape[1], *batch["entities"].shape[2:])
        return self.decode(flat, ent)


@torch.no_grad()
def sample_sharded(sample_fn: Callable[[Tensor, int], Tensor], latents: Tensor, group=None, dst: Optional[int] = 0) -> Optional[Tensor]:
    """Trajectories are independent end to end (no op mixes batch elements), so the batch is split
    contiguously over the ranks with no collective on the data path; ONE gather of the final latents onto rank
    ``dst`` closes the job (BASELINE north_star: "a single RCCL gather over xGMI at the end"; with backend "nccl" it
    is RCCL).  ``sample_fn(local_latents, first_global_index) -> local_final``.  Rank ``dst`` returns the full result in
    the original order, every other rank ``None`` - nobody needs the other shards back, and a gather moves 1 / world
    of what an all_gather does.  ``dst=None``: every rank gets the full result (an all_gather)."""
    if not (dist.is_available() and dist.is_initialized()):
        return sample_fn(latents, 0)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    B = latents.shape[0]
    lo, hi = shard_bounds(B, world, rank)
    # (a rank with an empty shard still calls sample_fn: stateful samplers count their calls to key the noise streams, and every rank
    # has to stay on the same call number as the unsharded run)
    local = sample_fn(latents[lo:hi], lo)
    sizes = [shard_bounds(B, world, r)[1] - shard_bounds(B, world, r)[0] for r in range(world)]
    cap = max(sizes)  # (shards differ by at most one trajectory: the collective moves equal-sized buffers)
    pad = local.new_zeros((cap,) + tuple(latents.shape[1:]))
    pad[: hi - lo] = local
    if dst is None:
        gathered = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(gathered, pad.contiguous(), group=group)
    else:
        gathered = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad.contiguous(), gathered, dst=dist.get_global_rank(group, dst) if group is not None else dst, group=group)
        if rank != dst:
            return None
    return torch.cat([g[:s] for g, s in zip(gathered, sizes)], dim=0)


@torch.no_grad()
def min_ade_fde(trajectories: Tensor, target: Tensor) -> Tuple[Tensor, Tensor]:
    """Best-of-K displacement errors on the device (second_stage/pedestrian.py:178-185 ``_compute_errors``).
    trajectories: [N, K, T, D] predicted positions of N agents; target: [N, T, D].
    Returns (ADE, FDE), each [N]: the minimum over K of the time-averaged / final-frame L2 error."""
    err = torch.norm(trajectories - target[:, None], dim=-1)  # [N, K, T]
    return err.mean(dim=-1).min(dim=1).values, err[..., -1].min(dim=1).values


@torch.no_grad()
def sample_rollout(sample_positions: Callable[[Tensor], Tensor], cond_pos: Tensor, num_rollouts: int = 1, shift: float = 0.0,
                   scale: float = 1.0) -> Tensor:
    """Chained rollouts of one system (modules/sampling.py:44-63, ``SIAtom14SamplingWrapper.sample_rollout``): each rollout is
    conditioned on the last frame of the previous one; everything stays on the device of ``cond_pos`` (no host round trips
    between rollouts).  ``sample_positions(pos [R, A, D]) -> [T, R, A, D]`` is one conditioned sample (encode -> sampler -> decode,
    e.g. ``SecondStageSampler.sample`` on the batch built from ``pos``).  Rollouts of ONE system are sequential by construction;
    shard over systems, not over rollouts."""
    cond = (cond_pos - shift) / scale
    pos = cond.clone()
    rollouts = []
    for _ in range(num_rollouts):
        pred = sample_positions(pos)
        rollouts.append(pred)
        pos = pred[-1].clone()
    positions = torch.cat(rollouts)
    positions[0] = cond
    return positions * scale + shift


@torch.no_grad()
def best_of_k_errors(drv: "SecondStageSampler", latents: Tensor, target_pos: Tensor, K: int, decode: Callable[[Tensor], Tensor],
                     agent_mask: Optional[Tensor] = None, y: Optional[Tensor] = None, inits: Optional[Tensor] = None,
                     num_runs: Optional[int] = None) -> Tuple[Tensor, Tensor]:
    """The evaluation tail of the trajectory models on the device (second_stage/pedestrian.py:186-212, nba.py:205-225): K samples per
    scene, decoded, future frames only, one row per real agent, best-of-K ADE / FDE.  The reference runs K sequential ``sample()``
    calls (re-encoding the same batch each time) and stacks the results on the host side of the loop; here the K samples are one fused
    call (``sample_latents_k``), one decode, and the reductions of :func:`min_ade_fde`, with no host round trip in between.

    latents [B,T,L,C] stage-1 latents; target_pos [B, T - c1, A, D] true future positions; ``decode(latents [K*B,T,L,C]) ->
    positions [K*B, T, A, D]``; agent_mask [B, A] bool (``attention_mask[:, -1]``) or None.  Returns (ADE, FDE) per real agent."""
    B = latents.shape[0]
    c1 = drv.cond_idx[1]
    final = drv.sample_latents_k(latents, K, y=y, inits=inits)            # [K, B, T, L, C]
    pos = decode(final.reshape(K * B, *final.shape[2:]))                   # [K*B, T, A, D]
    pos = pos.reshape(K, B, *pos.shape[1:])[:, :, c1:]                     # future frames
    traj = pos.permute(1, 3, 0, 2, 4).reshape(B * pos.shape[3], K, pos.shape[2], pos.shape[4])   # "(B A) K T D"
    tgt = target_pos.permute(0, 2, 1, 3).reshape(B * target_pos.shape[2], target_pos.shape[1], target_pos.shape[3])
    if agent_mask is not None:
        keep = agent_mask.reshape(-1).bool()
        traj, tgt = traj[keep], tgt[keep]
    if num_runs is not None:
        traj = traj[:, :num_runs]
    return min_ade_fde(traj, tgt)


class RolloutSampler:
    """Counterpart of ``SIAtom14SamplingWrapper`` (modules/sampling.py:16-63) for the tensors it handles: ``create_batch`` builds the
    one-system batch whose every frame repeats the conditioning frame, ``sample_rollout`` chains ``num_rollouts`` samples, each
    conditioned on the last frame of the previous one.  ``model`` is anything with ``sample(batch) -> {"atom14_pos": [1*T, R, A, D] or
    [1, T, R, A, D]}``, ``shift``, ``scale`` and ``hparams.n_timesteps`` (or ``n_timesteps``) - the reference's peptide LightningModule,
    or a :class:`SecondStageSampler` wired with device encode / decode.  Everything stays on ``cond_pos``'s device; the mdtraj
    conversion (``sample_traj``) stays the reference's."""

    def __init__(self, model, key: str = "atom14_pos"):
        self.model = model
        self.key = key

    def _T(self) -> int:
        hp = getattr(self.model, "hparams", None)
        if hp is not None and hasattr(hp, "n_timesteps"):
            return int(hp.n_timesteps)
        return int(self.model.n_timesteps)

    def create_batch(self, pos: Tensor, res: Tensor, res_mask: Tensor) -> Dict[str, Tensor]:
        T = self._T()
        pos = pos * res_mask[..., None].to(pos.device)
        R = res.shape[0]
        return {
            self.key: pos[None, None].expand(1, T, *pos.shape),
            "aatype": res[None, None].expand(1, T, R),
            "attention_mask": torch.ones(1, T, R, dtype=torch.bool, device=res.device),
            "entities": torch.arange(R, device=res.device).expand(T, -1).unsqueeze(0),
        }

    @torch.no_grad()
    def sample_rollout(self, cond_pos: Tensor, res: Tensor, res_mask: Tensor, num_rollouts: int = 1) -> Tensor:
        def one(pos):
            out = self.model.sample(self.create_batch(pos=pos, res=res, res_mask=res_mask))[self.key]
            return out.squeeze(0) if out.dim() == cond_pos.dim() + 2 else out

        return sample_rollout(one, cond_pos, num_rollouts, shift=self.model.shift, scale=self.model.scale)
