"""lam_slide_amd: MI355X (gfx950) implementation of LaM-SLidE's second-stage latent SiT sampling path.

Public surface (mirrors the reference's for this path only):
  LatentSIV3                       <- src.models.components.latent.latent_si_v31.LatentSIV3
  CreateTransport, Transport, Sampler, ModelType, PathType
                                   <- src.modules.transport
  SecondStageSampler, setup_conditioning, sample_sharded
                                   <- SecondStageCondLightningBase.{sample, setup_conditioning} + batch sharding
  Stage1Decoder                    <- first_stage.decode = Decoder(post_quant(latents), entities) (frozen, after the sampler)
  Stage1Encoder                    <- quant(Encoder(prepare_inputs(batch), entities, mask)) (frozen, before the sampler)
  best_of_k_errors, min_ade_fde    <- the K-sample test loops + _compute_errors (second_stage/pedestrian.py:178-212)
  RolloutSampler, sample_rollout   <- SIAtom14SamplingWrapper.{create_batch, sample_rollout} (modules/sampling.py:16-63)
  install()                        <- rebinds the reference's module-level ``Sampler`` (lightning_base.py:10); see dropin.py
The compute lives in liblamslide_hip.so (include/lsl_api.h); build it with ``__graft_entry__.build()``.
"""
from . import _lib, dropin
from .dropin import install, uninstall
from .decoder import Stage1Decoder
from .encoder import Stage1Encoder
from .latent_si import LatentSIV3
from .sampling import (RolloutSampler, SecondStageSampler, best_of_k_errors, min_ade_fde, sample_rollout, sample_sharded,
                       setup_conditioning, shard_bounds)
from .transport import (CreateTransport, ModelType, PathType, Sampler, SampleResult, Transport, WeightType, as_transport, device_randn,
                        mix_seed)

__all__ = ["LatentSIV3", "CreateTransport", "Transport", "Sampler", "SampleResult", "ModelType", "PathType", "WeightType",
           "SecondStageSampler", "setup_conditioning", "sample_sharded", "shard_bounds", "min_ade_fde", "sample_rollout", "best_of_k_errors",
           "RolloutSampler", "Stage1Decoder", "Stage1Encoder", "as_transport", "device_randn", "mix_seed", "install", "uninstall", "dropin", "_lib"]
