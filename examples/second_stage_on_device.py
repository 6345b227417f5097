#!/usr/bin/env python3
"""The second-stage evaluation chain of LaM-SLidE on one MI355X with synthetic (seeded) weights: frozen stage-1 encode -> conditioning ->
K samples per scene in ONE fused sampler call -> frozen stage-1 decode -> best-of-K ADE / FDE, everything on the device.

Mirrors what `second_stage/pedestrian.py:186-226` does around `SecondStageCondLightningBase.sample` (lightning_base.py:217-238); with a
trained checkpoint, pass its state dicts instead of the seeded ones (same parameter names) and the dataset's `prepare_inputs` output as `x`.

    python examples/second_stage_on_device.py [--scenes 8] [--K 20]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler, Stage1Decoder, Stage1Encoder, min_ade_fde  # noqa: E402
from lam_slide_amd.synthetic import seeded_decoder_state_dict, seeded_encoder_state_dict, seeded_state_dict  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scenes", type=int, default=8)
    ap.add_argument("--K", type=int, default=20)
    ap.add_argument("--T", type=int, default=20)
    ap.add_argument("--agents", type=int, default=11)
    args = ap.parse_args()
    assert torch.cuda.is_available(), "needs the MI355X (the package has no CPU path)"
    dev = torch.device("cuda:0")
    B, K, T, A, L = args.scenes, args.K, args.T, args.agents, 8

    # frozen first stage (NBA-like shape: 8 latents of width 32 per frame) and the second-stage backbone
    enc = Stage1Encoder(seeded_encoder_state_dict(num_latents=L, seed=1), num_head_cross=8, dim_head_cross=16, num_head_latent=2, dim_head_latent=16)
    dec = Stage1Decoder(seeded_decoder_state_dict(out_dim=2, seed=2), num_head_latent=2, dim_head_latent=16, num_head_cross=8, dim_head_cross=16)
    net = LatentSIV3(depth=6, in_dim=32, hidden_size=256, num_heads=16, mlp_ratio=4, reset_parameters=False)
    net.load_state_dict(seeded_state_dict(net, seed=0))
    net.to(dev)

    g = torch.Generator().manual_seed(3)
    x = torch.randn(B * T, A, 128, generator=g).to(dev)                  # prepare_inputs(batch) of the dataset's first stage
    entities = torch.arange(A)[None].expand(B * T, A).contiguous().to(dev)
    mask = torch.ones(B * T, A, dtype=torch.bool, device=dev)
    target = torch.randn(B * A, T, 2, generator=g).to(dev)               # ground-truth positions of every agent

    def encode(_):
        z = enc.encode(x, entities, mask)
        return z.reshape(B, T, L, 32)

    drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=(0, 5), mask_cond_mean=True,
                             sampling_kwargs={"sampling_method": "euler", "num_steps": 51}, encode=encode, decode=dec)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    latents = drv.encode(None)                                            # once, not K times
    samples = drv.sample_latents_k(latents, K)                            # [K, B, T, L, C] from one fused call
    pos = dec.decode(samples.reshape(K * B * T, L, 32), entities.repeat(K, 1))   # [K*B*T, A, 2]
    pos = pos.reshape(K, B, T, A, 2).permute(1, 3, 0, 2, 4).reshape(B * A, K, T, 2)
    ade, fde = min_ade_fde(pos, target)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{B} scenes x {K} samples x {T} frames x {A} agents: {dt * 1e3:.1f} ms  ({B * K / dt:.0f} trajectories/s)  "
          f"min-ADE {float(ade.mean()):.3f}  min-FDE {float(fde.mean()):.3f}  (random weights: the numbers only show the plumbing)")


if __name__ == "__main__":
    main()
