#!/usr/bin/env python3
"""BASELINE configs[2] end to end on the device, per scene: frozen stage-1 encode -> K = 20 second-stage samples (one fused call, the
trajectory-resident kernel) -> frozen stage-1 decode of the K * T frames -> best-of-K ADE / FDE (second_stage/pedestrian.py:186-212).
Wall time per stage (host clock around synchronised calls, median of 30) with seeded synthetic weights of the pedestrian shapes.
    python tools/cfg3_end_to_end.py [agents per scene] [scenes per call]"""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS  # noqa: E402
from lam_slide_amd import CreateTransport, LatentSIV3, SecondStageSampler, Stage1Decoder, Stage1Encoder, min_ade_fde  # noqa: E402
from lam_slide_amd.synthetic import seeded_decoder_state_dict, seeded_encoder_state_dict, seeded_state_dict  # noqa: E402


def main():
    A = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    kw, T, L, cond_idx, method, skw, _ = WORKLOADS["pedestrian_scene"]
    K, C = 20, kw["in_dim"]
    dev = torch.device("cuda", 0)
    net = LatentSIV3(reset_parameters=False, **kw)
    net.load_state_dict(seeded_state_dict(net, seed=0))
    net.to(dev)
    drv = SecondStageSampler(net, CreateTransport("GVP", "data")(), cond_idx=cond_idx, mask_cond_mean=True, sampling_method=method,
                             sampling_kwargs=skw, seed=7)
    enc = Stage1Encoder(seeded_encoder_state_dict(num_latents=L, seed=6), num_head_cross=8, dim_head_cross=16, num_head_latent=2, dim_head_latent=16)
    dec = Stage1Decoder(seeded_decoder_state_dict(seed=7), num_head_latent=2, dim_head_latent=16, num_head_cross=8, dim_head_cross=16)
    g = torch.Generator().manual_seed(1)
    feats = torch.randn(S * T, A, 128, generator=g).to(dev)
    ent = torch.arange(A, device=dev)[None].expand(S * T, A).contiguous()
    emask = torch.ones(S * T, A, dtype=torch.bool, device=dev)
    y = torch.randn(S, kw["vec_in_dim"], generator=g).to(dev)
    target = torch.randn(S * A, T - cond_idx[1], 3, generator=g).to(dev)
    entk = ent.repeat(K, 1)

    stages = {}

    def timed(name, fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        stages.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return out

    for rep in range(35):
        z = timed("encode", lambda: enc.encode(feats, ent, emask).reshape(S, T, L, C))
        fin = timed("sample_k20", lambda: drv.sample_latents_k(z, K, y=y))                        # [K, S, T, L, C]
        pos = timed("decode", lambda: dec.decode(fin.reshape(K * S * T, L, C), entk).reshape(K, S, T, A, 3))
        timed("best_of_k", lambda: min_ade_fde(pos[:, :, cond_idx[1]:].permute(1, 3, 0, 2, 4).reshape(S * A, K, T - cond_idx[1], 3), target))
        if rep == 4:
            stages.clear()  # warm-up
    assert drv.last_sampler.last_kernels == "resident"
    tot = 0.0
    print(f"cfg 3 end to end: {S} scene(s) x {A} agents, T = {T}, L = {L}, K = {K} samples per scene (median of 30, ms)")
    for k, v in stages.items():
        m = statistics.median(v)
        tot += m
        print(f"  {k:12s} {m:8.3f}   (min {min(v):.3f})")
    print(f"  {'total':12s} {tot:8.3f}   = {S / tot * 1e3:.0f} scenes/s, {S * K / tot * 1e3:.0f} sampled trajectories/s end to end")


if __name__ == "__main__":
    main()
