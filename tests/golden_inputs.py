"""Seeded input generators shared by tools/make_fixtures.py (build container) and the tests: inputs too large to store in a fixture are
regenerated from the seed the fixture holds (torch's CPU generator is the same stream on every box)."""
import torch


def peptide_frames(seed: int, B: int, T: int, R: int):
    """F13: atom14 coordinates of B systems x T frames x R residues - a base conformation plus a slow random walk over the frames (frames of one
    trajectory, not independent noise) -, residue types, entity ids, mask.  -> dict of the batch keys second_stage/peptide.py:85-95 reads."""
    g = torch.Generator().manual_seed(seed)
    base = torch.randn(B, 1, R, 14, 3, generator=g)
    drift = torch.cumsum(torch.randn(B, T, R, 14, 3, generator=g) * 0.03, dim=1)
    return {"atom14_pos": base + drift,
            "aatype": torch.randint(0, 20, (B, 1, R), generator=g).expand(B, T, R).contiguous(),
            "entities": torch.arange(R)[None, None].expand(B, T, R).contiguous(),
            "attention_mask": torch.ones(B, T, R, dtype=torch.bool)}
