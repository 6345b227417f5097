"""Build-container-only environment in which the reference's REAL second-stage LightningModule imports and runs.

The image has neither Lightning / Hydra / omegaconf / torchmetrics nor torchdiffeq, and ``src/utils/__init__.py`` pulls in rich / wandb
tooling.  None of that is on the sampling path; what is on it - ``models/composites/lightning_base.py`` (``sample``, ``prepare_batch``,
``setup_conditioning``), ``models/composites/second_stage/md17.py`` (``Wrapper.__init__`` / ``encode`` / ``decode``), ``modules/transport/*``,
the backbone and the stage-1 encoder / decoder - is imported from ``/root/reference`` UNCHANGED.  This module provides the smallest
stand-ins that let those files import:

    lightning.LightningModule      nn.Module + save_hyperparameters / hparams / device / freeze (what the two files use)
    hydra.utils.instantiate        ``_target_`` / ``_partial_`` / ``_recursive_`` on plain dicts
    omegaconf.DictConfig           = AttrDict (dict with attribute access)
    torchmetrics.MeanMetric        an nn.Module that is never updated here
    torchdiffeq.odeint             fixed-grid Euler with the package's published semantics (grid == t, states stacked; method == "euler")
    torch_kmeans.KMeans            a name only (second_stage/pedestrian.py imports it; used only with post_process=True, which no config sets)
    src.datasets.pedestrian        only ``dataset_cond_indices`` (the five scene names; the real module imports the dataset stack)
    src.utils (package shell)      real ``pylogger`` / ``tensor_utils`` are imported from the reference, ``__init__`` is not executed;
                                   ``src.utils.utils.load_class`` is the reference's four lines of importlib (checkpoint plumbing)
    src.datasets.md17              only ``dataset_cond_indices`` (a dict of 8 molecule names; the real module imports the dataset stack)
    lightning_utilities.core.rank_zero   the two names pylogger imports

Used by ``tools/make_fixtures.py f9 f11`` and by ``tests/test_dropin.py`` (skipped where /root/reference does not exist).  Nothing here ships.
"""
from __future__ import annotations

import functools
import importlib
import inspect
import os
import sys
import types

import torch
from torch import nn

REF = "/root/reference"


class AttrDict(dict):
    """dict with attribute access, nested (what the two files need of omegaconf.DictConfig / Lightning's AttributeDict)."""

    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v

    def __setattr__(self, k, v):
        self[k] = v


def instantiate(cfg, *args, **kwargs):
    """hydra.utils.instantiate for plain dict configs: `_target_`, `_partial_`, `_recursive_` (default True), `_convert_` ignored."""
    cfg = dict(cfg)
    target = cfg.pop("_target_")
    partial = cfg.pop("_partial_", False)
    recursive = cfg.pop("_recursive_", True)
    cfg.pop("_convert_", None)
    mod, name = target.rsplit(".", 1)
    fn = getattr(importlib.import_module(mod), name)
    kw = {}
    for k, v in cfg.items():
        if isinstance(v, dict) and "_target_" in v and recursive:
            v = instantiate(v)
        elif isinstance(v, dict):
            v = AttrDict(v)
        kw[k] = v
    kw.update(kwargs)
    return functools.partial(fn, *args, **kw) if partial else fn(*args, **kw)


class LightningModule(nn.Module):
    def __init__(self):
        super().__init__()
        self._hparams = AttrDict()
        self._device = torch.device("cpu")

    @property
    def hparams(self):
        return self._hparams

    @property
    def device(self):
        return self._device

    def save_hyperparameters(self, *a, logger=True, **k):
        frame = inspect.currentframe().f_back
        init_args = {n: v for n, v in frame.f_locals.items() if n not in ("self", "__class__") and not n.startswith("_")}
        if "kwargs" in init_args and isinstance(init_args["kwargs"], dict):
            init_args.update(init_args.pop("kwargs"))
        self._hparams.update(init_args)

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def log(self, *a, **k):
        pass

    def log_dict(self, *a, **k):
        pass


class _Ema:
    """What `first_stage_model.load_ema_weights()` reads (lightning_base.py:63-70): an object whose state_dict()["params"] loads into the model."""

    def __init__(self, model):
        self._params = {k: v.detach().clone() for k, v in model.state_dict().items()}

    def state_dict(self):
        return {"params": self._params}


_done = False


def setup():
    """Install the stand-ins in sys.modules and put /root/reference on sys.path.  Idempotent.  Returns the namespace of real modules."""
    global _done
    if not os.path.isdir(REF):
        raise RuntimeError("the reference checkout is not available (build container only)")
    if not _done:
        if REF not in sys.path:
            sys.path.insert(0, REF)

        def mod(name, **attrs):
            m = types.ModuleType(name)
            m.__dict__.update(attrs)
            sys.modules[name] = m
            return m

        mod("lightning", LightningModule=LightningModule)
        mod("hydra", utils=mod("hydra.utils", instantiate=instantiate))
        mod("omegaconf", DictConfig=AttrDict)

        class MeanMetric(nn.Module):
            def forward(self, *a, **k):
                return None

        mod("torchmetrics", MeanMetric=MeanMetric)

        def odeint(f, y0, t, method=None, atol=None, rtol=None, **kw):
            assert method == "euler", "the stand-in knows the fixed-grid Euler only"
            ys, y = [y0], y0
            for i in range(len(t) - 1):
                y = y + (t[i + 1] - t[i]) * f(t[i], y)
                ys.append(y)
            return torch.stack(ys)

        mod("torchdiffeq", odeint=odeint)

        class KMeans:  # (second_stage/pedestrian.py:8; instantiated only under post_process=True)
            def __init__(self, *a, **k):
                raise RuntimeError("torch_kmeans is not available in the build container (post_process=True is not covered)")

        mod("torch_kmeans", KMeans=KMeans)
        mod("src.datasets.pedestrian", dataset_cond_indices={"zara1": 0, "zara2": 1, "univ": 2, "hotel": 3, "eth": 4})  # (datasets/pedestrian.py:14-20)
        rz = lambda fn: fn  # noqa: E731
        rz.rank = 0
        mod("lightning_utilities", core=mod("lightning_utilities.core", rank_zero=mod(
            "lightning_utilities.core.rank_zero", rank_zero_only=rz, rank_prefixed_message=lambda msg, rank: f"[rank: {rank}] {msg}")))
        # src.utils: a package shell over the real directory (pylogger / tensor_utils import from the reference; __init__ is skipped)
        pkg = mod("src.utils")
        pkg.__path__ = [os.path.join(REF, "src", "utils")]

        def load_class(class_string):  # the reference's own four lines (src/utils/utils.py:125-129)
            module_name, class_name = class_string.rsplit(".", 1)
            return getattr(importlib.import_module(module_name), class_name)

        mod("src.utils.utils", load_class=load_class)
        pkg.RankedLogger = importlib.import_module("src.utils.pylogger").RankedLogger  # (what src/utils/__init__.py re-exports; modules/geometry.py:6)
        # second_stage/peptide.py imports geometry / residue constants (not on the sampling path): their third-party imports get shells
        class PDBParser:  # (modules/protein.py:23; only instantiated when a PDB file is parsed)
            def __init__(self, *a, **k):
                raise RuntimeError("Bio.PDB is not available in the build container")

        bio = mod("Bio")
        bio.PDB = mod("Bio.PDB", PDBParser=PDBParser)

        def map_structure(fn, st):  # (dm-tree's map_structure on the nested lists of utils/residue_constants.py:1065)
            if isinstance(st, (list, tuple)):
                return type(st)(map_structure(fn, x) for x in st)
            if isinstance(st, dict):
                return {k: map_structure(fn, v) for k, v in st.items()}
            return fn(st)

        mod("tree", map_structure=map_structure)
        mod("src.datasets.nba", dataset_cond_indices={"score": 0, "rebound": 1})  # (datasets/nba.py:26-29; the module itself needs easydict / joblib)
        mod("src.datasets.md17", dataset_cond_indices={n: i for i, n in enumerate(
            ("aspirin", "benzene", "ethanol", "malonaldehyde", "naphthalene", "salicylic", "toluene", "uracil"))})
        _done = True
    ns = types.SimpleNamespace()
    ns.lightning_base = importlib.import_module("src.models.composites.lightning_base")
    ns.md17 = importlib.import_module("src.models.composites.second_stage.md17")
    ns.transport_pkg = importlib.import_module("src.modules.transport")
    ns.transport = importlib.import_module("src.modules.transport.transport")
    ns.latent = importlib.import_module("src.models.components.latent.latent_si_v31")
    ns.encoder = importlib.import_module("src.models.components.encoder")
    ns.decoder = importlib.import_module("src.models.components.decoder")
    ns.entity = importlib.import_module("src.modules.entity_embeddings")
    ns.torch_modules = importlib.import_module("src.modules.torch_modules")
    return ns


# ---- a frozen first-stage model built from the reference's own classes (BackboneBase / FirstStageLightningBase / Encoder / Decoder) ----

F9 = dict(B=2, T=6, A=5, L=16, dim_input=128, dim_latent=32, n_entities=32, cond_idx=[0, 2], num_steps=6,
          backbone=dict(depth=2, in_dim=32, hidden_size=64, mlp_ratio=2, num_heads=4))
_STAGE1 = {}


def build_first_stage(ns, seed=21, num_latents=None, lift=None):
    """FirstStageLightningBase (real class) around a BackboneBase (real class) with the real Encoder / Decoder; `prepare_inputs` - the
    first stage's atom / position embedding, not on this path - hands through batch["pos"], which the F9 inputs carry already merged."""
    from functools import partial
    lb = ns.lightning_base
    torch.manual_seed(seed)
    emb = ns.entity.EntityEmbeddingOrthogonal(n_entiy_embeddings=F9["n_entities"], embedding_dim=128, max_norm=1)
    act = partial(ns.torch_modules.GELU)
    enc = ns.encoder.Encoder(dim_input=F9["dim_input"], dim_latent=F9["dim_latent"], dim_head_cross=16, dim_head_latent=16, num_latents=num_latents or F9["L"],
                             num_head_cross=8, num_head_latent=2, num_block_cross=1, num_block_attn=1, qk_norm=True, entity_embedding=emb, act=act)
    dec = ns.decoder.Decoder(outputs={"pos": 3}, dim_query=128, dim_latent=F9["dim_latent"], entity_embedding=emb, dim_head_cross=16,
                             dim_head_latent=16, num_head_cross=8, num_head_latent=2, num_block_cross=0, num_block_attn=1, dropout_query=0.1,
                             qk_norm=True, act=act)
    with torch.no_grad():
        emb.embedding.weight.mul_(torch.linspace(0.5, 1.8, F9["n_entities"])[:, None])  # some rows above unit norm: the max_norm path

    class Backbone(lb.BackboneBase):
        def prepare_inputs(self, batch):  # (F11: positions [.., 3] lifted to the encoder's input width by a fixed matrix stored in the fixture)
            return batch["pos"] if lift is None else batch["pos"] @ lift

    class FirstStage(lb.FirstStageLightningBase):
        def __init__(self, backbone):
            super().__init__()
            self.hparams.update(shift=0.0, scale=1.0, ema=None)
            self.backbone = backbone

        @classmethod
        def load_from_checkpoint(cls, path, map_location=None):  # what second_stage/md17.py:46-48 calls; the "checkpoint" is the seeded model
            model = _STAGE1[path]
            model.ema = _Ema(model)
            return model

    model = FirstStage(Backbone(dim_latent=F9["dim_latent"], encoder=enc, decoder=dec))
    return model, FirstStage


def build_wrapper(ns, backbone_target, transport_target, first_stage, first_stage_cls, seed=22):
    """The reference's real second-stage Wrapper (second_stage/md17.py:17-64), constructed by ITS OWN __init__ from the reference's own
    YAML block (configs/model/md17/second-stage.yaml) with the F9 sizes and the given `_target_`s."""
    import yaml
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/model/md17/second-stage.yaml")))
    cfg.pop("_target_"), cfg.pop("_recursive_")
    cfg.update(compile=False, n_atom_types=10, num_timesteps=F9["T"], cond_idx=list(F9["cond_idx"]), ema=None, scheduler=None,
               sampling_method="ODE", sampling_kwargs={"sampling_method": "euler", "num_steps": F9["num_steps"]})
    cfg["backbone"] = dict(cfg["backbone"], _target_=backbone_target, **F9["backbone"])
    cfg["transport"] = dict(cfg["transport"], _target_=transport_target)
    key = f"f9-stage1-{id(first_stage)}"
    _STAGE1[key] = first_stage
    mod_name = "_lsl_f9_first_stage"
    sys.modules.setdefault(mod_name, types.ModuleType(mod_name)).FirstStage = first_stage_cls
    cfg["first_stage_model"] = {"class_name": f"{mod_name}.FirstStage", "path": key}
    torch.manual_seed(seed)
    return ns.md17.Wrapper(**{k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})


def f9_batch(seed=23):
    g = torch.Generator().manual_seed(seed)
    B, T, A = F9["B"], F9["T"], F9["A"]
    return {"pos": torch.randn(B, T, A, F9["dim_input"], generator=g),  # (already merged stage-1 inputs: see build_first_stage)
            "entities": torch.stack([torch.randperm(F9["n_entities"], generator=g)[:A] for _ in range(B)])[:, None].expand(B, T, A).contiguous(),
            "attention_mask": torch.ones(B, T, A, dtype=torch.bool)}


class fixed_randn_like:
    """Context manager: torch.randn_like returns the given tensor (the reference draws its initial state with it, lightning_base.py:231)."""

    def __init__(self, noise):
        self.noise = noise

    def __enter__(self):
        self._orig = torch.randn_like
        torch.randn_like = lambda x, **kw: self.noise.to(x.dtype).clone()
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig


# ---- F11: the reference's real pedestrian CondWrapper (second_stage/pedestrian.py), its prepare_batch and its K-sample test_step ----

F11 = dict(B=3, T=20, A=4, L=2, K=20, cond_idx=[0, 8], num_steps=11, n_classes=5, vec_in_dim=256,
           backbone=dict(depth=2, in_dim=32, hidden_size=128, mlp_ratio=2, num_heads=4, normalize=True, vec_in_dim=256))


def build_pedestrian_wrapper(ns, first_stage, first_stage_cls, seed=32):
    """second_stage/pedestrian.py CondWrapper, constructed by ITS OWN __init__ from the reference's own YAML blocks
    (configs/model/pedestrian/second-stage.yaml + second-stage_cond.yaml) with the F11 sizes (true T = 20, L = 2; depth 2 instead of 6)."""
    import yaml
    ped = importlib.import_module("src.models.composites.second_stage.pedestrian")
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/model/pedestrian/second-stage.yaml")))
    cond = yaml.safe_load(open(os.path.join(REF, "configs/model/pedestrian/second-stage_cond.yaml")))
    for k in ("_target_", "_recursive_", "defaults"):
        cfg.pop(k, None)
    assert cond["_target_"].endswith("pedestrian.CondWrapper") and cfg["K"] == 20 and cfg["num_runs"] == 20 and cfg["cond_idx"] == F11["cond_idx"]
    cfg.update(compile=False, num_timesteps=F11["T"], ema=None, scheduler=None, n_classes=cond["n_classes"], vec_in_dim=cond["vec_in_dim"],
               sampling_method="ODE", sampling_kwargs={"sampling_method": "euler", "num_steps": F11["num_steps"]})
    bb = {k: v for k, v in cfg["backbone"].items() if k != "n_timesteps"}
    cfg["backbone"] = dict(bb, n_timesteps=F11["T"], **F11["backbone"])
    key = f"f11-stage1-{id(first_stage)}"
    _STAGE1[key] = first_stage
    mod_name = "_lsl_f11_first_stage"
    sys.modules.setdefault(mod_name, types.ModuleType(mod_name)).FirstStage = first_stage_cls
    cfg["first_stage_model"] = {"class_name": f"{mod_name}.FirstStage", "path": key}
    torch.manual_seed(seed)
    w = ped.CondWrapper(**{k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
    # what Lightning provides around test_step: the datamodule's name of the dataloader (second_stage/pedestrian.py:162)
    w.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(dataloader_names=lambda idx: "eth"))
    return w


def f11_batch(seed=33):
    g = torch.Generator().manual_seed(seed)
    B, T, A = F11["B"], F11["T"], F11["A"]
    am = torch.ones(B, T, A, dtype=torch.bool)
    am[1, :, 3] = False  # a padded agent: its rows are dropped from the error statistics (pedestrian.py:171-173)
    am[2, :, 2:] = False
    return {"pos": torch.randn(B, T, A, F9["dim_input"], generator=g),
            "entities": torch.stack([torch.randperm(F9["n_entities"], generator=g)[:A] for _ in range(B)])[:, None].expand(B, T, A).contiguous(),
            "attention_mask": am, "cond_scene": torch.tensor([4, 0, 2])}


# ---- F12: the reference's real NBA CondWrapper (second_stage/nba.py): prepare_batch and the K = 60 / num_runs = 20 test_step ----

F12 = dict(B=2, T=20, A=5, L=8, K=60, num_runs=20, cond_idx=[0, 8], num_steps=6, n_classes=2, vec_in_dim=256,
           backbone=dict(depth=2, in_dim=32, hidden_size=256, mlp_ratio=4, num_heads=16, normalize=True, vec_in_dim=256))


def build_nba_wrapper(ns, first_stage, first_stage_cls, seed=42):
    """second_stage/nba.py CondWrapper, constructed by ITS OWN __init__ from the reference's own YAML blocks (configs/model/nba/second-stage.yaml
    + second-stage_cond.yaml) with the F12 sizes (true T = 20, L = 8, hidden 256, 16 heads, mlp 4; depth 2 instead of 6).  K and num_runs are
    not in the YAML: the class defaults (60 / 20, nba.py:33-35) apply."""
    import yaml
    nba = importlib.import_module("src.models.composites.second_stage.nba")
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/model/nba/second-stage.yaml")))
    cond = yaml.safe_load(open(os.path.join(REF, "configs/model/nba/second-stage_cond.yaml")))
    for k in ("_target_", "_recursive_", "defaults"):
        cfg.pop(k, None)
    assert cond["_target_"].endswith("nba.CondWrapper") and "K" not in cfg and "num_runs" not in cfg and cfg["cond_idx"] == F12["cond_idx"]
    cfg.update(compile=False, num_timesteps=F12["T"], ema=None, scheduler=None, n_classes=cond["n_classes"], vec_in_dim=cond["vec_in_dim"],
               sampling_method="ODE", sampling_kwargs={"sampling_method": "euler", "num_steps": F12["num_steps"]})
    cfg["backbone"] = dict(cfg["backbone"], **F12["backbone"])
    key = f"f12-stage1-{id(first_stage)}"
    _STAGE1[key] = first_stage
    mod_name = "_lsl_f12_first_stage"
    sys.modules.setdefault(mod_name, types.ModuleType(mod_name)).FirstStage = first_stage_cls
    cfg["first_stage_model"] = {"class_name": f"{mod_name}.FirstStage", "path": key}
    torch.manual_seed(seed)
    w = nba.CondWrapper(**{k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})
    assert w.hparams.K == F12["K"] and w.hparams.num_runs == F12["num_runs"]
    w.trainer = types.SimpleNamespace(datamodule=types.SimpleNamespace(dataloader_names=lambda idx: "score"))
    return w


def f12_batch(seed=43):
    g = torch.Generator().manual_seed(seed)
    B, T, A = F12["B"], F12["T"], F12["A"]
    am = torch.ones(B, T, A, dtype=torch.bool)
    am[1, :, 4] = False  # a padded agent: dropped from the error statistics (nba.py:184-186)
    return {"pos": torch.randn(B, T, A, 3, generator=g),
            "entities": torch.stack([torch.randperm(F9["n_entities"], generator=g)[:A] for _ in range(B)])[:, None].expand(B, T, A).contiguous(),
            "attention_mask": am, "cond_scene": torch.tensor([1, 0])}


# ---- F13: the reference's real peptide second-stage Wrapper (second_stage/peptide.py) at the peptide shape ----

F13 = dict(B=2, T=1000, R=4, L=2, cond_idx=[0, 1], num_steps=4, n_entities=32, dim_input=256, dim_latent=96,
           backbone=dict(depth=2, in_dim=96, hidden_size=384, mlp_ratio=4, num_heads=16))


def build_peptide_first_stage(ns, lift, seed=51):
    """FirstStageLightningBase (real class) around a BackboneBase (real class) with the real Encoder / DecoderQuerySplitter at the sizes of
    configs/model/peptide/first-stage.yaml (dim_input 256, dim_latent 96, 2 latents, 2 + 2 heads of 16, num_split 8, heads atom14_pos (42)
    and aatype (20)); `encode` passes mask=None like first_stage/peptide.py:77-80; `prepare_inputs` - the residue / position embedding
    of the first stage, not on this path - lifts the flattened atom14 coordinates with a fixed matrix stored in the fixture."""
    from functools import partial
    lb = ns.lightning_base
    torch.manual_seed(seed)
    emb = ns.entity.EntityEmbeddingOrthogonal(n_entiy_embeddings=F13["n_entities"], embedding_dim=128, max_norm=1)
    act = partial(ns.torch_modules.GELU)
    enc = ns.encoder.Encoder(dim_input=F13["dim_input"], dim_latent=F13["dim_latent"], dim_head_cross=16, dim_head_latent=16, num_latents=F13["L"],
                             num_head_cross=2, num_head_latent=2, num_block_cross=1, num_block_attn=1, qk_norm=True, entity_embedding=emb, act=act)
    dec = ns.decoder.DecoderQuerySplitter(outputs={"atom14_pos": 42, "aatype": 20}, dim_query=128, dim_latent=F13["dim_latent"], entity_embedding=emb,
                                          dim_head_cross=16, dim_head_latent=16, num_head_cross=2, num_head_latent=2, num_block_cross=0,
                                          num_block_attn=1, dropout_query=0.1, num_split=8, qk_norm=True, act=act)
    with torch.no_grad():
        emb.embedding.weight.mul_(torch.linspace(0.5, 1.8, F13["n_entities"])[:, None])

    class Backbone(lb.BackboneBase):
        def prepare_inputs(self, batch):
            return batch["atom14_pos"].flatten(-2) @ lift

        def encode(self, batch):  # (first_stage/peptide.py:77-80: no entity mask)
            return self.quant(self.encoder(x=self.prepare_inputs(batch), entities=batch["entities"], mask=None))

    class FirstStage(lb.FirstStageLightningBase):
        def __init__(self, backbone):
            super().__init__()
            self.hparams.update(shift=0.0, scale=1.0, ema=None)
            self.backbone = backbone

        @classmethod
        def load_from_checkpoint(cls, path, map_location=None):
            return _STAGE1[path]  # (ema stays None: second_stage/peptide.py:53-54 skips load_ema_weights then)

    return FirstStage(Backbone(dim_latent=F13["dim_latent"], encoder=enc, decoder=dec)), FirstStage


def build_peptide_wrapper(ns, first_stage, first_stage_cls, seed=52):
    """second_stage/peptide.py Wrapper, constructed by ITS OWN __init__ from the reference's own YAML block
    (configs/model/peptide/second-stage.yaml: cond_idx [0, 1], mask_cond_mean, hidden 384, 16 heads of 24, mlp 4) with the true T = 1000;
    depth 2 instead of 7."""
    import yaml
    pep = importlib.import_module("src.models.composites.second_stage.peptide")
    cfg = yaml.safe_load(open(os.path.join(REF, "configs/model/peptide/second-stage.yaml")))
    for k in ("_target_", "_recursive_", "defaults"):
        cfg.pop(k, None)
    assert cfg["cond_idx"] == F13["cond_idx"] and cfg["mask_cond_mean"] is True and cfg["backbone"]["hidden_size"] == 384
    cfg.update(n_timesteps=F13["T"], ema=None, scheduler=None, sampling_method="ODE",
               sampling_kwargs={"sampling_method": "euler", "num_steps": F13["num_steps"]})
    cfg["backbone"] = dict(cfg["backbone"], **F13["backbone"])
    key = f"f13-stage1-{id(first_stage)}"
    _STAGE1[key] = first_stage
    mod_name = "_lsl_f13_first_stage"
    sys.modules.setdefault(mod_name, types.ModuleType(mod_name)).FirstStage = first_stage_cls
    cfg["first_stage_model"] = {"class_name": f"{mod_name}.FirstStage", "path": key}
    torch.manual_seed(seed)
    return pep.Wrapper(**{k: (AttrDict(v) if isinstance(v, dict) else v) for k, v in cfg.items()})


def f13_batch(seed=53):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from golden_inputs import peptide_frames  # (shared with the tests: the fixture stores the seed, not the 1.3 MB of coordinates)
    return peptide_frames(seed, F13["B"], F13["T"], F13["R"])


class randn_like_sequence:
    """Context manager: the i-th torch.randn_like call returns noises[i] (one initial state per sample() call of the K-loop)."""

    def __init__(self, noises):
        self.noises, self.i = noises, 0

    def __enter__(self):
        self._orig = torch.randn_like

        def fake(x, **kw):
            out = self.noises[self.i].to(x.dtype).clone()
            self.i += 1
            return out

        torch.randn_like = fake
        return self

    def __exit__(self, *exc):
        torch.randn_like = self._orig
