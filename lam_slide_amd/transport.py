"""``CreateTransport`` / ``Transport`` / ``Sampler``: drop-in for ``src.modules.transport`` on the sampling
side (``configs/model/*/second-stage.yaml`` ``transport:`` blocks, lightning_base.py:219-234).

``Sampler(transport).get_sample_fn(method, kwargs)`` returns ``fn(init, model, **model_kwargs)`` exactly like
the reference (transport.py:475-503).  When ``model`` resolves to a :class:`lam_slide_amd.LatentSIV3`
(the module itself, its bound ``forward``, or a bound method of an object whose ``.backbone`` is one, which is
what ``SecondStageCondLightningBase.forward`` is) and the solver is fixed-grid Euler / Euler-Maruyama, the
whole loop runs inside liblamslide_hip.so (``lsl_sample``): every step is the affine update
``x <- ax x + am net(x,t) + aw w`` with (ax, am, aw) derived here in float64 from the reference's formulas.
Any other callable or solver goes through the generic per-step loop below, which mirrors
integrators.py:7-120 and calls ``model`` once per drift evaluation.

Preserved semantics: ``num_steps=N`` means N-1 Euler updates on ``linspace(t0, t1, N)``; the SDE result has
``len == num_steps``; (t0, t1) come from ``check_interval`` (transport.py:69-101); errors are the reference's
(KeyError unknown path, NotImplementedError unknown sampler / diffusion form / last step, AssertionError t0<t1).
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from collections.abc import Sequence
from typing import Any, Callable, Dict, List, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from .latent_si import LatentSIV3


class ModelType(enum.Enum):
    NOISE = enum.auto()
    SCORE = enum.auto()
    VELOCITY = enum.auto()
    DATA = enum.auto()


class PathType(enum.Enum):
    LINEAR = enum.auto()
    GVP = enum.auto()
    VP = enum.auto()


class WeightType(enum.Enum):
    NONE = enum.auto()
    VELOCITY = enum.auto()
    LIKELIHOOD = enum.auto()


_SMIN, _SMAX = 0.1, 20.0


class _Schedule:
    """Scalar (float64) alpha/sigma schedule of one path type: path.py:21-206 evaluated at one t."""

    def __init__(self, path_type: PathType):
        self.kind = path_type

    def alpha(self, t: float) -> Tuple[float, float]:
        if self.kind is PathType.LINEAR:
            return t, 1.0
        if self.kind is PathType.GVP:
            return math.sin(t * math.pi / 2), math.pi / 2 * math.cos(t * math.pi / 2)
        a = math.exp(self._lm(t))
        return a, a * self._dlm(t)

    def sigma(self, t: float) -> Tuple[float, float]:
        if self.kind is PathType.LINEAR:
            return 1 - t, -1.0
        if self.kind is PathType.GVP:
            return math.cos(t * math.pi / 2), -math.pi / 2 * math.sin(t * math.pi / 2)
        p = 2 * self._lm(t)
        s = math.sqrt(1 - math.exp(p))
        return s, math.exp(p) * (2 * self._dlm(t)) / (-2 * s)

    def _lm(self, t):
        return -0.25 * ((1 - t) ** 2) * (_SMAX - _SMIN) - 0.5 * (1 - t) * _SMIN

    def _dlm(self, t):
        return 0.5 * (1 - t) * (_SMAX - _SMIN) + 0.5 * _SMIN

    def drift_terms(self, t: float) -> Tuple[float, float]:
        """(k, var): the reference's compute_drift returns (-k x ... ) i.e. drift_mean = k * x, drift_var = var."""
        if self.kind is PathType.VP:
            beta = _SMIN + (1 - t) * (_SMAX - _SMIN)
            return -0.5 * beta, beta / 2
        if self.kind is PathType.LINEAR:
            ratio = 1 / t if t != 0 else math.inf  # torch gives inf here too (Linear path evaluated at t0 = 0)
        else:
            ratio = math.pi / (2 * math.tan(t * math.pi / 2))
        s, ds = self.sigma(t)
        return -ratio, ratio * s * s - s * ds

    def diffusion(self, t: float, form: str, norm: float) -> float:
        if form == "constant":
            return norm
        if form == "SBDM":
            return norm * self.drift_terms(t)[1]
        if form == "sigma":
            return norm * self.sigma(t)[0]
        if form == "linear":
            return norm * (1 - t)
        if form == "decreasing":
            return 0.25 * (norm * math.cos(math.pi * t) + 1) ** 2
        if form == "inccreasing-decreasing":
            return norm * math.sin(math.pi * t) ** 2
        raise NotImplementedError(f"Diffusion form {form} not implemented")


class Transport:
    def __init__(self, *, model_type, path_type, loss_type, train_eps, sample_eps):
        self.loss_type = loss_type
        self.model_type = model_type
        self.path_type = path_type
        self.schedule = _Schedule(path_type)
        self.train_eps = train_eps
        self.sample_eps = sample_eps

    def check_interval(self, train_eps, sample_eps, *, diffusion_form="SBDM", sde=False, reverse=False, eval=False,
                       last_step_size=0.0):
        t0, t1 = 0, 1
        eps = train_eps if not eval else sample_eps
        if self.path_type is PathType.VP:
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        elif self.model_type != ModelType.VELOCITY or sde:
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.model_type != ModelType.VELOCITY else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1

    # velocity(x, m) = vx * x + vm * m and score(x, m) = sx * x + sm * m for network output m (transport.py:158-226)
    def velocity_coeffs(self, t: float) -> Tuple[float, float]:
        if self.model_type is ModelType.VELOCITY:
            return 0.0, 1.0
        k, var = self.schedule.drift_terms(t)  # v = -drift_mean + var * score, drift_mean = k * x
        sx, sm = self.score_coeffs(t)
        return -k + var * sx, var * sm

    def score_coeffs(self, t: float) -> Tuple[float, float]:
        sch = self.schedule
        if self.model_type is ModelType.SCORE:
            return 0.0, 1.0
        if self.model_type is ModelType.NOISE:
            return 0.0, -1.0 / sch.sigma(t)[0]
        if self.model_type is ModelType.DATA:
            s, _ = sch.sigma(t)
            a, _ = sch.alpha(t)
            return -1.0 / (s * s), a / (s * s)
        a, da = sch.alpha(t)
        s, ds = sch.sigma(t)
        rev = a / da
        var = s * s - rev * ds * s
        return -1.0 / var, rev / var


class CreateTransport:
    """Same keywords as the reference factory (transport/__init__.py:7-77); call it to get the Transport."""

    def __init__(self, path_type="Linear", prediction="velocity", loss_weight=None, train_eps=None, sample_eps=None):
        from . import dropin
        dropin.install()  # no-op unless a reference checkout's sampler modules are already imported (dropin.py)
        self.path_type = path_type
        self.prediction = prediction
        self.loss_weight = loss_weight
        self.train_eps = train_eps
        self.sample_eps = sample_eps

    def __call__(self) -> Transport:
        model_type = {"noise": ModelType.NOISE, "score": ModelType.SCORE, "data": ModelType.DATA}.get(self.prediction, ModelType.VELOCITY)
        loss_type = {"velocity": WeightType.VELOCITY, "likelihood": WeightType.LIKELIHOOD}.get(self.loss_weight, WeightType.NONE)
        path_type = {"Linear": PathType.LINEAR, "GVP": PathType.GVP, "VP": PathType.VP}[self.path_type]  # KeyError as the reference
        if path_type is PathType.VP:
            train_eps = 1e-5 if self.train_eps is None else self.train_eps
            sample_eps = 1e-3 if self.sample_eps is None else self.sample_eps
        elif model_type != ModelType.VELOCITY:
            train_eps = 1e-3 if self.train_eps is None else self.train_eps
            sample_eps = 1e-3 if self.sample_eps is None else self.sample_eps
        else:
            train_eps = 0
            sample_eps = 0
        return Transport(model_type=model_type, path_type=path_type, loss_type=loss_type, train_eps=train_eps, sample_eps=sample_eps)


def as_transport(obj) -> "Transport":
    """Accept the reference's own ``Transport`` object (src/modules/transport/transport.py:40-58) wherever this module wants one: a
    LightningModule built from the ORIGINAL ``transport._target_: src.modules.transport.CreateTransport`` hands ``Sampler`` such an
    object.  It is read by duck typing - enum member NAMES of ``model_type`` / ``loss_type``, the class name of ``path_sampler``
    (ICPlan / GVPCPlan / VPCPlan, path.py:21-206) - and restated as a :class:`Transport` with the same eps values."""
    if isinstance(obj, Transport):
        return obj
    try:
        model_type = ModelType[obj.model_type.name]
        loss_type = WeightType[getattr(getattr(obj, "loss_type", None), "name", "NONE")]
        if hasattr(obj, "path_type"):
            path_type = PathType[obj.path_type.name]
        else:
            path_type = {"ICPlan": PathType.LINEAR, "GVPCPlan": PathType.GVP, "VPCPlan": PathType.VP}[type(obj.path_sampler).__name__]
        return Transport(model_type=model_type, path_type=path_type, loss_type=loss_type, train_eps=obj.train_eps, sample_eps=obj.sample_eps)
    except (AttributeError, KeyError) as e:
        raise TypeError(f"cannot interpret {type(obj).__name__} as a stochastic-interpolant Transport: {e}") from None


_MASK64 = (1 << 64) - 1


def mix_seed(seed: int, index: int) -> int:
    """splitmix64 of (seed, index): the seed of call number ``index`` of an object seeded with ``seed``."""
    z = (int(seed) + 0x9E3779B97F4A7C15 * (int(index) + 1)) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


def device_randn(shape, device, seed: int, elem_offset: int = 0) -> Tensor:
    """Standard-normal tensor from the library's counter stream (``lsl_randn``): element e of the result is global element
    ``elem_offset + e`` of stream ``seed``, so a rank that holds rows [lo, hi) of a batch draws exactly rows [lo, hi) of the
    unsharded draw.  Counterpart of ``torch.randn_like(x_cond)`` in lightning_base.py:231."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("device_randn draws on the GPU (HIP kernel); there is no CPU fallback")
    with torch.cuda.device(device):
        out = torch.empty(tuple(shape), dtype=torch.float32, device=device)
        _lib.check(_lib.load().lsl_randn(out.data_ptr(), out.numel(), int(seed) & _MASK64, int(elem_offset),
                                         torch.cuda.current_stream(device).cuda_stream))
    return out


class SampleResult(Sequence):
    """What the fused samplers return: indexable like the reference's stacked tensor / list of states.
    ``[-1]`` (the only element the reference's callers read, lightning_base.py:230-234) is always there;
    earlier states only when the sampler was built with ``keep_trajectory=True``."""

    def __init__(self, final: Tensor, n: int, trace: Optional[Tensor] = None, first: Optional[Tensor] = None, offset: int = 0):
        self.final, self.n, self.trace, self.first, self.offset = final, n, trace, first, offset

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if i == self.n - 1:
            return self.final
        if not 0 <= i < self.n:
            raise IndexError(i)
        if self.trace is None:
            raise IndexError("intermediate states were not kept: build the sampler with keep_trajectory=True")
        if self.offset and i == 0:
            return self.first
        return self.trace[i - self.offset]


def _unwrap(mod):
    """``torch.compile`` wraps a module in an ``OptimizedModule`` that keeps the original as ``_orig_mod`` (the reference compiles its
    backbone when ``compile: True``, second_stage/md17.py:53-55).  The HIP path has nothing for a tracing compiler to do; the fused
    loop looks through the wrapper instead of silently losing the native path."""
    seen = 0
    while mod is not None and not isinstance(mod, LatentSIV3) and hasattr(mod, "_orig_mod") and seen < 4:
        mod = mod._orig_mod
        seen += 1
    return mod


def resolve_backbone(model: Callable) -> Optional[LatentSIV3]:
    model = _unwrap(model)
    if isinstance(model, LatentSIV3):
        return model
    tagged = _unwrap(getattr(model, "lsl_backbone", None))
    if isinstance(tagged, LatentSIV3):
        return tagged
    owner = _unwrap(getattr(model, "__self__", None))
    if isinstance(owner, LatentSIV3) and getattr(model, "__name__", "") in ("forward", "__call__"):
        return owner
    if owner is not None and getattr(model, "__name__", "") == "forward":
        inner = _unwrap(getattr(owner, "backbone", None))
        if isinstance(inner, LatentSIV3):
            return inner  # LightningModule.forward == backbone(x=xt, t=t, **kw) (lightning_base.py:173-174)
    return None


def _f32(v: float) -> float:
    return float(torch.tensor(v, dtype=torch.float32))


# ---- adaptive Dormand-Prince 5(4), the reference's default ODE method (transport.py:486-494: torchdiffeq.odeint(method="dopri5")) ----------
# torchdiffeq is a third-party dependency that is neither vendored in the reference nor installed here (parity unpinned); this
# restates its published algorithm (Dormand-Prince / Shampine tableau, Hairer's initial step, RMS error norm over the whole state, step
# factor clamp(0.9 err^-1/5, 0.2 .. 10), first-same-as-last, 4th-order dense output through the midpoint weights) so that the default
# configuration of the reference runs on the HIP network without it.  The network evaluations - all the cost - go through `f`.
_DP_ALPHA = (1 / 5, 3 / 10, 4 / 5, 8 / 9, 1.0, 1.0)
_DP_BETA = ((1 / 5,), (3 / 40, 9 / 40), (44 / 45, -56 / 15, 32 / 9), (19372 / 6561, -25360 / 2187, 64448 / 6561, -212 / 729),
            (9017 / 3168, -355 / 33, 46732 / 5247, 49 / 176, -5103 / 18656), (35 / 384, 0.0, 500 / 1113, 125 / 192, -2187 / 6784, 11 / 84))
_DP_C_ERR = (35 / 384 - 1951 / 21600, 0.0, 500 / 1113 - 22642 / 50085, 125 / 192 - 451 / 720, -2187 / 6784 + 12231 / 42400,
             11 / 84 - 649 / 6300, -1 / 60)
_DP_C_MID = (6025192743 / 30085553152 / 2, 0.0, 51252292925 / 65400821598 / 2, -2691868925 / 45128329728 / 2,
             187940372067 / 1594534317056 / 2, -1776094331 / 19743644256 / 2, 11237099 / 235043384 / 2)


def _rms(v: Tensor) -> float:
    return float(v.double().pow(2).mean().sqrt())


class _RkOps:
    """The arithmetic of a Runge-Kutta step on whole states: linear combinations, the controller's error ratio, the quartic dense output.
    fp32 CUDA tensors go through the library (``lsl_rk_lincomb`` / ``lsl_rk_error_ratio`` / ``lsl_rk_dense``: one launch per combination, terms
    added in list order, deterministic reduction); anything else (CPU tensors of the host-logic tests, other dtypes) through torch operations
    in the same order.  ``terms`` = [(coefficient, tensor), ...] with at most 8 entries."""

    def __init__(self, like: Tensor):
        self.hip = like.is_cuda and like.dtype == torch.float32
        if self.hip:
            self.lib = _lib.load()
            self.dev = like.device
            self.scratch = torch.empty(_lib.RK_SCRATCH_BYTES // 4, dtype=torch.float32, device=like.device)
            self.ratio = torch.empty(1, dtype=torch.float32, device=like.device)

    def _pack(self, terms):
        xs = [x.contiguous() for _, x in terms]
        ptrs = (C.c_void_p * len(xs))(*[x.data_ptr() for x in xs])
        cs = (C.c_float * len(xs))(*[float(c) for c, _ in terms])
        return xs, ptrs, cs

    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def lincomb(self, terms) -> Tensor:
        if not self.hip:
            rnd = _f32 if terms[0][1].dtype == torch.float32 else float  # (the kernels take fp32 coefficients)
            acc = terms[0][1] * rnd(terms[0][0])
            for c, x in terms[1:]:
                acc = acc + x * rnd(c)
            return acc
        xs, ptrs, cs = self._pack(terms)
        out = torch.empty_like(xs[0])
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.lsl_rk_lincomb(out.data_ptr(), ptrs, cs, len(xs), out.numel(), self._stream()))
        return out

    def error_ratio(self, y0: Tensor, y1: Tensor, terms, atol: float, rtol: float) -> float:
        """sqrt(mean((sum_j c_j k_j / (atol + rtol max(|y0|, |y1|)))^2)) as a host scalar (the one device sync of a step)."""
        if not self.hip:
            return _rms(self.lincomb(terms) / (atol + rtol * torch.maximum(y0.abs(), y1.abs())))
        xs, ptrs, cs = self._pack(terms)
        y0, y1 = y0.contiguous(), y1.contiguous()
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.lsl_rk_error_ratio(self.ratio.data_ptr(), y0.data_ptr(), y1.data_ptr(), ptrs, cs, len(xs), float(atol), float(rtol),
                                                   y0.numel(), self.scratch.data_ptr(), self._stream()))
        return float(self.ratio.item())

    def poly4(self, a: Tensor, b: Tensor, c: Tensor, d: Tensor, e: Tensor, x: float) -> Tensor:
        if not self.hip:
            xf = _f32(x) if e.dtype == torch.float32 else float(x)
            return e + xf * (d + xf * (c + xf * (b + xf * a)))
        out = torch.empty_like(e)
        with torch.cuda.device(self.dev):
            _lib.check(self.lib.lsl_rk_dense(out.data_ptr(), a.data_ptr(), b.data_ptr(), c.data_ptr(), d.data_ptr(), e.data_ptr(), float(x), out.numel(),
                                             self._stream()))
        return out


def dopri5_solve(f: Callable[[float, Tensor], Tensor], y0: Tensor, grid: Sequence[float], rtol: float, atol: float,
                 max_steps: int = 100000) -> Tuple[List[Tensor], Dict[str, int]]:
    """Solution of y' = f(t, y) at the (increasing) times `grid`, grid[0] being the initial time: [y(grid[0]), ..., y(grid[-1])] and
    counters.  Steps are chosen by the error controller alone and run past the output times; outputs are interpolated.  An output time
    that does not exceed the current time (repeated grid points, a single-point grid) returns the current state.  Time arithmetic runs in
    Python float64 (torchdiffeq keeps t in the dtype of the state's time tensor: accepted-step sequences agree in accuracy class, not in
    the last bit).  The state arithmetic of a step is ten launches of the library's Runge-Kutta kernels (`_RkOps`: six stage states, the
    error ratio, and for an accepted step the mid-point and three dense-output coefficients); the error ratio is the one host scalar, and the
    one device sync, of a step."""
    y0 = y0.contiguous()
    ops = _RkOps(y0)
    t0 = float(grid[0])
    f0 = f(t0, y0)
    nfe = 1
    # initial step (Hairer, Norsett, Wanner I, II.4), order 4 estimate; scale = atol + |y0| rtol
    d0, d1 = ops.error_ratio(y0, y0, [(1.0, y0)], atol, rtol), ops.error_ratio(y0, y0, [(1.0, f0)], atol, rtol)
    h0 = 1e-6 if d0 < 1e-5 or d1 < 1e-5 else 0.01 * d0 / d1
    f1 = f(t0 + h0, ops.lincomb([(1.0, y0), (h0, f0)]))
    nfe += 1
    d2 = ops.error_ratio(y0, y0, [(1.0, f1), (-1.0, f0)], atol, rtol) / h0
    h1 = max(1e-6, h0 * 1e-3) if d1 <= 1e-15 and d2 <= 1e-15 else (0.01 / max(d1, d2)) ** (1.0 / 5.0)
    dt = min(100 * h0, h1)

    out = [y0]
    t, y, fy = t0, y0, f0
    t_lo, coeff = t0, None  # dense output of the last accepted step [t_lo, t]
    accepted = rejected = 0
    for tn in [float(g) for g in grid[1:]]:
        while tn > t:
            if accepted + rejected >= max_steps:
                raise RuntimeError("dopri5: max_steps reached")
            k = [fy]
            yi = y
            for a, beta in zip(_DP_ALPHA, _DP_BETA):
                yi = ops.lincomb([(1.0, y)] + [(dt * b, kj) for b, kj in zip(beta, k) if b != 0.0])
                k.append(f(t + a * dt, yi))
            nfe += 6
            y1 = yi  # the last stage IS the 5th-order solution (beta[-1] = c_sol): first same as last
            ratio = ops.error_ratio(y, y1, [(dt * c, kj) for c, kj in zip(_DP_C_ERR, k) if c != 0.0], atol, rtol)
            if ratio <= 1.0:
                y_mid = ops.lincomb([(1.0, y)] + [(dt * c, kj) for c, kj in zip(_DP_C_MID, k) if c != 0.0])
                fa, fb = k[0], k[-1]
                coeff = (y, ops.lincomb([(dt, fa)]),
                         ops.lincomb([(dt, fb), (-4 * dt, fa), (-11.0, y), (-5.0, y1), (16.0, y_mid)]),
                         ops.lincomb([(5 * dt, fa), (-3 * dt, fb), (18.0, y), (14.0, y1), (-32.0, y_mid)]),
                         ops.lincomb([(2 * dt, fb), (-2 * dt, fa), (-8.0, y1), (-8.0, y), (16.0, y_mid)]))
                t_lo, t, y, fy = t, t + dt, y1, k[-1]
                accepted += 1
            else:
                rejected += 1
            if ratio == 0.0:
                dt = dt * 10.0
            else:
                dt = dt * min(10.0, max(0.9 / ratio ** 0.2, 1.0 if ratio < 1.0 else 0.2))
        if coeff is None or t <= t_lo:  # an output time at (or before) the current time with no accepted step to interpolate: the state itself
            out.append(y)
            continue
        e, d, c, b, a = coeff
        out.append(ops.poly4(a, b, c, d, e, (tn - t_lo) / (t - t_lo)))
    return out, {"nfe": nfe, "accepted": accepted, "rejected": rejected}



# ---- torchdiffeq's other fixed-grid explicit methods (integrators.py:119 passes any `method` through) ------------------------------------
# Restated from the published schemes, as torchdiffeq's fixed_grid.py / rk_common.py define them (parity unpinned like dopri5: the package is
# absent): the grid is the output grid itself, one step per interval.  "rk4" is torchdiffeq's default fourth-order step, the 3/8 rule.
FIXED_GRID_RK_METHODS = ("midpoint", "heun3", "rk4")


def fixed_grid_rk_solve(f: Callable[[float, Tensor], Tensor], y0: Tensor, grid: Sequence[float], method: str) -> List[Tensor]:
    """[y(grid[0]), ..., y(grid[-1])] of y' = f(t, y) with one explicit Runge-Kutta step of `method` per grid interval; stage states and
    the update through `_RkOps` (library kernels for fp32 CUDA states)."""
    if method not in FIXED_GRID_RK_METHODS:
        raise NotImplementedError(f"fixed-grid method {method!r}")
    y = y0.contiguous()
    ops = _RkOps(y)
    out = [y]
    ts = [float(g) for g in grid]
    for t0, t1 in zip(ts[:-1], ts[1:]):
        dt = t1 - t0
        k1 = f(t0, y)
        if method == "midpoint":
            k2 = f(t0 + dt / 2, ops.lincomb([(1.0, y), (dt / 2, k1)]))
            y = ops.lincomb([(1.0, y), (dt, k2)])
        elif method == "heun3":
            k2 = f(t0 + dt / 3, ops.lincomb([(1.0, y), (dt / 3, k1)]))
            k3 = f(t0 + 2 * dt / 3, ops.lincomb([(1.0, y), (2 * dt / 3, k2)]))
            y = ops.lincomb([(1.0, y), (dt / 4, k1), (3 * dt / 4, k3)])
        else:  # rk4: 3/8 rule
            k2 = f(t0 + dt / 3, ops.lincomb([(1.0, y), (dt / 3, k1)]))
            k3 = f(t0 + 2 * dt / 3, ops.lincomb([(1.0, y), (dt, k2), (-dt / 3, k1)]))
            k4 = f(t1, ops.lincomb([(1.0, y), (dt, k1), (-dt, k2), (dt, k3)]))
            y = ops.lincomb([(1.0, y), (dt / 8, k1), (3 * dt / 8, k2), (3 * dt / 8, k3), (dt / 8, k4)])
        out.append(y)
    return out


class Sampler:
    """``Sampler(transport)`` as in the reference (transport.py:229-244; rebuilt on every ``sample()`` call, lightning_base.py:219).

    Device noise (Euler-Maruyama steps without an explicit ``noise=`` tensor): every sampling CALL draws from a stream of its own, as
    the reference draws fresh ``randn`` per step and per call (integrators.py:30).  ``seed=None`` (default): the call's stream seed
    comes from torch's global generator, so ``torch.manual_seed`` / ``seed_everything`` make a run reproducible and consecutive calls
    differ.  ``seed=int``: call number k of this object uses ``mix_seed(seed, k)`` - reproducible by rebuilding the Sampler with the
    same seed."""

    def __init__(self, transport: Transport, fused: Optional[bool] = None, keep_trajectory: bool = False, seed: Optional[int] = None):
        self._given_transport = transport  # (as passed: the reference's Transport object when installed into a reference checkout)
        self.transport = as_transport(transport)
        self.fused = fused
        self.keep_trajectory = keep_trajectory
        self.seed = seed
        self.calls = 0
        self.last_seed: Optional[int] = None  # stream seed of the most recent fused call
        self.last_path: Optional[str] = None
        self.last_kernels: Optional[str] = None  # "general" | "resident" (lsl_sampler_path) after a fused call
        self.elem_offset = 0  # global element index of this rank's first state element (device noise stream)
        self._rk_ops: Dict[Any, "_RkOps"] = {}  # per device: the library's Runge-Kutta arithmetic (scratch buffers allocated once)

    def next_call_seed(self, draw: bool = True) -> int:
        """Seed of this call's device-noise stream.  The call counter advances on every call (ranks of a sharded run stay in step even
        when one of them has an empty shard); the global generator is consumed only when ``draw`` (the call really uses device noise)."""
        k = self.calls
        self.calls += 1
        if self.seed is None:
            return int(torch.randint(0, 1 << 62, (1,)).item()) if draw else 0
        return mix_seed(self.seed, k)

    # ---- step tables ------------------------------------------------------------------------------------
    def ode_steps(self, num_steps: int, reverse: bool = False) -> Tuple[List[Tuple[float, float, float, float]], Tensor]:
        tr = self.transport
        t0, t1 = tr.check_interval(tr.train_eps, tr.sample_eps, sde=False, eval=True, reverse=reverse, last_step_size=0.0)
        assert t0 < t1, "ODE sampler has to be in forward time"
        grid = torch.linspace(t0, t1, num_steps)
        steps = []
        for i in range(num_steps - 1):
            ti = float(grid[i])
            dt = float(grid[i + 1] - grid[i])  # fp32 difference, as torchdiffeq's fixed grid forms it
            te = _f32(1 - ti) if reverse else ti
            vx, vm = tr.velocity_coeffs(te)
            steps.append((te, 1.0 + dt * vx, dt * vm, 0.0))
        return steps, grid

    def sde_steps(self, *, diffusion_form, diffusion_norm, last_step, last_step_size, num_steps):
        tr = self.transport
        if last_step is None:
            last_step_size = 0.0
        t0, t1 = tr.check_interval(tr.train_eps, tr.sample_eps, diffusion_form=diffusion_form, sde=True, eval=True, reverse=False,
                                   last_step_size=last_step_size)
        assert t0 < t1, "SDE sampler has to be in forward time"
        grid = torch.linspace(t0, t1, num_steps)
        dt = float(grid[1] - grid[0])
        sch = tr.schedule

        def drift(t):
            vx, vm = tr.velocity_coeffs(t)
            sx, sm = tr.score_coeffs(t)
            g = sch.diffusion(t, diffusion_form, diffusion_norm)
            return vx + g * sx, vm + g * sm, g

        steps = []
        for i in range(num_steps - 1):
            ti = float(grid[i])
            dx, dm, g = drift(ti)
            steps.append((ti, 1.0 + dt * dx, dt * dm, math.sqrt(2 * g) * math.sqrt(dt)))
        t1f = _f32(t1)
        if last_step is None:
            pass
        elif last_step == "Mean":
            dx, dm, _ = drift(t1f)
            steps.append((t1f, 1.0 + last_step_size * dx, last_step_size * dm, 0.0))
        elif last_step == "Euler":
            vx, vm = tr.velocity_coeffs(t1f)
            steps.append((t1f, 1.0 + last_step_size * vx, last_step_size * vm, 0.0))
        elif last_step == "Tweedie":
            a = sch.alpha(t1f)[0]
            s = sch.sigma(t1f)[0]
            sx, sm = tr.score_coeffs(t1f)
            steps.append((t1f, 1.0 / a + s * s / a * sx, s * s / a * sm, 0.0))
        else:
            raise NotImplementedError()
        return steps, grid

    def heun_records(self, *, diffusion_form, diffusion_norm, last_step, last_step_size, num_steps):
        """Extended step records (``lsl_step_ex``: t, ax, am, aw, as, flags, noise slice, trace slice) of the stochastic Heun sampler
        (integrators.py:39-51) followed by the reference's last step.  drift(x, t) = a(t) x + b(t) net(x, t), so per step
            x_hat = x + sqrt(2 g(t) dt) w                       no network, state kept
            x_p   = (1 + dt a(t)) x_hat + dt b(t) net(x_hat, t)
            x'    = x_hat + dt/2 (K1 + K2) = x_hat / 2 + (1/2 + dt a(t')/2) x_p + dt b(t')/2 net(x_p, t'),   t' = t + dt
        (dt K1 = x_p - x_hat).  Returned with the EM table's last step appended (same last-step rule as Euler-Maruyama)."""
        em, grid = self.sde_steps(diffusion_form=diffusion_form, diffusion_norm=diffusion_norm, last_step=last_step,
                                  last_step_size=last_step_size, num_steps=num_steps)
        tr, sch = self.transport, self.transport.schedule
        dt = float(grid[1] - grid[0])

        def drift(t):
            vx, vm = tr.velocity_coeffs(t)
            sx, sm = tr.score_coeffs(t)
            g = sch.diffusion(t, diffusion_form, diffusion_norm)
            return vx + g * sx, vm + g * sm, g

        rec = []
        for i in range(num_steps - 1):
            ti = float(grid[i])
            tn = _f32(_f32(ti) + _f32(dt))  # t_cur + dt in fp32, as the reference forms it
            dx, dm, g = drift(ti)
            dx2, dm2, _ = drift(tn)
            rec.append((ti, 1.0, 0.0, math.sqrt(2 * g) * math.sqrt(dt), 0.0, _lib.STEP_NO_NETWORK | _lib.STEP_SAVE, i, -1))
            rec.append((ti, 1.0 + dt * dx, dt * dm, 0.0, 0.0, 0, 0, -1))
            rec.append((tn, 0.5 + 0.5 * dt * dx2, 0.5 * dt * dm2, 0.0, 0.5, 0, 0, i))
        if len(em) == num_steps:  # the last step (Mean / Euler / Tweedie)
            te, ax, am, _ = em[-1]
            rec.append((te, ax, am, 0.0, 0.0, 0, 0, num_steps - 1))
        return rec, grid

    # ---- fused execution ----------------------------------------------------------------------------------
    def run_fused(self, net: LatentSIV3, init: Tensor, steps, n_result: int, model_kwargs: Dict[str, Any],
                  noise: Optional[Tensor] = None, duplicate_last: bool = False, records=None) -> SampleResult:
        """``steps``: plain (t, ax, am, aw) records -> lsl_sample; ``records``: extended ones (heun_records) -> lsl_sample_ex, with
        ``steps`` then only giving the number of kept states."""
        extra = set(model_kwargs) - {"x_cond", "x_cond_mask", "y"}
        if extra:
            raise TypeError(f"unexpected model kwargs {sorted(extra)}")
        net._require_gpu(init)
        lib = _lib.load()
        dev = init.device
        for name in ("x_cond", "x_cond_mask"):
            if name not in model_kwargs:
                raise TypeError(f"missing model kwarg {name!r}")
        for name, ten in model_kwargs.items():
            if ten is not None and ten.device != dev:
                raise RuntimeError(f"Expected all tensors to be on the same device, but {name} is on {ten.device} and the state is on {dev}")
        # The call's noise-stream seed is drawn only when the call uses device noise (an Euler-Maruyama step without an explicit noise
        # tensor).  ODE calls and SDE calls with stored noise leave torch's global generator untouched, like the reference.
        table = records if records is not None else steps
        needs_device_noise = noise is None and any(s[3] != 0.0 for s in table)
        call_seed = self.next_call_seed(draw=needs_device_noise)
        self.last_seed = call_seed if needs_device_noise else None
        with torch.cuda.device(dev):
            net.ensure_packed(dev)
            # the state is updated in place by the library: always a private copy (persistent buffers only for calls the library may replay as a hipGraph, see staged())
            replay = net.graph_replay_enabled(tokens=int(init.shape[0]) * int(init.shape[1]) * int(init.shape[2]))
            x = net.staged("state", init, torch.float32, dev, fresh=True, persistent=replay)
            xc = net.staged("x_cond", model_kwargs["x_cond"], torch.float32, dev, persistent=replay)
            xm = net.staged("x_cond_mask", model_kwargs["x_cond_mask"], torch.int64, dev, persistent=replay)
            yv = model_kwargs.get("y")
            if yv is not None:
                yv = net.staged("y", yv, torch.float32, dev, persistent=replay)
            io, keep = net.make_io(x, xc, xm, yv)
            ws = net.workspace(io.B, io.T, io.L, dev)
            if records is None:
                arr = (_lib.Step * len(steps))(*[_lib.Step(*s) for s in steps])
            else:
                arr = (_lib.StepEx * len(records))(*[_lib.StepEx(*r) for r in records])
            trace = None
            if self.keep_trajectory:
                trace = torch.empty((len(steps),) + tuple(x.shape), dtype=torch.float32, device=dev)
            nz = None
            if noise is not None:  # slice s belongs to step s, like the reference's one draw per EM / Heun step
                nz = noise.detach().float().contiguous().to(dev)
                if records is None:
                    need = max([i + 1 for i, s in enumerate(steps) if s[3] != 0.0], default=0)
                else:
                    need = max([r[6] + 1 for r in records if r[3] != 0.0], default=0)
                if nz.shape[0] < need or tuple(nz.shape[1:]) != tuple(x.shape):
                    raise ValueError(f"noise must be [>={need}, {tuple(x.shape)}], got {tuple(nz.shape)}")
            stream = torch.cuda.current_stream(dev).cuda_stream
            head = (net._handle, C.byref(io), arr, len(arr), nz.data_ptr() if nz is not None else None,
                    nz.shape[0] if nz is not None else 0, call_seed, self.elem_offset, trace.data_ptr() if trace is not None else None)
            if records is None:
                _lib.check(lib.lsl_sample(*head, ws.data_ptr(), ws.numel(), stream))
            else:  # (the library checks every record's trace slice against the slices the buffer really has)
                _lib.check(lib.lsl_sample_ex(*head, trace.shape[0] if trace is not None else 0, ws.data_ptr(), ws.numel(), stream))
        net.last_path = "hip"
        self.last_path = "fused"
        # (extended records always run the general kernels: the trajectory-resident kernel implements the plain affine step only)
        self.last_kernels = "resident" if records is None and lib.lsl_sampler_path(net._handle, io.T, io.L) == 1 else "general"
        del keep
        if replay:
            x = x.clone()  # the persistent buffer is overwritten by the next call
        if x.dtype != init.dtype:
            x = x.to(init.dtype)
        if trace is None:
            return SampleResult(x, n_result)
        if duplicate_last:  # last_step=None: the reference appends xs[-1] again (transport.py:353-357)
            trace = torch.cat([trace, trace[-1:]], dim=0)
            return SampleResult(x, n_result, trace, None, 0)
        if n_result == len(steps) + 1:  # ODE: state 0 is the initial noise
            return SampleResult(x, n_result, trace, init, 1)
        return SampleResult(x, n_result, trace, None, 0)

    # ---- generic execution (any callable; mirrors integrators.py) -------------------------------------------
    def _vel(self, x, t, model, **kw):
        tr = self.transport
        out = model(x, t, **kw)
        tf = float(t.flatten()[0])
        vx, vm = tr.velocity_coeffs(tf)
        v = None
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or out.requires_grad)  # (the library result carries no grad_fn)
        if x.is_cuda and x.dtype == torch.float32 and out.dtype == torch.float32 and out.shape == x.shape and not needs_grad:
            # one library launch instead of three element-wise kernels; the same two rounded products and their rounded sum
            ops = self._rk_ops.get(x.device)
            if ops is None:
                try:
                    ops = _RkOps(x)
                except _lib.LibraryMissing:  # an arbitrary callable on a GPU needs no library: the torch expression below, same values
                    ops = False              # (remembered per device; HIP errors and out-of-memory conditions propagate)
                self._rk_ops[x.device] = ops
            if ops:
                v = ops.lincomb([(vx, x), (vm, out)])
        if v is None:
            v = vx * x + vm * out
        assert v.shape == x.shape, "Output shape from ODE solver must match input shape"
        return v, out

    def _sde_drift(self, x, t, model, form, norm, **kw):
        tr = self.transport
        v, out = self._vel(x, t, model, **kw)
        tf = float(t.flatten()[0])
        sx, sm = tr.score_coeffs(tf)
        g = tr.schedule.diffusion(tf, form, norm)
        return v + g * (sx * x + sm * out), g

    # ---- reference API --------------------------------------------------------------------------------------
    def sample_ode(self, *, sampling_method="dopri5", num_steps=50, atol=1e-6, rtol=1e-3, reverse=False):
        if sampling_method == "dopri5" or sampling_method in FIXED_GRID_RK_METHODS:
            return self._sample_ode_dopri5(num_steps=num_steps, atol=atol, rtol=rtol, reverse=reverse, method=sampling_method)
        if sampling_method != "euler":
            # Other torchdiffeq solvers: not implemented here.  When this class stands in for the reference's Sampler (dropin.install) and
            # was given the reference's own Transport object, hand the call to the class it replaced instead of breaking a flow that worked
            # before the install.
            from . import dropin
            orig = dropin.original_sampler()
            if orig is not None and hasattr(self._given_transport, "get_drift"):
                return orig(self._given_transport).sample_ode(sampling_method=sampling_method, num_steps=num_steps, atol=atol, rtol=rtol,
                                                              reverse=reverse)
            raise NotImplementedError(f"ODE solver {sampling_method!r}: torchdiffeq's fixed-grid 'euler' / 'midpoint' / 'heun3' / 'rk4' and adaptive 'dopri5' are "
                                      "implemented (SURVEY.md 8c)")
        steps, grid = self.ode_steps(num_steps, reverse)

        def _sample(init, model, **model_kwargs):
            net = resolve_backbone(model) if self.fused is not False else None
            if net is not None and init.is_cuda:
                return self.run_fused(net, init, steps, num_steps, model_kwargs)
            if self.fused:
                raise RuntimeError("fused sampling requested but the model is not a lam_slide_amd.LatentSIV3 on a GPU")
            self.last_path = "generic"
            x = init
            xs = [x]
            for (te, _, _, _), i in zip(steps, range(num_steps - 1)):
                tv = torch.ones(x.size(0), device=x.device) * te
                v, _ = self._vel(x, tv, model, **model_kwargs)
                x = x + (grid[i + 1] - grid[i]).to(x.device) * v
                xs.append(x)
            return torch.stack(xs)

        return _sample

    def _sample_ode_dopri5(self, *, num_steps, atol, rtol, reverse, method="dopri5"):
        """The reference's default ODE sampler (transport.py:486-494, integrators.py:67-78 with method "dopri5"): adaptive steps, the solution
        reported at linspace(t0, t1, num_steps).  Every network evaluation runs on the HIP path (LatentSIV3.forward); the stage
        combinations and the error norm are a handful of element-wise device operations per step."""
        tr = self.transport
        t0, t1 = tr.check_interval(tr.train_eps, tr.sample_eps, sde=False, eval=True, reverse=reverse, last_step_size=0.0)
        assert t0 < t1, "ODE sampler has to be in forward time"
        grid = [float(g) for g in torch.linspace(t0, t1, num_steps)]

        def _sample(init, model, **model_kwargs):
            self.last_path = method

            def f(t, x):
                tv = torch.ones(x.size(0), device=x.device) * (_f32(1 - t) if reverse else t)
                return self._vel(x, tv, model, **model_kwargs)[0]

            if method == "dopri5":
                ys, self.last_ode_stats = dopri5_solve(f, init, grid, rtol, atol)
            else:  # (torchdiffeq's fixed-grid midpoint / heun3 / rk4: one step per output interval)
                ys = fixed_grid_rk_solve(f, init, grid, method)
            return torch.stack(ys)

        return _sample

    def sample_sde(self, *, sampling_method="Euler", diffusion_form="SBDM", diffusion_norm=1.0, last_step="Mean",
                   last_step_size=0.04, num_steps=250, noise: Optional[Tensor] = None):
        if sampling_method not in ("Euler", "Heun"):
            raise NotImplementedError("Smapler type not implemented.")
        if last_step not in (None, "Mean", "Tweedie", "Euler"):
            raise NotImplementedError()
        steps, grid = self.sde_steps(diffusion_form=diffusion_form, diffusion_norm=diffusion_norm, last_step=last_step,
                                     last_step_size=last_step_size, num_steps=num_steps)
        dt = grid[1] - grid[0]
        records = None
        if sampling_method == "Heun":
            records, _ = self.heun_records(diffusion_form=diffusion_form, diffusion_norm=diffusion_norm, last_step=last_step,
                                           last_step_size=last_step_size, num_steps=num_steps)

        def _sample(init, model, **model_kwargs):
            net = resolve_backbone(model) if self.fused is not False else None
            if net is not None and init.is_cuda:
                return self.run_fused(net, init, steps, num_steps, model_kwargs, noise=noise, duplicate_last=last_step is None,
                                      records=records)
            if self.fused:
                raise RuntimeError("fused sampling requested but unavailable for this model / solver")
            self.last_path = "generic"
            x = init
            xs = []
            for i in range(num_steps - 1):
                ti = grid[i]
                w = noise[i].to(x) if noise is not None else torch.randn(x.size()).to(x)
                dw = w * torch.sqrt(dt).to(x)
                tv = torch.ones(x.size(0)).to(x) * ti.to(x)
                if sampling_method == "Euler":
                    d, g = self._sde_drift(x, tv, model, diffusion_form, diffusion_norm, **model_kwargs)
                    x = x + d * dt.to(x) + math.sqrt(2 * g) * dw
                else:
                    g = self.transport.schedule.diffusion(float(ti), diffusion_form, diffusion_norm)
                    xhat = x + math.sqrt(2 * g) * dw
                    k1, _ = self._sde_drift(xhat, tv, model, diffusion_form, diffusion_norm, **model_kwargs)
                    xp = xhat + dt.to(x) * k1
                    k2, _ = self._sde_drift(xp, tv + dt.to(x), model, diffusion_form, diffusion_norm, **model_kwargs)
                    x = xhat + 0.5 * dt.to(x) * (k1 + k2)
                xs.append(x)
            if last_step is None:
                xs.append(xs[-1])
            else:
                te, ax, am, _ = steps[-1]
                tv = torch.ones(init.size(0), device=x.device) * te
                out = model(xs[-1], tv, **model_kwargs)
                xs.append(ax * xs[-1] + am * out)
            assert len(xs) == num_steps, "Samples does not match the number of steps"
            return xs

        return _sample

    def get_sample_fn(self, sampling_method: str = "ODE", sampling_kwargs: Dict[str, Any] = {}):
        sde_kwargs = {"sampling_method": "Euler", "diffusion_form": "linear", "diffusion_norm": 1.0, "last_step": "Mean",
                      "last_step_size": 0.04, "num_steps": 250}
        ode_kwargs = {"sampling_method": "dopri5", "num_steps": 50, "atol": 1e-6, "rtol": 1e-3, "reverse": False}
        if sampling_method == "SDE":
            sde_kwargs.update(sampling_kwargs)
            return self.sample_sde(**sde_kwargs)
        if sampling_method == "ODE":
            ode_kwargs.update(sampling_kwargs)
            return self.sample_ode(**ode_kwargs)
        return None
