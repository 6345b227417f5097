"""CPU restatement of the stochastic-interpolant transport and its samplers.

Test infrastructure only (see ``oracle/__init__.py``).

Reference lines restated (all under /root/reference/src/modules/transport/):
  __init__.py:23-77        string options -> model/path type and the eps defaults
  path.py:7-15             broadcast of t over the state
  path.py:21-72            linear plan: alpha, sigma, d_alpha/alpha, drift, diffusion forms
  path.py:90-94            score from a data prediction
  path.py:149-186          VP plan
  path.py:188-206          GVP plan
  transport.py:69-101      integration interval
  transport.py:158-226     drift of the probability-flow ODE and score, per prediction type
  transport.py:246-363     SDE sampler factory: drift + g*score, last step, list of states
  transport.py:365-411     ODE sampler factory
  transport.py:475-503     defaults merged by get_sample_fn
  integrators.py:7-78      Euler-Maruyama / Heun loops (noise drawn on the host)
  integrators.py:81-120    ODE wrapper around torchdiffeq.odeint

Third-party arithmetic: ``torchdiffeq.odeint(..., method="euler")`` is not vendored in the reference
and its version is not pinned anywhere.  Its published fixed-grid algorithm, used with no
``step_size`` option so the solver grid equals the output grid ``t``, is
    y_{i+1} = y_i + (t_{i+1} - t_i) * f(t_i, y_i)
and the states at every t_i are returned stacked.  ``odeint_euler`` below restates exactly that; this
boundary is "parity unpinned" by the reference (it has no test of it).
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Tuple

import torch
from torch import Tensor

PATHS = ("Linear", "GVP", "VP")
PREDICTIONS = ("velocity", "noise", "score", "data")

_VP_SMIN, _VP_SMAX = 0.1, 20.0


def _bt(t: Tensor, x: Tensor) -> Tensor:
    return t.view(t.shape[0], *([1] * (x.dim() - 1)))  # path.py:7-15


class Plan:
    """alpha_t, sigma_t and friends for one path type (path.py:21-206)."""

    def __init__(self, path_type: str):
        if path_type not in PATHS:
            raise KeyError(path_type)  # reference: dict lookup, transport/__init__.py:52-58
        self.kind = path_type

    def alpha(self, t):
        if self.kind == "Linear":
            return t, 1
        if self.kind == "GVP":
            return torch.sin(t * math.pi / 2), math.pi / 2 * torch.cos(t * math.pi / 2)
        a = torch.exp(self._log_mean(t))
        return a, a * self._d_log_mean(t)

    def sigma(self, t):
        if self.kind == "Linear":
            return 1 - t, -1
        if self.kind == "GVP":
            return torch.cos(t * math.pi / 2), -math.pi / 2 * torch.sin(t * math.pi / 2)
        p = 2 * self._log_mean(t)
        s = torch.sqrt(1 - torch.exp(p))
        return s, torch.exp(p) * (2 * self._d_log_mean(t)) / (-2 * s)

    def _log_mean(self, t):
        return -0.25 * ((1 - t) ** 2) * (_VP_SMAX - _VP_SMIN) - 0.5 * (1 - t) * _VP_SMIN

    def _d_log_mean(self, t):
        return 0.5 * (1 - t) * (_VP_SMAX - _VP_SMIN) + 0.5 * _VP_SMIN

    def d_alpha_over_alpha(self, t):
        if self.kind == "Linear":
            return 1 / t
        if self.kind == "GVP":
            return math.pi / (2 * torch.tan(t * math.pi / 2))
        return self._d_log_mean(t)

    def drift_terms(self, x: Tensor, t: Tensor):
        """(drift_mean, drift_var) with the reference's sign convention (path.py:39-47, 181-185)."""
        tb = _bt(t, x)
        if self.kind == "VP":
            beta = _VP_SMIN + (1 - tb) * (_VP_SMAX - _VP_SMIN)
            return -0.5 * beta * x, beta / 2
        r = self.d_alpha_over_alpha(tb)
        s, ds = self.sigma(tb)
        return -(r * x), r * (s ** 2) - s * ds

    def diffusion(self, x: Tensor, t: Tensor, form: str, norm: float):
        tb = _bt(t, x)
        if form == "constant":
            return norm
        if form == "SBDM":
            # path.py:61 passes the already expanded t to compute_drift
            return norm * self.drift_terms(x, tb)[1]
        if form == "sigma":
            return norm * self.sigma(tb)[0]
        if form == "linear":
            return norm * (1 - tb)
        if form == "decreasing":
            return 0.25 * (norm * torch.cos(math.pi * tb) + 1) ** 2
        if form == "inccreasing-decreasing":  # sic, spelled this way in path.py:64
            return norm * torch.sin(math.pi * tb) ** 2
        raise NotImplementedError(f"Diffusion form {form} not implemented")

    def score_from_velocity(self, v, x, t):
        tb = _bt(t, x)
        a, da = self.alpha(tb)
        s, ds = self.sigma(tb)
        rev = a / da
        var = s ** 2 - rev * ds * s
        return (rev * v - x) / var

    def score_from_data(self, d, x, t):
        tb = _bt(t, x)
        s, _ = self.sigma(tb)
        a, _ = self.alpha(tb)
        return -(1 / s ** 2) * (x - a * d)


class Transport:
    """Counterpart of CreateTransport(...)() for sampling (training losses are not on the path)."""

    def __init__(self, path_type="Linear", prediction="velocity", train_eps=None, sample_eps=None):
        self.plan = Plan(path_type)
        self.prediction = prediction if prediction in ("noise", "score", "data") else "velocity"
        if path_type == "VP":
            self.train_eps = 1e-5 if train_eps is None else train_eps
            self.sample_eps = 1e-3 if sample_eps is None else sample_eps
        elif self.prediction != "velocity":
            self.train_eps = 1e-3 if train_eps is None else train_eps
            self.sample_eps = 1e-3 if sample_eps is None else sample_eps
        else:
            self.train_eps = 0
            self.sample_eps = 0

    def interval(self, *, diffusion_form="SBDM", sde=False, reverse=False, evaluate=True, last_step_size=0.0):
        t0, t1 = 0, 1
        eps = self.sample_eps if evaluate else self.train_eps
        if self.plan.kind == "VP":
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        elif self.prediction != "velocity" or sde:
            t0 = eps if (diffusion_form == "SBDM" and sde) or self.prediction != "velocity" else 0
            t1 = 1 - eps if (not sde or last_step_size == 0) else 1 - last_step_size
        if reverse:
            t0, t1 = 1 - t0, 1 - t1
        return t0, t1

    # -- conversions of the network output (transport.py:158-226) --------------------------------
    def velocity(self, x, t, model, **kw):
        out = model(x, t, **kw)
        if self.prediction == "velocity":
            v = out
        else:
            mean, var = self.plan.drift_terms(x, t)
            if self.prediction == "score":
                score = out
            elif self.prediction == "noise":
                score = out / -self.plan.sigma(_bt(t, x))[0]
            else:
                s, _ = self.plan.sigma(_bt(t, x))
                a, _ = self.plan.alpha(_bt(t, x))
                score = -(1 / s ** 2) * (x - a * out)
            v = -mean + var * score
        assert v.shape == x.shape, "Output shape from ODE solver must match input shape"
        return v

    def score(self, x, t, model, **kw):
        out = model(x, t, **kw)
        if self.prediction == "noise":
            return out / -self.plan.sigma(_bt(t, x))[0]
        if self.prediction == "score":
            return out
        if self.prediction == "velocity":
            return self.plan.score_from_velocity(out, x, t)
        return self.plan.score_from_data(out, x, t)


def odeint_euler(f: Callable[[Tensor, Tensor], Tensor], y0: Tensor, t: Tensor) -> Tensor:
    """Fixed-grid explicit Euler on the grid t (restated torchdiffeq behaviour, see module docstring)."""
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        y = y + (t[i + 1] - t[i]) * f(t[i], y)
        ys.append(y)
    return torch.stack(ys)


def sample_ode(tr: Transport, init: Tensor, model: Callable, *, num_steps=50, sampling_method="euler",
               reverse=False, **model_kwargs) -> Tensor:
    """Returns the stacked states [num_steps, ...]; callers take [-1] (lightning_base.py:230-234)."""
    if sampling_method != "euler":
        raise NotImplementedError("oracle restates only the fixed-grid euler solver")
    t0, t1 = tr.interval(sde=False, reverse=reverse, last_step_size=0.0)
    assert t0 < t1, "ODE sampler has to be in forward time"
    grid = torch.linspace(t0, t1, num_steps)

    def fn(t, x):
        tv = torch.ones(x.shape[0]) * t  # fp32 time vector whatever the state dtype (integrators.py:107-114)
        if reverse:
            tv = torch.ones_like(tv) * (1 - tv)
        return tr.velocity(x, tv, model, **model_kwargs)

    return odeint_euler(fn, init, grid)


def sample_sde(tr: Transport, init: Tensor, model: Callable, *, noise: Optional[List[Tensor]] = None,
               sampling_method="Euler", diffusion_form="linear", diffusion_norm=1.0, last_step="Mean",
               last_step_size=0.04, num_steps=250, single_eval=False, **model_kwargs) -> List[Tensor]:
    """Euler-Maruyama / Heun loop plus the last step.  Returns the list of num_steps states.

    noise: optional list of num_steps-1 standard-normal tensors used instead of drawing them
    (the reference draws ``th.randn(x.size())`` on the host each step, integrators.py:30,40).
    single_eval: evaluate the network once per drift (algebraically identical; the reference evaluates it
    twice with identical inputs because drift and score each call the model, transport.py:259-261).
    """
    if last_step is None:
        last_step_size = 0.0
    t0, t1 = tr.interval(diffusion_form=diffusion_form, sde=True, last_step_size=last_step_size)
    assert t0 < t1, "SDE sampler has to be in forward time"
    grid = torch.linspace(t0, t1, num_steps)
    dt = grid[1] - grid[0]

    def g(x, t):
        return tr.plan.diffusion(x, t, diffusion_form, diffusion_norm)

    def sde_drift(x, t):
        if single_eval:
            memo = {}

            def once(xx, tt, **kw):
                if "o" not in memo:
                    memo["o"] = model(xx, tt, **kw)
                return memo["o"]

            return tr.velocity(x, t, once, **model_kwargs) + g(x, t) * tr.score(x, t, once, **model_kwargs)
        return tr.velocity(x, t, model, **model_kwargs) + g(x, t) * tr.score(x, t, model, **model_kwargs)

    if sampling_method not in ("Euler", "Heun"):
        raise NotImplementedError("Smapler type not implemented.")

    x = init
    xs: List[Tensor] = []
    for i, ti in enumerate(grid[:-1]):
        w = noise[i].to(x) if noise is not None else torch.randn(x.size()).to(x)
        dw = w * torch.sqrt(dt)
        tv = torch.ones(x.size(0)).to(x) * ti
        if sampling_method == "Euler":
            d = sde_drift(x, tv)
            x = (x + d * dt) + torch.sqrt(2 * g(x, tv)) * dw
        else:
            xhat = x + torch.sqrt(2 * g(x, tv)) * dw
            k1 = sde_drift(xhat, tv)
            xp = xhat + dt * k1
            k2 = sde_drift(xp, tv + dt)
            x = xhat + 0.5 * dt * (k1 + k2)
        xs.append(x)

    ts = torch.ones(init.size(0)) * t1
    if last_step is None:
        last = xs[-1]
    elif last_step == "Mean":
        last = xs[-1] + sde_drift(xs[-1], ts) * last_step_size
    elif last_step == "Euler":
        last = xs[-1] + tr.velocity(xs[-1], ts, model, **model_kwargs) * last_step_size
    elif last_step == "Tweedie":
        a = tr.plan.alpha(ts)[0][0]
        s = tr.plan.sigma(ts)[0][0]
        last = xs[-1] / a + (s ** 2) / a * tr.score(xs[-1], ts, model, **model_kwargs)
    else:
        raise NotImplementedError()
    xs.append(last)
    assert len(xs) == num_steps, "Samples does not match the number of steps"
    return xs


ODE_DEFAULTS = {"sampling_method": "dopri5", "num_steps": 50, "atol": 1e-6, "rtol": 1e-3, "reverse": False}
SDE_DEFAULTS = {"sampling_method": "Euler", "diffusion_form": "linear", "diffusion_norm": 1.0,
                "last_step": "Mean", "last_step_size": 0.04, "num_steps": 250}


def get_sample_fn(tr: Transport, sampling_method="ODE", sampling_kwargs=None, **extra):
    """transport.py:475-503: merge kwargs over the defaults and return fn(init, model, **model_kwargs)."""
    kw = dict(sampling_kwargs or {})
    if sampling_method == "SDE":
        merged = {**SDE_DEFAULTS, **kw}
        return lambda init, model, **mk: sample_sde(tr, init, model, **merged, **extra, **mk)
    if sampling_method == "ODE":
        merged = {**ODE_DEFAULTS, **kw}
        merged.pop("atol"), merged.pop("rtol")
        return lambda init, model, **mk: sample_ode(tr, init, model, **merged, **mk)
    return None
