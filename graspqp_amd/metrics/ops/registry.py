"""``SpanMetricWrapper`` / ``GraspSpanMetricFactory`` with the reference's surface (metrics/ops/registry.py:18-140).

``MetricType.GRASPQP`` (== ``--energy_type graspqp`` in scripts/fit.py:348-356) maps to the fused HIP op; the
returned callable is used exactly like the reference's ``energy_fnc`` (core/energy.py:35-42):

    E_fc, x = energy_fnc(contact_pts=, contact_normals=, sdf=, cog=, with_solution=True, svd_gain=)
"""

from enum import Enum

import torch

from ... import ops
from ..solver.qp_solver import SQPLsqSolver
from .span import OverallFrictionConeSpanMetric


def _require_hip_solver(solver_cls):
    """The reference lets ``solver_cls`` choose the QP back end (span.py:23-37,299; registry.py:112).  Here the solve is
    fused into the force-closure kernels, so the only class that can be honoured is the HIP ``SQPLsqSolver`` (or a
    subclass); anything else is refused loudly instead of being silently replaced."""
    if solver_cls is None or (isinstance(solver_cls, type) and issubclass(solver_cls, SQPLsqSolver)):
        return
    raise NotImplementedError(
        f"solver_cls={getattr(solver_cls, '__name__', solver_cls)!r}: graspqp_amd runs the force-closure QP inside its HIP "
        "kernels and honours only graspqp_amd.metrics.SQPLsqSolver here; for another solver use the reference's torch metric")


class SpanMetricWrapper(torch.nn.Module):
    def __init__(self, metric=OverallFrictionConeSpanMetric, metric_kwargs: dict = {}):
        super().__init__()
        self.metric = metric
        self._initialized = False
        self.metric_kwargs = dict(metric_kwargs)
        self.last_n_iter = None

    def forward(self, contact_pts, contact_normals, cog=None, contact_threshold: float = 0.0, torque_weight: float = 5.0,
                **kwargs):
        svd_gain = kwargs.pop("svd_gain", 0.1)
        values_gain = kwargs.pop("values_gain", 2.0)
        with_solution = kwargs.pop("with_solution", False)
        e, xs = ops.fc_energy(contact_pts, contact_normals, cog, **self.fc_config(svd_gain, values_gain, torque_weight))
        if with_solution:
            return e, xs
        return e

    def fc_config(self, svd_gain=0.1, values_gain=2.0, torque_weight=5.0) -> dict:
        """Keyword arguments of ``ops.fc_energy`` for this wrapper's metric_kwargs (resolved on first use, like the
        reference's lazy metric construction, registry.py:44-52)."""
        if not self._initialized:
            self._max_limit = self.metric_kwargs.pop("max_limit", None)
            self._friction = self.metric_kwargs.pop("friction", None)
            self._n_cone = self.metric_kwargs.pop("n_cone_vecs", 4)
            _require_hip_solver(self.metric_kwargs.pop("solver_cls", None))
            if self.metric is not OverallFrictionConeSpanMetric and not (
                    isinstance(self.metric, type) and issubclass(self.metric, OverallFrictionConeSpanMetric)):
                raise NotImplementedError(f"SpanMetricWrapper: metric {self.metric} is not the friction-cone span metric "
                                          "the HIP kernels implement (metrics/ops/span.py:298)")
            self._initialized = True
        max_limit = 50.0 if self._max_limit is None else self._max_limit  # span.py:28 default
        return dict(friction=0.2 if self._friction is None else self._friction, n_cone_vecs=self._n_cone,
                    torque_weight=torque_weight, max_limit=max_limit, svd_gain=svd_gain, values_gain=values_gain)


class GraspSpanMetricFactory:
    class MetricType(Enum):
        DEXGRASP = 1
        TDG = 2
        GRASPQP = 3
        GRASPQP_SCIPY = 4
        GRASPQP_EUCLIDIAN_SCIPY = 5

    @staticmethod
    def create(metric_type, solver_kwargs: dict = {}):
        MT = GraspSpanMetricFactory.MetricType
        if metric_type == MT.GRASPQP:
            return SpanMetricWrapper(
                OverallFrictionConeSpanMetric,
                metric_kwargs={
                    "solver_cls": SQPLsqSolver,
                    "friction": solver_kwargs.pop("friction", None),
                    "max_limit": solver_kwargs.pop("max_limit", None),
                    **solver_kwargs,
                },
            )
        if metric_type == MT.DEXGRASP:  # registry.py:104-105
            from .dexgrasp import DexgraspSpanMetric

            return DexgraspSpanMetric()
        if metric_type == MT.TDG:  # registry.py:106-107
            from .tdg import TDGSpanMetric

            return TDGSpanMetric(device="cuda")
        raise NotImplementedError(
            f"{metric_type}: the scipy-backed variants (GRASPQP_SCIPY, GRASPQP_EUCLIDIAN_SCIPY) are CPU solvers the "
            "reference keeps for visualisation only; graspqp_amd has no CPU path"
        )
