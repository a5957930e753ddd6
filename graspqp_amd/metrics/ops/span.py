"""Force-closure span metric with the reference's class surface (metrics/ops/span.py:298-415).

``OverallFrictionConeSpanMetric`` here is a thin module over the fused HIP op ``ops.fc_energy`` pieces: the
friction cone, grasp matrix, box-QP and svd scale all run in the kernels of csrc/fc.hip + csrc/qp*.hip.
``forward`` returns ``(res (B,1), basis (B,1,6), svd_scales (B,1)[, values (B,n)])`` like the reference.
"""

import torch

from ... import ops
from ..solver.qp_solver import SQPLsqSolver


class OverallFrictionConeSpanMetric(torch.nn.Module):
    n_basis_vectors = 1

    def __init__(self, solver_cls=SQPLsqSolver, friction=0.2, n_cone_vecs=4):
        super().__init__()
        self._mu = friction if friction is not None else 0.2
        self.n_cone_vecs = n_cone_vecs
        self._max_limit_value = 50  # span.py:28 default; SpanMetricWrapper overrides it
        if not (isinstance(solver_cls, type) and issubclass(solver_cls, SQPLsqSolver)):
            raise NotImplementedError(
                f"solver_cls={getattr(solver_cls, '__name__', solver_cls)!r}: the QP is solved inside the HIP force-closure "
                "kernels; only graspqp_amd.metrics.SQPLsqSolver (or a subclass) can be honoured (reference span.py:23-37)")
        self._solver_cls = solver_cls
        self._cache = {}

    @classmethod
    def from_dim(cls, num_wrenches, wrench_dim, batch_size=1, device="cuda", solver_cls=SQPLsqSolver, **kwargs):
        friction = kwargs.pop("friction", 0.2)
        n_cone_vecs = kwargs.pop("n_cone_vecs", 4)
        if len(kwargs) > 0:
            print("WARNING: Unknown kwargs", kwargs.keys())
        if wrench_dim != 6:
            raise NotImplementedError("graspqp_amd span metric supports 3-D contacts (wrench_dim=6) only")
        return cls(solver_cls=solver_cls, friction=friction, n_cone_vecs=n_cone_vecs)

    def forward(self, contact_pts, contact_normals, cog=None, contact_threshold=0.0, reg=0.0, env_ids=None,
                return_solution=True, torque_weight=5):
        B = contact_pts.shape[0]
        # values_gain=1, eps_add folded out: ask the fused op for val and svd separately
        e, xs, val, svd = _fc_parts(contact_pts, contact_normals, cog, self._mu, self.n_cone_vecs, torque_weight,
                                    self._max_limit_value)
        basis = torch.zeros(B, 1, 6, device=contact_pts.device)
        if not return_solution:
            return val.unsqueeze(-1), basis, svd.unsqueeze(-1)
        return val.unsqueeze(-1), basis, svd.unsqueeze(-1), xs


class _FcParts(torch.autograd.Function):
    """(val, svd) = (1/2|Fx|^2, (prod sigma)^(1/6)) as separately differentiable outputs.

    Implemented with two fused-energy evaluations' worth of backward: d val = backward with svd_gain = 0,
    d svd via E = exp(-svd) trick is avoided by calling the kernel backward twice with unit gains.
    """

    @staticmethod
    def forward(ctx, cp, cn, cog, mu, k, tw, max_limit):
        cfg = dict(ops.FC_DEFAULTS)
        cfg.update(friction=mu, n_cone_vecs=k, torque_weight=tw, max_limit=max_limit, svd_gain=0.0, values_gain=1.0)
        with torch.enable_grad():
            cpr = cp.detach().requires_grad_()
            e0, xs = ops.fc_energy(cpr, cn, cog, **cfg)  # e0 = val + 0.01
        cfg1 = dict(cfg)
        cfg1.update(svd_gain=1.0)
        with torch.enable_grad():
            cpr1 = cp.detach().requires_grad_()
            e1, _ = ops.fc_energy(cpr1, cn, cog, **cfg1)  # e1 = (val + 0.01) exp(-svd)
        val = e0.detach() - 1e-2
        svd = -torch.log(e1.detach() / e0.detach())
        ctx.graphs = (cpr, e0, cpr1, e1)
        ctx.save_for_backward(e0.detach(), e1.detach())
        ctx.mark_non_differentiable(xs)
        return e0.detach(), xs, val, svd

    @staticmethod
    def backward(ctx, g_e, g_xs, g_val, g_svd):
        cpr, e0, cpr1, e1 = ctx.graphs
        e0d, e1d = ctx.saved_tensors
        g0 = torch.zeros_like(e0d) if g_val is None else g_val.clone()
        if g_e is not None:
            g0 = g0 + g_e
        gp = torch.zeros_like(cpr)
        # svd = log e0 - log e1
        if g_svd is not None:
            (ga,) = torch.autograd.grad(e1, cpr1, -g_svd / e1d, retain_graph=False)
            gp = gp + ga
            g0 = g0 + g_svd / e0d
        (gb,) = torch.autograd.grad(e0, cpr, g0, retain_graph=False)
        return gp + gb, None, None, None, None, None, None


def _fc_parts(cp, cn, cog, mu, k, tw, max_limit):
    return _FcParts.apply(cp, cn, cog, mu, k, tw, max_limit)
