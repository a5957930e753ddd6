"""HIP-backed drop-in for reference ``graspqp/metrics/solver/qp_solver.py`` (SQPLsqSolver) and for the
``qpth.qp.QPFunction`` callable it wraps.  Same constructor / ``from_mat`` / ``build_solver`` / ``solve``
signatures and the same (value, x) return shapes; gradients flow to ``A`` (and ``b``) through the implicit KKT
backward exactly as with qpth."""

import torch

from ... import ops


class QPFunction:
    """``QPFunction(verbose, maxIter, eps)(Q, p, G, h, A, b) -> x`` for G = [I; -I] (box constraints).

    reference call site: qp_solver.py:8,125.  ``G`` must be the stacked identity the reference builds
    (qp_solver.py:105-111); anything else, or equality constraints, raises -- the HIP kernel is a box-QP solver.
    """

    def __init__(self, verbose=False, maxIter=12, eps=5e-2, notImprovedLim=3, check_Q_spd=False):
        self.maxIter, self.eps, self.notImprovedLim = maxIter, eps, notImprovedLim
        self.last = None

    def __call__(self, Q, p, G, h, A=None, b=None):
        if A is not None and A.numel() > 0:
            raise NotImplementedError("graspqp_amd QPFunction: equality constraints are not supported")
        nz = Q.shape[-1]
        Gd = G if G.dim() == 2 else G[0]
        eye = torch.eye(nz, device=Gd.device, dtype=Gd.dtype)
        if Gd.shape != (2 * nz, nz) or not torch.equal(Gd, torch.cat([eye, -eye], dim=0)):
            raise NotImplementedError("graspqp_amd QPFunction: G must be [I; -I] (box constraints)")
        if Q.dim() == 2:
            Q = Q.unsqueeze(0)
        B = max(Q.shape[0], p.shape[0] if p.dim() == 2 else 1, h.shape[0] if h.dim() == 2 else 1)
        Q = Q.expand(B, nz, nz)
        p = p.expand(B, nz) if p.dim() == 2 else p.unsqueeze(0).expand(B, nz)
        h = h.expand(B, 2 * nz) if h.dim() == 2 else h.unsqueeze(0).expand(B, 2 * nz)
        upper, lower = h[:, :nz], -h[:, nz:]
        x, lam, slack = ops.box_qp(Q, p, lower, upper, self.eps, self.maxIter, self.notImprovedLim)
        self.last = (lam, slack)
        return x


class SQPLsqSolver:
    def __init__(self, sum_to_one=False):
        if sum_to_one:
            raise NotImplementedError("sum_to_one (equality constraint) is not supported by the HIP box-QP")
        self._sum_to_one = sum_to_one
        self._max_iter, self._eps = 12, 5e-2  # qp_solver.py:8

    @classmethod
    def from_mat(cls, A, b, step_size=0.15, solver_kwargs={}):
        solver = cls(solver_kwargs.pop("sum_to_one", False))
        solver.build_solver_from_mat(A, b, step_size=step_size, solver_kwargs=solver_kwargs)
        return solver

    def build_solver_from_mat(self, A, b, step_size=0.15, solver_kwargs={}):
        if A.ndim == 2:
            A = A.unsqueeze(0)
        if b.ndim == 1:
            b = b.unsqueeze(0)
        batch_size = A.shape[0] * A.shape[1] if A.ndim == 4 else A.shape[0]
        self.build_solver(A.shape[-1], b.shape[-1], batch_size, device=A.device, step_size=step_size,
                          solver_kwargs=solver_kwargs)

    def to(self, device):
        self._device = device

    def build_solver(self, num_wrenches, wrench_dim, batch_size=1, step_size=0.15, device="cuda", solver_kwargs={}):
        self._num_wrenches, self._wrench_dim = num_wrenches, wrench_dim
        self._batch_size, self._device, self._step_size = batch_size, device, step_size

    def __call__(self, A, b, **kwargs):
        return self.solve(A, b, **kwargs)

    def solve(self, A, b, init=None, min_bound=-1e4, max_bound=1e4, return_solution=False, **kwargs):
        """min 1/2 |A x - b|^2 + 5e-5 |x|^2 s.t. min_bound <= x <= max_bound (qp_solver.py:60-134).
        ``init`` is accepted and ignored, as in the reference (it is never passed to qpth)."""
        if len(kwargs) > 0:
            print("WARNING: Unknown kwargs passed to solver", SQPLsqSolver.__name__, list(kwargs.keys()))
        batch_shape = (A.shape[0],)
        if A.ndim == 4:
            batch_shape = A.shape[0], A.shape[1]
            if b.shape[0] != A.shape[0]:
                b = b.expand(A.shape[0], -1, -1)
            A = A.flatten(0, 1)
            b = b.flatten(0, 1)
        x = ops.lsq_box_qp(A, b, float(min_bound), float(max_bound), 1e-4, self._eps, self._max_iter)
        value = 0.5 * torch.sum((b - (A @ x.unsqueeze(-1)).squeeze(-1)).pow(2), -1)
        x = x.view(*batch_shape, self._num_wrenches)
        value = value.view(*batch_shape)
        if return_solution:
            return value, x
        return value
