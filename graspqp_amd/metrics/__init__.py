from .ops.registry import GraspSpanMetricFactory, SpanMetricWrapper
from .solver.qp_solver import QPFunction, SQPLsqSolver

GraspQPSpanMetric = SpanMetricWrapper
__all__ = ["GraspSpanMetricFactory", "SpanMetricWrapper", "GraspQPSpanMetric", "SQPLsqSolver", "QPFunction"]
