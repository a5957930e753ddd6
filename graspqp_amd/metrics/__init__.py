from .ops.dexgrasp import DexgraspSpanMetric
from .ops.registry import GraspSpanMetricFactory, SpanMetricWrapper
from .ops.tdg import TDGSpanMetric
from .solver.qp_solver import QPFunction, SQPLsqSolver

GraspQPSpanMetric = SpanMetricWrapper
__all__ = ["GraspSpanMetricFactory", "SpanMetricWrapper", "GraspQPSpanMetric", "SQPLsqSolver", "QPFunction",
           "DexgraspSpanMetric", "TDGSpanMetric"]
