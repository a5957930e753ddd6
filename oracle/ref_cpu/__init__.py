"""ref_cpu -- CPU restatement of the GraspQP hot path.  TEST INFRASTRUCTURE ONLY.

This package is the *oracle*: a plain-torch (CPU, fp32 or fp64) restatement of the reference's
per-MALA*-iteration algorithm, function by function, each citing the reference file:line it
follows.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product (``graspqp_amd``) never does and fails loudly without its HIP library.

Pinning status (see DESIGN.md "Oracle"):
  * friction cone / grasp matrix F / svd scale / E_fc formula / calculate_energy composition /
    MalaStar propose+accept: pinned against the reference's own files executed in the build
    container (``tools/make_golden.py`` -> ``tests/golden/*.npz``).
  * bounded least-squares KATs of the reference tests (``tests/metrics/test_solver.py``): pinned.
  * the third-party arithmetic that is absent from the reference tree -- qpth 0.0.18 PDIPM iterate,
    TorchSDF closest-feature sign/tie rules, pytorch_kinematics URDF FK, roma Gram-Schmidt -- is
    restated from the published algorithms: PARITY UNPINNED for those outputs.
"""

from . import export, init, kin, mala, metrics_alt, models, qp, sdf, span  # noqa: F401
from .energy import calculate_energy, total_energy  # noqa: F401
