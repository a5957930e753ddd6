"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/graspqp_hip.h declares, the host mirror has the reference's surface, and nothing falls back to the CPU."""
import os
import re

import numpy as np
import pytest
import torch

from graspqp_amd import _C


def test_library_exports_every_declared_symbol():
    protos = _C.parse_header()
    assert len(protos) >= 30
    lib = _C.lib()
    for name in protos:
        assert hasattr(lib, name), name
    assert lib.gq_version() >= 100
    for required in ("gq_sdf_forward", "gq_sdf_backward", "gq_boxqp_forward", "gq_boxqp_backward", "gq_fk_forward",
                     "gq_fk_backward", "gq_hand_pen_forward", "gq_hand_pen_backward", "gq_self_pen_forward",
                     "gq_mala_propose", "gq_mala_accept", "gq_fc_forward", "gq_fc_backward"):
        assert required in protos, required


def test_argument_validation_without_gpu():
    """Size helpers are pure host code: callable without a GPU, bad arguments give an error code + message."""
    import ctypes

    lib = _C.lib()
    out = ctypes.c_size_t(0)
    assert lib.gq_boxqp_workspace_bytes(256, 48, 12, ctypes.byref(out)) == 0 and out.value > 256 * 12 * 5 * 48 * 4
    assert lib.gq_boxqp_workspace_bytes(256, 0, 12, ctypes.byref(out)) != 0
    assert b"bad arguments" in lib.gq_last_error()
    assert lib.gq_fc_workspace_bytes(4, 4, 4, 12, ctypes.byref(out)) == 0 and out.value > 0
    assert lib.gq_sdf_workspace_bytes(1000, ctypes.byref(out)) == 0 and out.value >= 64000


def test_no_cpu_fallback():
    from graspqp_amd import ops

    with pytest.raises(RuntimeError):
        ops.compute_sdf(torch.zeros(4, 3), torch.zeros(2, 3, 3))
    from graspqp_amd.core.hand_model import HandModel
    from graspqp_amd.hands import get_hand_spec

    with pytest.raises(RuntimeError):
        HandModel(get_hand_spec("allegro"), device="cpu")


def test_product_does_not_import_oracle():
    root = os.path.join(os.path.dirname(__file__), "..", "graspqp_amd")
    for dp, _, fs in os.walk(root):
        for f in fs:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+(ref_cpu|oracle)", src, flags=re.M), f"{f} imports the oracle"
                assert "sys.path" not in src or "oracle" not in src, f"{f} reaches into oracle/"


def test_hand_specs():
    from graspqp_amd.hands import AVAILABLE_HANDS, get_hand_spec

    expect = {"allegro": (16, 14, 92, 25), "shadow_hand": (24, 18, 80, 22), "robotiq3": (11, 12, 48, 12)}
    for h in AVAILABLE_HANDS:
        s = get_hand_spec(h)
        assert (s.n_dofs, s.n_links, s.n_contact_candidates, s.n_spheres) == expect[h]
        assert (s.node_parent < np.arange(s.n_dofs)).all()
        assert np.allclose(np.linalg.norm(s.cand_nrm, axis=1), 1.0, atol=1e-4)
        assert (s.joints_lower < s.joints_upper).all()


def test_metric_factory_surface():
    from graspqp_amd.metrics import GraspSpanMetricFactory as GF
    from graspqp_amd.metrics import SQPLsqSolver

    fn = GF.create(GF.MetricType.GRASPQP, {"friction": 0.2, "max_limit": 20.0, "n_cone_vecs": 4})
    assert callable(fn)
    with pytest.raises(NotImplementedError):
        GF.create(GF.MetricType.TDG)
    s = SQPLsqSolver.from_mat(torch.zeros(2, 1, 6, 48), torch.zeros(2, 1, 6))
    assert s._batch_size == 2 and s._num_wrenches == 48


def test_descriptor_struct_layouts_match_the_header(tmp_path):
    """The ctypes mirrors of the descriptor structs (graspqp_amd/_C.py) must have the size and field offsets that a C
    compiler gives the structs of include/graspqp_hip.h -- a silent mismatch would hand garbage pointers to kernels."""
    import ctypes
    import shutil
    import subprocess

    from graspqp_amd import _C

    if shutil.which("gcc") is None:
        pytest.skip("no C compiler")
    pairs = {"gqHandDesc": _C.HandDesc, "gqRowEnergyDesc": _C.RowEnergyDesc, "gqFcStepDesc": _C.FcStepDesc,
             "gqPenStepDesc": _C.PenStepDesc, "gqProposeDesc": _C.ProposeDesc, "gqAcceptDesc": _C.AcceptDesc,
             "gqSdfDesc": _C.SdfDesc, "gqInitDesc": _C.InitDesc}
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "graspqp_hip.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in cls._fields_:
            m = re.fullmatch(r"(\w+_axis)(\d)", fname)  # float[3] members are mirrored element by element
            cfield = f"{m.group(1)}[{m.group(2)}]" if m else fname
            lines.append(f'  printf(" %zu", offsetof({cname}, {cfield}));')
        lines.append('  printf("\\n");')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include")
    subprocess.check_call(["gcc", "-std=c11", "-I", inc, str(src), "-o", str(exe)])
    out = subprocess.check_output([str(exe)]).decode().strip().splitlines()
    for line in out:
        name, *nums = line.split()
        cls = pairs[name]
        want = [ctypes.sizeof(cls)] + [getattr(cls, f).offset for f, _ in cls._fields_]
        assert [int(x) for x in nums] == want, f"{name}: C {nums} vs ctypes {want}"
