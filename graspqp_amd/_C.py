"""ctypes binding of libgraspqp_hip.so -- the only way compute enters this package.

The prototypes are read from ``include/graspqp_hip.h`` (single source of truth for the C ABI); a test checks
that every declared symbol is exported.  There is deliberately no fallback: if the library is missing or a call
fails, a ``RuntimeError`` is raised.
"""

from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GRASPQP_HIP_LIB") or os.path.join(_HERE, "lib", "libgraspqp_hip.so")  # env: A/B builds
HEADER_PATH = os.path.join(_HERE, "..", "include", "graspqp_hip.h")

_SCALARS = {
    "int": ctypes.c_int,
    "int32_t": ctypes.c_int32,
    "int64_t": ctypes.c_int64,
    "float": ctypes.c_float,
    "size_t": ctypes.c_size_t,
    "uint64_t": ctypes.c_uint64,
}


def parse_header(path: str = HEADER_PATH) -> Dict[str, Tuple[object, List[object]]]:
    """name -> (restype, argtypes) for every ``gq_*`` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"typedef struct (\w+) \{.*?\} \1;", " ", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"(const char\*|int)\s+(gq_\w+)\s*\(([^;{]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        argtypes = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a:
                    argtypes.append(ctypes.c_void_p)
                else:
                    ty = a.replace("const ", "").split(" ")[0]
                    argtypes.append(_SCALARS[ty])
        protos[name] = (ctypes.c_char_p if ret.startswith("const char") else ctypes.c_int, argtypes)
    return protos


class HandDesc(ctypes.Structure):
    _fields_ = [
        ("n_dofs", ctypes.c_int32),
        ("n_links", ctypes.c_int32),
        ("n_cand", ctypes.c_int32),
        ("n_spheres", ctypes.c_int32),
        ("node_parent", ctypes.c_void_p),
        ("node_type", ctypes.c_void_p),
        ("node_pre", ctypes.c_void_p),
        ("node_axis", ctypes.c_void_p),
        ("link_node", ctypes.c_void_p),
        ("link_offset", ctypes.c_void_p),
        ("cand_pos", ctypes.c_void_p),
        ("cand_nrm", ctypes.c_void_p),
        ("cand_link", ctypes.c_void_p),
        ("sphere", ctypes.c_void_p),
        ("sphere_link", ctypes.c_void_p),
        ("joints_lower", ctypes.c_void_p),
        ("joints_upper", ctypes.c_void_p),
        ("n_actuated", ctypes.c_int32),
        ("coupling", ctypes.c_void_p),
        ("coupling_offset", ctypes.c_void_p),
    ]


class RowEnergyDesc(ctypes.Structure):
    _fields_ = [(k, ctypes.c_void_p) for k in ("dist_sq", "sign", "obj_dir", "hand_normals", "joints_lower",
                                                "joints_upper", "e_fc", "e_pen", "e_spen")] + [
        ("n", ctypes.c_int32)] + [(k, ctypes.c_float) for k in ("w_dis", "w_fc", "w_pen", "w_spen", "w_joints")] + [
        (k, ctypes.c_void_p) for k in ("e_dis", "e_joints", "total")]


def _struct(name, fields):
    ty = {"p": ctypes.c_void_p, "i": ctypes.c_int32, "l": ctypes.c_int64, "f": ctypes.c_float, "z": ctypes.c_size_t}
    return type(name, (ctypes.Structure,), {"_fields_": [(n, ty[t]) for n, t in fields]})


# field order = include/graspqp_hip.h
FcStepDesc = _struct("FcStepDesc", [
    ("dist_sq", "p"), ("sign", "p"), ("obj_dir", "p"), ("closest", "p"), ("contact_pts", "p"), ("hand_normals", "p"),
    ("cog", "p"), ("batch", "l"), ("n_contact", "i"), ("n_cone", "i"), ("friction", "f"), ("torque_weight", "f"),
    ("max_limit", "f"), ("svd_gain", "f"), ("values_gain", "f"), ("eps", "f"), ("max_iter", "i"), ("w_dis", "f"),
    ("w_fc", "f"), ("obj_normal", "p"), ("g_contact_pts", "p"), ("g_hand_normals", "p"), ("e_fc", "p"), ("x_sum", "p"),
    ("n_iter", "p"), ("workspace", "p"), ("workspace_bytes", "z")])
PenStepDesc = _struct("PenStepDesc", [
    ("links", "p"), ("surface_points", "p"), ("n_obj", "l"), ("n_surface", "l"), ("batch_each", "l"), ("hand_pose", "p"),
    ("pose_dim", "i"), ("Rg", "p"), ("link_T", "p"), ("dis", "p"), ("link", "p"), ("gvec", "p"), ("link_wrench", "p"),
    ("gRt", "p"), ("w_pen", "f"), ("e_pen", "p"), ("span", "p"), ("span_acc", "p"), ("hand", "p"), ("w_spen", "f"),
    ("e_spen", "p"), ("g_sphere_centers", "p"), ("sphere_centers", "p"), ("grid", "p"), ("patch_spheres", "p")])
AltFcDesc = _struct("AltFcDesc", [
    ("dist_sq", "p"), ("sign", "p"), ("obj_dir", "p"), ("closest", "p"), ("contact_pts", "p"), ("hand_normals", "p"),
    ("cog", "p"), ("batch", "l"), ("n_contact", "i"), ("energy", "i"), ("torque_weight", "f"), ("directions", "p"),
    ("n_directions", "i"), ("friction", "f"), ("obb_length", "f"), ("enable_density", "i"), ("scale", "f"), ("w_dis", "f"),
    ("w_fc", "f"), ("obj_normal", "p"), ("g_contact_pts", "p"), ("g_hand_normals", "p"), ("e_fc", "p")])

ProposeDesc = _struct("ProposeDesc", [
    ("hand_pose", "p"), ("grad", "p"), ("contact_idx", "p"), ("u_switch", "p"), ("new_idx", "p"), ("ema", "p"),
    ("step", "p"), ("step_size_out", "p"), ("g2_scratch", "p"), ("energy", "p"), ("batch_each", "l"), ("z_out", "p"), ("step_size", "f"),
    ("stepsize_period", "i"), ("decay", "f"), ("mu", "f"), ("switch_possibility", "f"), ("clip_grad", "i"),
    ("slot_ctr", "p"), ("slots", "i")])
SdfDesc = _struct("SdfDesc", [("meshes", "p"), ("queries_per_mesh", "l"), ("dist_sq", "p"), ("sign", "p"),
                              ("obj_dir", "p"), ("closest", "p")])
AcceptDesc = _struct("AcceptDesc", [
    ("u_accept", "p"), ("z", "p"), ("reset_mask", "p"), ("step", "p"), ("starting_temperature", "f"), ("decay", "f"),
    ("annealing_period", "i"), ("energy", "p"), ("pose", "p"), ("idx", "p"), ("grad", "p"), ("accept", "p"),
    ("temperature", "p"), ("n_terms", "i"), ("terms_new", "p"), ("terms", "p"), ("slot_ctr", "p"), ("slots", "i")])

InitDesc = _struct("InitDesc", [
    ("hull_face_verts", "p"), ("hull_cdf", "p"), ("hull_offsets", "p"), ("n_obj", "l"), ("batch_each", "l"),
    ("samples_per_object", "l"), ("n_dofs", "i"), ("inflate", "f"),
    ("forward_axis0", "f"), ("forward_axis1", "f"), ("forward_axis2", "f"), ("up_axis0", "f"), ("up_axis1", "f"), ("up_axis2", "f"),
    ("default_state", "p"), ("joints_lower", "p"), ("joints_upper", "p"),
    ("jitter_strength", "f"), ("distance_lower", "f"), ("distance_upper", "f"), ("rotate_lower", "f"), ("rotate_upper", "f"),
    ("pitch_lower", "f"), ("pitch_upper", "f"), ("tilt_lower", "f"), ("tilt_upper", "f"),
    ("u_face", "p"), ("u_len", "p"), ("u_pose", "p"), ("u_joint", "p"), ("hand_pose", "p"), ("shell_points", "p"),
    ("shell_dirs", "p"), ("workspace", "p"), ("workspace_bytes", "z")])

_lib = None
_protos = None


def lib() -> ctypes.CDLL:
    """Load (once) and return the library; raises if it is not built."""
    global _lib, _protos
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(graspqp_amd has no CPU fallback)"
            )
        l = ctypes.CDLL(LIB_PATH)
        _protos = parse_header()
        for name, (res, args) in _protos.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def prototypes():
    lib()
    return _protos


def call(name: str, *args):
    """Call an ``int``-returning entry point; non-zero status -> RuntimeError with the library's message."""
    l = lib()
    rc = getattr(l, name)(*args)
    if rc != 0:
        msg = l.gq_last_error()
        raise RuntimeError(f"{name} failed (code {rc}): {msg.decode() if msg else ''}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> ctypes.c_void_p:
    """hipStream_t of torch's current stream on the current device (the raw handle: building a torch.cuda.Stream object
    for every C-ABI call costs ~5 us of host time, a dozen times per iteration of a class-surface loop)."""
    if _raw_stream is not None:
        return ctypes.c_void_p(_raw_stream(torch.cuda.current_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t, dtype=None):
    """Device pointer of a contiguous CUDA tensor (None -> NULL)."""
    if t is None:
        return ctypes.c_void_p(0)
    if not t.is_cuda:
        raise RuntimeError("graspqp_amd ops need CUDA (ROCm) tensors; got a CPU tensor")
    if not t.is_contiguous():
        raise RuntimeError("graspqp_amd ops need contiguous tensors")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"expected dtype {dtype}, got {t.dtype}")
    return ctypes.c_void_p(t.data_ptr())


def f32(t):
    return ptr(t, torch.float32)


def i32(t):
    return ptr(t, torch.int32)


def i64(t):
    return ptr(t, torch.int64)


def u8(t):
    return ptr(t, torch.uint8)
