"""initialize_convex_hull with the reference's signature (core/initializations.py:15-193) on the HIP kernels.

Translation / rotation from the inflated convex hull of every object (surface samples -> farthest-point sampling ->
look-at rotation -> random stand-off distance, roll, pitch, tilt), joint angles from a truncated normal around the hand's
default state, random contact indices; handed to ``HandModel.set_parameters(..., env_mask=...)`` at the end, so a call
with ``env_mask`` re-initialises just those rows (scripts/fit.py:408-422).  Everything runs on the device: no trimesh /
pytorch3d / transforms3d, no host round trip (the hull itself is set-up data of the ObjectModel).
"""

from __future__ import annotations

import ctypes
import math

import torch

from .. import _C, ops

# scripts/fit.py:59-71
DEFAULT_INIT_ARGS = dict(jitter_strength=0.1, distance_lower=0.05, distance_upper=0.1, rotate_lower=-math.pi,
                         rotate_upper=math.pi, pitch_lower=-15 * math.pi / 180, pitch_upper=15 * math.pi / 180,
                         tilt_lower=-45 * math.pi / 180, tilt_upper=45 * math.pi / 180)


def _arg(args, name):
    if args is not None and hasattr(args, name):
        return float(getattr(args, name))
    if isinstance(args, dict) and name in args:
        return float(args[name])
    return float(DEFAULT_INIT_ARGS[name])


def convex_hull_poses(hand_spec, hulls, n_obj, batch_each, args=None, generator=None, device="cuda", draws=None,
                      samples_per_object=None, return_shell=False):
    """The pose part of initialize_convex_hull for all n_obj * batch_each rows -> hand_pose (B, 9 + J) on the device.
    ``hulls`` = ObjectModel.convex_hulls(); ``draws`` = dict(u_face, u_len, u_pose, u_joint) to inject the uniforms."""
    dev = torch.device(device)
    fv, cdf, off = hulls
    J = hand_spec.n_dofs
    B = n_obj * batch_each
    M = int(samples_per_object or 100 * batch_each)  # initializations.py:57
    if draws is None:
        r = lambda *s: torch.rand(*s, device=dev, generator=generator)
        draws = {"u_face": r(n_obj, M), "u_len": r(n_obj, M, 2), "u_pose": r(B, 4), "u_joint": r(B, J)}
    f32 = lambda t: t.to(dev, torch.float32).contiguous()
    d = {k: f32(v) for k, v in draws.items()}
    nb = ops._size_call("gq_init_workspace_bytes", ctypes.c_int64(n_obj), ctypes.c_int64(M), ctypes.c_int64(batch_each))
    ws = ops._ws(nb, dev)
    pose = torch.empty(B, 9 + J, device=dev)
    shell_p = torch.empty(B, 3, device=dev) if return_shell else None
    shell_n = torch.empty(B, 3, device=dev) if return_shell else None
    consts = [torch.tensor(v, dtype=torch.float32, device=dev) for v in
              (hand_spec.default_state, hand_spec.joints_lower, hand_spec.joints_upper)]
    de = _C.InitDesc()
    de.hull_face_verts, de.hull_cdf, de.hull_offsets = fv.data_ptr(), cdf.data_ptr(), off.data_ptr()
    de.n_obj, de.batch_each, de.samples_per_object, de.n_dofs, de.inflate = n_obj, batch_each, M, J, 0.01
    for i in range(3):
        setattr(de, f"forward_axis{i}", float(hand_spec.forward_axis[i]))
        setattr(de, f"up_axis{i}", float(hand_spec.up_axis[i]))
    de.default_state, de.joints_lower, de.joints_upper = (c.data_ptr() for c in consts)
    for k in DEFAULT_INIT_ARGS:
        setattr(de, k, _arg(args, k))
    de.u_face, de.u_len, de.u_pose, de.u_joint = (d[k].data_ptr() for k in ("u_face", "u_len", "u_pose", "u_joint"))
    de.hand_pose = pose.data_ptr()
    de.shell_points = shell_p.data_ptr() if return_shell else None
    de.shell_dirs = shell_n.data_ptr() if return_shell else None
    de.workspace, de.workspace_bytes = ws.data_ptr(), nb
    _C.call("gq_init_convex_hull", ctypes.byref(de), _C.stream_ptr())
    return (pose, shell_p, shell_n) if return_shell else pose


def initialize_convex_hull(hand_model, object_model, args=None, env_mask=None, energy_checker=None, init_contacts=True,
                           generator=None):
    """Reference signature (initializations.py:15).  ``args`` carries the ranges of scripts/fit.py:59-71 and ``n_contact``
    (a Namespace or dict; missing entries take the reference defaults)."""
    n_obj = len(object_model.object_mesh_list)
    be = object_model.batch_size_each
    pose = convex_hull_poses(hand_model.spec, object_model.convex_hulls(), n_obj, be, args, generator, hand_model.device)
    if not init_contacts:  # initializations.py:183-184
        return pose[:, :3], pose[:, 3:9], pose[:, 9:]
    if hasattr(args, "n_contact"):
        n_contact = int(args.n_contact)
    elif isinstance(args, dict) and "n_contact" in args:
        n_contact = int(args["n_contact"])
    else:
        n_contact = 12  # scripts/fit.py:38
    idx = torch.randint(hand_model.n_contact_candidates, (n_obj * be, n_contact), device=hand_model.device, generator=generator)
    hand_model.set_parameters(pose.requires_grad_(), idx, env_mask=env_mask)
    return pose, idx
