#!/usr/bin/env python3
"""Build graspqp_amd/assets/hands/<hand>.npz from a reference-style asset directory.

Usage: python tools/build_hand_specs.py [/root/reference/graspqp/assets]
The output is plain numeric data (kinematic tree, baked triangle soups, contact candidates,
penetration spheres); it is what travels to the GPU box, where the reference tree does not exist.
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import importlib.util

spec_py = os.path.join(os.path.dirname(__file__), "..", "graspqp_amd", "hands", "__init__.py")
# import the sub-package without importing graspqp_amd/__init__ (which needs the HIP library)
import types

pkg = types.ModuleType("graspqp_amd")
pkg.__path__ = [os.path.join(os.path.dirname(__file__), "..", "graspqp_amd")]
sys.modules.setdefault("graspqp_amd", pkg)
from graspqp_amd.hands import AVAILABLE_HANDS, get_hand_spec  # noqa: E402

asset_dir = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/graspqp/assets"
out = os.path.join(os.path.dirname(__file__), "..", "graspqp_amd", "assets", "hands")
os.makedirs(out, exist_ok=True)
for h in AVAILABLE_HANDS:
    s = get_hand_spec(h, asset_dir)
    s.save(os.path.join(out, f"{h}.npz"))
    nf = s.link_face_offset[1:] - s.link_face_offset[:-1]
    print(f"{h}: dofs={s.n_dofs} links={s.n_links} faces={int(nf.sum())} {nf.tolist()} cands={s.n_contact_candidates} spheres={s.n_spheres}")
    print("   links:", s.link_names)
    print("   joints:", s.joint_names)
