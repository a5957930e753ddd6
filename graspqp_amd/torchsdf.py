"""``torchsdf``-compatible module backed by the HIP kernels (SDF_BACKEND=HIP).

Same two entry points and conventions as the TorchSDF extension the reference imports
(``from torchsdf import compute_sdf, index_vertices_by_faces``; reference core/hand_model.py:32,
core/object_model.py:18):

    index_vertices_by_faces(verts (V,3) f32, faces (F,3) i64) -> (F,3,3) f32
    compute_sdf(points (N,3) f32, face_verts (F,3,3) f32)
        -> dist_sq (N,) f32, sign (N,) int32 {+1 outside, -1 inside}, normal (N,3) f32, closest (N,3) f32
       autograd: only dist_sq w.r.t. points.

To use it from unmodified reference code: ``sys.modules['torchsdf'] = graspqp_amd.torchsdf`` before importing
``graspqp.core``, or set ``SDF_BACKEND=HIP`` with the two-line patch shown there.
"""

from .ops import compute_sdf, index_vertices_by_faces

__all__ = ["compute_sdf", "index_vertices_by_faces"]
