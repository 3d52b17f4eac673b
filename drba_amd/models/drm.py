"""DistanceRatioMap (DRM) operators on the HIP library; names/arguments as the reference's models/drm.py.

calc_drm_rife with linear=True (the only form the driver uses, infer.py:143) runs as two
fused kernels per direction (distance + ratio + scale + splat scatter, then normalise +
hole fill).  The remaining variants compose the generic HIP operators.
"""
import torch

from drba_amd import ops as _ops
from drba_amd.models.softsplat.softsplat import softsplat as warp


def get_drm_t(drm, t, precision=1e-3):
    """Non-linear retiming (reference drm.py:10-62)."""
    return _ops.drm_retime(drm, t, precision)


def _unaligned(flow10, flow12, t, linear, eps):
    drm10, drm12 = _ops.drm_ratio(flow10, flow12, eps)
    if linear:
        return drm10, _ops.affine(drm10, float(t) * 2.0, 0.0), _ops.affine(drm12, float(t) * 2.0, 0.0)
    return drm10, get_drm_t(drm10, t), get_drm_t(drm12, t)


def _aligned(value, flow, metric, mode, ones):
    """warp(value) and warp(ones) along the same (flow, metric): softsplat treats channels independently (same weights,
    same normaliser, same accumulation order), so both maps are splatted as one 2-channel tensor -- one sort and one
    gather instead of two, bit-identical to the reference's two calls (drm.py:96-104, :184-192)."""
    both = warp(torch.cat((value, ones), 1), flow, metric, mode)
    return _ops.fill_holes(both[:, 0:1].contiguous(), both[:, 1:2].contiguous(), value)


def calc_drm_rife(t, flow10, flow12, linear=False):
    """reference drm.py:65-107 -> {'drm_t1_t01', 'drm_t1_t12'}"""
    if linear:
        return {"drm_t1_t01": _ops.drm_rife_linear(flow10, flow12, t, 1e-4),
                "drm_t1_t12": _ops.drm_rife_linear(flow12, flow10, t, 1e-4)}
    drm10, u0, u1 = _unaligned(flow10, flow12, t, False, 1e-4)
    ones = _ops.affine(drm10, 0.0, 1.0)  # drm10 * 0 + 1 (drm.py:92)
    return {"drm_t1_t01": _aligned(u1, _ops.mul_map(flow10, u1), None, "avg", ones),
            "drm_t1_t12": _aligned(u0, _ops.mul_map(flow12, u0), None, "avg", ones)}


def calc_drm_rife_auxiliary(t, flow10, flow12, metric10, metric12, linear=False):
    """reference drm.py:158-195: as calc_drm_rife, 'soft' splats when both metrics are given."""
    mode = "soft" if (metric10 is not None and metric12 is not None) else "avg"
    if mode == "avg":
        return calc_drm_rife(t, flow10, flow12, linear)
    drm10, u0, u1 = _unaligned(flow10, flow12, t, linear, 1e-4)
    ones = _ops.affine(drm10, 0.0, 1.0)
    return {"drm_t1_t01": _aligned(u1, _ops.mul_map(flow10, u1), metric10, mode, ones),
            "drm_t1_t12": _aligned(u0, _ops.mul_map(flow12, u0), metric12, mode, ones)}


def calc_drm_gmfss(t, flow10, flow12, metric10, metric12, linear=False):
    """reference drm.py:110-155: no +1e-4 on distances; complementary maps splatted along the unscaled flows."""
    mode = "soft" if (metric10 is not None and metric12 is not None) else "avg"
    _, u0, u1 = _unaligned(flow10, flow12, t, linear, 0.0)
    drm1t_t01, drm1t_t12 = u1, u0
    c01 = _ops.affine(drm1t_t01, -1.0, 1.0)  # 1 - drm
    c12 = _ops.affine(drm1t_t12, -1.0, 1.0)
    a01 = warp(c01, flow10, metric10, mode)
    a12 = warp(c12, flow12, metric12, mode)
    ones = _ops.affine(a01, 0.0, 1.0)  # built from the splatted map (drm.py:135)
    cov01 = warp(ones, flow10, metric10, mode)
    cov12 = warp(ones, flow12, metric12, mode)
    return {"drm0t_t01": _ops.fill_holes(a01, cov01, c01), "drm1t_t01": drm1t_t01,
            "drm1t_t12": drm1t_t12, "drm2t_t12": _ops.fill_holes(a12, cov12, c12)}
