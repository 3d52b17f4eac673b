"""Oracle DistanceRatioMap (DRM) computation — restates models/drm.py.  (test infra)"""
import torch

from .ops import distance, softsplat


def drm_to_t(drm, t, precision=1e-3):
    """Non-linear DRM retiming — models/drm.py:10-62.

    A scalar x starts at 0.5 and bisects toward t inside the bracket (l, r); every
    scalar move is replayed on the whole map with the map itself as the step fraction,
    so each value keeps its proportion.  Both `if`s may fire in one iteration.
    """
    dt = drm.dtype
    x, frac = 0.5, 0.5
    lo, hi = 0, 1
    xm = drm.float().clone()
    fm = drm.float().clone()
    lom, him = xm * 0, xm * 0 + 1
    while abs(x - t) > precision:
        if x > t:
            hi = x
            x = x - (x - lo) * frac
            him = xm.clone()
            xm = xm - (xm - lom) * fm
        if x < t:
            lo = x
            x = x + (hi - x) * frac
            lom = xm.clone()
            xm = xm + (him - xm) * fm
    return xm.to(dt)


def _ratio_maps(flow10, flow12, eps):
    d10 = distance(flow10)
    d12 = distance(flow12)
    if eps:
        d10 = d10 + eps
        d12 = d12 + eps
    return d10 / (d10 + d12), d12 / (d10 + d12)


def _fill(aligned, value, ones, flow, metric, mode):
    """Holes = where the splat of `ones` along `flow` is < 0.999; they keep `value`."""
    cover = softsplat(ones, flow, metric, mode)
    return torch.where(cover < 0.999, value, aligned)


def calc_drm_rife(t, flow10, flow12, linear=False):
    """models/drm.py:65-107.  d = |flow| + 1e-4; ratio maps; x 2t (linear) or drm_to_t;
    each map is moved to time t by an 'avg' splat along flow*map; uncovered pixels keep
    the unaligned value.  Returns {'drm_t1_t01', 'drm_t1_t12'}."""
    drm10, drm12 = _ratio_maps(flow10, flow12, 1e-4)
    if linear:
        u0 = drm10 * t * 2
        u1 = drm12 * t * 2
    else:
        u0 = drm_to_t(drm10, t)
        u1 = drm_to_t(drm12, t)
    a01 = softsplat(u1, flow10 * u1, None, "avg")
    a12 = softsplat(u0, flow12 * u0, None, "avg")
    ones = drm10 * 0 + 1  # drm.py:92 (NaN-propagating "ones", kept as written)
    return {
        "drm_t1_t01": _fill(a01, u1, ones, flow10 * u1, None, "avg"),
        "drm_t1_t12": _fill(a12, u0, ones, flow12 * u0, None, "avg"),
    }


def calc_drm_rife_auxiliary(t, flow10, flow12, metric10, metric12, linear=False):
    """models/drm.py:158-195.  As calc_drm_rife but 'soft' splats when both metrics are given."""
    drm10, drm12 = _ratio_maps(flow10, flow12, 1e-4)
    if linear:
        u0 = drm10 * t * 2
        u1 = drm12 * t * 2
    else:
        u0 = drm_to_t(drm10, t)
        u1 = drm_to_t(drm12, t)
    mode = "soft" if (metric10 is not None and metric12 is not None) else "avg"
    a01 = softsplat(u1, flow10 * u1, metric10, mode)
    a12 = softsplat(u0, flow12 * u0, metric12, mode)
    ones = drm10 * 0 + 1  # drm.py:180
    return {
        "drm_t1_t01": _fill(a01, u1, ones, flow10 * u1, metric10, mode),
        "drm_t1_t12": _fill(a12, u0, ones, flow12 * u0, metric12, mode),
    }


def calc_drm_gmfss(t, flow10, flow12, metric10, metric12, linear=False):
    """models/drm.py:110-155.  No +1e-4 on distances (0/0 -> NaN where both flows are zero);
    the complementary maps (1 - drm) are splatted along the *unscaled* flows."""
    drm10, drm12 = _ratio_maps(flow10, flow12, 0.0)
    mode = "soft" if (metric10 is not None and metric12 is not None) else "avg"
    if linear:
        d1t_01 = drm12 * t * 2
        d1t_12 = drm10 * t * 2
    else:
        d1t_01 = drm_to_t(drm12, t)
        d1t_12 = drm_to_t(drm10, t)
    u01 = 1 - d1t_01
    u12 = 1 - d1t_12
    a01 = softsplat(u01, flow10, metric10, mode)
    a12 = softsplat(u12, flow12, metric12, mode)
    ones = a01 * 0 + 1  # drm.py:135: built from the *splatted* map, shared by both directions
    return {
        "drm0t_t01": _fill(a01, u01, ones, flow10, metric10, mode),
        "drm1t_t01": d1t_01,
        "drm1t_t12": d1t_12,
        "drm2t_t12": _fill(a12, u12, ones, flow12, metric12, mode),
    }
