"""Drop-in mirror of the reference's gauss_render.py -- rasteriser façade (filled in with the HIP renderer)."""


def get_renderer(renderer_type, xyz, opacities, colours, covariances, shs=None, visible_gaussian_threshold=0.0,
                 surface_distance_std=None, calculate_surface_distance=False):
    raise NotImplementedError("HIP rasteriser not built yet")
