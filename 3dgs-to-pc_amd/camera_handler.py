"""
Camera set-up for the HIP rasteriser behind the reference's ``camera_handler`` interface (``fov2focal``,
``focal2fov``, ``getProjectionMatrix``, ``Camera``, ``get_camera``; reference: camera_handler.py:8-108).

A view is a handful of 4x4 operations, so everything here is float32 host arithmetic: the results go to the kernels
by value inside ``G2pcCamera`` and a render call neither launches set-up kernels nor reads anything back.

Conventions kept from the reference (they are what the parity tests pin):
  * matrices are stored for ROW vectors: ``world_view_transform = inverse(c2w)^T``, ``projection = P^T``;
  * near / far are fixed at 10 / 100 and only shape the (unused) clip-space z;
  * "python" cameras use c2w as loaded (OpenGL: the camera looks down -z); "cuda"/"hip" cameras negate the y and z
    columns of c2w first (camera_handler.py:75), i.e. look down +z;
  * the render size is ``int(int(w) * s)`` with ``s = colour_resolution / int(w)`` (1 when a mask pins the size).
"""
import math

import torch

_ZNEAR, _ZFAR = 10, 100


def fov2focal(fov, pixels):
    return 0.5 * pixels / math.tan(0.5 * fov)


def focal2fov(focal, pixels):
    return 2.0 * math.atan(0.5 * pixels / focal)


def getProjectionMatrix(znear, zfar, fovX, fovY):
    """Perspective matrix of camera_handler.py:14-33: symmetric frustum, +z forward, w = z."""
    half_w = math.tan(0.5 * fovX) * znear
    half_h = math.tan(0.5 * fovY) * znear
    depth = zfar - znear
    rows = [[znear / half_w, 0.0, 0.0, 0.0],
            [0.0, znear / half_h, 0.0, 0.0],
            [0.0, 0.0, zfar / depth, -(zfar * znear) / depth],
            [0.0, 0.0, 1.0, 0.0]]
    P = torch.tensor(rows, dtype=torch.float32)
    # the reference builds 2n/(r-l) with r = -l = tan*n in float64 python scalars and stores float32: identical values
    P[0, 0] = 2.0 * znear / (half_w - (-half_w))
    P[1, 1] = 2.0 * znear / (half_h - (-half_h))
    return P


def _row_vector_view(c2w):
    """inverse(c2w)^T as float32 on the host."""
    return torch.linalg.inv(c2w).transpose(0, 1).contiguous()


def _host(m):
    return m.detach().to(device="cpu", dtype=torch.float32)


class Camera():
    """The python-renderer camera (camera_handler.py:35-50).  `view` / `projection`: matrices computed for a whole rig at once
    (get_cameras): inverse(c2w)^T of this camera and the shared projection of its intrinsics."""

    def __init__(self, width, height, focal_x, focal_y, c2w, znear=_ZNEAR, zfar=_ZFAR, view=None, projection=None):
        self.c2w = _host(c2w)
        self.znear, self.zfar = znear, zfar
        self.focal_x, self.focal_y = focal_x, focal_y
        self.image_width, self.image_height = int(width), int(height)
        self.FoVx, self.FoVy = focal2fov(focal_x, width), focal2fov(focal_y, height)
        self.world_view_transform = _row_vector_view(self.c2w) if view is None else view
        self.projection_matrix = (getProjectionMatrix(znear, zfar, self.FoVx, self.FoVy).transpose(0, 1)
                                  if projection is None else projection)
        self._derived = None

    # full_proj_transform / camera_center (camera_handler.py:47-50) are not read by any renderer: formed on first use
    def _derive(self):
        if self._derived is None:
            self._derived = (self.world_view_transform @ self.projection_matrix, self.world_view_transform.inverse()[3, :3])
        return self._derived

    @property
    def full_proj_transform(self):
        return self._derive()[0]

    @property
    def camera_center(self):
        return self._derive()[1]


def _render_geometry(cam_intrinsic, colour_resolution, mask):
    w0, h0 = int(cam_intrinsic[0]), int(cam_intrinsic[1])
    scale = 1 if (colour_resolution is None or mask is not None) else colour_resolution / w0
    return int(w0 * scale), int(h0 * scale), float(cam_intrinsic[2]) * scale, float(cam_intrinsic[3]) * scale


def get_camera(renderer_type, transform, cam_intrinsic, colour_resolution=None, sh_degree=3, white_bkgd=True, mask=None):
    """camera_handler.py:53-108 ('hip' is the native spelling of the reference's 'cuda')."""
    if mask is not None:
        if mask.shape[1] != int(cam_intrinsic[0]) or mask.shape[0] != int(cam_intrinsic[1]):
            raise Exception("Size of mask must match size of input image")
        mask = mask.flatten()
    width, height, fx, fy = _render_geometry(cam_intrinsic, colour_resolution, mask)

    if renderer_type == "python":
        return Camera(width, height, fx, fy, transform)

    if renderer_type in ("cuda", "hip"):
        from gaussian_pointcloud_rasterization import GaussianRasterizationSettings
        c2w = _host(transform).clone()
        c2w[:, 1:3].neg_()                                   # OpenGL -> OpenCV camera axes
        fov_x, fov_y = focal2fov(fx, width), focal2fov(fy, height)
        view = _row_vector_view(c2w)
        proj = getProjectionMatrix(_ZNEAR, _ZFAR, fov_x, fov_y).transpose(0, 1)
        background = torch.ones(3) if white_bkgd else torch.zeros(3)
        return GaussianRasterizationSettings(
            image_height=height, image_width=width, tanfovx=math.tan(0.5 * fov_x), tanfovy=math.tan(0.5 * fov_y),
            bg=background, scale_modifier=1.0, viewmatrix=view, projmatrix=view @ proj, sh_degree=sh_degree,
            campos=view.inverse()[3, :3], mask=mask, prefiltered=False, debug=True, antialiasing=False)

    raise Exception(f"Renderer of type {renderer_type} is not supported")


def get_cameras(renderer_type, transforms, intrinsics, colour_resolution=None, sh_degree=3, white_bkgd=True):
    """Camera set-up for a whole rig in one pass (the reference calls get_camera once per image inside its
    render loop, gauss_to_pc.py:437-454): ONE batched inverse of all camera-to-world matrices -- LAPACK factors every 4x4 of
    the batch by itself, so each view matrix is bit for bit the one get_camera computes --, one projection matrix per distinct
    intrinsics.  Returns {name: camera} in the order of `transforms`.  Masked cameras (the mask pins the render size) go
    through get_camera."""
    names = list(transforms)
    if not names:
        return {}
    c2w = torch.stack([_host(torch.as_tensor(transforms[k])) for k in names])              # [C, 4, 4]
    cuda_like = renderer_type in ("cuda", "hip")
    if not cuda_like and renderer_type != "python":
        raise Exception(f"Renderer of type {renderer_type} is not supported")
    if cuda_like:
        c2w = c2w.clone()
        c2w[:, :, 1:3].neg_()                                                               # OpenGL -> OpenCV camera axes
    views = torch.linalg.inv(c2w).transpose(1, 2).contiguous()                              # inverse(c2w)^T per camera
    proj_of, out = {}, {}
    for i, k in enumerate(names):
        width, height, fx, fy = _render_geometry(intrinsics[k], colour_resolution, None)
        key = (width, height, fx, fy)
        if key not in proj_of:
            fov_x, fov_y = focal2fov(fx, width), focal2fov(fy, height)
            proj_of[key] = (getProjectionMatrix(_ZNEAR, _ZFAR, fov_x, fov_y).transpose(0, 1), fov_x, fov_y)
        proj, fov_x, fov_y = proj_of[key]
        if not cuda_like:
            out[k] = Camera(width, height, fx, fy, c2w[i], view=views[i], projection=proj)
        else:
            from gaussian_pointcloud_rasterization import GaussianRasterizationSettings
            view = views[i]
            out[k] = GaussianRasterizationSettings(
                image_height=height, image_width=width, tanfovx=math.tan(0.5 * fov_x), tanfovy=math.tan(0.5 * fov_y),
                bg=torch.ones(3) if white_bkgd else torch.zeros(3), scale_modifier=1.0, viewmatrix=view,
                projmatrix=view @ proj, sh_degree=sh_degree, campos=view.inverse()[3, :3], mask=None, prefiltered=False,
                debug=True, antialiasing=False)
    return out
