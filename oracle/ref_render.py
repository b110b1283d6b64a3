"""
TEST INFRASTRUCTURE ONLY (oracle) -- CPU restatement of the reference's pure-torch renderer
(renderer_type="python"), the parity target of the HIP rasteriser.  Never imported by the product path; only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

Follows (file:line under /root/reference):
    camera_handler.py:8-50     fov/focal, projection matrix, Camera
    camera_handler.py:53-70    get_camera("python"): resolution scale, int() truncation
    gauss_render.py:101-148    build_covariance_2d
    gauss_render.py:151-168    projection_ndc (in_mask = z_view <= -1e-6)
    gauss_render.py:171-193    get_radius (3*ceil(sqrt(lambda_max))), get_rect (clip to the image)
    gauss_render.py:266-402    render(): FIFO quad-tree, strict overlap, sort+flip, dense blend, per-Gaussian
                               max contribution + arg-max pixel, strict-> running update, flipped image
    gauss_render.py:404-465    __call__ (tile limits pinned to render()'s defaults 60 / 60000)
    gauss_render.py:237-264    getters

Written with torch CPU kernels in the reference's operation order (float32, colours float64 as in the
reference's python-renderer path) so that it is bit-comparable with the reference run under oracle/ref_shim.py;
pinned by tests/test_oracle_render.py against tests/golden/render_py_n6000.npz.
"""
import math
from math import ceil, floor

import torch


class Camera:
    def __init__(self, width, height, focal_x, focal_y, c2w, znear=10, zfar=100):
        self.focal_x, self.focal_y = focal_x, focal_y
        self.FoVx = 2 * math.atan(width / (2 * focal_x))
        self.FoVy = 2 * math.atan(height / (2 * focal_y))
        self.image_width, self.image_height = int(width), int(height)
        self.world_view_transform = torch.linalg.inv(c2w).permute(1, 0)
        ty, tx = math.tan(self.FoVy / 2), math.tan(self.FoVx / 2)
        top, right = ty * znear, tx * znear
        P = torch.zeros(4, 4)
        P[0, 0] = 2.0 * znear / (right - (-right))
        P[1, 1] = 2.0 * znear / (top - (-top))
        P[0, 2] = (right + (-right)) / (right - (-right))
        P[1, 2] = (top + (-top)) / (top - (-top))
        P[3, 2] = 1.0
        P[2, 2] = 1.0 * zfar / (zfar - znear)
        P[2, 3] = -(zfar * znear) / (zfar - znear)
        self.projection_matrix = P.transpose(0, 1)


def get_camera(transform, cam_intrinsic, colour_resolution=None):
    diff = 1 if colour_resolution is None else colour_resolution / int(cam_intrinsic[0])
    w = int(int(cam_intrinsic[0]) * diff)
    h = int(int(cam_intrinsic[1]) * diff)
    return Camera(w, h, float(cam_intrinsic[2]) * diff, float(cam_intrinsic[3]) * diff, transform)


def cov2d(mean3d, cov3d, V, cam):
    tfx, tfy = math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5)
    t = (mean3d @ V[:3, :3]) + V[-1:, :3]
    tx = (t[..., 0] / t[..., 2]).clip(min=-tfx * 1.3, max=tfx * 1.3) * t[..., 2]
    ty = (t[..., 1] / t[..., 2]).clip(min=-tfy * 1.3, max=tfy * 1.3) * t[..., 2]
    tz = t[..., 2]
    J = torch.zeros(mean3d.shape[0], 3, 3).to(mean3d)
    J[..., 0, 0] = 1 / tz * cam.focal_x
    J[..., 0, 2] = -tx / (tz * tz) * cam.focal_x
    J[..., 1, 1] = 1 / tz * cam.focal_y
    J[..., 1, 2] = -ty / (tz * tz) * cam.focal_y
    W = V[:3, :3].T
    c = J @ W @ cov3d @ W.T @ J.permute(0, 2, 1)
    return c[:, :2, :2] + (torch.eye(2, 2).to(c) * 0.3)[None]


def quadtree_leaves(width, height, counts_fn, max_tile_size, max_gaussians):
    """FIFO replay of gauss_render.py:290-335; counts_fn(x0, y0, w, h) -> membership mask."""
    queue = [[[0, 0], [width, height]]]
    while queue:
        start, size = queue.pop(0)
        if size[0] <= 1 or size[1] <= 1:
            continue
        size[0] = min(size[0], width - start[1])
        size[1] = min(size[1], height - start[0])
        mask = counts_fn(start[1], start[0], size[0], size[1])
        cnt = int(mask.sum())
        if cnt <= 0:
            yield ("empty", start[1], start[0], size[0], size[1], None)
            continue
        if cnt > max_gaussians or size[0] > max_tile_size or size[1] > max_tile_size:
            size = [ceil(size[0] / 2), ceil(size[1] / 2)]
            s = list(start)
            queue.append([list(s), list(size)])
            s[0] += floor(size[1])
            queue.append([list(s), list(size)])
            s[0] -= floor(size[1])
            s[1] += floor(size[0])
            queue.append([list(s), list(size)])
            s[0] += floor(size[1])
            queue.append([list(s), list(size)])
            continue
        yield ("leaf", start[1], start[0], size[0], size[1], mask)


class PythonRendererOracle:
    def __init__(self, means3D, opacity, colour, cov3d, white_bkgd=True, threshold=0.0,
                 max_tile_size=60, max_gaussians_per_tile=60000):
        n = means3D.shape[0]
        self.means3D, self.opacity, self.colour, self.cov3d = means3D, opacity.float(), colour, cov3d
        self.bg = 1 if white_bkgd else 0
        self.threshold = threshold
        self.max_tile_size, self.max_gaussians = max_tile_size, max_gaussians_per_tile
        self.max_contribution = torch.zeros(n)
        self.colours = torch.zeros((n, 3), dtype=torch.double)

    def get_gaussian_colours(self):
        return self.colours * 255

    def get_visible_gaussians(self):
        return self.max_contribution > self.threshold

    def __call__(self, cam):
        W, H = cam.image_width, cam.image_height
        V = cam.world_view_transform
        pix = torch.stack(torch.meshgrid(torch.arange(W), torch.arange(H), indexing='xy'), dim=-1)
        c2 = cov2d(self.means3D, self.cov3d, V, cam)
        po = torch.cat([self.means3D, torch.ones_like(self.means3D[..., :1])], dim=-1)
        ph = po @ V @ cam.projection_matrix
        pw = 1.0 / (ph[..., -1:] + 0.000001)
        pproj = ph * pw
        pview = po @ V
        in_mask = pview[..., 2] <= -0.000001
        ndc, depths, c2 = pproj[in_mask], pview[in_mask][:, 2], c2[in_mask]
        opacity, colour = self.opacity[in_mask], self.colour[in_mask]
        mx = ((ndc[..., 0] + 1) * W - 1.0) * 0.5
        my = ((ndc[..., 1] + 1) * H - 1.0) * 0.5
        m2 = torch.stack([mx, my], dim=-1)
        det = c2[:, 0, 0] * c2[:, 1, 1] - c2[:, 0, 1] * c2[:, 1, 0]
        mid = 0.5 * (c2[:, 0, 0] + c2[:, 1, 1])
        l1 = mid + torch.sqrt((mid ** 2 - det).clip(min=0.1))
        l2 = mid - torch.sqrt((mid ** 2 - det).clip(min=0.1))
        radii = 3.0 * torch.sqrt(torch.max(l1, l2)).ceil()
        rmin, rmax = m2 - radii[:, None], m2 + radii[:, None]
        for r in (rmin, rmax):
            r[..., 0] = r[..., 0].clip(0, W - 1.0)
            r[..., 1] = r[..., 1].clip(0, H - 1.0)
        image = torch.ones(H, W, 3)
        idx_in = in_mask.nonzero(as_tuple=True)[0]

        def members(x0, y0, w, h):
            tlx, tly = rmin[..., 0].clip(min=x0), rmin[..., 1].clip(min=y0)
            brx, bry = rmax[..., 0].clip(max=x0 + w - 1), rmax[..., 1].clip(max=y0 + h - 1)
            return (brx > tlx) & (bry > tly)

        for kind, x0, y0, w, h, mask in quadtree_leaves(W, H, members, self.max_tile_size, self.max_gaussians):
            if kind == "empty":
                image[y0:y0 + h, x0:x0 + w, :] = self.bg
                continue
            coord = pix[y0:y0 + h, x0:x0 + w].flatten(0, -2)
            # DEPTH TIES: the reference calls torch.sort(depths) with stable=False (gauss_render.py:340); on an AVX-512 host
            # that is a quicksort whose order of EQUAL depths is a property of its partitioning (half of all tied neighbours
            # come out swapped).  Any tie order is a legal execution of the reference; this oracle -- like the 1 M fixture,
            # oracle/make_golden.py::gen_render_big, whose `tie_spread` records how far the as-is execution lands from it --
            # fixes the one deterministic rule: stable, i.e. ties in input order before the flip.
            _, index = torch.sort(depths[mask], stable=True)
            index = torch.flip(index, [0, ])
            inv_index = index.argsort(0)
            sm, sc2 = m2[mask][index], c2[mask][index]
            conic = sc2.inverse()
            so, scol = opacity[mask][index], colour[mask][index]
            dx = coord[:, None, :] - sm[None, :]
            wgt = torch.exp(-0.5 * (dx[:, :, 0] ** 2 * conic[:, 0, 0] + dx[:, :, 1] ** 2 * conic[:, 1, 1]
                                    + dx[:, :, 0] * dx[:, :, 1] * conic[:, 0, 1]
                                    + dx[:, :, 0] * dx[:, :, 1] * conic[:, 1, 0]))
            alpha = (wgt[..., None] * so[None]).clip(max=0.99)
            T = torch.cat([torch.ones_like(alpha[:, :1]), 1 - alpha[:, :-1]], dim=1).cumprod(dim=1)
            acc = (alpha * T).sum(dim=1)
            tile_colour = (T * alpha * scol[None]).sum(dim=1) + (1 - acc) * self.bg
            image[y0:y0 + h, x0:x0 + w] = tile_colour.reshape(h, w, -1).to(image.dtype)
            gids = idx_in[mask]
            contribution = (T * alpha).squeeze(2)[:, inv_index]
            best, best_pix = torch.max(contribution, 0)
            upd = best > self.max_contribution[gids]
            self.max_contribution[gids[upd]] = best[upd]
            self.colours[gids[upd]] = tile_colour[best_pix[upd]].to(torch.double)
        return torch.flip(image, [1, ])
