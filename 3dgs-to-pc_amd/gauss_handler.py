"""
Drop-in mirror of the reference's ``gauss_handler.py`` (same public names, arguments and results) whose
arithmetic runs in libg2pc.so on the MI355X.  Reference lines are cited per function
(``gauss_handler.py:NN`` = /root/reference/gauss_handler.py).

Differences that are deliberate and documented in DESIGN.md:
  * device-agnostic (tensors stay on the device they arrive on; the reference hard-codes "cuda");
  * eigen tests use closed-form fp64 eigenvalues instead of LAPACK's general ``eigvals``;
  * ``cull_large_gaussians`` implements the documented intent (keep the smallest fraction); the
    reference's version ANDs a bool mask with an index tensor (gauss_handler.py:248-250) and cannot run.
"""
from math import floor

import torch

from g2pc import ops


def strip_lowerdiag(L):
    """gauss_handler.py:12-21 -- (xx, xy, xz, yy, yz, zz) of a batch of 3x3 matrices (pure data movement)."""
    idx = torch.tensor([0, 1, 2, 4, 5, 8], device=L.device)
    return L.reshape(L.shape[0], 9).index_select(1, idx).to(torch.float)


def strip_symmetric(sym):
    """gauss_handler.py:23-24."""
    return strip_lowerdiag(sym)


def build_rotation(q):
    """gauss_handler.py:26-47 -- quaternion (r,x,y,z), used as given (no normalisation)."""
    zeros = torch.zeros((q.shape[0], 3), dtype=torch.float32, device=q.device)
    return ops.build_covariances(zeros, q, 1.0, want_rotmat=True)[3]


def build_scaling_rotation(s, r):
    """gauss_handler.py:49-58 -- L = R diag(exp(s)); the reference's scales are log-space."""
    R = build_rotation(r)
    return R * torch.exp(s.to(torch.float32)).unsqueeze(1)


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """gauss_handler.py:60-63 -- Sigma = L L^T in one fused HIP pass."""
    return ops.build_covariances(scaling, rotation, scaling_modifier)[0]


class _TensorKey:
    def __init__(self, t):
        import weakref
        self.ref, self.version = weakref.ref(t), t._version

    def __eq__(self, other):
        return isinstance(other, _TensorKey) and self.ref() is not None and self.ref() is other.ref() and self.version == other.version

    def __ne__(self, other):
        return not self.__eq__(other)


class Gaussians():
    """
    Manages all loaded gaussians in the renderer (gauss_handler.py:65-279)
    """

    def __init__(self, xyz, scales, rots, colours, opacities, shs=None):
        self.xyz = xyz
        self.scales = scales
        self.rots = rots
        self.opacities = opacities
        self.colours = colours
        self.shs = shs
        self.normals = None

        self.scaling_modifier = 1.0

        # 3D covariance matrices and (same pass, kept for calculate_normals) the normals
        self.covariances, _, self._normals_from_build = ops.build_covariances(
            scales, rots, self.scaling_modifier, want_normals=True)

        self.set_default_filter()

    def set_default_filter(self):
        self._filter = None                       # "keep everything": materialised on first use (filter_indices)

    @property
    def filter_indices(self):
        if self._filter is None:
            self._filter = torch.full((self.xyz.shape[0],), True, dtype=torch.bool, device=self.xyz.device)
        return self._filter

    @filter_indices.setter
    def filter_indices(self, value):
        self._filter = value

    def calculate_normals(self):
        """gauss_handler.py:89-106 -- the axis of the smallest scale, rotated by R."""
        if self._normals_from_build is None or self._normals_from_build.shape[0] != self.xyz.shape[0]:
            self._normals_from_build = ops.build_covariances(self.scales, self.rots, 1.0, want_normals=True)[2]
        self.normals = self._normals_from_build

    def non_posdef_covariances(self, covariances, epsilon: float = 1e-10):
        """gauss_handler.py:108-112 -- mask of matrices with an eigenvalue <= epsilon."""
        probe = covariances.to(torch.float32).contiguous().clone()
        keep = ops.validate_covariances_(probe, regularise=False, eps=epsilon, min_eps=epsilon, iters=0)
        return ~keep

    def clamp_covariances(self, covariances, mask=None, epsilon=1e-6):
        """gauss_handler.py:114-127 -- clamp eigenvalues to >= epsilon (one round), in place on the masked rows."""
        work = covariances.to(torch.float32).contiguous().clone()
        ops.validate_covariances_(work, regularise=False, eps=epsilon, min_eps=epsilon, iters=1)
        if mask is None:
            covariances[:] = work
        else:
            covariances[mask] = work[mask]
        return covariances

    def regularise_covariances(self, covariances, mask=None, epsilon=5e-7):
        """gauss_handler.py:129-140 -- Sigma += epsilon * I."""
        eye = epsilon * torch.eye(3, device=covariances.device, dtype=covariances.dtype)
        if mask is None:
            covariances += eye
        else:
            covariances[mask] += eye
        return covariances

    def validate_covariances(self, regularise=True, epsilon=1e-7, min_ps_epsilon=1e-8, num_clamp_iters=3, defer_cull=False):
        """gauss_handler.py:142-166 -- one fused kernel: regularise, up to num_clamp_iters clamp rounds,
        final test; culls what is still not positive definite and returns the keep mask."""
        cov = self.covariances.to(torch.float32).contiguous()
        keep, culled, area = ops.validate_covariances_(cov, regularise=regularise, reg_eps=5e-7, eps=epsilon,
                                                       min_eps=min_ps_epsilon, iters=num_clamp_iters, want_count=True,
                                                       want_area=True, defer_count=defer_cull)
        self.covariances = cov
        # sqrt(ellipsoid area) of the validated matrices, from the pass's own eigenvalues: get_gaussian_magnitudes multiplies
        # instead of decomposing again -- valid while self.covariances is this very tensor, unmodified
        self._sqrt_area, self._sqrt_area_of = area, self._cov_key()
        if defer_cull:
            # the culled count stays on the device: the caller carries on as if nothing was culled (the common case) and asks
            # resolve_deferred_cull() once its work is queued -- the host does not stall in the middle of the job
            # (the count travels to pinned host memory behind the validation on the same stream: whoever synchronises that
            # stream later -- the sampler does, for its point count -- finds it there without another round trip)
            host = ops._pinned_i64(cov.device, 1) if cov.device.type == "cuda" else None
            if host is not None:
                host.zero_()
                host.view(torch.int32)[:1].copy_(culled, non_blocking=True)
            self._deferred_cull = (keep, culled, host)
            self.last_validate_culled = False
            return keep
        self.last_validate_culled = culled > 0
        if self.last_validate_culled:
            self.add_gaussians_to_cull(keep)
            self.filter_gaussians()
        return keep

    def resolve_deferred_cull(self):
        """After validate_covariances(defer_cull=True): were rows culled after all?  (One 4-byte read-back, meant to be asked
        when the device has caught up anyway.)  If so the cull and the filter are applied now and True is returned -- whatever was
        computed from the unfiltered Gaussians in between has to be computed again."""
        pending, self._deferred_cull = getattr(self, "_deferred_cull", None), None
        if pending is None:
            return False
        keep, count, host = pending
        if host is not None:
            torch.cuda.current_stream(count.device).synchronize()       # (normally idle already: the sampler has just waited for it)
            culled = int(host.view(torch.int32)[0])
        else:
            culled = int(count.item())
        if culled == 0:
            return False
        self.last_validate_culled = True
        self.add_gaussians_to_cull(keep)
        self.filter_gaussians()
        return True

    def _cov_key(self):
        """Identity of the covariance tensor the kept sqrt(area) belongs to: the tensor OBJECT (held weakly) and its version
        counter -- a reassigned or in-place modified self.covariances no longer matches."""
        return _TensorKey(self.covariances)

    def add_gaussians_to_cull(self, indices_to_cull):
        self.filter_indices = indices_to_cull.clone() if self._filter is None else self._filter & indices_to_cull

    def filter_gaussians(self):
        """gauss_handler.py:171-193 -- stream compaction (scan + row gathers in HIP)."""
        filter_indices = torch.clone(self.filter_indices)
        index = ops.compact_index(filter_indices)
        self.last_filter_index = index        # int32 positions of the survivors: select(per_gaussian_tensor) reuses it

        # every per-Gaussian array through ONE gather launch (they share the index)
        area = getattr(self, "_sqrt_area", None)
        if area is not None and getattr(self, "_sqrt_area_of", None) != self._cov_key():
            area = None                                   # the covariances changed since the validation that kept it
        (self.xyz, self.scales, self.rots, self.colours, self.opacities, self.covariances, self.shs, self.normals) = \
            ops.gather_rows_multi([self.xyz, self.scales, self.rots, self.colours, self.opacities, self.covariances, self.shs,
                                   self.normals], index)
        self._normals_from_build = self.normals
        self._sqrt_area = ops.gather_rows(area, index) if area is not None else None
        self._sqrt_area_of = self._cov_key() if area is not None else None

        self.set_default_filter()

        return filter_indices

    def select(self, per_gaussian):
        """per_gaussian[mask of the LAST filter_gaussians()] as a row gather on the compaction index that filter already
        built (what the reference writes as tensor[culled_indices], gauss_to_pc.py:503,513)."""
        return ops.gather_rows(per_gaussian.contiguous(), self.last_filter_index)

    def apply_min_opacity(self, min_opacity):
        """gauss_handler.py:195-204."""
        if min_opacity > 0.0:
            m = self.filter_indices.to(torch.uint8)
            ops.cull_mask_(m, None, self.opacities, min_opacity, None, None)
            self.filter_indices = m.to(torch.bool)

    def apply_bounding_box(self, bounding_box_min, bounding_box_max):
        """gauss_handler.py:206-224 -- strict inequalities on xyz."""
        if bounding_box_min is None and bounding_box_max is None:
            return
        m = self.filter_indices.to(torch.uint8)
        ops.cull_mask_(m, self.xyz, None, None, bounding_box_min, bounding_box_max)
        self.filter_indices = m.to(torch.bool)

    def cull_large_gaussians(self, cull_gauss_size_percent):
        """gauss_handler.py:235-250.  The reference's line `filter_indices & culled_gaussians` ANDs a bool mask with int64
        INDICES (a shape / dtype error as written); the intent -- keep the floor(n * (1 - p)) smallest Gaussians by
        get_gaussian_magnitudes(), ties in index order -- is what runs here, ranked by the library's stable radix sort."""
        if cull_gauss_size_percent > 0.0:
            gaussian_sizes = self.get_gaussian_magnitudes()
            cull_index = floor(gaussian_sizes.shape[0] * (1 - cull_gauss_size_percent))
            order = ops.argsort_f64_nonnegative(gaussian_sizes)
            keep = torch.zeros((gaussian_sizes.shape[0],), dtype=torch.uint8, device=gaussian_sizes.device)
            ops.scatter_ones_u8(keep, order[:cull_index])
            self.filter_indices = self.filter_indices & keep.to(torch.bool)

    def get_gaussian_magnitudes(self, contributions=None):
        """gauss_handler.py:252-279 -- sqrt(ellipsoid area) x (contributions or opacities), float64."""
        if contributions is None:
            contributions = self.opacities
        area = getattr(self, "_sqrt_area", None)
        if area is not None and getattr(self, "_sqrt_area_of", None) == self._cov_key():
            return ops.gaussian_magnitudes_from_area(area, contributions)      # validate_covariances kept sqrt(area)
        return ops.gaussian_magnitudes(self.covariances, contributions)
