// Per-Gaussian arithmetic of the reference's PYTHON renderer (gauss_render.py:101-193, 404-437), shared by the
// rasteriser's k_preprocess_py (raster.hip) and the stand-alone helper kernels (project.hip), evaluated in the order
// torch's CPU build evaluates it -- found by bisecting the candidate orders against the untouched reference on 200 k
// Gaussians until every cov2d entry, p_view and p_hom agreed to the last bit (tools/torch_order_probe.py):
//   * a matmul with ONE shared right-hand matrix -- `mean3d @ V[:3,:3]`, `points_o @ V @ P`, `J @ W`, `... @ W.T`
//     (gauss_render.py:125,144,161; torch folds [N,3,3] @ [3,3] into one [3N,3] x [3,3] product) -- is an MKL sgemm: its
//     dot products run k = 0, 1, 2, ... as ONE fused chain, fma(a_k, b_k, acc), first product rounded on its own;
//   * a matmul of two BATCHED 3x3 operands -- `(J W) @ cov3d`, `... @ J^T` -- is ATen's small-matrix bmm kernel: plain
//     `acc += a_k * b_k`, every product and every sum rounded;
//   * everything else is elementwise torch arithmetic: one rounding per operation, python floats meet f32 tensors as f32.
// The library is compiled with -ffp-contract=off; the fused steps are spelled __builtin_fmaf.  What these values decide
// is INTEGER downstream (radius -> pixel rectangle -> strict overlap with tile edges, gauss_render.py:308-310): one ulp in
// a projected mean moves a Gaussian into or out of a tile and changes that tile's pixels by a whole term.
#pragma once

namespace g2pc {

__device__ __forceinline__ float mkl_dot3(float a0, float b0, float a1, float b1, float a2, float b2) {
    return __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0));
}
__device__ __forceinline__ float mkl_dot4(float a0, float b0, float a1, float b1, float a2, float b2, float a3, float b3) {
    return __builtin_fmaf(a3, b3, __builtin_fmaf(a2, b2, __builtin_fmaf(a1, b1, a0 * b0)));
}

// p_view = [x, y, z, 1] @ V (gauss_render.py:163); its first three components are also t = mean @ V[:3,:3] + V[3,:3] (:125):
// the last step of the 4-chain, fma(1, V[12+j], acc), and the separate addition of :125 are the same single rounding.
__device__ __forceinline__ void py_view(const float* V, float x, float y, float z, float pv[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) pv[j] = mkl_dot3(x, V[0 + j], y, V[4 + j], z, V[8 + j]) + V[12 + j];
}

// p_hom = p_view @ P (:161)
__device__ __forceinline__ void py_hom(const float* P, const float pv[4], float ph[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) ph[j] = mkl_dot4(pv[0], P[0 + j], pv[1], P[4 + j], pv[2], P[8 + j], pv[3], P[12 + j]);
}

// build_covariance_2d (:101-148): cov2d = (J W S W^T J^T)[:2,:2] + 0.3 I, left to right.  lim_x / lim_y = 1.3 tan(fov / 2)
// as the reference forms them -- in double, from python floats -- rounded to f32 when they meet the tensor (:128-129).
// S = the 3x3 covariance, row-major.  c = (c00, c01, c10, c11).
__device__ __forceinline__ void py_cov2d(const float* V, const float t[3], float lim_x, float lim_y, float focal_x,
                                         float focal_y, const float* S, float c[4]) {
    float qx = t[0] / t[2], qy = t[1] / t[2];
    qx = qx < -lim_x ? -lim_x : (qx > lim_x ? lim_x : qx);
    qy = qy < -lim_y ? -lim_y : (qy > lim_y ? lim_y : qy);
    const float tx = qx * t[2], ty = qy * t[2], tz = t[2];
    const float j00 = 1.0f / tz * focal_x, j02 = -tx / (tz * tz) * focal_x;          // J (:134-138)
    const float j11 = 1.0f / tz * focal_y, j12 = -ty / (tz * tz) * focal_y;
    // J @ W (sgemm; W[k][c] = V[c][k]).  Row 0 of J is (j00, 0, j02): fma(0, W1c, acc) = acc;
    // row 1 is (0, j11, j12): the chain starts from 0 * W0c = 0, so j11 * W1c is rounded on its own.
    float M0[3], M1[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        M0[k] = __builtin_fmaf(j02, V[4 * k + 2], j00 * V[4 * k + 0]);
        M1[k] = __builtin_fmaf(j12, V[4 * k + 2], j11 * V[4 * k + 1]);
    }
    float A0[3], A1[3];                                                               // (J W) @ Sigma  (bmm: plain)
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        A0[k] = M0[0] * S[0 + k] + M0[1] * S[3 + k] + M0[2] * S[6 + k];
        A1[k] = M1[0] * S[0 + k] + M1[1] * S[3 + k] + M1[2] * S[6 + k];
    }
    float B0[3], B1[3];                                                               // ... @ W^T  (sgemm; W^T[k][c] = V[k][c])
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        B0[k] = mkl_dot3(A0[0], V[0 + k], A0[1], V[4 + k], A0[2], V[8 + k]);
        B1[k] = mkl_dot3(A1[0], V[0 + k], A1[1], V[4 + k], A1[2], V[8 + k]);
    }
    // ... @ J^T (bmm: plain; the products with J's zeros add +0) and the 0.3 I low-pass filter (:147-148)
    c[0] = B0[0] * j00 + B0[2] * j02 + 0.3f;
    c[1] = B0[1] * j11 + B0[2] * j12;
    c[2] = B1[0] * j00 + B1[2] * j02;
    c[3] = B1[1] * j11 + B1[2] * j12 + 0.3f;
}

// get_radius (:171-180)
__device__ __forceinline__ float py_radius(const float c[4], float& det_out) {
    const float det = c[0] * c[3] - c[1] * c[2];
    const float mid = 0.5f * (c[0] + c[3]);
    float disc = mid * mid - det;
    disc = disc < 0.1f ? 0.1f : disc;                   // NaN stays NaN, as torch.clip does
    const float sq = sqrtf(disc);
    const float l1 = mid + sq, l2 = mid - sq;
    det_out = det;
    return 3.0f * ceilf(sqrtf(l1 > l2 ? l1 : l2));
}

}  // namespace g2pc
