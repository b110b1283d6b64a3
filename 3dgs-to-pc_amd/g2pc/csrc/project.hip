// Stand-alone forms of the python renderer's per-Gaussian helpers (the reference's public gauss_render API):
//   eval_sh              gauss_render.py:43-99    SH -> RGB, degree 0..4, any number of channels
//   build_covariance_2d  gauss_render.py:101-148  EWA splatting covariance J W S W^T J^T + 0.3 I
//   projection_ndc       gauss_render.py:151-168  homogeneous projection + "in front of the camera" mask
//   get_radius           gauss_render.py:171-180  3 * ceil(sqrt(largest eigenvalue))
//   get_rect             gauss_render.py:183-193  pixel rectangle clipped to the image
// The rasteriser's k_preprocess_py (raster.hip) evaluates the same expressions fused, in the same order; these kernels
// exist so that callers of the reference's helper functions find them behind the same names (gauss_render.py of this
// package) without a torch re-implementation.  All HBM-bound, one Gaussian per lane, AoS rows read as whole rows.
#include "g2pc_internal.h"
#include "py_project.inl"

namespace g2pc {

constexpr int PJ_T = 256;

__constant__ float kC2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                             0.5462742152960396f};
__constant__ float kC3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                             -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
__constant__ float kC4[9] = {2.5033429417967046f, -1.7701307697799304f, 0.9461746957575601f, -0.6690465435572892f,
                             0.10578554691520431f, -0.6690465435572892f, 0.47308734787878004f, -1.7701307697799304f,
                             0.6258357354491761f};

// sh f32[n, channels, coeffs] (the reference indexes sh[..., k] on the LAST axis), dirs f32[n, 3], out f32[n, channels].
// Terms are accumulated left to right exactly as the reference's expression reads (python floats there are doubles that
// torch rounds to f32 when they meet an f32 tensor, so every product below is an f32 product).
__global__ __launch_bounds__(PJ_T) void k_eval_sh(int deg, const float* __restrict__ sh, const float* __restrict__ dirs,
                                                 long n, int channels, int coeffs, float* __restrict__ out) {
    long t = (long)blockIdx.x * PJ_T + threadIdx.x;
    if (t >= n * channels) return;
    const long i = t / channels;
    const float* s = sh + (size_t)t * coeffs;
    float r = 0.28209479177387814f * s[0];
    if (deg > 0) {
        const float x = dirs[3 * i], y = dirs[3 * i + 1], z = dirs[3 * i + 2];
        const float c1 = 0.4886025119029199f;
        r = r - c1 * y * s[1] + c1 * z * s[2] - c1 * x * s[3];
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            r = r + kC2[0] * xy * s[4] + kC2[1] * yz * s[5] + kC2[2] * (2.0f * zz - xx - yy) * s[6] + kC2[3] * xz * s[7] +
                kC2[4] * (xx - yy) * s[8];
            if (deg > 2) {
                r = r + kC3[0] * y * (3.0f * xx - yy) * s[9] + kC3[1] * xy * z * s[10] +
                    kC3[2] * y * (4.0f * zz - xx - yy) * s[11] + kC3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * s[12] +
                    kC3[4] * x * (4.0f * zz - xx - yy) * s[13] + kC3[5] * z * (xx - yy) * s[14] +
                    kC3[6] * x * (xx - 3.0f * yy) * s[15];
                if (deg > 3) {
                    r = r + kC4[0] * xy * (xx - yy) * s[16] + kC4[1] * yz * (3.0f * xx - yy) * s[17] +
                        kC4[2] * xy * (7.0f * zz - 1.0f) * s[18] + kC4[3] * yz * (7.0f * zz - 3.0f) * s[19] +
                        kC4[4] * (zz * (35.0f * zz - 30.0f) + 3.0f) * s[20] + kC4[5] * xz * (7.0f * zz - 3.0f) * s[21] +
                        kC4[6] * (xx - yy) * (7.0f * zz - 1.0f) * s[22] + kC4[7] * xz * (xx - 3.0f * yy) * s[23] +
                        kC4[8] * (xx * (xx - 3.0f * yy) - yy * (3.0f * xx - yy)) * s[24];
                }
            }
        }
    }
    out[t] = r;
}

struct Mat16 { float m[16]; };

// cov2d f32[n,2,2] = (J W S W^T J^T)[:2,:2] + 0.3 I, J W S W^T J^T evaluated left to right (gauss_render.py:144)
__global__ __launch_bounds__(PJ_T) void k_cov2d_py(Mat16 Vm, float lim_x, float lim_y, float focal_x, float focal_y,
                                                  const float* __restrict__ means3D, const float* __restrict__ cov9,
                                                  long n, float* __restrict__ cov2d) {
    long i = (long)blockIdx.x * PJ_T + threadIdx.x;
    if (i >= n) return;
    float pv[4], c[4];
    py_view(Vm.m, means3D[3 * i], means3D[3 * i + 1], means3D[3 * i + 2], pv);
    py_cov2d(Vm.m, pv, lim_x, lim_y, focal_x, focal_y, cov9 + 9 * i, c);
#pragma unroll
    for (int j = 0; j < 4; ++j) cov2d[4 * i + j] = c[j];
}

// p_view = [x,1] V; p_hom = p_view P; p_proj = p_hom / (w + 1e-6); in_mask = p_view.z <= -1e-6
__global__ __launch_bounds__(PJ_T) void k_projection_ndc(Mat16 Vm, Mat16 Pm, const float* __restrict__ points, long n,
                                                        float* __restrict__ p_proj, float* __restrict__ p_view,
                                                        uint8_t* __restrict__ in_mask) {
    long i = (long)blockIdx.x * PJ_T + threadIdx.x;
    if (i >= n) return;
    float pv[4], ph[4];
    py_view(Vm.m, points[3 * i], points[3 * i + 1], points[3 * i + 2], pv);
    py_hom(Pm.m, pv, ph);
    const float pw = 1.0f / (ph[3] + 0.000001f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        p_proj[4 * i + j] = ph[j] * pw;
        p_view[4 * i + j] = pv[j];
    }
    in_mask[i] = pv[2] <= -0.000001f ? 1 : 0;
}

__global__ __launch_bounds__(PJ_T) void k_radius_py(const float* __restrict__ cov2d, long n, float* __restrict__ radius) {
    long i = (long)blockIdx.x * PJ_T + threadIdx.x;
    if (i >= n) return;
    const float c[4] = {cov2d[4 * i], cov2d[4 * i + 1], cov2d[4 * i + 2], cov2d[4 * i + 3]};
    float det;
    radius[i] = py_radius(c, det);
}

__global__ __launch_bounds__(PJ_T) void k_rect_py(const float* __restrict__ pix, const float* __restrict__ radii, long n,
                                                 float wmax, float hmax, float* __restrict__ rect_min,
                                                 float* __restrict__ rect_max) {
    long i = (long)blockIdx.x * PJ_T + threadIdx.x;
    if (i >= n) return;
    const float mx = pix[2 * i], my = pix[2 * i + 1], r = radii[i];
    rect_min[2 * i + 0] = fminf(fmaxf(mx - r, 0.0f), wmax);
    rect_min[2 * i + 1] = fminf(fmaxf(my - r, 0.0f), hmax);
    rect_max[2 * i + 0] = fminf(fmaxf(mx + r, 0.0f), wmax);
    rect_max[2 * i + 1] = fminf(fmaxf(my + r, 0.0f), hmax);
}

}  // namespace g2pc

extern "C" {

int g2pc_eval_sh(int32_t deg, const float* sh, const float* dirs, int64_t n, int32_t channels, int32_t coeffs, float* out,
                 void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(deg >= 0 && deg <= 4, G2PC_ERR_ARG, "SH degree must be in [0, 4]");          // gauss_render.py:56
    G2PC_REQUIRE(coeffs >= (deg + 1) * (deg + 1), G2PC_ERR_ARG, "too few SH coefficients");     // gauss_render.py:58
    G2PC_REQUIRE(n >= 0 && channels > 0, G2PC_ERR_ARG, "bad sizes");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(sh && out && (deg == 0 || dirs), G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_eval_sh, dim3(cdiv(n * channels, PJ_T)), dim3(PJ_T), 0, (hipStream_t)stream, (int)deg, sh, dirs,
                       (long)n, (int)channels, (int)coeffs, out);
    return check_launch("g2pc_eval_sh");
}

int g2pc_build_covariance_2d(const float* means3D, const float* cov9, int64_t n, const float* viewmatrix, float lim_x,
                             float lim_y, float focal_x, float focal_y, float* cov2d, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(means3D && cov9 && viewmatrix && cov2d, G2PC_ERR_ARG, "null pointer");
    Mat16 V;
    for (int i = 0; i < 16; ++i) V.m[i] = viewmatrix[i];
    hipLaunchKernelGGL(k_cov2d_py, dim3(cdiv(n, PJ_T)), dim3(PJ_T), 0, (hipStream_t)stream, V, lim_x, lim_y, focal_x,
                       focal_y, means3D, cov9, (long)n, cov2d);
    return check_launch("g2pc_build_covariance_2d");
}

int g2pc_projection_ndc(const float* points, int64_t n, const float* viewmatrix, const float* projmatrix, float* p_proj,
                        float* p_view, uint8_t* in_mask, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(points && viewmatrix && projmatrix && p_proj && p_view && in_mask, G2PC_ERR_ARG, "null pointer");
    Mat16 V, P;
    for (int i = 0; i < 16; ++i) { V.m[i] = viewmatrix[i]; P.m[i] = projmatrix[i]; }
    hipLaunchKernelGGL(k_projection_ndc, dim3(cdiv(n, PJ_T)), dim3(PJ_T), 0, (hipStream_t)stream, V, P, points, (long)n, p_proj,
                       p_view, in_mask);
    return check_launch("g2pc_projection_ndc");
}

int g2pc_get_radius(const float* cov2d, int64_t n, float* radius, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(cov2d && radius, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_radius_py, dim3(cdiv(n, PJ_T)), dim3(PJ_T), 0, (hipStream_t)stream, cov2d, (long)n, radius);
    return check_launch("g2pc_get_radius");
}

int g2pc_get_rect(const float* pix_coord, const float* radii, int64_t n, float width, float height, float* rect_min,
                  float* rect_max, void* stream) {
    using namespace g2pc;
    G2PC_REQUIRE(n >= 0, G2PC_ERR_ARG, "negative n");
    if (n == 0) return G2PC_OK;
    G2PC_REQUIRE(pix_coord && radii && rect_min && rect_max, G2PC_ERR_ARG, "null pointer");
    hipLaunchKernelGGL(k_rect_py, dim3(cdiv(n, PJ_T)), dim3(PJ_T), 0, (hipStream_t)stream, pix_coord, radii, (long)n,
                       width - 1.0f, height - 1.0f, rect_min, rect_max);
    return check_launch("g2pc_get_rect");
}
}
