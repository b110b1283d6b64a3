/*
 * TEST INFRASTRUCTURE ONLY (oracle) -- plain-C restatement of the reference's native rasteriser
 * (gaussian-pointcloud-rasterization/cuda_rasterizer), i.e. renderer_type="cuda".  Never linked into the product.
 *
 * PARITY UNPINNED: the reference's CUDA sources cannot be compiled or run here (no nvcc, no NVIDIA GPU) and the
 * reference ships no test vectors, so this file is a restatement checked only against the CUDA *source text*:
 *   forward.cu:22-73     computeColorFromSH           forward.cu:76-111  computeCov2D (glm column-major)
 *   forward.cu:153-271   preprocessCUDA               auxiliary.h:40-55  ndc2Pix (double arithmetic), getRect
 *   auxiliary.h:151-176  in_frustum (z_view <= 0.2)    rasterizer_impl.cu:69-110 duplicateWithKeys
 *   rasterizer_impl.cu:311-316 stable radix sort on (tile << 32 | depth bits)
 *   forward.cu:303-497   renderCUDA (batches of 256, power>0 skip, alpha<1/255 skip, T(1-a)<1e-4 stop, per-Gaussian
 *                        max contribution + pixel, surface distance against the per-batch expected depth)
 * It implements the DETERMINISTIC SPEC of SURVEY.md §8(a.5), which resolves the data races / UB of the CUDA kernel
 * (SURVEY §2.2 defects 1-4): ties of the per-Gaussian maximum go to the lowest pixel id; masked pixels take no
 * part in anything; out-of-image threads of partial tiles take part in the surface-distance minimum with an
 * expected depth of 0; a Gaussian whose 256-batch is never reached keeps FLT_MAX / 0.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/_build/libcuda_raster_ref.so oracle/cuda_raster_ref.c -lm
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define BX 16
#define BY 16
#define BLOCK 256

static const float SH_C0 = 0.28209479177387814f, SH_C1 = 0.4886025119029199f;
static const float SH_C2[] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f,
                              0.5462742152960396f};
static const float SH_C3[] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                              -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};

typedef struct { uint64_t key; uint32_t val; } Inst;
static int cmp_inst(const void* a, const void* b) {
    const Inst *x = (const Inst*)a, *y = (const Inst*)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->val < y->val ? -1 : (x->val > y->val ? 1 : 0);     /* stable LSD radix == ties by input order (index) */
}

static void color_from_sh(int idx, int deg, int M, const float* means, const float* campos, const float* shs, float out[3]) {
    float dx = means[3 * idx] - campos[0], dy = means[3 * idx + 1] - campos[1], dz = means[3 * idx + 2] - campos[2];
    float len = sqrtf(dx * dx + dy * dy + dz * dz);
    float x = dx / len, y = dy / len, z = dz / len;
    const float* sh = shs + (size_t)idx * M * 3;
    for (int c = 0; c < 3; ++c) {
        float r = SH_C0 * sh[c];
        if (deg > 0) {
            r = r - SH_C1 * y * sh[3 + c] + SH_C1 * z * sh[6 + c] - SH_C1 * x * sh[9 + c];
            if (deg > 1) {
                float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
                r = r + SH_C2[0] * xy * sh[12 + c] + SH_C2[1] * yz * sh[15 + c] + SH_C2[2] * (2.0f * zz - xx - yy) * sh[18 + c] +
                    SH_C2[3] * xz * sh[21 + c] + SH_C2[4] * (xx - yy) * sh[24 + c];
                if (deg > 2) {
                    r = r + SH_C3[0] * y * (3.0f * xx - yy) * sh[27 + c] + SH_C3[1] * xy * z * sh[30 + c] +
                        SH_C3[2] * y * (4.0f * zz - xx - yy) * sh[33 + c] +
                        SH_C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy) * sh[36 + c] +
                        SH_C3[4] * x * (4.0f * zz - xx - yy) * sh[39 + c] + SH_C3[5] * z * (xx - yy) * sh[42 + c] +
                        SH_C3[6] * x * (xx - 3.0f * yy) * sh[45 + c];
                }
            }
        }
        r += 0.5f;
        out[c] = r < 0.0f ? 0.0f : r;
    }
}

/* cov2D = (W J)^T Vrk^T (W J) with glm's column-major constructors (forward.cu:91-108) */
static void cov2d(const float m[3], float fx, float fy, float tfx, float tfy, const float* c6, const float* V, float out[3]) {
    float tx = V[0] * m[0] + V[4] * m[1] + V[8] * m[2] + V[12];
    float ty = V[1] * m[0] + V[5] * m[1] + V[9] * m[2] + V[13];
    float tz = V[2] * m[0] + V[6] * m[1] + V[10] * m[2] + V[14];
    float limx = 1.3f * tfx, limy = 1.3f * tfy;
    float txtz = tx / tz, tytz = ty / tz;
    tx = fminf(limx, fmaxf(-limx, txtz)) * tz;
    ty = fminf(limy, fmaxf(-limy, tytz)) * tz;
    /* math-convention matrices: glm mat3(a,b,c,d,e,f,g,h,i) has COLUMNS (a,b,c),(d,e,f),(g,h,i) */
    float J[3][3] = {{fx / tz, 0.0f, 0.0f}, {0.0f, fy / tz, 0.0f}, {-(fx * tx) / (tz * tz), -(fy * ty) / (tz * tz), 0.0f}};  /* J[row][col] */
    float Wm[3][3] = {{V[0], V[1], V[2]}, {V[4], V[5], V[6]}, {V[8], V[9], V[10]}};
    float T[3][3], Vrk[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[r][c] = Wm[r][0] * J[0][c] + Wm[r][1] * J[1][c] + Wm[r][2] * J[2][c];
    float A[3][3];  /* T^T Vrk^T */
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = T[0][r] * Vrk[c][0] + T[1][r] * Vrk[c][1] + T[2][r] * Vrk[c][2];
    float C[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r][c] = A[r][0] * T[0][c] + A[r][1] * T[1][c] + A[r][2] * T[2][c];
    /* glm cov[0][0], cov[0][1], cov[1][1] (column-major indexing; the matrix is symmetric) */
    out[0] = C[0][0]; out[1] = C[1][0]; out[2] = C[1][1];
}

static float ndc2pix(float v, int S) { return (float)(((v + 1.0) * S - 1.0) * 0.5); }

static void get_rect(float px, float py, int r, int gx, int gy, int rmin[2], int rmax[2]) {
    int a;
    a = (int)((px - r) / BX); rmin[0] = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py - r) / BY); rmin[1] = a < 0 ? 0 : (a > gy ? gy : a);
    a = (int)((px + r + BX - 1) / BX); rmax[0] = a < 0 ? 0 : (a > gx ? gx : a);
    a = (int)((py + r + BY - 1) / BY); rmax[1] = a < 0 ? 0 : (a > gy ? gy : a);
}

int cuda_ref_forward(int P, int D, int M, const float* bg, const float* means3D, const float* colors_precomp,
                     const float* opacities, const float* cov3D_precomp, const float* viewmatrix,
                     const float* projmatrix, float tan_fovx, float tan_fovy, int H, int W, const float* sh,
                     const float* campos, const int* mask, int calc_surf, float* out_color, float* out_depth,
                     float* out_invdepth, int* radii, float* gauss_contrib, float* gauss_surf, int* gauss_pixels) {
    const float focal_y = H / (2.0f * tan_fovy), focal_x = W / (2.0f * tan_fovx);
    const int gx = (W + BX - 1) / BX, gy = (H + BY - 1) / BY;
    float* depths = (float*)calloc(P, sizeof(float));
    float* xy = (float*)calloc(2 * (size_t)P, sizeof(float));
    float* conic_o = (float*)calloc(4 * (size_t)P, sizeof(float));
    float* rgb = (float*)calloc(3 * (size_t)P, sizeof(float));
    uint32_t* touched = (uint32_t*)calloc(P, sizeof(uint32_t));
    for (size_t i = 0; i < (size_t)3 * H * W; ++i) out_color[i] = 0.0f;
    for (size_t i = 0; i < (size_t)H * W; ++i) { out_depth[i] = 0.0f; out_invdepth[i] = 0.0f; }
    size_t L = 0;
    for (int i = 0; i < P; ++i) {
        radii[i] = 0; gauss_contrib[i] = 0.0f; gauss_surf[i] = FLT_MAX; gauss_pixels[i] = 0;
        const float* m = means3D + 3 * i;
        float pvz = viewmatrix[2] * m[0] + viewmatrix[6] * m[1] + viewmatrix[10] * m[2] + viewmatrix[14];
        if (pvz <= 0.2f) continue;
        float hx = projmatrix[0] * m[0] + projmatrix[4] * m[1] + projmatrix[8] * m[2] + projmatrix[12];
        float hy = projmatrix[1] * m[0] + projmatrix[5] * m[1] + projmatrix[9] * m[2] + projmatrix[13];
        float hw = projmatrix[3] * m[0] + projmatrix[7] * m[1] + projmatrix[11] * m[2] + projmatrix[15];
        float pw = 1.0f / (hw + 0.0000001f);
        float cov[3];
        cov2d(m, focal_x, focal_y, tan_fovx, tan_fovy, cov3D_precomp + 6 * (size_t)i, viewmatrix, cov);
        cov[0] += 0.3f; cov[2] += 0.3f;
        float det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0.0f) continue;
        float di = 1.f / det;
        float cx = cov[2] * di, cy = -cov[1] * di, cz = cov[0] * di;
        float mid = 0.5f * (cov[0] + cov[2]);
        float l1 = mid + sqrtf(fmaxf(0.1f, mid * mid - det)), l2 = mid - sqrtf(fmaxf(0.1f, mid * mid - det));
        float my_radius = ceilf(3.f * sqrtf(fmaxf(l1, l2)));
        float px = ndc2pix(hx * pw, W), py = ndc2pix(hy * pw, H);
        int rmin[2], rmax[2];
        get_rect(px, py, (int)my_radius, gx, gy, rmin, rmax);
        if ((rmax[0] - rmin[0]) * (rmax[1] - rmin[1]) == 0) continue;
        if (colors_precomp) { rgb[3 * i] = colors_precomp[3 * i]; rgb[3 * i + 1] = colors_precomp[3 * i + 1]; rgb[3 * i + 2] = colors_precomp[3 * i + 2]; }
        else color_from_sh(i, D, M, means3D, campos, sh, rgb + 3 * i);
        depths[i] = pvz; radii[i] = (int)my_radius; xy[2 * i] = px; xy[2 * i + 1] = py;
        conic_o[4 * i] = cx; conic_o[4 * i + 1] = cy; conic_o[4 * i + 2] = cz; conic_o[4 * i + 3] = opacities[i];
        touched[i] = (uint32_t)((rmax[1] - rmin[1]) * (rmax[0] - rmin[0]));
        L += touched[i];
    }
    Inst* inst = (Inst*)malloc((L + 1) * sizeof(Inst));
    size_t off = 0;
    for (int i = 0; i < P; ++i) {
        if (radii[i] <= 0) continue;
        int rmin[2], rmax[2];
        get_rect(xy[2 * i], xy[2 * i + 1], radii[i], gx, gy, rmin, rmax);
        uint32_t dbits; memcpy(&dbits, &depths[i], 4);
        for (int y = rmin[1]; y < rmax[1]; ++y)
            for (int x = rmin[0]; x < rmax[0]; ++x) {
                inst[off].key = ((uint64_t)(y * gx + x) << 32) | dbits;
                inst[off].val = (uint32_t)i;
                ++off;
            }
    }
    qsort(inst, L, sizeof(Inst), cmp_inst);
    size_t* tstart = (size_t*)calloc((size_t)gx * gy + 1, sizeof(size_t));
    for (size_t l = 0; l < L; ++l) tstart[(inst[l].key >> 32) + 1]++;
    for (int t = 0; t < gx * gy; ++t) tstart[t + 1] += tstart[t];

    float T[BLOCK], E[BLOCK], Ei[BLOCK], C[BLOCK][3];
    int done[BLOCK], inside[BLOCK], masked[BLOCK];
    float largest[BLOCK], smallest[BLOCK];
    int largest_pix[BLOCK];
    for (int ty = 0; ty < gy; ++ty)
        for (int tx = 0; tx < gx; ++tx) {
            int tile = ty * gx + tx;
            for (int t = 0; t < BLOCK; ++t) {
                int x = tx * BX + t % BX, y = ty * BY + t / BX;
                inside[t] = (x < W && y < H);
                masked[t] = inside[t] ? (mask[(size_t)W * y + x] == 0) : 0;
                done[t] = !inside[t];
                T[t] = 1.0f; E[t] = 0.0f; Ei[t] = 0.0f; C[t][0] = C[t][1] = C[t][2] = 0.0f;
            }
            size_t s = tstart[tile], e = tstart[tile + 1];
            for (size_t b = s; b < e; b += BLOCK) {
                int ndone = 0;
                for (int t = 0; t < BLOCK; ++t) ndone += (done[t] || masked[t]);      /* masked threads have left the loop */
                if (ndone == BLOCK) break;
                int cnt = (int)((e - b) < BLOCK ? (e - b) : BLOCK);
                for (int j = 0; j < cnt; ++j) { largest[j] = 0.0f; largest_pix[j] = -1; smallest[j] = FLT_MAX; }
                for (int t = 0; t < BLOCK; ++t) {
                    if (masked[t] || done[t]) continue;
                    int x = tx * BX + t % BX, y = ty * BY + t / BX;
                    int pix_id = W * y + x;
                    for (int j = 0; j < cnt && !done[t]; ++j) {
                        uint32_t id = inst[b + j].val;
                        float dx = xy[2 * id] - (float)x, dy = xy[2 * id + 1] - (float)y;
                        const float* co = conic_o + 4 * (size_t)id;
                        float power = -0.5f * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                        if (power > 0.0f) continue;
                        float alpha = fminf(0.99f, co[3] * expf(power));
                        if (alpha < 1.0f / 255.0f) continue;
                        float test_T = T[t] * (1 - alpha);
                        if (test_T < 0.0001f) { done[t] = 1; continue; }
                        for (int ch = 0; ch < 3; ++ch) C[t][ch] += rgb[3 * id + ch] * alpha * T[t];
                        Ei[t] += (1 / depths[id]) * alpha * T[t];
                        float contribution = alpha * T[t];
                        E[t] += depths[id] * contribution;
                        if (contribution > largest[j]) { largest[j] = contribution; largest_pix[j] = pix_id; }   /* ascending pix_id */
                        T[t] = test_T;
                    }
                }
                for (int j = 0; j < cnt; ++j) {
                    uint32_t id = inst[b + j].val;
                    if (largest[j] > gauss_contrib[id] ||
                        (largest[j] == gauss_contrib[id] && largest[j] > 0.0f && largest_pix[j] < gauss_pixels[id])) {
                        gauss_contrib[id] = largest[j];
                        gauss_pixels[id] = largest_pix[j];
                    }
                }
                if (calc_surf) {
                    for (int j = 0; j < cnt; ++j) {
                        uint32_t id = inst[b + j].val;
                        for (int t = 0; t < BLOCK; ++t) {
                            if (masked[t]) continue;
                            float d = fabsf(depths[id] - E[t]);
                            if (d < smallest[j]) smallest[j] = d;
                        }
                        if (smallest[j] < gauss_surf[id]) gauss_surf[id] = smallest[j];
                    }
                }
            }
            for (int t = 0; t < BLOCK; ++t) {
                if (!inside[t] || masked[t]) continue;
                int x = tx * BX + t % BX, y = ty * BY + t / BX;
                size_t pix_id = (size_t)W * y + x;
                for (int ch = 0; ch < 3; ++ch) out_color[(size_t)ch * H * W + pix_id] = C[t][ch] + T[t] * bg[ch];
                out_invdepth[pix_id] = Ei[t];
                out_depth[pix_id] = E[t];
            }
        }
    free(depths); free(xy); free(conic_o); free(rgb); free(touched); free(inst); free(tstart);
    return (int)L;
}
