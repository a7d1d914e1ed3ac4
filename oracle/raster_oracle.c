/*
 * ORACLE (test infrastructure, not product): plain-C CPU restatement of the reference's forward
 * rasterizer arithmetic.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product path (relightable-nr_amd/) never does.
 *
 * Follows, function by function:
 *   oracle_face_setup        <- rasterize_cuda_kernel.cu:24-67   (forward_face_index_map_cuda_kernel_1)
 *   oracle_face_index_map    <- rasterize_cuda_kernel.cu:70-169  (forward_face_index_map_cuda_kernel_2)
 *   oracle_texture_sampling  <- rasterize_cuda_kernel.cu:171-242 (forward_texture_sampling_cuda_kernel)
 *
 * Brute force O(pixels x faces), exactly like the reference: the HIP rasterizer's binned design is
 * checked against this for bit-exact face_index / weight / depth maps.
 *
 * Arithmetic contract (pinned by tests/golden/raster_*.npz, generated from the reference's own kernel
 * bodies by tests/golden/make_golden.py): IEEE-754 binary32, one rounding per source-level operation,
 * NO fused multiply-add (build with -ffp-contract=off), CUDA min/max == fminf/fmaxf (a NaN operand
 * loses), candidates visited in ascending face order with a strict `<` depth test.
 * The reference's double literals (`0.5 *`, `2. *`, `1. /`, `max(w, 0.)`) promote sub-expressions to
 * double; each such sub-expression is exactly representable or a single correctly rounded division,
 * so evaluating in float gives the same bits.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* NDC [-1,1] -> pixel-index space [0, is-1]  (kernel.cu:44-49) */
static inline float ndc_to_pix(float c, int is) {
    float s = (float)is;
    return 0.5f * (c * s + s - 1.0f);
}

/* back-face predicate, identical in kernel_1 (cu:40) and kernel_2 (cu:111) */
static inline int is_backface(const float* f) {
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

void oracle_face_setup(const float* faces, float* faces_inv, int batch_size, int num_faces, int image_size) {
    const long total = (long)batch_size * num_faces;
    for (long i = 0; i < total; i++) {
        const float* f = faces + 9 * i;
        float* out = faces_inv + 9 * i;
        if (is_backface(f)) continue; /* caller pre-zeroes faces_inv (rasterize.py:163) */
        float px[3], py[3];
        for (int v = 0; v < 3; v++) {
            px[v] = ndc_to_pix(f[3 * v + 0], image_size);
            py[v] = ndc_to_pix(f[3 * v + 1], image_size);
        }
        /* adjugate rows of [[x0 x1 x2],[y0 y1 y2],[1 1 1]]^-1 (cu:52-55) */
        float m[9];
        for (int v = 0; v < 3; v++) {
            const int a = (v + 1) % 3, b = (v + 2) % 3;
            m[3 * v + 0] = py[a] - py[b];
            m[3 * v + 1] = px[b] - px[a];
            m[3 * v + 2] = px[a] * py[b] - px[b] * py[a];
        }
        /* denominator, same association as cu:56-59: ((t2 + t0) + t1) */
        const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
        for (int k = 0; k < 9; k++) out[k] = m[k] / den;
    }
}

/* One candidate face at one pixel.  Returns 1 and fills w[3], *zp when the face passes the
 * back-face, inside, and near/far tests (cu:110-139); the caller applies the strict z-test. */
static inline int candidate(const float* f, const float* finv, float xp, float yp, int xi, int yi,
                            float near, float far, float* w, float* zp_out) {
    if (is_backface(f)) return 0;
    if ((yp - f[1]) * (f[3] - f[0]) < (xp - f[0]) * (f[4] - f[1])) return 0;
    if ((yp - f[4]) * (f[6] - f[3]) < (xp - f[3]) * (f[7] - f[4])) return 0;
    if ((yp - f[7]) * (f[0] - f[6]) < (xp - f[6]) * (f[1] - f[7])) return 0;
    const float fx = (float)xi, fy = (float)yi;
    float wsum = 0.0f;
    for (int k = 0; k < 3; k++) {
        float t = finv[3 * k + 0] * fx + finv[3 * k + 1] * fy + finv[3 * k + 2];
        t = fminf(fmaxf(t, 0.0f), 1.0f);
        w[k] = t;
        wsum += t;
    }
    for (int k = 0; k < 3; k++) w[k] /= wsum;
    const float zp = 1.0f / (w[0] / f[2] + w[1] / f[5] + w[2] / f[8]);
    if (zp <= near || far <= zp) return 0; /* NaN zp passes this reject and then loses the z-test */
    *zp_out = zp;
    return 1;
}

void oracle_face_index_map(const float* faces, const float* faces_inv, int32_t* face_index_map,
                           float* weight_map, float* depth_map, float* face_inv_map, int batch_size,
                           int num_faces, int image_size, float near, float far, int return_depth) {
    const int is = image_size;
    const long npix = (long)batch_size * is * is;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < npix; i++) {
        const int bn = (int)(i / ((long)is * is));
        const int pn = (int)(i % ((long)is * is));
        const int yi = pn / is, xi = pn % is;
        /* pixel centre in NDC (cu:93-94): (2*idx + 1 - is) / is, an exact small integer over is */
        const float yp = (float)(2 * yi + 1 - is) / (float)is;
        const float xp = (float)(2 * xi + 1 - is) / (float)is;
        const float* f = faces + (size_t)bn * num_faces * 9;
        const float* fi = faces_inv + (size_t)bn * num_faces * 9;
        float best_z = far, best_w[3] = {0, 0, 0};
        int best = -1;
        for (int fn = 0; fn < num_faces; fn++, f += 9, fi += 9) {
            float w[3], zp;
            if (!candidate(f, fi, xp, yp, xi, yi, near, far, w, &zp)) continue;
            if (zp < best_z) {
                best_z = zp;
                best = fn;
                best_w[0] = w[0];
                best_w[1] = w[1];
                best_w[2] = w[2];
            }
        }
        if (best >= 0) { /* untouched pixels keep the caller's fill: -1 / 0 / far (rasterize.py:50-52) */
            depth_map[i] = best_z;
            face_index_map[i] = best;
            for (int k = 0; k < 3; k++) weight_map[3 * i + k] = best_w[k];
            if (return_depth && face_inv_map) {
                const float* wfi = faces_inv + ((size_t)bn * num_faces + best) * 9;
                for (int k = 0; k < 9; k++) face_inv_map[9 * i + k] = wfi[k];
            }
        }
    }
}

void oracle_texture_sampling(const float* faces, const float* textures, const int32_t* face_index_map,
                             const float* weight_map, const float* depth_map, float* rgb_map,
                             int32_t* sampling_index_map, float* sampling_weight_map, int batch_size,
                             int num_faces, int image_size, int texture_size, float eps) {
    const int ts = texture_size;
    const long npix = (long)batch_size * image_size * image_size;
    for (long i = 0; i < npix; i++) {
        const int fidx = face_index_map[i];
        if (fidx < 0) continue;
        const int bn = (int)(i / ((long)image_size * image_size));
        const float* f = faces + ((size_t)bn * num_faces + fidx) * 9;
        const float* tex = textures + ((size_t)bn * num_faces + fidx) * ts * ts * ts * 3;
        const float depth = depth_map[i];
        float t[3];
        for (int k = 0; k < 3; k++) { /* cu:208-213: perspective-corrected barycentric texel coordinate */
            float v = weight_map[3 * i + k] * (float)(ts - 1) * (depth / f[3 * k + 2]);
            v = fmaxf(v, 0.0f);
            v = fminf(v, (float)(ts - 1) - eps);
            t[k] = v;
        }
        float acc[3] = {0, 0, 0};
        for (int corner = 0; corner < 8; corner++) { /* cu:217-236: trilinear blend of the 8 cube corners */
            float w = 1.0f;
            int ti[3];
            for (int k = 0; k < 3; k++) {
                const int base = (int)t[k];
                const float frac = t[k] - (float)base;
                if (((corner >> k) & 1) == 0) {
                    w *= 1.0f - frac;
                    ti[k] = base;
                } else {
                    w *= frac;
                    ti[k] = base + 1;
                }
            }
            const int isc = ti[0] * ts * ts + ti[1] * ts + ti[2];
            for (int k = 0; k < 3; k++) acc[k] += w * tex[isc * 3 + k];
            sampling_index_map[8 * i + corner] = isc;
            sampling_weight_map[8 * i + corner] = w;
        }
        for (int k = 0; k < 3; k++) rgb_map[3 * i + k] = acc[k];
    }
}

/* =====================================================================================================
 * Backward passes (rasterize_cuda_kernel.cu:244-592).  Same arithmetic contract as above.
 *   oracle_backward_pixel_map  <- cu:244-498   oracle_backward_textures <- cu:500-535
 *   oracle_backward_depth_map  <- cu:537-592
 * float -> int conversions follow the GPU rule the reference runs under (NaN -> 0, saturating), which
 * x86's cvttss2si does not: gpu_int() makes it explicit.
 * ===================================================================================================== */
static inline int gpu_int(float x) {
    if (x != x) return 0;
    if (x >= 2147483520.0f) return 2147483647;
    if (x <= -2147483648.0f) return -2147483647 - 1;
    return (int)x;
}

typedef struct {
    const int32_t* fim; const float* rgb; const float* alpha; const float* g_rgb; const float* g_alpha;
    int is, use_rgb, use_alpha; float eps;
} sweep_ctx;

/* d(loss)/d(pixel m) when pixel m would take the colour of reference pixel `ref` (cu:357-364, 454-461) */
static float colour_delta(const sweep_ctx* c, long m, long ref) {
    float d = 0.0f;
    if (c->use_alpha) d += (c->alpha[m] - c->alpha[ref]) * c->g_alpha[m];
    if (c->use_rgb)
        for (int k = 0; k < 3; k++) d += (c->rgb[m * 3 + k] - c->rgb[ref * 3 + k]) * c->g_rgb[m * 3 + k];
    return d;
}

/* One edge (a -> b, opposite vertex o) swept along one axis; q[.][0] is the sweep coordinate, q[.][1] the
 * other one, all in pixel-index space.  ga / gb accumulate the gradient of vertex a / b's OTHER coordinate. */
static void sweep_edge(const sweep_ctx* c, int fn, long base, int axis, const float q[3][2], float* ga, float* gb) {
    const int is = c->is;
    const float fis = (float)is;
    const int dir = (axis == 0) ? (q[0][0] < q[1][0] ? -1 : 1) : (q[0][0] < q[1][0] ? 1 : -1);
    const long step1 = axis == 0 ? is : 1, step0 = axis == 0 ? 1 : is;
    const int from = gpu_int(fmaxf(ceilf(fminf(q[0][0], q[1][0])), 0.0f));
    const int to = gpu_int(fminf(fmaxf(q[0][0], q[1][0]), fis - 1.0f));
    for (int d0 = from; d0 <= to; d0++) {
        const float x = (float)d0;
        const float cross = (q[1][1] - q[0][1]) / (q[1][0] - q[0][0]) * (x - q[0][0]) + q[0][1];
        const int in = gpu_int(dir > 0 ? floorf(cross) : ceilf(cross));
        const int out = in + dir;
        if (in < 0 || in >= is || out < 0 || out >= is) continue;
        const long line = base + d0 * step0;
        const long m_in = line + in * step1, m_out = line + out * step1;
        for (int pass = 0; pass < 2; pass++) {
            int lim;
            if (pass == 0) {                                   /* outside run, needs the inner pixel to show fn */
                if (c->fim[m_in] != fn) continue;
                lim = dir > 0 ? is - 1 : 0;
            } else {                                           /* inside run, up to the opposite edge */
                float far_cross;
                if ((x - q[0][0]) * (x - q[2][0]) < 0)
                    far_cross = (q[2][1] - q[0][1]) / (q[2][0] - q[0][0]) * (x - q[0][0]) + q[0][1];
                else
                    far_cross = (q[1][1] - q[2][1]) / (q[1][0] - q[2][0]) * (x - q[2][0]) + q[2][1];
                lim = gpu_int(dir > 0 ? ceilf(far_cross) : floorf(far_cross));
            }
            const int start = pass == 0 ? out : in;
            int lo = start < lim ? start : lim, hi = start < lim ? lim : start;
            if (lo < 0) lo = 0;
            if (hi > is - 1) hi = is - 1;
            for (int d1 = lo; d1 <= hi; d1++) {
                const long m = line + d1 * step1;
                if (pass == 1 && c->fim[m] != fn) continue;
                const float dg = colour_delta(c, m, pass == 0 ? m_in : m_out);
                if (dg <= 0) continue;
                const float off = (float)d1 - cross;
                if (q[1][0] != x) {
                    float dist = (q[1][0] - q[0][0]) / (q[1][0] - x) * off * 2.0f / fis;
                    dist = 0 < dist ? dist + c->eps : dist - c->eps;
                    *ga -= dg / dist;
                }
                if (q[0][0] != x) {
                    float dist = (q[1][0] - q[0][0]) / (x - q[0][0]) * off * 2.0f / fis;
                    dist = 0 < dist ? dist + c->eps : dist - c->eps;
                    *gb -= dg / dist;
                }
            }
        }
    }
}

void oracle_backward_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                               const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                               float* grad_faces, int batch_size, int num_faces, int image_size, float eps,
                               int return_rgb, int return_alpha) {
    if (!return_rgb && !return_alpha) return;
    const sweep_ctx c = {face_index_map, rgb_map, alpha_map, grad_rgb_map, grad_alpha_map, image_size, return_rgb,
                         return_alpha, eps};
    const long total = (long)batch_size * num_faces;
#pragma omp parallel for schedule(dynamic, 64)
    for (long i = 0; i < total; i++) {
        const float* f = faces + i * 9;
        if (is_backface(f)) continue;                     /* grad_faces row left as the caller filled it */
        float g[9] = {0};
        const long base = (i / num_faces) * (long)image_size * image_size;
        for (int e = 0; e < 3; e++) {
            const int a = e, b = (e + 1) % 3, o = (e + 2) % 3;
            for (int axis = 0; axis < 2; axis++) {
                float q[3][2];
                const int v[3] = {a, b, o};
                for (int n = 0; n < 3; n++)
                    for (int d = 0; d < 2; d++) q[n][d] = ndc_to_pix(f[3 * v[n] + ((d + axis) & 1)], image_size);
                sweep_edge(&c, (int)(i % num_faces), base, axis, q, &g[3 * a + (1 - axis)], &g[3 * b + (1 - axis)]);
            }
        }
        for (int k = 0; k < 9; k++) grad_faces[i * 9 + k] = g[k];
    }
}

/* serial (the reference's atomics have no defined order; tests compare with a tolerance) */
void oracle_backward_textures(const int32_t* face_index_map, const float* sampling_weight_map,
                              const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                              int batch_size, int num_faces, int image_size, int texture_size) {
    const long pix = (long)image_size * image_size, cube = (long)texture_size * texture_size * texture_size * 3;
    for (long i = 0; i < batch_size * pix; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        float* gt = grad_textures + ((i / pix) * num_faces + fn) * cube;
        for (int s = 0; s < 8; s++)
            for (int k = 0; k < 3; k++)
                gt[(long)sampling_index_map[i * 8 + s] * 3 + k] += sampling_weight_map[i * 8 + s] * grad_rgb_map[i * 3 + k];
    }
}

void oracle_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                               const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                               float* grad_faces, int batch_size, int num_faces, int image_size) {
    const long pix = (long)image_size * image_size;
    for (long i = 0; i < batch_size * pix; i++) {
        const int fn = face_index_map[i];
        if (fn < 0) continue;
        const long fi = (i / pix) * num_faces + fn;
        const float* f = faces + fi * 9;
        const float* inv = face_inv_map + i * 9;
        const float* w = weight_map + i * 3;
        const float d2 = depth_map[i] * depth_map[i], gd = grad_depth_map[i];
        float* g = grad_faces + fi * 9;
        for (int k = 0; k < 3; k++) g[3 * k + 2] += gd * w[k] * d2 / (f[3 * k + 2] * f[3 * k + 2]);
        float t[2] = {0.0f, 0.0f};
        for (int k = 0; k < 2; k++)
            for (int l = 0; l < 3; l++) t[k] += -inv[3 * l + k] / f[3 * l + 2];
        for (int k = 0; k < 3; k++)
            for (int l = 0; l < 2; l++) g[3 * k + l] += -gd * t[l] * w[k] * d2 * (float)image_size / 2.0f;
    }
}

/* =====================================================================================================
 * Texture-cube <-> image kernels (load_textures_cuda_kernel.cu:6-115, create_texture_image_cuda_kernel.cu:8-116).
 * oracle_load_textures wraps each uv coordinate ONCE (the reference re-wraps in place from every texel thread:
 * identical for non-integer coordinates, racy for exact integers).
 * ===================================================================================================== */
static float uv_mod(float x, float y) { return x > 0 ? fmodf(x, y) : y + fmodf(x, y); }

void oracle_load_textures(const float* image, float* faces, float* textures, const int32_t* is_update, int num_faces,
                          int ts, int ih, int iw, int wrapping, int use_bilinear) {
    for (int c = 0; c < num_faces * 6; c++) {
        if (!is_update[c / 6]) continue;
        float f = faces[c];
        if (wrapping == 0) f = uv_mod(f, 1.0f);
        else if (wrapping == 1) f = uv_mod(f, 2.0f) < 1 ? uv_mod(f, 1.0f) : 1 - uv_mod(f, 1.0f);
        else if (wrapping == 2) f = fmaxf(fminf(f, 1.0f), 0.0f);
        faces[c] = f;
    }
    const long cube = (long)ts * ts * ts;
    for (long i = 0; i < num_faces * cube; i++) {
        const int fn = (int)(i / cube);
        if (!is_update[fn]) continue;
        float* t = textures + i * 3;
        if (wrapping == 3) { t[0] = t[1] = t[2] = 0.0f; continue; }
        float b[3] = {(float)((i / (ts * ts)) % ts) / (float)(ts - 1), (float)((i / ts) % ts) / (float)(ts - 1),
                      (float)(i % ts) / (float)(ts - 1)};
        if (0 < b[0] + b[1] + b[2]) {
            const float s = b[0] + b[1] + b[2];
            b[0] /= s; b[1] /= s; b[2] /= s;
        }
        const float* f = faces + (long)fn * 6;
        const float px = (f[0] * b[0] + f[2] * b[1] + f[4] * b[2]) * (float)(iw - 1);
        const float py = (f[1] * b[0] + f[3] * b[1] + f[5] * b[2]) * (float)(ih - 1);
        if (use_bilinear) {
            const int x0 = (int)px, y0 = (int)py;
            int x1 = x0 + 1, y1 = (int)(py + 1);
            if (x1 > iw - 1) x1 = iw - 1;
            if (y1 > ih - 1) y1 = ih - 1;
            const float wx1 = px - (float)x0, wx0 = 1 - wx1, wy1 = py - (float)y0, wy0 = 1 - wy1;
            for (int k = 0; k < 3; k++) {
                float c = 0;
                c += image[((long)y0 * iw + x0) * 3 + k] * (wx0 * wy0);
                c += image[((long)y1 * iw + x0) * 3 + k] * (wx0 * wy1);
                c += image[((long)y0 * iw + x1) * 3 + k] * (wx1 * wy0);
                c += image[((long)y1 * iw + x1) * 3 + k] * (wx1 * wy1);
                t[k] = c;
            }
        } else {
            const int xi = (int)roundf(px), yi = (int)roundf(py);
            for (int k = 0; k < 3; k++) t[k] = image[((long)yi * iw + xi) * 3 + k];
        }
    }
}

void oracle_create_texture_image(const float* vertices_all, const float* textures, float* image, int num_faces, int tsi,
                                 int image_height, int image_width, float eps) {
    int tw = 1;
    while ((long)tw * tw <= (long)num_faces - 1) tw++;
    const int tso = image_width / tw;
    for (int y = 0; y < image_height; y++)
        for (int x = 0; x < image_width; x++) {
            const int fn = x / tso + (y / tso) * tw;
            if (fn >= num_faces) continue;              /* the reference reads out of bounds here */
            const float* tex = textures + (long)fn * tsi * tsi * tsi * 3;
            const float* p0 = vertices_all + (long)fn * 6; const float* p1 = p0 + 2; const float* p2 = p0 + 4;
            float inv[9] = {p1[1] - p2[1], p2[0] - p1[0], p1[0] * p2[1] - p2[0] * p1[1],
                            p2[1] - p0[1], p0[0] - p2[0], p2[0] * p0[1] - p0[0] * p2[1],
                            p0[1] - p1[1], p1[0] - p0[0], p0[0] * p1[1] - p1[0] * p0[1]};
            const float den = p2[0] * (p0[1] - p1[1]) + p0[0] * (p1[1] - p2[1]) + p1[0] * (p2[1] - p0[1]);
            for (int k = 0; k < 9; k++) inv[k] /= den;
            float w[3], ws = 0;
            for (int k = 0; k < 3; k++) { w[k] = inv[3 * k] * (float)x + inv[3 * k + 1] * (float)y + inv[3 * k + 2]; ws += w[k]; }
            float tf[3];
            for (int k = 0; k < 3; k++) {
                w[k] /= (ws + eps);
                tf[k] = fminf(fmaxf(w[k] * (float)(tsi - 1), 0.0f), (float)(tsi - 1) - eps);
            }
            float px[3] = {0, 0, 0};
            for (int pn = 0; pn < 8; pn++) {
                float wt = 1; int ti[3];
                for (int k = 0; k < 3; k++) {
                    const int fl = (int)tf[k];
                    if (((pn >> k) & 1) == 0) { wt *= 1 - (tf[k] - (float)fl); ti[k] = fl; }
                    else { wt *= tf[k] - (float)fl; ti[k] = fl + 1; }
                }
                const int isc = ti[0] * tsi * tsi + ti[1] * tsi + ti[2];
                for (int k = 0; k < 3; k++) px[k] += wt * tex[isc * 3 + k];
            }
            for (int k = 0; k < 3; k++) image[((long)y * image_width + x) * 3 + k] = px[k];
        }
    for (int y = 0; y < image_height; y++)
        for (int x = 0; x < image_width; x++)
            if ((y % tso + 1) == (x % tso))
                for (int k = 0; k < 3; k++)
                    image[((long)y * image_width + x) * 3 + k] = image[((long)y * image_width + x - 1) * 3 + k];
}
