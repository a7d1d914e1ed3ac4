// Shading-side kernels of the hot path (everything between the rasterizer and the U-Net, and after it).
//
//   project_vertices_kernel   nr.projection                      projection.py:6-53
//   face_tangents_kernel      per-face tangent of get_TBN_map    render.py:135-150
//   shade_inputs_kernel       TBN + view dir + SH(lmax 2) + 4-level neural texture + 26 rays + channel
//                             assembly -> channel-last network input, written once, coalesced
//                             (render.py:152-166, camera.py:5-45, test_rnr.py:303-356, network.py:67-91,
//                              network.py:445-472, misc.py:5-42, sph_harm.py:41-71)
//   ray_render_kernel         out-layer bias + tanh + RayRenderer (network.py:253, 481-527; test_rnr.py:357-359)
//   sh_basis / sh_reconstruct / sh_fit / interpolate_bilinear / layout helpers
//
// Built with -ffp-contract=off: the integer tap indices of the bilinear sampler (misc.py:16-25) must be the
// bits the reference's float32 expressions produce (u*(S-1), (S-1) - v*(S-1), floor), so no FMA contraction.
#include "rnr_internal.h"

namespace rnr {

#define RNR_PI_F 3.14159265358979323846f

__device__ __forceinline__ float3 f3(float x, float y, float z) { return make_float3(x, y, z); }
__device__ __forceinline__ float dot3(float3 a, float3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float3 cross3(float3 a, float3 b) {
    return f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// torch.nn.functional.normalize: x / max(||x||_2, 1e-12)
__device__ __forceinline__ float3 normalize3(float3 a) {
    const float n = fmaxf(sqrtf(a.x * a.x + a.y * a.y + a.z * a.z), 1e-12f);
    return f3(a.x / n, a.y / n, a.z / n);
}

// same with the hardware reciprocal square root (v_rsq_f32, 1 ulp) instead of a correctly rounded sqrt and three
// correctly rounded divisions (~45 VALU instructions): for the fused shading kernel, whose unit vectors only feed
// float-tolerance quantities (ray directions, SH basis), never an integer index
__device__ __forceinline__ float3 normalize3_fast(float3 a) {
    const float inv = fminf(__builtin_amdgcn_rsqf(a.x * a.x + a.y * a.y + a.z * a.z), 1e12f);
    return f3(a.x * inv, a.y * inv, a.z * inv);
}

// ------------------------------------------------------------------------------------------------
// nr.projection for vertex i of the view batch (projection.py:6-53).  POSE: R and t are read straight from the [N,4,4] pose
// matrices (R = pose[:, :3, :3], t = pose[:, :3, 3]) — the same operands, the same arithmetic, the same bits as with the
// separate [N,3,3] / [N,3] tensors the reference builds (test_rnr.py:283-284).
template <bool POSE>
__device__ __forceinline__ void project_vertex(long i, const float* __restrict__ vertices, const float* __restrict__ K,
                                               const float* __restrict__ R, const float* __restrict__ t,
                                               const float* __restrict__ dist, const float* __restrict__ offset,
                                               const float* __restrict__ scale, float* __restrict__ out, int nviews, int nv,
                                               float orig_size, float eps) {
    if (i >= (long)nviews * nv) return;
    const int n = (int)(i / nv), vi = (int)(i % nv);
    const float* p = vertices + (size_t)vi * 3;
    constexpr int RS = POSE ? 4 : 3;            // row stride of R
    const float* Rn = R + n * (POSE ? 16 : 9);
    const float* Kn = K + n * 9;
    const float t0 = POSE ? Rn[3] : t[n * 3 + 0], t1 = POSE ? Rn[7] : t[n * 3 + 1], t2 = POSE ? Rn[11] : t[n * 3 + 2];
    const float vx = p[0], vy = p[1], vz = p[2];
    // vertices . R^T + t   (projection.py:22)
    const float x = vx * Rn[0] + vy * Rn[1] + vz * Rn[2] + t0;
    const float y = vx * Rn[RS + 0] + vy * Rn[RS + 1] + vz * Rn[RS + 2] + t1;
    const float z = vx * Rn[2 * RS + 0] + vy * Rn[2 * RS + 1] + vz * Rn[2 * RS + 2] + t2;
    const float xn = x / (z + eps), yn = y / (z + eps);
    float k1 = 0.f, k2 = 0.f, p1 = 0.f, p2 = 0.f, k3 = 0.f;
    if (dist) { k1 = dist[n * 5 + 0]; k2 = dist[n * 5 + 1]; p1 = dist[n * 5 + 2]; p2 = dist[n * 5 + 3]; k3 = dist[n * 5 + 4]; }
    const float r = sqrtf(xn * xn + yn * yn);
    const float r2 = r * r, r4 = r2 * r2, r6 = r4 * r2;
    const float radial = 1.0f + k1 * r2 + k2 * r4 + k3 * r6;
    const float xd = xn * radial + 2.0f * p1 * xn * yn + p2 * (r2 + 2.0f * xn * xn);
    const float yd = yn * radial + p1 * (r2 + 2.0f * yn * yn) + 2.0f * p2 * xn * yn;
    float u = xd * Kn[0] + yd * Kn[1] + Kn[2];
    float v = xd * Kn[3] + yd * Kn[4] + Kn[5];
    if (offset && scale) {  // projection.py:42-46 (note the swapped component order)
        u = (u + offset[n * 2 + 1]) * scale[n * 2 + 1];
        v = (v + offset[n * 2 + 0]) * scale[n * 2 + 0];
    }
    v = orig_size - v;
    u = 2.0f * (u - orig_size / 2.0f) / orig_size;
    v = 2.0f * (v - orig_size / 2.0f) / orig_size;
    out[i * 3 + 0] = u;
    out[i * 3 + 1] = v;
    out[i * 3 + 2] = z;
}

__global__ void __launch_bounds__(256)
project_vertices_kernel(const float* __restrict__ vertices, const float* __restrict__ K,
                        const float* __restrict__ R, const float* __restrict__ t,
                        const float* __restrict__ dist, const float* __restrict__ offset,
                        const float* __restrict__ scale, float* __restrict__ out, int nviews, int nv,
                        float orig_size, float eps) {
    project_vertex<false>((long)blockIdx.x * blockDim.x + threadIdx.x, vertices, K, R, t, dist, offset, scale, out, nviews, nv,
                          orig_size, eps);
}

// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void face_tangent(int i, const rnr_mesh& mesh, float* __restrict__ out) {
    if (i >= mesh.num_faces) return;
    const int32_t* vi = mesh.f_v_idx + (size_t)i * 3;
    const int32_t* ti = mesh.f_vt_idx + (size_t)i * 3;
    const float* a = mesh.v + (size_t)vi[0] * 3;
    const float* b = mesh.v + (size_t)vi[1] * 3;
    const float* c = mesh.v + (size_t)vi[2] * 3;
    const float* ta = mesh.vt + (size_t)ti[0] * 2;
    const float* tb = mesh.vt + (size_t)ti[1] * 2;
    const float* tc = mesh.vt + (size_t)ti[2] * 2;
    const float3 e1 = f3(b[0] - a[0], b[1] - a[1], b[2] - a[2]);
    const float3 e2 = f3(c[0] - a[0], c[1] - a[1], c[2] - a[2]);
    const float d1x = tb[0] - ta[0], d1y = tb[1] - ta[1];
    const float d2x = tc[0] - ta[0], d2y = tc[1] - ta[1];
    const float f = 1.0f / fmaxf(d1x * d2y - d2x * d1y, 1e-8f);   // clamp(min=1e-8), render.py:143
    const float3 tn = normalize3(f3(f * (d2y * e1.x - d1y * e2.x), f * (d2y * e1.y - d1y * e2.y),
                                    f * (d2y * e1.z - d1y * e2.z)));
    out[i * 3 + 0] = tn.x; out[i * 3 + 1] = tn.y; out[i * 3 + 2] = tn.z;
}

__global__ void __launch_bounds__(256)
face_tangents_kernel(rnr_mesh mesh, float* __restrict__ out) {
    face_tangent(blockIdx.x * blockDim.x + threadIdx.x, mesh, out);
}

// ------------------------------------------------------------------------------------------------
// bilinear taps of misc.interpolate_bilinear (misc.py:14-40)
struct Taps {
    int x0, y0, x1, y1;
    float w00, w10, w01, w11;
};
__device__ __forceinline__ Taps bilinear_taps(float x, float y, int W, int H) {
    Taps t;
    const float valid = (x >= 0.0f && x <= (float)(W - 1) && y >= 0.0f && y <= (float)(H - 1)) ? 1.0f : 0.0f;
    // floor -> int64 in the reference; clamp in float first so NaN / huge coordinates stay defined
    const float fx = floorf(x), fy = floorf(y);
    int x0 = (int)fminf(fmaxf(fx, -2.0f), (float)W + 1.0f);
    int y0 = (int)fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
    int x1 = x0 + 1, y1 = y0 + 1;
    x0 = min(max(x0, 0), W - 1); x1 = min(max(x1, 0), W - 1);
    y0 = min(max(y0, 0), H - 1); y1 = min(max(y1, 0), H - 1);
    t.x0 = x0; t.y0 = y0; t.x1 = x1; t.y1 = y1;
    const float x0w = (float)(x0 - (x0 == x1 ? 1 : 0)), y0w = (float)(y0 - (y0 == y1 ? 1 : 0));
    const float x1f = (float)x1, y1f = (float)y1;
    t.w00 = (x1f - x) * (y1f - y) * valid;
    t.w10 = (x1f - x) * (y - y0w) * valid;
    t.w01 = (x - x0w) * (y1f - y) * valid;
    t.w11 = (x - x0w) * (y - y0w) * valid;
    return t;
}

constexpr int SH_PIX = 32;        // pixels per workgroup (32 beat 16 / 64 / 128 on the GPU: 17 KB of LDS, 9 workgroups per CU)
constexpr int SH_THREADS = 256;
constexpr int MAX_LEVELS = 8;
constexpr int MAX_RAYS = 32;

struct ShadeParams {
    const int32_t* face_index_map;
    const float* alpha;
    const float* uv_map;
    const float* normal_map;
    const float* tangents;
    int num_faces;
    const float* proj_inv;
    const float* R_inv;
    const float* tex[MAX_LEVELS];
    int tex_size[MAX_LEVELS];
    int num_levels, C, sh_start;
    float piv_spec[MAX_RAYS * 3];   // [r][xyz]
    float piv_diff[MAX_RAYS * 3];
    int n_spec, n_diff;
    float* net_in;
    int c_pad;
    float* rays_uv;
    float* neural_img;
    float* sh_basis_map;
    long npix;                      // N*H*W
    int H, W;
};

// sum over the levels of the bilinear fetch of channel quad q at (u, v)  (TextureMapper.forward, network.py:71-85; the
// integer taps follow misc.py:16-35 exactly: this file is built without FMA contraction)
__device__ __forceinline__ float4 texture_quad(const ShadeParams& P, float u, float v, int q, int quads) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int l = 0; l < P.num_levels; l++) {
        const int s = P.tex_size[l];
        const float sm1 = (float)(s - 1);
        const float x = u * sm1;
        const float y = sm1 - v * sm1;
        const Taps t = bilinear_taps(x, y, s, s);
        // r04: a workgroup-uniform buffer resource per level and one 32-bit lane offset per tap (a level of <= 2 GiB): the flat
        // form spent ~6 VALU instructions of 64-bit address arithmetic on each of the 16 gathers of an item
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.tex[l]), 0, 0x7fffffff, 0x27000);
        const unsigned texel = (unsigned)quads * 16u, rowb = (unsigned)s * texel;
        const unsigned r0 = (unsigned)t.y0 * rowb + (unsigned)q * 16u, r1 = (unsigned)t.y1 * rowb + (unsigned)q * 16u;
        const unsigned c0 = (unsigned)t.x0 * texel, c1 = (unsigned)t.x1 * texel;
        auto tap = [&](unsigned off) { return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)off, 0, 0)); };
        const float4 i00 = tap(r0 + c0);
        const float4 i10 = tap(r1 + c0);
        const float4 i01 = tap(r0 + c1);
        const float4 i11 = tap(r1 + c1);
        float4 lv;   // I00*w00 + I10*w10 + I01*w01 + I11*w11 (misc.py:42)
        lv.x = i00.x * t.w00 + i10.x * t.w10 + i01.x * t.w01 + i11.x * t.w11;
        lv.y = i00.y * t.w00 + i10.y * t.w10 + i01.y * t.w01 + i11.y * t.w11;
        lv.z = i00.z * t.w00 + i10.z * t.w10 + i01.z * t.w01 + i11.z * t.w11;
        lv.w = i00.w * t.w00 + i10.w * t.w10 + i01.w * t.w01 + i11.w * t.w11;
        if (l == 0) acc = lv;
        else { acc.x += lv.x; acc.y += lv.y; acc.z += lv.z; acc.w += lv.w; }
    }
    return acc;
}

// geometry record per pixel in LDS: T(3) B(3) N(3) vtan(3) uv(2) alpha(1) sh(9) = 24 floats
constexpr int GEO = 24;

__global__ void __launch_bounds__(SH_THREADS)
shade_inputs_kernel(const ShadeParams P) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* tile = smem;                               // [SH_PIX][c_pad]
    float* geo = smem + SH_PIX * P.c_pad;             // [SH_PIX][GEO]
    const int tid = threadIdx.x;
    const long pix0 = (long)blockIdx.x * SH_PIX;
    const int cp = P.c_pad;
    const int n_rays = P.n_spec + P.n_diff;
    const int c_geo = 3 * n_rays;                     // first channel of normal

    // zero the padding channels once
    const int c_in = c_geo + 6 + P.C;
    for (int i = tid; i < SH_PIX * (cp - c_in); i += SH_THREADS) {
        const int p = i / (cp - c_in), c = c_in + i % (cp - c_in);
        tile[p * cp + c] = 0.0f;
    }

    // ---- phase 2a (waves 1..3): the texture gathers, issued FIRST ----
    // Phase 0 is a chain of dependent loads on 32 lanes of wave 0 (face index -> tangent -> frame); the 16 bilinear gathers
    // of a (pixel, channel quad) item depend on the uv map only.  With C = 24 the 32 x 6 items are exactly the 192 threads of
    // waves 1..3, which fetch and blend them while wave 0 walks its chain: the two latency chains of a workgroup overlap
    // instead of following each other (the kernel is latency-bound).  The SH factor (phase 0's result) is
    // applied after the barrier; more than 192 items fall back to the loop behind phase 1.
    const int quads = P.C / 4;
    const bool tex_early = SH_PIX * quads <= SH_THREADS - 64;
    float4 tex_acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int tex_p = -1, tex_q = 0;
    if (tex_early && tid >= 64 && tid - 64 < SH_PIX * quads) {
        const int i = tid - 64;
        tex_p = i / quads; tex_q = i - tex_p * quads;
        const long pix = pix0 + tex_p;
        float u = 0.f, v = 0.f;
        if (pix < P.npix) { u = P.uv_map[pix * 2 + 0]; v = P.uv_map[pix * 2 + 1]; }
        tex_acc = texture_quad(P, u, v, tex_q, quads);
    }

    // ---- phase 0: one lane per pixel: TBN, view direction, SH basis ----
    if (tid < SH_PIX) {
        const long pix = pix0 + tid;
        float* g = geo + tid * GEO;
        if (pix < P.npix) {
            const int hw = P.H * P.W;
            const int n = (int)(pix / hw);
            const int rem = (int)(pix % hw);
            const int row = rem / P.W, col = rem % P.W;
            int fi = P.face_index_map[pix];
            if (fi < 0) fi += P.num_faces;            // torch negative index wrap (render.py:152)
            const float a = P.alpha[pix];
            const float3 tg = f3(P.tangents[fi * 3 + 0], P.tangents[fi * 3 + 1], P.tangents[fi * 3 + 2]);
            const float3 nm_in = f3(P.normal_map[pix * 3 + 0], P.normal_map[pix * 3 + 1], P.normal_map[pix * 3 + 2]);
            const float3 nm = normalize3_fast(nm_in);                        // render.py:155
            const float3 bt = normalize3_fast(cross3(nm, tg));               // render.py:156-157
            const float3 tt = normalize3_fast(cross3(bt, nm));               // render.py:160-161
            // view direction (camera.py:19-30)
            const float* Pi = P.proj_inv + n * 9;
            const float* Ri = P.R_inv + n * 9;
            const float pu = (float)col + 0.5f, pv = (float)row + 0.5f;
            float3 dc = f3(-(Pi[0] * pu + Pi[1] * pv + Pi[2]), -(Pi[3] * pu + Pi[4] * pv + Pi[5]),
                           -(Pi[6] * pu + Pi[7] * pv + Pi[8]));
            dc = normalize3_fast(dc);
            float3 vd = f3(Ri[0] * dc.x + Ri[1] * dc.y + Ri[2] * dc.z, Ri[3] * dc.x + Ri[4] * dc.y + Ri[5] * dc.z,
                           Ri[6] * dc.x + Ri[7] * dc.y + Ri[8] * dc.z);
            vd = normalize3_fast(vd);
            // tangent-space view direction = normalize(TBN^T v) (test_rnr.py:314-315)
            const float3 vt = normalize3_fast(f3(dot3(tt, vd), dot3(bt, vd), dot3(nm, vd)));
            g[0] = tt.x; g[1] = tt.y; g[2] = tt.z;
            g[3] = bt.x; g[4] = bt.y; g[5] = bt.z;
            g[6] = nm.x; g[7] = nm.y; g[8] = nm.z;
            g[9] = vt.x; g[10] = vt.y; g[11] = vt.z;
            g[12] = P.uv_map[pix * 2 + 0]; g[13] = P.uv_map[pix * 2 + 1];
            g[14] = a;
            // real SH, lmax = 2, orthonormal, no Condon-Shortley phase, colatitude from +z
            const float3 d = normalize3_fast(vd);
            float* sh = g + 15;
            sh[0] = 0.28209479177387814f;
            sh[1] = 0.4886025119029199f * d.y;
            sh[2] = 0.4886025119029199f * d.z;
            sh[3] = 0.4886025119029199f * d.x;
            sh[4] = 1.0925484305920792f * d.x * d.y;
            sh[5] = 1.0925484305920792f * d.y * d.z;
            sh[6] = 0.31539156525252005f * (3.0f * d.z * d.z - 1.0f);
            sh[7] = 1.0925484305920792f * d.x * d.z;
            sh[8] = 0.5462742152960396f * (d.x * d.x - d.y * d.y);
            if (P.sh_basis_map) {
#pragma unroll
                for (int k = 0; k < 9; k++) P.sh_basis_map[pix * 9 + k] = sh[k];
            }
            float* tp = tile + tid * cp + c_geo;      // channels: normal (3), view_dir (3) (test_rnr.py:351-352)
            tp[0] = nm_in.x; tp[1] = nm_in.y; tp[2] = nm_in.z;
            tp[3] = vd.x; tp[4] = vd.y; tp[5] = vd.z;
        } else {
            for (int k = 0; k < GEO; k++) g[k] = 0.0f;
        }
    }
    __syncthreads();

    // ---- phase 1: (pixel, ray) items: reflect / diffuse directions (network.py:455-465) ----
    // lanes run over the rays of a pixel first: a wave stores 3-float records at a stride of 3 floats (3 is coprime with
    // the 32 banks: conflict-free), and reads the geometry record of only ~3 pixels (LDS broadcast).  With the pixel
    // fastest every lane of a wave hit one of two banks (row stride 112 floats = 16 mod 32): 16-way conflicts,
    // 72 M conflict cycles per dispatch in the round-1 profile.
    const float inv_rays = 1.0f / (float)n_rays;
    for (int i = tid; i < SH_PIX * n_rays; i += SH_THREADS) {
        const int p = (int)(((float)i + 0.5f) * inv_rays), r = i - p * n_rays;      // exact i / n_rays for i < 2^20

        const float* g = geo + p * GEO;
        const float3 tt = f3(g[0], g[1], g[2]), bt = f3(g[3], g[4], g[5]), nm = f3(g[6], g[7], g[8]);
        const float a = g[14];
        float3 lt;   // direction in tangent space
        if (r < P.n_spec) {
            const float3 pv = f3(P.piv_spec[r * 3 + 0], P.piv_spec[r * 3 + 1], P.piv_spec[r * 3 + 2]);
            const float3 v = f3(g[9], g[10], g[11]);
            const float s = dot3(pv, v) * 2.0f;                         // camera.py:43
            lt = normalize3_fast(f3(s * pv.x - v.x, s * pv.y - v.y, s * pv.z - v.z));
            lt = f3(lt.x * a, lt.y * a, lt.z * a);
        } else {
            const int rd = r - P.n_spec;
            lt = f3(P.piv_diff[rd * 3 + 0], P.piv_diff[rd * 3 + 1], P.piv_diff[rd * 3 + 2]);
        }
        float3 d = f3(tt.x * lt.x + bt.x * lt.y + nm.x * lt.z, tt.y * lt.x + bt.y * lt.y + nm.y * lt.z,
                      tt.z * lt.x + bt.z * lt.y + nm.z * lt.z);        // TBN . lt (columns T,B,N)
        d = normalize3_fast(d);
        float* tp = tile + p * cp + 3 * r;                              // ray-major, xyz inner (test_rnr.py:350)
        tp[0] = d.x; tp[1] = d.y; tp[2] = d.z;
        if (P.rays_uv) {
            const long pix = pix0 + p;
            if (pix < P.npix) {                                         // render.py:96-102, network.py:469-470
                float u = atan2f(d.z, d.x) * 0.5f / RNR_PI_F + 0.5f;
                float v = acosf(d.y) * 1.0f / RNR_PI_F;
                const float bg = (a == 0.0f) ? 1.0f : 0.0f;
                u = u * a - bg;
                v = v * a - bg;
                P.rays_uv[(pix * 2 + 0) * n_rays + r] = u;
                P.rays_uv[(pix * 2 + 1) * n_rays + r] = v;
            }
        }
    }

    // ---- phase 2: (pixel, channel-quad) items: sum over levels of bilinear fetches (network.py:71-85) ----
    const float inv_quads = 1.0f / (float)quads;
    auto finish_quad = [&](float4 acc, int p, int q) {
        const float* g = geo + p * GEO;
        if (P.sh_start >= 0) {  // output[:, s:s+9] *= sh_basis (network.py:88-89)
            const float* sh = g + 15;
            const int c0 = 4 * q - P.sh_start;
            if (c0 + 0 >= 0 && c0 + 0 < 9) acc.x *= sh[c0 + 0];
            if (c0 + 1 >= 0 && c0 + 1 < 9) acc.y *= sh[c0 + 1];
            if (c0 + 2 >= 0 && c0 + 2 < 9) acc.z *= sh[c0 + 2];
            if (c0 + 3 >= 0 && c0 + 3 < 9) acc.w *= sh[c0 + 3];
        }
        float* tp = tile + p * cp + c_geo + 6 + 4 * q;
        if (((c_geo + 6) & 3) == 0) *reinterpret_cast<float4*>(tp) = acc;     // one ds_write_b128 (rows are 16-byte aligned)
        else { tp[0] = acc.x; tp[1] = acc.y; tp[2] = acc.z; tp[3] = acc.w; }
    };
    if (tex_early) {
        if (tex_p >= 0) finish_quad(tex_acc, tex_p, tex_q);
    } else {
        for (int i = tid; i < SH_PIX * quads; i += SH_THREADS) {
            const int p = (int)(((float)i + 0.5f) * inv_quads), q = i - p * quads;
            const float* g = geo + p * GEO;
            finish_quad(texture_quad(P, g[12], g[13], q, quads), p, q);
        }
    }
    __syncthreads();

    // ---- phase 3: one coalesced sweep of the tile to HBM ----
    const long valid_pix = min((long)SH_PIX, P.npix - pix0);
    const int n4 = (int)(valid_pix * cp / 4);
    float4* dst = reinterpret_cast<float4*>(P.net_in + pix0 * cp);
    const float4* src = reinterpret_cast<const float4*>(tile);
    for (int i = tid; i < n4; i += SH_THREADS) dst[i] = src[i];
    if (P.neural_img) {  // [N, C, H, W] copy for the API (TextureMapper.forward's return value)
        const int hw = P.H * P.W;
        for (int i = tid; i < (int)valid_pix * P.C; i += SH_THREADS) {
            const int p = i % SH_PIX, c = i / SH_PIX;
            if (p < valid_pix) {
                const long pix = pix0 + p;
                const long n = pix / hw, rem = pix % hw;
                P.neural_img[(n * P.C + c) * hw + rem] = tile[p * cp + c_geo + 6 + c];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ray renderer: 32 lanes per pixel (one ray each), 16-lane segmented shuffle reductions.
// lane layout inside a 32-lane half: lanes 0..15 -> specular rays 0..15, lanes 16..31 -> diffuse rays 0..15
// ------------------------------------------------------------------------------------------------
// Polynomial atan2 / acos for the fused ray renderer (max abs error 3e-8 / 8e-8 rad, i.e. float rounding level; the
// uv they feed is only used to pick env-map taps).  ocml's atan2f/acosf cost ~100 instructions each and made this
// kernel transcendental-bound (26 rays per pixel).  The stand-alone ray-sampler operator keeps the ocml versions.
__device__ __forceinline__ float fast_atan2f(float y, float x) {
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float a = mx > 0.0f ? mn * __builtin_amdgcn_rcpf(mx) : 0.0f;      // 1 ulp reciprocal: the result only picks env-map taps
    const float s = a * a;
    // explicit FMAs: this file is compiled with -ffp-contract=off (bit-exact tap indices elsewhere), which would turn
    // every Horner step into a multiply and an add — the kernel is VALU-bound
    float p = 0.002899040700867772f;
    p = __builtin_fmaf(p, s, -0.01637016236782074f);
    p = __builtin_fmaf(p, s, 0.04338274151086807f);
    p = __builtin_fmaf(p, s, -0.07582952827215195f);
    p = __builtin_fmaf(p, s, 0.10688958317041397f);
    p = __builtin_fmaf(p, s, -0.14219146966934204f);
    p = __builtin_fmaf(p, s, 0.19995006918907166f);
    p = __builtin_fmaf(p, s, -0.3333321213722229f);
    p = __builtin_fmaf(p, s, 1.0f);
    float r = p * a;
    if (ay > ax) r = 1.57079632679489662f - r;
    if (x < 0.0f) r = 3.14159265358979324f - r;
    return y < 0.0f ? -r : r;
}
__device__ __forceinline__ float fast_acosf(float x) {
    const float ax = fminf(fabsf(x), 1.0f);
    float p = -0.001102376147173345f;
    p = __builtin_fmaf(p, ax, 0.006096228025853634f);
    p = __builtin_fmaf(p, ax, -0.01627347804605961f);
    p = __builtin_fmaf(p, ax, 0.03031114861369133f);
    p = __builtin_fmaf(p, ax, -0.049957286566495895f);
    p = __builtin_fmaf(p, ax, 0.08893882483243942f);
    p = __builtin_fmaf(p, ax, -0.2145957499742508f);
    p = __builtin_fmaf(p, ax, 1.570796251296997f);
    const float r = __builtin_amdgcn_sqrtf(1.0f - ax) * p;
    return x < 0.0f ? 3.14159265358979324f - r : r;
}

struct RayParams {
    const float* unet_raw; int c_out_pad;
    const float* bias;
    const float* net_in; int c_pad;
    const float* alpha;
    const float* lp; int lp_h, lp_w;
    int n_spec, n_diff, alb_diff_ch, alb_spec_ch;
    float* image;
    long npix; int hw;
    int ni_need;        // floats of a net_in row the kernel reads (directions, normal / view, albedo), multiple of 4
};

// sum over each aligned group of 16 lanes, result in every lane of the group: four DPP adds on the VALU (quad_perm
// swaps 1 and 2 apart, row_half_mirror and row_mirror fold 8 and 16 lanes) instead of four ds_bpermute round trips
__device__ __forceinline__ float dpp_add(float v, const int ctrl_sel) {
    const int x = __builtin_bit_cast(int, v);
    int y;
    if (ctrl_sel == 0) y = __builtin_amdgcn_update_dpp(0, x, 0xB1, 0xF, 0xF, false);        // quad_perm [1,0,3,2]
    else if (ctrl_sel == 1) y = __builtin_amdgcn_update_dpp(0, x, 0x4E, 0xF, 0xF, false);   // quad_perm [2,3,0,1]
    else if (ctrl_sel == 2) y = __builtin_amdgcn_update_dpp(0, x, 0x141, 0xF, 0xF, false);  // row_half_mirror
    else y = __builtin_amdgcn_update_dpp(0, x, 0x140, 0xF, 0xF, false);                     // row_mirror
    return v + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float seg16_sum(float v) {
    v = dpp_add(v, 0);
    v = dpp_add(v, 1);
    v = dpp_add(v, 2);
    v = dpp_add(v, 3);
    return v;
}

#ifndef RNR_RR_PIX
#define RNR_RR_PIX 4
#endif
constexpr int RR_PIX = RNR_RR_PIX;     // pixels per lane: independent load chains in flight (the kernel is latency-bound)

__global__ void __launch_bounds__(256)
ray_render_kernel(const RayParams P) {
    // a 32-lane half-wave owns RR_PIX consecutive pixels; a workgroup owns 8 * RR_PIX consecutive pixels, so every
    // global address is a workgroup-uniform base (SGPR pair) plus a small 32-bit lane offset: no 64-bit VALU math
    constexpr int PIX_PER_WG = 8 * RR_PIX;
    const long wg_pix0 = (long)blockIdx.x * PIX_PER_WG;
    const int sub = (int)(threadIdx.x & 31);
    const int lp0 = (int)(threadIdx.x >> 5) * RR_PIX;       // first pixel of this half-wave inside the workgroup
    const bool is_diff = sub >= 16;
    const int rr = sub & 15;
    const bool ray_live = is_diff ? rr < P.n_diff : rr < P.n_spec;
    const int r = is_diff ? P.n_spec + rr : rr;
    const long pix0 = wg_pix0 + lp0;
    const long wg_n0 = wg_pix0 / P.hw;                       // workgroup-uniform (SALU)
    const int wg_rem0 = (int)(wg_pix0 - wg_n0 * P.hw);
    const float* wg_net_in = P.net_in + wg_pix0 * P.c_pad;
    const float* wg_raw = P.unet_raw + wg_pix0 * P.c_out_pad;
    const float* wg_alpha = P.alpha + wg_pix0;
    // Stage the workgroup's rows in LDS with coalesced float4 loads: the vector memory path retires one wave instruction
    // per ~16 cycles whatever its width, and 3-float-per-lane reads of the rows cost 7 instructions per (pixel, lane
    // group) — 112 per 32 pixels against 22 for the float4 sweep.  ni_need = ray directions + normal / view + the albedo
    // channels, rounded up to whole float4s.
    extern __shared__ __attribute__((aligned(16))) float rr_smem[];
    const int ni_need = P.ni_need;                          // multiple of 4, <= c_pad
    float* s_raw = rr_smem;                                 // [PIX_PER_WG][c_out_pad]
    float* s_ni = rr_smem + PIX_PER_WG * P.c_out_pad;       // [PIX_PER_WG][ni_need]
    const int wg_valid = (int)min((long)PIX_PER_WG, P.npix - wg_pix0);
    float al[RR_PIX];
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) al[k] = (lp0 + k < wg_valid) ? wg_alpha[lp0 + k] : 0.0f;
    {
        // r04: a workgroup whose 32 pixels are all background writes its zeros and is done — the frame is exactly 0 there
        // whatever the network produced (rays_uv = -1 masks every env-map tap, network.py:469-470, 497), so neither the 320 +
        // 368 bytes per pixel of unet_raw / net_in nor the 26 rays are touched (bit-identical frames; about half of the bench
        // scene's pixels)
        int fg = 0;
#pragma unroll
        for (int k = 0; k < RR_PIX; k++) fg |= al[k] != 0.0f ? 1 : 0;
        if (!__syncthreads_or(fg)) {
            if (sub == 0) {
#pragma unroll
                for (int k = 0; k < RR_PIX; k++) {
                    if (lp0 + k >= wg_valid) continue;
                    int rem = wg_rem0 + lp0 + k;
                    long n = wg_n0;
                    while (rem >= P.hw) { rem -= P.hw; n += 1; }
#pragma unroll
                    for (int c = 0; c < 3; c++) P.image[(n * 3 + c) * P.hw + rem] = 0.0f;
                }
            }
            return;
        }
    }
    {
        const int q_raw = P.c_out_pad >> 2, q_ni = ni_need >> 2;
        const float4* g_raw = reinterpret_cast<const float4*>(wg_raw);
        for (int i = threadIdx.x; i < wg_valid * q_raw; i += 256) reinterpret_cast<float4*>(s_raw)[i] = g_raw[i];
        const float inv_q = 1.0f / (float)q_ni;
        for (int i = threadIdx.x; i < wg_valid * q_ni; i += 256) {
            const int p = (int)(((float)i + 0.5f) * inv_q), q4 = i - __mul24(p, q_ni);
            reinterpret_cast<float4*>(s_ni)[i] = *reinterpret_cast<const float4*>(wg_net_in + (unsigned)(__mul24(p, P.c_pad) + 4 * q4));
        }
    }
    __syncthreads();
    float dx[RR_PIX], dy[RR_PIX], dz[RR_PIX], y0[RR_PIX], y1[RR_PIX], y2[RR_PIX];
    bool live[RR_PIX];
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) {
        live[k] = ray_live && (lp0 + k < wg_valid);
        dx[k] = dy[k] = dz[k] = y0[k] = y1[k] = y2[k] = 0.f;
        if (live[k]) {      // lanes of a half-wave read 3-float records at a stride of 3 floats: conflict-free
            const float* d = s_ni + __mul24(lp0 + k, ni_need) + 3 * r;
            const float* yr = s_raw + __mul24(lp0 + k, P.c_out_pad) + 3 * r;
            dx[k] = d[0]; dy[k] = d[1]; dz[k] = d[2];
            y0[k] = yr[0]; y1[k] = yr[1]; y2[k] = yr[2];
        }
    }
    float b0 = 0.f, b1 = 0.f, b2 = 0.f;
    if (ray_live) { b0 = P.bias[3 * r + 0]; b1 = P.bias[3 * r + 1]; b2 = P.bias[3 * r + 2]; }
    Taps tp[RR_PIX];
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) {
        // rays_uv (render.py:96-102; network.py:469-470)
        float u = __builtin_fmaf(fast_atan2f(dz[k], dx[k]), 0.5f / RNR_PI_F, 0.5f);
        float v = fast_acosf(dy[k]) * (1.0f / RNR_PI_F);
        const float bg = (al[k] == 0.0f) ? 1.0f : 0.0f;
        u = u * al[k] - bg;
        v = v * al[k] - bg;
        // env-map taps (network.py:497; misc.py:5-42): clamp(max=) only, then the validity mask zeroes uv = -1
        const float x = fminf(u * (float)P.lp_w, (float)(P.lp_w - 1));
        const float y = fminf(v * (float)P.lp_h, (float)(P.lp_h - 1));
        tp[k] = bilinear_taps(x, y, P.lp_w, P.lp_h);
    }
    // mean over the rays of a group as a multiplication by the reciprocal (one division per thread instead of six
    // correctly rounded ones per pixel; <= 1 ulp from network.py:505-513's `.sum(1) / num_ray`)
    const int lp_w3 = P.lp_w * 3;
    const float inv_spec = 1.0f / (float)P.n_spec;
    const float inv_diff = P.n_diff > 0 ? 1.0f / (float)P.n_diff : 0.0f;
    float c0[RR_PIX], c1[RR_PIX], c2[RR_PIX];
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) {
        c0[k] = c1[k] = c2[k] = 0.f;
        if (live[k]) {
            const Taps& t = tp[k];
            // texel offsets with 24-bit multiplies (full rate; v_mul_lo_u32 runs at a quarter of it): the probe has far
            // fewer than 2^24 floats
            const unsigned r0 = __umul24((unsigned)t.y0, (unsigned)lp_w3), r1 = __umul24((unsigned)t.y1, (unsigned)lp_w3);
            const unsigned q0 = __umul24((unsigned)t.x0, 3u), q1 = __umul24((unsigned)t.x1, 3u);
            const float* l00 = P.lp + (r0 + q0);
            const float* l10 = P.lp + (r1 + q0);
            const float* l01 = P.lp + (r0 + q1);
            const float* l11 = P.lp + (r1 + q1);
            // (explicit FMAs, see fast_atan2f; the reference's torch ops round every product, <= 1 ulp apart)
            const float col0 = __builtin_fmaf(l11[0], t.w11, __builtin_fmaf(l01[0], t.w01, __builtin_fmaf(l10[0], t.w10, l00[0] * t.w00)));
            const float col1 = __builtin_fmaf(l11[1], t.w11, __builtin_fmaf(l01[1], t.w01, __builtin_fmaf(l10[1], t.w10, l00[1] * t.w00)));
            const float col2 = __builtin_fmaf(l11[2], t.w11, __builtin_fmaf(l01[2], t.w01, __builtin_fmaf(l10[2], t.w10, l00[2] * t.w00)));
            // network.py:253 tanh; test_rnr.py:359 (y*0.5+0.5)*2 = y + 1 (the two scalings by 2 are exact)
            c0[k] = fast_tanh_plus1f(y0[k] + b0) * col0;
            c1[k] = fast_tanh_plus1f(y1[k] + b1) * col1;
            c2[k] = fast_tanh_plus1f(y2[k] + b2) * col2;
            // background pixels contribute exactly 0 whatever the network produced there (col = 0 by the uv = -1
            // mask); select rather than multiply: the out layer may have skipped the tile (rnr_conv2d_masked) and
            // left non-finite garbage.  Done here, after every load has landed, so the loads stay independent.
            if (al[k] == 0.0f) { c0[k] = 0.f; c1[k] = 0.f; c2[k] = 0.f; }
        }
    }
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) {
        const float s0 = seg16_sum(c0[k]), s1 = seg16_sum(c1[k]), s2 = seg16_sum(c2[k]);
        // lane 0 of each 16-lane segment holds its segment's sum; bring the diffuse sum to the group's first lane
        const float d0 = __shfl_down(s0, 16, 64), d1 = __shfl_down(s1, 16, 64), d2 = __shfl_down(s2, 16, 64);
        const long pix = pix0 + k;
        if (sub == 0 && pix < P.npix) {
            const float* ni = s_ni + __mul24(lp0 + k, ni_need) + 3 * (P.n_spec + P.n_diff) + 6;
            // view / in-view pixel from the workgroup's (scalar) quotient: no per-lane 64-bit division
            int rem = wg_rem0 + lp0 + k;
            long n = wg_n0;
            while (rem >= P.hw) { rem -= P.hw; n += 1; }      // at most once unless a view has fewer pixels than a workgroup
            const float o[3] = {s0, s1, s2}, dd[3] = {d0, d1, d2};
#pragma unroll
            for (int c = 0; c < 3; c++) {
                float out = ni[P.alb_spec_ch + c] * (o[c] * inv_spec);
                if (P.n_diff > 0) out = out + ni[P.alb_diff_ch + c] * (dd[c] * inv_diff);
                P.image[(n * 3 + c) * P.hw + rem] = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Ray weights: the part of the ray renderer that does not depend on the U-Net (rnr_ray_weights; the rest runs in the
// out-layer convolution's epilogue, rnr_conv2d_ray).  For pixel p, ray r, colour channel c
//     W[p][3 r + c] = albedo_group(r)[p][c] * env-map colour(direction of ray r)[c] / rays in the group
// so that the frame is  image[p][c] = sum_r (tanh(y[p][3 r + c] + b[3 r + c]) + 1) * W[p][3 r + c]  (network.py:253, 481-527,
// test_rnr.py:357-359).  Same lane layout, uv arithmetic and taps as ray_render_kernel; background pixels get W = 0.
// ------------------------------------------------------------------------------------------------
struct RayWeightParams {
    const float* net_in; int c_pad;
    const float* alpha;
    const float* lp; int lp_h, lp_w;
    int n_spec, n_diff, alb_diff_ch, alb_spec_ch;
    float* ray_w; int c_w;          // [npix][c_w], c_w >= 3 * rays (padding columns are written as zeros)
    long npix;
    int ni_need;
};

__global__ void __launch_bounds__(256)
ray_weights_kernel(const RayWeightParams P) {
    constexpr int PIX_PER_WG = 8 * RR_PIX;
    const long wg_pix0 = (long)blockIdx.x * PIX_PER_WG;
    const int sub = (int)(threadIdx.x & 31);
    const int lp0 = (int)(threadIdx.x >> 5) * RR_PIX;
    const bool is_diff = sub >= 16;
    const int rr = sub & 15;
    const bool ray_live = is_diff ? rr < P.n_diff : rr < P.n_spec;
    const int r = is_diff ? P.n_spec + rr : rr;
    const float* wg_net_in = P.net_in + wg_pix0 * P.c_pad;
    const float* wg_alpha = P.alpha + wg_pix0;
    extern __shared__ __attribute__((aligned(16))) float rw_smem[];
    const int ni_need = P.ni_need;
    float* s_ni = rw_smem;                                  // [PIX_PER_WG][ni_need]
    float* s_w = rw_smem + PIX_PER_WG * ni_need;            // [PIX_PER_WG][c_w]: the weights leave as coalesced float4 rows
    const int wg_valid = (int)min((long)PIX_PER_WG, P.npix - wg_pix0);
    {
        const int q_ni = ni_need >> 2;
        const float inv_q = 1.0f / (float)q_ni;
        for (int i = threadIdx.x; i < wg_valid * q_ni; i += 256) {
            const int p = (int)(((float)i + 0.5f) * inv_q), q4 = i - __mul24(p, q_ni);
            reinterpret_cast<float4*>(s_ni)[i] = *reinterpret_cast<const float4*>(wg_net_in + (unsigned)(__mul24(p, P.c_pad) + 4 * q4));
        }
    }
    float al[RR_PIX];
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) al[k] = (lp0 + k < wg_valid) ? wg_alpha[lp0 + k] : 0.0f;
    __syncthreads();
    const int lp_w3 = P.lp_w * 3;
    const float inv_n = is_diff ? (P.n_diff > 0 ? 1.0f / (float)P.n_diff : 0.0f) : 1.0f / (float)P.n_spec;
    const int alb_ch = 3 * (P.n_spec + P.n_diff) + 6 + (is_diff ? P.alb_diff_ch : P.alb_spec_ch);
    const int n_cols = 3 * (P.n_spec + P.n_diff);
#pragma unroll
    for (int k = 0; k < RR_PIX; k++) {
        if (lp0 + k >= wg_valid) continue;
        float* out = s_w + __mul24(lp0 + k, P.c_w);
        if (sub == 31) for (int c = n_cols; c < P.c_w; c++) out[c] = 0.0f;        // padding columns (lane 31 never holds a ray)
        if (!ray_live) continue;
        const float* d = s_ni + __mul24(lp0 + k, ni_need) + 3 * r;
        float u = __builtin_fmaf(fast_atan2f(d[2], d[0]), 0.5f / RNR_PI_F, 0.5f);      // render.py:96-102; network.py:469-470
        float v = fast_acosf(d[1]) * (1.0f / RNR_PI_F);
        const float bg = (al[k] == 0.0f) ? 1.0f : 0.0f;
        u = u * al[k] - bg;
        v = v * al[k] - bg;
        const float x = fminf(u * (float)P.lp_w, (float)(P.lp_w - 1));                  // network.py:497; misc.py:5-42
        const float y = fminf(v * (float)P.lp_h, (float)(P.lp_h - 1));
        const Taps t = bilinear_taps(x, y, P.lp_w, P.lp_h);
        const unsigned r0 = __umul24((unsigned)t.y0, (unsigned)lp_w3), r1 = __umul24((unsigned)t.y1, (unsigned)lp_w3);
        const unsigned q0 = __umul24((unsigned)t.x0, 3u), q1 = __umul24((unsigned)t.x1, 3u);
        const float* l00 = P.lp + (r0 + q0);
        const float* l10 = P.lp + (r1 + q0);
        const float* l01 = P.lp + (r0 + q1);
        const float* l11 = P.lp + (r1 + q1);
        const float* alb = s_ni + __mul24(lp0 + k, ni_need) + alb_ch;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float col = __builtin_fmaf(l11[c], t.w11, __builtin_fmaf(l01[c], t.w01, __builtin_fmaf(l10[c], t.w10, l00[c] * t.w00)));
            out[3 * r + c] = (al[k] == 0.0f) ? 0.0f : alb[c] * (col * inv_n);
        }
    }
    __syncthreads();
    {
        const int n4 = wg_valid * P.c_w / 4;               // c_w is a multiple of 4 (the out layer's channel stride)
        float4* dst = reinterpret_cast<float4*>(P.ray_w + wg_pix0 * P.c_w);
        for (int i = threadIdx.x; i < n4; i += 256) dst[i] = reinterpret_cast<const float4*>(s_w)[i];
    }
}

// ------------------------------------------------------------------------------------------------
// spherical harmonics, float64 internally like the reference's numpy/pyshtools path
// ------------------------------------------------------------------------------------------------
constexpr int SH_LMAX_MAX = 16;

__global__ void __launch_bounds__(128)
sh_basis_kernel(const float* __restrict__ dirs, float* __restrict__ out, int n, int lmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = dirs[i * 3 + 0], y = dirs[i * 3 + 1], z = dirs[i * 3 + 2];
    // sph_harm.py:14-15, 54-57: azimuth = atan2(y,x); colatitude = pi/2 - atan2(z, sqrt(x^2+y^2))
    const double rho = sqrt(x * x + y * y);
    const double phi = atan2(y, x);
    const double rr = sqrt(rho * rho + z * z);
    const double ct = rr > 0.0 ? z / rr : 1.0, st = rr > 0.0 ? rho / rr : 0.0;
    const int nb = (lmax + 1) * (lmax + 1);
    float* o = out + (size_t)i * nb;
    const double fourpi = 12.566370614359172;
    // P_m^m upward in m; for each m run the l-recurrence and emit columns (l, +m) and (l, -m)
    double pmm = 1.0;
    for (int m = 0; m <= lmax; m++) {
        if (m > 0) pmm *= (2 * m - 1) * st;      // no Condon-Shortley phase
        const double cm = cos(m * phi), sm = sin(m * phi);
        double p_lm2 = 0.0, p_lm1 = 0.0;
        for (int l = m; l <= lmax; l++) {
            double p;
            if (l == m) p = pmm;
            else if (l == m + 1) p = ct * (2 * m + 1) * pmm;
            else p = ((2 * l - 1) * ct * p_lm1 - (l + m - 1) * p_lm2) / (l - m);
            p_lm2 = p_lm1; p_lm1 = p;
            // (l-m)!/(l+m)!
            double ratio = 1.0;
            for (int k = l - m + 1; k <= l + m; k++) ratio /= (double)k;
            const double norm = sqrt((m == 0 ? 1.0 : 2.0) * (2 * l + 1) / fourpi * ratio);
            const int base = l * l + l;          // column of (l, 0)
            o[base + m] = (float)(norm * p * cm);
            if (m > 0) o[base - m] = (float)(norm * p * sm);
        }
    }
}

// out[s,c] = sum_b basis[s,b] * coeff[b,c].  64 samples per workgroup: their basis rows are one contiguous block of
// 64*nb floats, staged through LDS with coalesced loads (a lane-per-output version reads rows nb floats apart).
constexpr int SHR_ROWS = 64;
__device__ __forceinline__ void sh_reconstruct_block(int block, float* sh_lds, const float* __restrict__ basis,
                                                     const float* __restrict__ coeff, float* __restrict__ out, int ns, int nb, int nc) {
    float* bl = sh_lds;                             // [SHR_ROWS*nb] basis rows, then [nb*nc] coefficients
    float* cl = sh_lds + SHR_ROWS * nb;
    const int s0 = block * SHR_ROWS;
    const int rows = min(SHR_ROWS, ns - s0);
    for (int i = threadIdx.x; i < rows * nb; i += blockDim.x) bl[i] = basis[(size_t)s0 * nb + i];
    for (int i = threadIdx.x; i < nb * nc; i += blockDim.x) cl[i] = coeff[i];
    __syncthreads();
    for (int o = threadIdx.x; o < rows * nc; o += blockDim.x) {
        const int s = o / nc, c = o - s * nc;
        float acc = 0.f;
        for (int b = 0; b < nb; b++) acc += bl[s * nb + b] * cl[b * nc + c];
        out[(size_t)(s0 + s) * nc + c] = acc;
    }
}

__global__ void __launch_bounds__(256)
sh_reconstruct_kernel(const float* __restrict__ basis, const float* __restrict__ coeff, float* __restrict__ out,
                      int ns, int nb, int nc) {
    extern __shared__ float sh_lds[];
    sh_reconstruct_block(blockIdx.x, sh_lds, basis, coeff, out, ns, nb, nc);
}

// ------------------------------------------------------------------------------------------------
// rnr_frame_prepare: the per-call preliminaries of a frame batch in ONE launch (r04; at one view per call they were six
// launches of ~5 us each in front of a 2.5 ms frame): block ranges of one grid run
//   [0, b_proj)            vertex projection straight from the [N,4,4] poses (no R / t copies);
//   [b_proj, b_tan)        per-face tangents, recomputed per call as render.get_TBN_map does (render.py:135-150);
//   [b_tan, b_lp)          the light probe reconstructed from SH coefficients (LightingSH.reconstruct_lp, network.py:622-627);
//   [b_lp, b_clear)        the per-call clearing of the rasterizer workspace (tile counters 0, depth keys ~0).
// The parts are independent of each other; each runs the code of its stand-alone kernel (same bits).
struct FramePrepParams {
    rnr_mesh mesh;
    const float* K; const float* pose; float* v_uvz; int nviews; float orig_size, eps;
    float* tangents;
    const float* lp_basis; const float* lp_coeff; float* lp_out; int lp_ns, lp_nb, lp_nc;
    uint4* counters; long counter_vec; uint4* keys; long key_vec;
    int b_proj, b_tan, b_lp, b_clear;
};
__global__ void __launch_bounds__(256)
frame_prepare_kernel(const FramePrepParams P) {
    extern __shared__ float sh_lds[];
    const int b = blockIdx.x;
    if (b < P.b_proj) {
        project_vertex<true>((long)b * 256 + threadIdx.x, P.mesh.v, P.K, P.pose, nullptr, nullptr, nullptr, nullptr, P.v_uvz,
                             P.nviews, P.mesh.num_vertices, P.orig_size, P.eps);
    } else if (b < P.b_tan) {
        face_tangent((b - P.b_proj) * 256 + threadIdx.x, P.mesh, P.tangents);
    } else if (b < P.b_lp) {
        sh_reconstruct_block(b - P.b_tan, sh_lds, P.lp_basis, P.lp_coeff, P.lp_out, P.lp_ns, P.lp_nb, P.lp_nc);
    } else {
        const long i = (long)(b - P.b_lp) * 256 + threadIdx.x;
        if (i < P.counter_vec) P.counters[i] = make_uint4(0u, 0u, 0u, 0u);
        else if (i - P.counter_vec < P.key_vec) P.keys[i - P.counter_vec] = make_uint4(~0u, ~0u, ~0u, ~0u);
    }
}

__global__ void __launch_bounds__(256)
sh_fit_kernel(const float* __restrict__ samples, const float* __restrict__ basis, float* __restrict__ out,
              int ns, int nb, int nc) {
    __shared__ float red[256];
    const int o = blockIdx.x;            // output (b, c)
    const int b = o / nc, c = o % nc;
    float acc = 0.f;
    for (int s = threadIdx.x; s < ns; s += blockDim.x) acc += samples[(size_t)s * nc + c] * basis[(size_t)s * nb + b];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[o] = red[0] * (4.0f * RNR_PI_F / (float)ns);
}

__global__ void __launch_bounds__(256)
interpolate_bilinear_kernel(const float* __restrict__ data, int h, int w, int c, const float* __restrict__ x,
                            const float* __restrict__ y, float* __restrict__ out, int32_t* __restrict__ taps, int n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)n * c) return;
    const int s = (int)(i / c), ch = (int)(i % c);
    const Taps t = bilinear_taps(x[s], y[s], w, h);
    const float v = data[((size_t)t.y0 * w + t.x0) * c + ch] * t.w00 + data[((size_t)t.y1 * w + t.x0) * c + ch] * t.w10 +
                    data[((size_t)t.y0 * w + t.x1) * c + ch] * t.w01 + data[((size_t)t.y1 * w + t.x1) * c + ch] * t.w11;
    out[i] = v;
    if (taps && ch == 0) {
        taps[s * 4 + 0] = t.x0; taps[s * 4 + 1] = t.y0; taps[s * 4 + 2] = t.x1; taps[s * 4 + 3] = t.y1;
    }
}

// ------------------------------------------------------------------------------------------------
// Stand-alone operator kernels for the drop-in Python API (network.py / render.py / camera.py functions called one
// at a time).  The fused pipeline does not use them: shade_inputs_kernel / ray_render_kernel compute the same
// quantities without the HBM round trips.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
view_dir_map_kernel(const float* __restrict__ proj_inv, const float* __restrict__ R_inv, float* __restrict__ out_world,
                    float* __restrict__ out_cam, long npix, int H, int W) {
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    const int hw = H * W;
    const int n = (int)(pix / hw), rem = (int)(pix % hw);
    const int row = rem / W, col = rem % W;
    const float* Pi = proj_inv + n * 9;
    const float* Ri = R_inv + n * 9;
    const float pu = (float)col + 0.5f, pv = (float)row + 0.5f;    // camera.py:19-20
    float3 dc = f3(-(Pi[0] * pu + Pi[1] * pv + Pi[2]), -(Pi[3] * pu + Pi[4] * pv + Pi[5]),
                   -(Pi[6] * pu + Pi[7] * pv + Pi[8]));
    dc = normalize3(dc);
    float3 vd = f3(Ri[0] * dc.x + Ri[1] * dc.y + Ri[2] * dc.z, Ri[3] * dc.x + Ri[4] * dc.y + Ri[5] * dc.z,
                   Ri[6] * dc.x + Ri[7] * dc.y + Ri[8] * dc.z);
    vd = normalize3(vd);
    out_world[pix * 3 + 0] = vd.x; out_world[pix * 3 + 1] = vd.y; out_world[pix * 3 + 2] = vd.z;
    if (out_cam) { out_cam[pix * 3 + 0] = dc.x; out_cam[pix * 3 + 1] = dc.y; out_cam[pix * 3 + 2] = dc.z; }
}

__global__ void __launch_bounds__(256)
tbn_map_kernel(const float* __restrict__ normal_map, const int32_t* __restrict__ face_index_map,
               const float* __restrict__ tangents, int num_faces, float* __restrict__ out, long npix) {
    const long pix = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    int fi = face_index_map[pix];
    if (fi < 0) fi += num_faces;
    const float3 tg = f3(tangents[fi * 3 + 0], tangents[fi * 3 + 1], tangents[fi * 3 + 2]);
    const float3 nm = normalize3(f3(normal_map[pix * 3 + 0], normal_map[pix * 3 + 1], normal_map[pix * 3 + 2]));
    const float3 bt = normalize3(cross3(nm, tg));
    const float3 tt = normalize3(cross3(bt, nm));
    float* o = out + pix * 9;      // [3,3] row-major, columns (T, B, N)  (render.py:164)
    o[0] = tt.x; o[1] = bt.x; o[2] = nm.x;
    o[3] = tt.y; o[4] = bt.y; o[5] = nm.y;
    o[6] = tt.z; o[7] = bt.z; o[8] = nm.z;
}

struct RaySamplerParams {
    const float* tbn; const float* view_tangent; const float* alpha;
    float* rays_dir; float* rays_uv; float* rays_dir_tangent;
    float piv[MAX_RAYS * 3];
    int n_rays, reflect;
    long npix;
};

__global__ void __launch_bounds__(256)
ray_sampler_kernel(const RaySamplerParams P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.npix * P.n_rays) return;
    const long pix = i / P.n_rays;
    const int r = (int)(i % P.n_rays);
    const float* M = P.tbn + pix * 9;
    const float a = P.alpha[pix];
    const float3 pv = f3(P.piv[r * 3 + 0], P.piv[r * 3 + 1], P.piv[r * 3 + 2]);
    float3 lt;
    if (P.reflect) {
        const float3 v = f3(P.view_tangent[pix * 3 + 0], P.view_tangent[pix * 3 + 1], P.view_tangent[pix * 3 + 2]);
        const float s = dot3(pv, v) * 2.0f;
        lt = normalize3(f3(s * pv.x - v.x, s * pv.y - v.y, s * pv.z - v.z));
        lt = f3(lt.x * a, lt.y * a, lt.z * a);
    } else {
        lt = pv;
    }
    float3 d = f3(M[0] * lt.x + M[1] * lt.y + M[2] * lt.z, M[3] * lt.x + M[4] * lt.y + M[5] * lt.z,
                  M[6] * lt.x + M[7] * lt.y + M[8] * lt.z);
    d = normalize3(d);
    const int R = P.n_rays;
    P.rays_dir[(pix * 3 + 0) * R + r] = d.x;
    P.rays_dir[(pix * 3 + 1) * R + r] = d.y;
    P.rays_dir[(pix * 3 + 2) * R + r] = d.z;
    if (P.rays_dir_tangent) {
        P.rays_dir_tangent[(pix * 3 + 0) * R + r] = lt.x;
        P.rays_dir_tangent[(pix * 3 + 1) * R + r] = lt.y;
        P.rays_dir_tangent[(pix * 3 + 2) * R + r] = lt.z;
    }
    float u = atan2f(d.z, d.x) * 0.5f / RNR_PI_F + 0.5f;
    float v = acosf(d.y) * 1.0f / RNR_PI_F;
    const float bg = (a == 0.0f) ? 1.0f : 0.0f;
    P.rays_uv[(pix * 2 + 0) * R + r] = u * a - bg;
    P.rays_uv[(pix * 2 + 1) * R + r] = v * a - bg;
}

struct TexMapParams {
    const float* uv_map; const float* sh;   // sh may be NULL
    const float* tex[MAX_LEVELS];
    int tex_size[MAX_LEVELS];
    int num_levels, C, sh_start;
    float* out;                              // [N,C,H,W]
    long npix; int hw;
};

// any channel count (DNR uses C = 30): one lane per (channel, pixel), pixel fastest so the NCHW store coalesces
__global__ void __launch_bounds__(256)
texture_mapper_kernel(const TexMapParams P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P.npix * P.C) return;
    const long n = i / ((long)P.C * P.hw);
    const long rem = i % ((long)P.C * P.hw);
    const int c = (int)(rem / P.hw);
    const long p = rem % P.hw;
    const long pix = n * P.hw + p;
    const float u = P.uv_map[pix * 2 + 0], v = P.uv_map[pix * 2 + 1];
    float acc = 0.f;
    for (int l = 0; l < P.num_levels; l++) {
        const int s = P.tex_size[l];
        const float sm1 = (float)(s - 1);
        const Taps t = bilinear_taps(u * sm1, sm1 - v * sm1, s, s);
        const float* tex = P.tex[l];
        const float lv = tex[((size_t)t.y0 * s + t.x0) * P.C + c] * t.w00 + tex[((size_t)t.y1 * s + t.x0) * P.C + c] * t.w10 +
                         tex[((size_t)t.y0 * s + t.x1) * P.C + c] * t.w01 + tex[((size_t)t.y1 * s + t.x1) * P.C + c] * t.w11;
        acc = (l == 0) ? lv : acc + lv;
    }
    if (P.sh && c >= P.sh_start && c < P.sh_start + 9) acc *= P.sh[pix * 9 + (c - P.sh_start)];
    P.out[i] = acc;
}

// RayRenderer.forward with API-shaped inputs (network.py:481-527): one lane per (pixel, channel)
struct RayApiParams {
    const float* rays_uv;    // [N,H,W,2,R]
    const float* rays_lt;    // [N,R,C,H,W]
    const float* lp;         // [Nl,Hl,Wl,C], Nl = 1 or N
    const float* alb_spec;   // [N,C,H,W]
    const float* alb_diff;   // [N,C,H,W] or NULL
    int lp_n, lp_h, lp_w, C, R, n_diff, no_albedo, separate;
    float lp_scale;
    float* out; float* out_spec; float* out_diff; float* ltt_spec; float* ltt_diff; float* rays_color;  // last may be NULL
    long npix; int hw;
};

__global__ void __launch_bounds__(256)
ray_renderer_api_kernel(const RayApiParams P) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over N*C*hw
    if (i >= P.npix * P.C) return;
    const long n = i / ((long)P.C * P.hw);
    const long rem = i % ((long)P.C * P.hw);
    const int c = (int)(rem / P.hw);
    const long p = rem % P.hw;
    const long pix = n * P.hw + p;
    const int n_spec = P.R - P.n_diff;
    const float* lp = P.lp + (P.lp_n == 1 ? 0 : (size_t)n * P.lp_h * P.lp_w * P.C);
    float ss = 0.f, sd = 0.f;
    for (int r = 0; r < P.R; r++) {
        const float u = P.rays_uv[(pix * 2 + 0) * P.R + r], v = P.rays_uv[(pix * 2 + 1) * P.R + r];
        const float x = fminf(u * (float)P.lp_w, (float)(P.lp_w - 1));
        const float y = fminf(v * (float)P.lp_h, (float)(P.lp_h - 1));
        const Taps t = bilinear_taps(x, y, P.lp_w, P.lp_h);
        const float col = (lp[((size_t)t.y0 * P.lp_w + t.x0) * P.C + c] * P.lp_scale) * t.w00 +
                          (lp[((size_t)t.y1 * P.lp_w + t.x0) * P.C + c] * P.lp_scale) * t.w10 +
                          (lp[((size_t)t.y0 * P.lp_w + t.x1) * P.C + c] * P.lp_scale) * t.w01 +
                          (lp[((size_t)t.y1 * P.lp_w + t.x1) * P.C + c] * P.lp_scale) * t.w11;
        const size_t li = (((size_t)n * P.R + r) * P.C + c) * P.hw + p;
        if (P.rays_color) P.rays_color[li] = col;
        const float prod = P.rays_lt[li] * col;
        if (r < n_spec) ss += prod; else sd += prod;
    }
    const float ls = ss / (float)n_spec;
    const float as = P.alb_spec[i];
    const float os = P.no_albedo ? ls : as * ls;
    float ld = 0.f, od = 0.f;
    if (P.n_diff > 0) {
        ld = sd / (float)P.n_diff;
        od = P.no_albedo ? ld : ((P.separate && P.alb_diff) ? P.alb_diff[i] : as) * ld;
    }
    P.out[i] = os + od;
    if (P.out_spec) P.out_spec[i] = os;
    if (P.out_diff) P.out_diff[i] = od;
    if (P.ltt_spec) P.ltt_spec[i] = ls;
    if (P.ltt_diff) P.ltt_diff[i] = ld;
}

// The same operator for <= 4 colour channels (the repo's probes have 3): a workgroup owns 64 consecutive pixels; thread (pixel lane,
// ray quarter q) takes the rays q, q + 4, ... of its pixel for ALL channels — the taps of a (pixel, ray) are computed once, rays_lt /
// rays_color are touched along the pixels (128-byte rows per wave instruction), the uv rows come through an LDS tile (row stride
// 2 R + 1: conflict-free) — and the four partial sums of a pixel meet in LDS.  Sums over the rays are four interleaved partial
// sums instead of one chain: <= 1e-6 from the one-thread-per-(channel, pixel) form above (which reads uv with a stride of 2 R floats
// and recomputes the taps per channel).
__global__ void __launch_bounds__(256)
ray_renderer_api_tiled_kernel(const RayApiParams P) {
    extern __shared__ float ra_sm[];
    const int R = P.R, C = P.C, n_spec = P.R - P.n_diff;
    const long pix0 = (long)blockIdx.x * 64;
    const int valid = (int)min((long)64, P.npix - pix0);
    const int lane = threadIdx.x & 63, qtr = threadIdx.x >> 6;
    const int urow = 2 * R + 1;
    for (int i = threadIdx.x; i < valid * 2 * R; i += 256) {
        const int px = i / (2 * R), k = i - px * 2 * R;
        ra_sm[px * urow + k] = P.rays_uv[pix0 * 2 * R + i];
    }
    __syncthreads();
    const long pix = pix0 + lane;
    const bool live = lane < valid;
    const long n = live ? pix / P.hw : 0, p = live ? pix % P.hw : 0;
    const float* lp = P.lp + (P.lp_n == 1 ? 0 : (size_t)n * P.lp_h * P.lp_w * C);
    float ss[4] = {0.f, 0.f, 0.f, 0.f}, sd[4] = {0.f, 0.f, 0.f, 0.f};
    if (live) {
        for (int r = qtr; r < R; r += 4) {
            const float u = ra_sm[lane * urow + r], v = ra_sm[lane * urow + R + r];
            const float x = fminf(u * (float)P.lp_w, (float)(P.lp_w - 1));
            const float y = fminf(v * (float)P.lp_h, (float)(P.lp_h - 1));
            const Taps t = bilinear_taps(x, y, P.lp_w, P.lp_h);
            const float* l00 = lp + ((size_t)t.y0 * P.lp_w + t.x0) * C;
            const float* l10 = lp + ((size_t)t.y1 * P.lp_w + t.x0) * C;
            const float* l01 = lp + ((size_t)t.y0 * P.lp_w + t.x1) * C;
            const float* l11 = lp + ((size_t)t.y1 * P.lp_w + t.x1) * C;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (c >= C) break;
                const float col = (l00[c] * P.lp_scale) * t.w00 + (l10[c] * P.lp_scale) * t.w10 + (l01[c] * P.lp_scale) * t.w01 +
                                  (l11[c] * P.lp_scale) * t.w11;
                const size_t li = (((size_t)n * R + r) * C + c) * P.hw + p;
                if (P.rays_color) P.rays_color[li] = col;
                const float prod = P.rays_lt[li] * col;
                if (r < n_spec) ss[c] += prod; else sd[c] += prod;
            }
        }
    }
    __syncthreads();        // the uv tile is dead: the partial sums take its place, [quarter][spec | diff][channel][lane]
#pragma unroll
    for (int c = 0; c < 4; c++) {
        ra_sm[((qtr * 2 + 0) * 4 + c) * 64 + lane] = ss[c];
        ra_sm[((qtr * 2 + 1) * 4 + c) * 64 + lane] = sd[c];
    }
    __syncthreads();
    const int c = qtr;      // thread (lane, c) finishes channel c of its pixel
    if (!live || c >= C) return;
    float s_spec = 0.f, s_diff = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < 4; q4++) {
        s_spec += ra_sm[((q4 * 2 + 0) * 4 + c) * 64 + lane];
        s_diff += ra_sm[((q4 * 2 + 1) * 4 + c) * 64 + lane];
    }
    const long i = (n * C + c) * P.hw + p;
    const float ls = s_spec / (float)n_spec;
    const float as = P.alb_spec[i];
    const float os = P.no_albedo ? ls : as * ls;
    float ld = 0.f, od = 0.f;
    if (P.n_diff > 0) {
        ld = s_diff / (float)P.n_diff;
        od = P.no_albedo ? ld : ((P.separate && P.alb_diff) ? P.alb_diff[i] : as) * ld;
    }
    P.out[i] = os + od;
    if (P.out_spec) P.out_spec[i] = os;
    if (P.out_diff) P.out_diff[i] = od;
    if (P.ltt_spec) P.ltt_spec[i] = ls;
    if (P.ltt_diff) P.ltt_diff[i] = ld;
}

// ---- layout helpers --------------------------------------------------------------------------------
// one element per thread: the form for channel counts whose 64-pixel tile does not fit 64 KB of LDS (c_pad > 240)
__global__ void __launch_bounds__(256)
nchw_to_nhwc_wide_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int hw, int c_pad, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*hw*c_pad
    if (i >= total) return;
    const int ch = (int)(i % c_pad);
    const long np = i / c_pad;
    const long n = np / hw, p = np % hw;
    out[i] = ch < c ? in[(n * c + ch) * hw + p] : 0.0f;
}
__global__ void __launch_bounds__(256)
nhwc_to_nchw_wide_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ bias,
                         int apply_tanh, int c, int hw, int c_pad, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*c*hw (output order)
    if (i >= total) return;
    const long p = i % hw;
    const long nc = i / hw;
    const int ch = (int)(nc % c);
    const long n = nc / c;
    float v = in[(n * hw + p) * c_pad + ch];
    if (bias) v += bias[ch];
    if (apply_tanh) v = tanhf(v);
    out[i] = v;
}

__global__ void __launch_bounds__(256)
nchw_to_nhwc_kernel(const float* __restrict__ in, float* __restrict__ out, int c, int hw, int c_pad, long total) {
    // r05: through an LDS tile of 64 pixels x c_pad channels — reads coalesced along the pixels of a channel (NCHW rows), writes
    // coalesced along the channels of a pixel (the one-element-per-thread form reads with a stride of hw floats: 0.143 ms for the
    // 108-channel network input at 512^2, the layout change in front of the drop-in RenderingNet)
    extern __shared__ float lt_tile[];          // [c_pad][65]
    const long pix0 = (long)blockIdx.x * 64;    // over n * hw (hw is a multiple of 64 or the tail is masked)
    const long npix = total / c_pad;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long pp = pix0 + lane;
    const long n = pp / hw, p = pp % hw;
    for (int ch = w; ch < c_pad; ch += 4)
        lt_tile[ch * 65 + lane] = (ch < c && pp < npix) ? in[(n * c + ch) * hw + p] : 0.0f;
    __syncthreads();
    const int per = 64 * c_pad;
    float* dst = out + pix0 * c_pad;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int px = i / c_pad, ch = i - px * c_pad;
        if (pix0 + px < npix) dst[i] = lt_tile[ch * 65 + px];
    }
}

__global__ void __launch_bounds__(256)
nhwc_to_nchw_kernel(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ bias,
                    int apply_tanh, int c, int hw, int c_pad, long total) {
    // the reverse through the same tile: reads coalesced along the channels of a pixel, writes along the pixels of a channel
    extern __shared__ float lt_tile[];          // [c_pad][65]
    const long npix = total / c;                // total = n * c * hw
    const long pix0 = (long)blockIdx.x * 64;
    const int per = 64 * c_pad;
    const float* src = in + pix0 * c_pad;
    for (int i = threadIdx.x; i < per; i += 256) {
        const int px = i / c_pad, ch = i - px * c_pad;
        if (pix0 + px < npix) lt_tile[ch * 65 + px] = src[i];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long pp = pix0 + lane;
    if (pp >= npix) return;
    const long n = pp / hw, p = pp % hw;
    for (int ch = w; ch < c; ch += 4) {
        float v = lt_tile[ch * 65 + lane];
        if (bias) v += bias[ch];
        if (apply_tanh) v = tanhf(v);
        out[(n * c + ch) * hw + p] = v;
    }
}

}  // namespace rnr

using namespace rnr;

extern "C" int rnr_project_vertices(const float* vertices, const float* K, const float* R, const float* t,
                                    const float* dist_coeffs, const float* offset, const float* scale,
                                    float* out, int num_views, int num_vertices, float orig_size, float eps,
                                    void* stream) {
    RNR_REQUIRE(vertices && K && R && t && out, "rnr_project_vertices: null pointer argument");
    RNR_REQUIRE((offset == nullptr) == (scale == nullptr), "rnr_project_vertices: offset and scale go together");
    RNR_REQUIRE(num_views > 0 && num_vertices > 0, "rnr_project_vertices: bad sizes");
    const long total = (long)num_views * num_vertices;
    hipLaunchKernelGGL(project_vertices_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), vertices, K, R, t, dist_coeffs, offset, scale, out, num_views,
                       num_vertices, orig_size, eps);
    return check_launch("project_vertices_kernel");
}

extern "C" int rnr_face_tangents(const rnr_mesh* mesh, float* out, void* stream) {
    RNR_REQUIRE(mesh && out && mesh->v && mesh->vt && mesh->f_v_idx && mesh->f_vt_idx && mesh->num_faces > 0,
                "rnr_face_tangents: incomplete mesh");
    hipLaunchKernelGGL(face_tangents_kernel, dim3((mesh->num_faces + 255) / 256), dim3(256), 0,
                       as_stream(stream), *mesh, out);
    return check_launch("face_tangents_kernel");
}

extern "C" int rnr_shade_inputs(const int32_t* face_index_map, const float* alpha, const float* uv_map,
                                const float* normal_map, const float* face_tangents, int num_faces,
                                const float* proj_inv, const float* R_inv, const float* const* textures_host,
                                const int* tex_sizes_host, int num_levels, int tex_channels, int sh_start_ch,
                                const rnr_rays* rays, float* net_in, int c_pad, float* rays_uv,
                                float* neural_img, float* sh_basis_map, int num_views, int height, int width,
                                void* stream) {
    RNR_REQUIRE(face_index_map && alpha && uv_map && normal_map && face_tangents && proj_inv && R_inv &&
                    textures_host && tex_sizes_host && rays && net_in,
                "rnr_shade_inputs: null pointer argument");
    RNR_REQUIRE(num_levels >= 1 && num_levels <= MAX_LEVELS, "rnr_shade_inputs: num_levels %d not in [1,%d]",
                num_levels, MAX_LEVELS);
    RNR_REQUIRE(tex_channels > 0 && tex_channels % 4 == 0, "rnr_shade_inputs: texture channels must be a multiple of 4");
    RNR_REQUIRE(rays->num_spec >= 0 && rays->num_spec <= MAX_RAYS && rays->num_diff >= 0 && rays->num_diff <= MAX_RAYS,
                "rnr_shade_inputs: at most %d rays per sampler", MAX_RAYS);
    const int c_in = 3 * (rays->num_spec + rays->num_diff) + 6 + tex_channels;
    RNR_REQUIRE(c_pad >= c_in && c_pad % 4 == 0, "rnr_shade_inputs: c_pad %d < %d or not a multiple of 4", c_pad, c_in);
    RNR_REQUIRE((3 * (rays->num_spec + rays->num_diff) + 6) % 4 == 0,
                "rnr_shade_inputs: 3*rays+6 must be a multiple of 4 (texture channels are written as float4)");
    RNR_REQUIRE(sh_start_ch < 0 || sh_start_ch + 9 <= tex_channels, "rnr_shade_inputs: sh_start_ch + 9 > channels");
    ShadeParams P = {};
    P.face_index_map = face_index_map; P.alpha = alpha; P.uv_map = uv_map; P.normal_map = normal_map;
    P.tangents = face_tangents; P.num_faces = num_faces; P.proj_inv = proj_inv; P.R_inv = R_inv;
    for (int l = 0; l < num_levels; l++) {
        RNR_REQUIRE(textures_host[l] && tex_sizes_host[l] >= 2, "rnr_shade_inputs: bad texture level %d", l);
        RNR_REQUIRE((size_t)tex_sizes_host[l] * tex_sizes_host[l] * tex_channels * sizeof(float) < (1ull << 31),
                    "rnr_shade_inputs: texture level %d exceeds 2 GiB (32-bit texel offsets)", l);
        P.tex[l] = textures_host[l];
        P.tex_size[l] = tex_sizes_host[l];
    }
    P.num_levels = num_levels; P.C = tex_channels; P.sh_start = sh_start_ch;
    P.n_spec = rays->num_spec; P.n_diff = rays->num_diff;
    for (int r = 0; r < rays->num_spec; r++)
        for (int k = 0; k < 3; k++) P.piv_spec[r * 3 + k] = rays->pivots_spec_host[k * rays->num_spec + r];
    for (int r = 0; r < rays->num_diff; r++)
        for (int k = 0; k < 3; k++) P.piv_diff[r * 3 + k] = rays->pivots_diff_host[k * rays->num_diff + r];
    P.net_in = net_in; P.c_pad = c_pad; P.rays_uv = rays_uv; P.neural_img = neural_img; P.sh_basis_map = sh_basis_map;
    P.npix = (long)num_views * height * width; P.H = height; P.W = width;
    const size_t lds = (size_t)(SH_PIX * c_pad + SH_PIX * GEO) * sizeof(float);
    RNR_REQUIRE(lds <= 160 * 1024, "rnr_shade_inputs: c_pad too large for LDS");
    const unsigned blocks = (unsigned)((P.npix + SH_PIX - 1) / SH_PIX);
    hipLaunchKernelGGL(shade_inputs_kernel, dim3(blocks), dim3(SH_THREADS), lds, as_stream(stream), P);
    return check_launch("shade_inputs_kernel");
}

extern "C" int rnr_ray_render(const float* unet_raw, int c_out_pad, const float* bias, const float* net_in,
                              int c_pad, const float* alpha, const float* lp, int lp_h, int lp_w, int num_spec,
                              int num_diff, int albedo_diff_ch, int albedo_spec_ch, float* image, int num_views,
                              int height, int width, void* stream) {
    RNR_REQUIRE(unet_raw && bias && net_in && alpha && lp && image, "rnr_ray_render: null pointer argument");
    RNR_REQUIRE(num_spec >= 1 && num_spec <= 16 && num_diff >= 0 && num_diff <= 16,
                "rnr_ray_render: ray counts must be <= 16 per group (got %d, %d)", num_spec, num_diff);
    RNR_REQUIRE(lp_h >= 2 && lp_w >= 2, "rnr_ray_render: bad light-probe size");
    RayParams P;
    P.unet_raw = unet_raw; P.c_out_pad = c_out_pad; P.bias = bias; P.net_in = net_in; P.c_pad = c_pad;
    P.alpha = alpha; P.lp = lp; P.lp_h = lp_h; P.lp_w = lp_w; P.n_spec = num_spec; P.n_diff = num_diff;
    P.alb_diff_ch = albedo_diff_ch; P.alb_spec_ch = albedo_spec_ch; P.image = image;
    P.npix = (long)num_views * height * width; P.hw = height * width;
    const int alb_hi = (albedo_diff_ch > albedo_spec_ch ? albedo_diff_ch : albedo_spec_ch) + 3;
    P.ni_need = (3 * (num_spec + num_diff) + 6 + alb_hi + 3) / 4 * 4;
    RNR_REQUIRE(P.ni_need <= c_pad && c_pad % 4 == 0 && c_out_pad % 4 == 0, "rnr_ray_render: channel strides must be multiples of 4 and cover the albedo channels");
    const long lanes = (P.npix + RR_PIX - 1) / RR_PIX * 32;
    const size_t lds = (size_t)(8 * RR_PIX) * (size_t)(c_out_pad + P.ni_need) * sizeof(float);
    RNR_REQUIRE(lds <= 64 * 1024, "rnr_ray_render: rows too wide for the LDS staging (%zu bytes)", lds);
    hipLaunchKernelGGL(ray_render_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), lds, as_stream(stream), P);
    return check_launch("ray_render_kernel");
}

extern "C" int rnr_ray_weights(const float* net_in, int c_pad, const float* alpha, const float* lp, int lp_h, int lp_w,
                               int num_spec, int num_diff, int albedo_diff_ch, int albedo_spec_ch, float* ray_w, int c_w,
                               int num_views, int height, int width, void* stream) {
    RNR_REQUIRE(net_in && alpha && lp && ray_w, "rnr_ray_weights: null pointer argument");
    RNR_REQUIRE(num_spec >= 1 && num_spec <= 16 && num_diff >= 0 && num_diff <= 15,
                "rnr_ray_weights: at most 16 specular / 15 diffuse rays (got %d, %d)", num_spec, num_diff);
    RNR_REQUIRE(lp_h >= 2 && lp_w >= 2, "rnr_ray_weights: bad light-probe size");
    RNR_REQUIRE(c_w >= 3 * (num_spec + num_diff), "rnr_ray_weights: c_w %d < 3 * rays", c_w);
    RayWeightParams P;
    P.net_in = net_in; P.c_pad = c_pad; P.alpha = alpha; P.lp = lp; P.lp_h = lp_h; P.lp_w = lp_w;
    P.n_spec = num_spec; P.n_diff = num_diff; P.alb_diff_ch = albedo_diff_ch; P.alb_spec_ch = albedo_spec_ch;
    P.ray_w = ray_w; P.c_w = c_w; P.npix = (long)num_views * height * width;
    const int alb_hi = (albedo_diff_ch > albedo_spec_ch ? albedo_diff_ch : albedo_spec_ch) + 3;
    P.ni_need = (3 * (num_spec + num_diff) + 6 + alb_hi + 3) / 4 * 4;
    RNR_REQUIRE(P.ni_need <= c_pad && c_pad % 4 == 0, "rnr_ray_weights: channel stride must be a multiple of 4 and cover the albedo channels");
    const long lanes = (P.npix + RR_PIX - 1) / RR_PIX * 32;
    RNR_REQUIRE(c_w % 4 == 0, "rnr_ray_weights: c_w must be a multiple of 4");
    const size_t lds = (size_t)(8 * RR_PIX) * (size_t)(P.ni_need + c_w) * sizeof(float);
    hipLaunchKernelGGL(ray_weights_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), lds, as_stream(stream), P);
    return check_launch("ray_weights_kernel");
}

extern "C" int rnr_sh_basis(const float* dirs, float* out, int n, int lmax, void* stream) {
    RNR_REQUIRE(dirs && out && n > 0, "rnr_sh_basis: bad arguments");
    RNR_REQUIRE(lmax >= 0 && lmax <= SH_LMAX_MAX, "rnr_sh_basis: lmax %d not in [0,%d]", lmax, SH_LMAX_MAX);
    hipLaunchKernelGGL(sh_basis_kernel, dim3((n + 127) / 128), dim3(128), 0, as_stream(stream), dirs, out, n, lmax);
    return check_launch("sh_basis_kernel");
}

extern "C" int rnr_sh_reconstruct(const float* basis, const float* coeff, float* out, int num_samples,
                                  int num_basis, int num_channels, void* stream) {
    RNR_REQUIRE(basis && coeff && out && num_samples > 0 && num_basis > 0 && num_channels > 0,
                "rnr_sh_reconstruct: bad arguments");
    const size_t lds = (size_t)(SHR_ROWS * num_basis + num_basis * num_channels) * sizeof(float);
    RNR_REQUIRE(lds <= 64 * 1024, "rnr_sh_reconstruct: num_basis * (64 + num_channels) floats must fit 64 KiB of LDS");
    hipLaunchKernelGGL(sh_reconstruct_kernel, dim3((unsigned)((num_samples + SHR_ROWS - 1) / SHR_ROWS)), dim3(256), lds,
                       as_stream(stream), basis, coeff, out, num_samples, num_basis, num_channels);
    return check_launch("sh_reconstruct_kernel");
}

extern "C" int rnr_frame_prepare(const rnr_mesh* mesh, const float* K, const float* pose, int num_views, int image_size,
                                 float eps, float* v_uvz, float* tangents, const float* lp_basis, const float* lp_coeff,
                                 float* light_probe, int lp_samples, int lp_num_basis, int lp_channels,
                                 void* gbuffer_workspace, void* stream) {
    RNR_REQUIRE(mesh && mesh->v && mesh->num_vertices > 0 && mesh->num_faces > 0, "rnr_frame_prepare: incomplete mesh");
    RNR_REQUIRE(num_views > 0 && image_size > 0, "rnr_frame_prepare: bad sizes");
    RNR_REQUIRE(!v_uvz || (K && pose), "rnr_frame_prepare: the projection needs K and pose");
    RNR_REQUIRE(!tangents || (mesh->vt && mesh->f_v_idx && mesh->f_vt_idx), "rnr_frame_prepare: tangents need vt and the index arrays");
    RNR_REQUIRE(!light_probe || (lp_basis && lp_coeff && lp_samples > 0 && lp_num_basis > 0 && lp_channels > 0),
                "rnr_frame_prepare: the light probe needs basis, coefficients and sizes");
    FramePrepParams P = {};
    P.mesh = *mesh; P.K = K; P.pose = pose; P.v_uvz = v_uvz; P.nviews = num_views; P.orig_size = (float)image_size; P.eps = eps;
    P.tangents = tangents;
    P.lp_basis = lp_basis; P.lp_coeff = lp_coeff; P.lp_out = light_probe; P.lp_ns = lp_samples; P.lp_nb = lp_num_basis;
    P.lp_nc = lp_channels;
    size_t lds = 0;
    long nb = 0;
    if (v_uvz) nb += ((long)num_views * mesh->num_vertices + 255) / 256;
    P.b_proj = (int)nb;
    if (tangents) nb += (mesh->num_faces + 255) / 256;
    P.b_tan = (int)nb;
    if (light_probe) {
        lds = (size_t)(SHR_ROWS * lp_num_basis + lp_num_basis * lp_channels) * sizeof(float);
        RNR_REQUIRE(lds <= 64 * 1024, "rnr_frame_prepare: num_basis * (64 + num_channels) floats must fit 64 KiB of LDS");
        nb += (lp_samples + SHR_ROWS - 1) / SHR_ROWS;
    }
    P.b_lp = (int)nb;
    if (gbuffer_workspace) {
        gbuffer_clear_regions(gbuffer_workspace, num_views, mesh->num_faces, image_size, &P.counters, &P.counter_vec, &P.keys,
                              &P.key_vec);
        nb += (P.counter_vec + P.key_vec + 255) / 256;
    }
    P.b_clear = (int)nb;
    RNR_REQUIRE(nb > 0 && nb < (1L << 31), "rnr_frame_prepare: nothing to do / grid too large");
    hipLaunchKernelGGL(frame_prepare_kernel, dim3((unsigned)nb), dim3(256), lds, as_stream(stream), P);
    return check_launch("frame_prepare_kernel");
}

extern "C" int rnr_sh_fit(const float* samples, const float* basis, float* out, int num_samples, int num_basis,
                          int num_channels, void* stream) {
    RNR_REQUIRE(samples && basis && out && num_samples > 0 && num_basis > 0 && num_channels > 0,
                "rnr_sh_fit: bad arguments");
    hipLaunchKernelGGL(sh_fit_kernel, dim3(num_basis * num_channels), dim3(256), 0, as_stream(stream), samples,
                       basis, out, num_samples, num_basis, num_channels);
    return check_launch("sh_fit_kernel");
}

extern "C" int rnr_interpolate_bilinear(const float* data, int h, int w, int c, const float* x, const float* y,
                                        float* out, int32_t* taps, int n, void* stream) {
    RNR_REQUIRE(data && x && y && out && h > 0 && w > 0 && c > 0 && n > 0, "rnr_interpolate_bilinear: bad arguments");
    const long total = (long)n * c;
    hipLaunchKernelGGL(interpolate_bilinear_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       as_stream(stream), data, h, w, c, x, y, out, taps, n);
    return check_launch("interpolate_bilinear_kernel");
}

// ------------------------------------------------------------------------------------------------
// Area-average image resize: what LightingLP.__init__ asks of cv2.resize(..., interpolation=cv2.INTER_AREA)
// (network.py:667) for the 1600 x 3200 light probes.  Restated from OpenCV's published algorithm (imgproc/resize.cpp):
//   * shrinking on both axes (scale = src/dst >= 1): separable box filter with fractional cell coverage
//     (computeResizeAreaTab): cell [d*scale, (d+1)*scale) of the source axis; a partially covered first pixel weighs
//     (ceil(f1) - f1), whole pixels 1, a partially covered last pixel min(f2 - floor(f2), 1), all divided by
//     cellWidth = min(scale, ssize - f1); partial coverage below 1e-3 is dropped.  Integer ratios reduce to the plain
//     box mean.
//   * otherwise (enlarging on an axis): OpenCV switches to its "area-mode" bilinear: s = floor(d*scale),
//     f = (d+1) - (s+1)/scale, f <= 0 ? 0 : f - floor(f); out = (1-f)*src[s] + f*src[min(s+1, ssize-1)].
// cv2 is absent from this image: parity with OpenCV is UNPINNED (the oracle pins the integer-ratio case against a
// numpy box mean).  One lane per (output pixel, channel); an init-time operator, not on the per-view path.
__device__ __forceinline__ void area_cell(int d, double scale, int ssize, int& s1, int& s2, float& a_first, float& a_mid,
                                          float& a_last) {
    const double f1 = d * scale, f2 = f1 + scale;
    const double cell = fmin(scale, (double)ssize - f1);
    s1 = (int)ceil(f1);
    s2 = (int)floor(f2);
    s2 = min(s2, ssize - 1);
    s1 = min(s1, s2);
    a_first = (s1 - f1 > 1e-3) ? (float)((s1 - f1) / cell) : 0.0f;          // weight of pixel s1 - 1
    a_mid = (float)(1.0 / cell);                                            // pixels s1 .. s2 - 1
    a_last = (f2 - s2 > 1e-3) ? (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell) : 0.0f;   // weight of pixel s2
}

__global__ void __launch_bounds__(256)
resize_area_kernel(const float* __restrict__ src, float* __restrict__ dst, int sh, int sw, int dh, int dw, int c) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)dh * dw * c) return;
    const int ch = (int)(i % c);
    const int dx = (int)((i / c) % dw), dy = (int)(i / ((long)c * dw));
    const double scx = (double)sw / dw, scy = (double)sh / dh;
    float acc = 0.0f;
    if (scx >= 1.0 && scy >= 1.0) {
        int x1, x2, y1, y2;
        float xa, xm, xl, ya, ym, yl;
        area_cell(dx, scx, sw, x1, x2, xa, xm, xl);
        area_cell(dy, scy, sh, y1, y2, ya, ym, yl);
        for (int y = y1 - 1; y <= y2; y++) {
            const float wy = y < y1 ? ya : (y < y2 ? ym : yl);
            if (wy == 0.0f || y < 0) continue;
            float row = 0.0f;
            for (int x = x1 - 1; x <= x2; x++) {
                const float wx = x < x1 ? xa : (x < x2 ? xm : xl);
                if (wx == 0.0f || x < 0) continue;
                row += src[((size_t)y * sw + x) * c + ch] * wx;
            }
            acc += row * wy;
        }
    } else {
        int sx = (int)floor(dx * scx), sy = (int)floor(dy * scy);
        float fx = (float)((dx + 1) - (sx + 1) / scx), fy = (float)((dy + 1) - (sy + 1) / scy);
        fx = fx <= 0.0f ? 0.0f : fx - floorf(fx);
        fy = fy <= 0.0f ? 0.0f : fy - floorf(fy);
        sx = min(sx, sw - 1); sy = min(sy, sh - 1);
        const int sx1 = min(sx + 1, sw - 1), sy1 = min(sy + 1, sh - 1);
        const float v00 = src[((size_t)sy * sw + sx) * c + ch], v01 = src[((size_t)sy * sw + sx1) * c + ch];
        const float v10 = src[((size_t)sy1 * sw + sx) * c + ch], v11 = src[((size_t)sy1 * sw + sx1) * c + ch];
        acc = (v00 * (1.0f - fx) + v01 * fx) * (1.0f - fy) + (v10 * (1.0f - fx) + v11 * fx) * fy;
    }
    dst[i] = acc;
}

extern "C" int rnr_resize_area(const float* src, float* dst, int src_h, int src_w, int dst_h, int dst_w, int channels,
                               void* stream) {
    RNR_REQUIRE(src && dst && src_h > 0 && src_w > 0 && dst_h > 0 && dst_w > 0 && channels > 0, "rnr_resize_area: bad arguments");
    const long total = (long)dst_h * dst_w * channels;
    hipLaunchKernelGGL(resize_area_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), src, dst,
                       src_h, src_w, dst_h, dst_w, channels);
    return check_launch("resize_area_kernel");
}

extern "C" int rnr_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int c_pad, void* stream) {
    RNR_REQUIRE(in && out && n > 0 && c > 0 && c_pad >= c, "rnr_nchw_to_nhwc: bad arguments");
    const long total = (long)n * h * w * c_pad;
    const long npix = (long)n * h * w;
    if (c_pad <= 240)       // 64-pixel x c_pad tile in LDS (<= 62.4 KB)
        hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), (size_t)c_pad * 65 * sizeof(float),
                           as_stream(stream), in, out, c, h * w, c_pad, total);
    else
        hipLaunchKernelGGL(nchw_to_nhwc_wide_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                           in, out, c, h * w, c_pad, total);
    return check_launch("nchw_to_nhwc_kernel");
}

extern "C" int rnr_nhwc_to_nchw(const float* in, float* out, const float* bias, int apply_tanh, int n, int c,
                                int h, int w, int c_pad, void* stream) {
    RNR_REQUIRE(in && out && n > 0 && c > 0 && c_pad >= c, "rnr_nhwc_to_nchw: bad arguments");
    const long total = (long)n * c * h * w;
    const long npix = (long)n * h * w;
    if (c_pad <= 240)
        hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(256), (size_t)c_pad * 65 * sizeof(float),
                           as_stream(stream), in, out, bias, apply_tanh, c, h * w, c_pad, total);
    else
        hipLaunchKernelGGL(nhwc_to_nchw_wide_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                           in, out, bias, apply_tanh, c, h * w, c_pad, total);
    return check_launch("nhwc_to_nchw_kernel");
}

extern "C" int rnr_view_dir_map(const float* proj_inv, const float* R_inv, float* out_world, float* out_cam,
                                int num_views, int height, int width, void* stream) {
    RNR_REQUIRE(proj_inv && R_inv && out_world && num_views > 0 && height > 0 && width > 0, "rnr_view_dir_map: bad arguments");
    const long npix = (long)num_views * height * width;
    hipLaunchKernelGGL(view_dir_map_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream),
                       proj_inv, R_inv, out_world, out_cam, npix, height, width);
    return check_launch("view_dir_map_kernel");
}

extern "C" int rnr_tbn_map(const float* normal_map, const int32_t* face_index_map, const float* face_tangents,
                           int num_faces, float* out, int num_views, int height, int width, void* stream) {
    RNR_REQUIRE(normal_map && face_index_map && face_tangents && out && num_faces > 0, "rnr_tbn_map: bad arguments");
    const long npix = (long)num_views * height * width;
    hipLaunchKernelGGL(tbn_map_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, as_stream(stream), normal_map,
                       face_index_map, face_tangents, num_faces, out, npix);
    return check_launch("tbn_map_kernel");
}

// out[p][i] = sum_j M[p][i][j] v[p][j] with M = tbn[p] (transposed = 0) or tbn[p]^T (transposed = 1): the per-pixel 3x3 products of
// test_rnr.py:314 (torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1), view_dir_map.reshape((-1, 3, 1)))), which torch hands to
// rocBLAS as 262 144 batched 3 x 3 GEMMs (1.9 ms per 512^2 view; this launch: a few us).  One pixel per thread; the 36 + 12 bytes of
// a pixel are read with 4-byte loads that the L1 merges (consecutive threads, consecutive records).
__global__ void __launch_bounds__(256)
tbn_matvec_kernel(const float* __restrict__ tbn, const float* __restrict__ v, float* __restrict__ out, long npix, int transposed) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= npix) return;
    const float* m = tbn + p * 9;
    const float x = v[p * 3 + 0], y = v[p * 3 + 1], z = v[p * 3 + 2];
    float o[3];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float a = transposed ? m[i] : m[3 * i], b = transposed ? m[3 + i] : m[3 * i + 1], c = transposed ? m[6 + i] : m[3 * i + 2];
        o[i] = __builtin_fmaf(c, z, __builtin_fmaf(b, y, a * x));
    }
    out[p * 3 + 0] = o[0]; out[p * 3 + 1] = o[1]; out[p * 3 + 2] = o[2];
}

extern "C" int rnr_tbn_matvec(const float* tbn, const float* vec, float* out, long num_pixels, int transposed, void* stream) {
    RNR_REQUIRE(tbn && vec && out && num_pixels > 0, "rnr_tbn_matvec: bad arguments");
    hipLaunchKernelGGL(tbn_matvec_kernel, dim3((unsigned)((num_pixels + 255) / 256)), dim3(256), 0, as_stream(stream), tbn, vec, out,
                       num_pixels, transposed);
    return check_launch("tbn_matvec_kernel");
}

extern "C" int rnr_ray_sampler(int reflect, const float* pivots_host, int num_rays, const float* tbn,
                               const float* view_tangent, const float* alpha, float* rays_dir, float* rays_uv,
                               float* rays_dir_tangent, long num_pixels, void* stream) {
    RNR_REQUIRE(pivots_host && tbn && alpha && rays_dir && rays_uv, "rnr_ray_sampler: null pointer argument");
    RNR_REQUIRE(!reflect || view_tangent, "rnr_ray_sampler: reflect mode needs view_tangent");
    RNR_REQUIRE(num_rays >= 1 && num_rays <= MAX_RAYS, "rnr_ray_sampler: 1..%d rays", MAX_RAYS);
    RaySamplerParams P;
    P.tbn = tbn; P.view_tangent = view_tangent; P.alpha = alpha; P.rays_dir = rays_dir; P.rays_uv = rays_uv;
    P.rays_dir_tangent = reflect ? rays_dir_tangent : nullptr;
    for (int r = 0; r < num_rays; r++)
        for (int k = 0; k < 3; k++) P.piv[r * 3 + k] = pivots_host[k * num_rays + r];
    P.n_rays = num_rays; P.reflect = reflect; P.npix = num_pixels;
    const long total = num_pixels * num_rays;
    hipLaunchKernelGGL(ray_sampler_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), P);
    return check_launch("ray_sampler_kernel");
}

extern "C" int rnr_texture_mapper(const float* uv_map, const float* sh_basis_map, const float* const* textures_host,
                                  const int* tex_sizes_host, int num_levels, int tex_channels, int sh_start_ch,
                                  float* out, int num_views, int height, int width, void* stream) {
    RNR_REQUIRE(uv_map && textures_host && tex_sizes_host && out, "rnr_texture_mapper: null pointer argument");
    RNR_REQUIRE(num_levels >= 1 && num_levels <= MAX_LEVELS && tex_channels > 0, "rnr_texture_mapper: bad sizes");
    RNR_REQUIRE(!sh_basis_map || (sh_start_ch >= 0 && sh_start_ch + 9 <= tex_channels),
                "rnr_texture_mapper: sh_start_ch + 9 > channels");
    TexMapParams P = {};
    P.uv_map = uv_map; P.sh = sh_basis_map;
    for (int l = 0; l < num_levels; l++) { P.tex[l] = textures_host[l]; P.tex_size[l] = tex_sizes_host[l]; }
    P.num_levels = num_levels; P.C = tex_channels; P.sh_start = sh_start_ch; P.out = out;
    P.npix = (long)num_views * height * width; P.hw = height * width;
    const long total = P.npix * tex_channels;
    hipLaunchKernelGGL(texture_mapper_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), P);
    return check_launch("texture_mapper_kernel");
}

extern "C" int rnr_ray_renderer(const float* rays_uv, const float* rays_lt, const float* lp, int lp_n, int lp_h,
                                int lp_w, const float* albedo_specular, const float* albedo_diffuse, int channels,
                                int num_rays, int num_ray_diffuse, int no_albedo, int seperate_albedo,
                                float lp_scale_factor, float* out, float* out_specular, float* out_diffuse,
                                float* ltt_specular, float* ltt_diffuse, float* rays_color, int num_views, int height,
                                int width, void* stream) {
    RNR_REQUIRE(rays_uv && rays_lt && lp && albedo_specular && out, "rnr_ray_renderer: null pointer argument");
    RNR_REQUIRE(num_rays > num_ray_diffuse && num_ray_diffuse >= 0, "rnr_ray_renderer: bad ray counts");
    RNR_REQUIRE(lp_n == 1 || lp_n == num_views, "rnr_ray_renderer: lp batch must be 1 or N");
    RayApiParams P;
    P.rays_uv = rays_uv; P.rays_lt = rays_lt; P.lp = lp; P.alb_spec = albedo_specular; P.alb_diff = albedo_diffuse;
    P.lp_n = lp_n; P.lp_h = lp_h; P.lp_w = lp_w; P.C = channels; P.R = num_rays; P.n_diff = num_ray_diffuse;
    P.no_albedo = no_albedo; P.separate = seperate_albedo; P.lp_scale = lp_scale_factor;
    P.out = out; P.out_spec = out_specular; P.out_diff = out_diffuse; P.ltt_spec = ltt_specular; P.ltt_diff = ltt_diffuse;
    P.rays_color = rays_color; P.npix = (long)num_views * height * width; P.hw = height * width;
    const long total = P.npix * channels;
    if (channels <= 4 && num_rays <= 64) {
        // r05: 64 pixels per workgroup, (pixel, ray quarter) per thread: uv rows staged through LDS, taps once per (pixel, ray) for
        // all channels, every global access coalesced along the pixels (0.48 -> ms per 512^2 view in the drop-in loop)
        const size_t lds = sizeof(float) * (size_t)std::max(64 * (2 * num_rays + 1), 4 * 2 * 4 * 64);
        hipLaunchKernelGGL(ray_renderer_api_tiled_kernel, dim3((unsigned)((P.npix + 63) / 64)), dim3(256), lds, as_stream(stream), P);
    } else {
        hipLaunchKernelGGL(ray_renderer_api_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream), P);
    }
    return check_launch("ray_renderer_api_kernel");
}
