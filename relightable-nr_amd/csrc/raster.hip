// Tiled, order-preserving z-resolve rasterizer for gfx950 + fused G-buffer attribute interpolation.
//
// Replaces (does not translate) the reference's brute-force forward kernels
//   forward_face_index_map_cuda_kernel_1/2  rasterize_cuda_kernel.cu:24-169   (O(pixels x faces))
//   forward_texture_sampling_cuda_kernel    rasterize_cuda_kernel.cu:171-242
// and, in the fused entry point, the torch glue of network.Rasterizer.forward (network.py:156-214).
//
// Design:
//   1. face_setup_kernel, one lane per (view, face): back-face predicate, the 3x3 barycentric inverse
//      (same IEEE binary32 operation sequence as the reference, this file is built with
//      -ffp-contract=off), and a conservative pixel bounding box.  Faces whose box cannot be trusted
//      (degenerate / sliver / non-finite) are flagged and decided by the exact tile test below.
//      (On the product path the setup shares the launch and the registers of step 2: setup_splat_faces_kernel.)
//   2. splat_faces_kernel, one lane per (view, face) (near >= 0, the product path): a face whose trusted box
//      covers <= 256 pixels walks them itself and folds (depth bits, face index) into a per-pixel 64-bit key
//      with one atomicMin; a bigger trusted box is binned into the 16x16-pixel tiles it touches (exact,
//      rounding-monotone tile test); untrusted boxes and huge faces go to a per-view wide list, which
//      every tile of raster_tile_kernel tests against itself (records of six coordinates + index, 1024 per round
//      trip).  (near < 0: bin_faces_kernel bins everything.)  With at most two views in the launch four lanes share
//      a face (box rows modulo 4).
//   3. raster_tile_kernel, one 256-thread workgroup per tile: evaluates the reference's per-candidate
//      arithmetic for its binned candidates on LDS-broadcast face records — unordered lists, so the
//      reference's "ascending faces, strict <" rule is applied in its order-free form (smallest zp, ties ->
//      smallest face index) —, walks the zero-area wide faces (two coincident vertices: the pole faces of a UV sphere)
//      along their line — only the two pixels next to the line's crossing of a row / column can pass the reference's
//      inside test, see walk_lines —, merges the winner with the pixel's key, and writes either the extension's maps
//      (drop-in mode) or the perspective-corrected attribute interpolation of network.py:176-214, already
//      vertically flipped.  A tile whose list overflowed rescans every face box itself (the in-order ballot
//      scan of round 1, kept as the fallback: capacity never changes results).
//   face_index_map, weight_map and depth_map are bit-identical to the reference evaluated without FMA
//   contraction (tests/test_gpu_raster.py, tests/golden/raster_*.npz).
#include "rnr_internal.h"

namespace rnr {

constexpr int TILE = 16;
constexpr int RTHREADS = 256;
constexpr int SCAN_ITEMS = 4;
constexpr int SCAN_CHUNK = RTHREADS * SCAN_ITEMS;
constexpr int FLUSH_EVERY = 4;                          // scan steps between queue-level checks (one barrier each)
constexpr int QCAP = (FLUSH_EVERY + 2) * SCAN_CHUNK;    // a flush is forced once more than 2 chunks are queued
constexpr int STAGE = 128;                             // candidates staged in LDS per evaluation batch
constexpr int STAGE_FLOATS = 24;

struct __attribute__((aligned(8))) FaceBox {
    short xlo, xhi, ylo, yhi;  // inclusive pixel-index bounds, rows in the kernel's native (unflipped) order
};
// xlo == BOX_EXACT : bounding box not trustworthy, the exact tile test decides
// empty_box()     : never a candidate (back face, or entirely off-screen)
constexpr short BOX_EXACT = -2;
// a box that overlaps no tile under the interval test `lo <= t_hi && hi >= t_lo` (lo > every tile index, hi < 0)
__host__ __device__ __forceinline__ FaceBox empty_box() {
    FaceBox b;
    b.xlo = 32767; b.xhi = -1; b.ylo = 32767; b.yhi = -1;
    return b;
}

__device__ __forceinline__ bool backface(const float* f) {
    // rasterize_cuda_kernel.cu:40 / :111
    return (f[7] - f[1]) * (f[3] - f[0]) < (f[4] - f[1]) * (f[6] - f[0]);
}

__device__ __forceinline__ float pix_center(int i, int is) {
    // rasterize_cuda_kernel.cu:93-94: (2*i + 1 - is) / is  (exact small integer, one correctly rounded divide)
    return (float)(2 * i + 1 - is) / (float)is;
}

// ------------------------------------------------------------------------------------------------
// 1. per-face setup
// ------------------------------------------------------------------------------------------------
// Setup of face i: writes faces_out (GATHER) / faces_inv / boxes and leaves the same values in f / inv / box for a caller
// that goes on with them (setup_splat_faces_kernel).  False = back face (inv undefined).
template <bool GATHER>
__device__ __forceinline__ bool face_setup(long i, const float* __restrict__ faces_in, const float* __restrict__ v_uvz,
                                           const int32_t* __restrict__ fidx, float* __restrict__ faces_out,
                                           float* __restrict__ faces_inv, FaceBox* __restrict__ boxes, int nf, int nv, int is,
                                           float (&f)[9], float (&inv)[9], FaceBox& box, bool write = true) {
    if (GATHER) {  // vertices_to_faces.py:4-25 fused in
        const int bn = (int)(i / nf), fn = (int)(i % nf);
#pragma unroll
        for (int v = 0; v < 3; v++) {
            const int vi = fidx[3 * fn + v];
            const float* p = v_uvz + ((long)bn * nv + vi) * 3;
            f[3 * v + 0] = p[0];
            f[3 * v + 1] = p[1];
            f[3 * v + 2] = p[2];
        }
#pragma unroll
        for (int k = 0; k < 9; k++) if (write) faces_out[i * 9 + k] = f[k];
    } else {
#pragma unroll
        for (int k = 0; k < 9; k++) f[k] = faces_in[i * 9 + k];
    }
    if (backface(f)) {  // reference returns before writing: caller's zero fill stays (rasterize.py:163)
        box = empty_box();
        if (write) boxes[i] = box;
        if (GATHER && write) {
#pragma unroll
            for (int k = 0; k < 9; k++) faces_inv[i * 9 + k] = 0.0f;
        }
        return false;
    }
    // ---- barycentric inverse, rasterize_cuda_kernel.cu:44-66 (operation order preserved) ----
    const float s = (float)is;
    float px[3], py[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
        px[v] = 0.5f * (f[3 * v + 0] * s + s - 1.0f);
        py[v] = 0.5f * (f[3 * v + 1] * s + s - 1.0f);
    }
    float m[9];
    m[0] = py[1] - py[2]; m[1] = px[2] - px[1]; m[2] = px[1] * py[2] - px[2] * py[1];
    m[3] = py[2] - py[0]; m[4] = px[0] - px[2]; m[5] = px[2] * py[0] - px[0] * py[2];
    m[6] = py[0] - py[1]; m[7] = px[1] - px[0]; m[8] = px[0] * py[1] - px[1] * py[0];
    const float den = px[2] * (py[0] - py[1]) + px[0] * (py[1] - py[2]) + px[1] * (py[2] - py[0]);
#pragma unroll
    for (int k = 0; k < 9; k++) { inv[k] = m[k] / den; if (write) faces_inv[i * 9 + k] = inv[k]; }

    // ---- conservative pixel bounding box (double; DESIGN.md §Rasterizer gives the bound) ----
    const double x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    const double minx = fmin(x0, fmin(x1, x2)), maxx = fmax(x0, fmax(x1, x2));
    const double miny = fmin(y0, fmin(y1, y2)), maxy = fmax(y0, fmax(y1, y2));
    const double maxabs = fmax(fmax(fabs(minx), fabs(maxx)), fmax(fabs(miny), fabs(maxy)));
    const double ax = x1 - x0, ay = y1 - y0, bx = x2 - x0, by = y2 - y0, cx = x2 - x1, cy = y2 - y1;
    const double area2 = fabs(ax * by - bx * ay);
    const double l2 = fmax(ax * ax + ay * ay, fmax(bx * bx + by * by, cx * cx + cy * cy));
    const double sin_min = area2 / l2;                         // <= sin(smallest interior angle)
    const double u = 5.9604644775390625e-08;                   // 2^-24
    const double pad_ndc = 4.0 * (32.0 * u * (1.0 + maxabs) / sin_min);
    const double pad_px = pad_ndc * is * 0.5 + 1.0;
    const bool finite = isfinite(maxabs);
    if (!finite || !(sin_min > 0.0) || !(pad_px <= 16.0)) {
        box.xlo = BOX_EXACT; box.xhi = (short)(is - 1); box.ylo = 0; box.yhi = (short)(is - 1);
    } else {
        // pixel i has its centre at NDC (2i + 1 - is) / is, i.e. NDC x is the pixel coordinate (x * is + is - 1) / 2: the pixels
        // whose centre lies within pad_ndc (>= 10 x the proven distance bound) of the face's extent are exactly
        // [ceil(lo), floor(hi)].  (Until r06 the box carried up to four further pixels per side — the +1 of pad_px, which
        // belongs to the trust criterion above, a -1 / +1 here, and floor / ceil the other way round — and a pixel-sized face
        // of the 65 536-face sphere walked 80 pixels instead of 30.)
        double lx = ceil(((minx - pad_ndc) * is + is - 1) * 0.5);
        double hx = floor(((maxx + pad_ndc) * is + is - 1) * 0.5);
        double ly = ceil(((miny - pad_ndc) * is + is - 1) * 0.5);
        double hy = floor(((maxy + pad_ndc) * is + is - 1) * 0.5);
        lx = fmax(lx, 0.0); ly = fmax(ly, 0.0);
        hx = fmin(hx, (double)(is - 1)); hy = fmin(hy, (double)(is - 1));
        if (lx > hx || ly > hy) {
            box = empty_box();
        } else {
            box.xlo = (short)lx; box.xhi = (short)hx; box.ylo = (short)ly; box.yhi = (short)hy;
        }
    }
    if (write) boxes[i] = box;
    return true;
}

template <bool GATHER>
__global__ void __launch_bounds__(256)
face_setup_kernel(const float* __restrict__ faces_in, const float* __restrict__ v_uvz,
                  const int32_t* __restrict__ fidx, float* __restrict__ faces_out,
                  float* __restrict__ faces_inv, FaceBox* __restrict__ boxes, int batch, int nf, int nv,
                  int is) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * nf) return;
    float f[9], inv[9];
    FaceBox box;
    face_setup<GATHER>(i, faces_in, v_uvz, fidx, faces_out, faces_inv, boxes, nf, nv, is, f, inv, box);
}

// Exact conservative tile test.  The reference rejects pixel p for edge a->b iff
//     fl(fl(yp - ya) * fl(xb - xa)) < fl(fl(xp - xa) * fl(yb - ya))          (rasterize_cuda_kernel.cu:115-117)
// Rounding is monotone, so over a tile the left product lies between its values at the tile's first/last
// row and the right product between its values at the first/last column.  If all four corner comparisons
// say "reject", every pixel of the tile rejects.  Any NaN makes a comparison false => tile kept.
__device__ __forceinline__ bool edge_rejects_tile(float xa, float ya, float xb, float yb, float xlo,
                                                  float xhi, float ylo, float yhi) {
    const float dx = xb - xa, dy = yb - ya;
    const float p0 = (ylo - ya) * dx, p1 = (yhi - ya) * dx;
    const float q0 = (xlo - xa) * dy, q1 = (xhi - xa) * dy;
    return (p0 < q0) && (p0 < q1) && (p1 < q0) && (p1 < q1);
}


// ------------------------------------------------------------------------------------------------
// 1b. face-parallel binning: O(faces x tiles-per-face) instead of every tile scanning every face.
//     Each kept face is appended (LDS-free, one global atomic per (face, tile)) to the candidate list of every tile
//     whose exact tile test it survives.  Lists are unordered; the z-resolve is order-free.  Faces whose box is not
//     trustworthy (BOX_EXACT) or spans many tiles go to a per-view "wide" list, which every tile of raster_tile_kernel
//     tests against itself (a launch of its own until r04: 10 - 26 us for a list that is empty on the bench scene).  A tile
//     whose list overflows BIN_CAP falls back to scanning all boxes itself, so capacity never affects results.
// ------------------------------------------------------------------------------------------------
constexpr int BIN_CAP = 2048;
constexpr int WIDE_TILES = 64;

__device__ __forceinline__ bool face_may_touch_tile(const float* f, int tx, int ty, int is) {
    const int tx0 = tx * TILE, ty0 = ty * TILE;
    const int tx1 = min(tx0 + TILE - 1, is - 1), ty1 = min(ty0 + TILE - 1, is - 1);
    const float xlo = pix_center(tx0, is), xhi = pix_center(tx1, is);
    const float ylo = pix_center(ty0, is), yhi = pix_center(ty1, is);
    const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
    return !(edge_rejects_tile(x0, y0, x1, y1, xlo, xhi, ylo, yhi) || edge_rejects_tile(x1, y1, x2, y2, xlo, xhi, ylo, yhi) ||
             edge_rejects_tile(x2, y2, x0, y0, xlo, xhi, ylo, yhi));
}

__device__ __forceinline__ void bin_append(int* tile_count, int* tile_list, int tile, int fn) {
    const int slot = atomicAdd(tile_count + tile, 1);
    if (slot < BIN_CAP) tile_list[(size_t)tile * BIN_CAP + slot] = fn;
}
// The same test on the six coordinates of a wide-list record (below).
__device__ __forceinline__ bool coords_may_touch_tile(float x0, float y0, float x1, float y1, float x2, float y2, float xlo,
                                                      float xhi, float ylo, float yhi) {
    return !(edge_rejects_tile(x0, y0, x1, y1, xlo, xhi, ylo, yhi) || edge_rejects_tile(x1, y1, x2, y2, xlo, xhi, ylo, yhi) ||
             edge_rejects_tile(x2, y2, x0, y0, xlo, xhi, ylo, yhi));
}
// A face on the wide list leaves, beside its index, the record every tile tests: (x0, y0, x1, y1) (x2, y2, face index, -) —
// two 16-byte loads at consecutive addresses per lane, where the tile kernel used to chase index -> box -> nine floats at
// a stride of 36 bytes through three dependent round trips per 256 wide faces (the 65 536-face sphere has 520 of them, its
// zero-area pole faces: 42 of the tile kernel's 52 us at one view per call, r06).
constexpr int WIDE_REC_FLOATS = 8;
__device__ __forceinline__ void wide_append(int* wide_count, int* wide_list, float* wide_rec, int bn, int nf, int fn, float x0,
                                            float y0, float x1, float y1, float x2, float y2) {
    const int pos = atomicAdd(wide_count + bn, 1);
    wide_list[(size_t)bn * nf + pos] = fn;
    float4* r = reinterpret_cast<float4*>(wide_rec + ((size_t)bn * nf + pos) * WIDE_REC_FLOATS);
    r[0] = make_float4(x0, y0, x1, y1);
    r[1] = make_float4(x2, y2, __builtin_bit_cast(float, fn), 0.0f);
}

__global__ void __launch_bounds__(256)
bin_faces_kernel(const float* __restrict__ faces, const FaceBox* __restrict__ boxes, int* __restrict__ tile_count,
                 int* __restrict__ tile_list, int* __restrict__ wide_count, int* __restrict__ wide_list,
                 float* __restrict__ wide_rec, int batch, int nf, int is) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * nf) return;
    const int bn = (int)(i / nf), fn = (int)(i % nf);
    const FaceBox b = boxes[i];
    if (b.xlo != BOX_EXACT && b.xlo > b.xhi) return;                    // empty_box(): culled / off-screen
    const int tiles_x = (is + TILE - 1) / TILE;
    const int ntiles = tiles_x * tiles_x;
    const int txa = max(b.xlo, (short)0) / TILE, txb = b.xhi / TILE, tya = max(b.ylo, (short)0) / TILE, tyb = b.yhi / TILE;
    const float* f = faces + i * 9;
    if (b.xlo == BOX_EXACT || (txb - txa + 1) * (tyb - tya + 1) > WIDE_TILES) {
        wide_append(wide_count, wide_list, wide_rec, bn, nf, fn, f[0], f[1], f[3], f[4], f[6], f[7]);
        return;
    }
    int* tc = tile_count + (size_t)bn * ntiles;
    int* tl = tile_list + (size_t)bn * ntiles * BIN_CAP;
    for (int ty = tya; ty <= tyb; ty++)
        for (int tx = txa; tx <= txb; tx++)
            if (face_may_touch_tile(f, tx, ty, is)) bin_append(tc, tl, ty * tiles_x + tx, fn);
}

// The reference's per-(face, pixel) candidate arithmetic (rasterize_cuda_kernel.cu:115-139) on a 24-float face record
//   r0 = (x0, y0, x1, y1)  r1 = (x2, y2, x1-x0, y1-y0)  r2 = (x2-x1, y2-y1, x0-x2, y0-y2)
//   r3 = (inv0..3)  r4 = (inv4..7)  r5 = (inv8, z0, z1, z2)
// Used by the tile kernel (LDS-broadcast records) and by the face-parallel splat kernel: one definition, so both paths
// produce the same bits for the same (face, pixel).
__device__ __forceinline__ bool cand_inside(const float4 r0, const float4 r1, const float4 r2, float xp, float yp) {
    // inside test (rasterize_cuda_kernel.cu:115-118)
    if ((yp - r0.y) * r1.z < (xp - r0.x) * r1.w) return false;
    if ((yp - r0.w) * r2.x < (xp - r0.z) * r2.y) return false;
    if ((yp - r1.y) * r2.z < (xp - r1.x) * r2.w) return false;
    return true;
}
__device__ __forceinline__ bool cand_depth(const float4 r3, const float4 r4, const float4 r5, float fxi, float fyi,
                                           float near_, float far_, float& zp, float& w0, float& w1, float& w2) {
    // w = face_inv * (xi, yi, 1), clamp, renormalise (cu:121-134)
    w0 = r3.x * fxi + r3.y * fyi + r3.z;
    w1 = r3.w * fxi + r4.x * fyi + r4.y;
    w2 = r4.z * fxi + r4.w * fyi + r5.x;
    w0 = fminf(fmaxf(w0, 0.0f), 1.0f);
    w1 = fminf(fmaxf(w1, 0.0f), 1.0f);
    w2 = fminf(fmaxf(w2, 0.0f), 1.0f);
    float wsum = 0.0f;
    wsum += w0; wsum += w1; wsum += w2;
    w0 /= wsum; w1 /= wsum; w2 /= wsum;
    zp = 1.0f / (w0 / r5.y + w1 / r5.z + w2 / r5.w);   // cu:136
    return !(zp <= near_ || far_ <= zp);                // cu:137-139 (a NaN zp is rejected by neither test: kept, never wins)
}

// ------------------------------------------------------------------------------------------------
// 1c. face-parallel z-resolve for small faces ("splat"): one lane per (view, face) walks the few pixels of the face's
//     trusted bounding box, runs the reference's candidate arithmetic on each and folds the result into a per-pixel
//     64-bit key with ONE atomic:  key = (bits of zp) << 32 | face index,  atomicMin.
//     For zp > 0 the bit pattern of a float orders like the float, so the minimum key is "smallest zp, ties -> smallest
//     face index" — exactly the order-free form of the reference's rule (ascending faces, strict `<`, cu:142-153) the
//     tile kernel applies.  zp > near >= 0 is guaranteed by the near test; a NaN zp never passes `zp < far`-style
//     comparisons and is skipped here just as it never wins there (see cand_depth: NaN is kept by the reject test, so
//     it is filtered explicitly).  The path is only taken when near >= 0.
//     Work: faces x box pixels (a pixel-sized face of the bench mesh: ~16-36 tests) instead of tiles x candidates x 256.
//     Faces whose box cannot be trusted (BOX_EXACT) or covers more than SPLAT_MAX_PIX pixels go to the wide list and
//     to the tile kernel as before; the tile kernel merges both results per pixel.
// ------------------------------------------------------------------------------------------------
constexpr int SPLAT_MAX_PIX = 256;
constexpr unsigned long long KEY_EMPTY = ~0ull;

// The pixel walk of one face.  A row is scanned in 32-column pieces: first the cheap inside tests of the piece (a bit per
// pixel), then the depth arithmetic — seven correctly rounded divisions — for the set bits only: a wave then runs it as
// often as its busiest lane has inside pixels, not once per column in which ANY lane is inside.
// POW2: `is` is a power of two, where (2i + 1 - is) / is == (2i + 1 - is) * (1 / is) exactly (both are exact), without the
// division sequence per pixel.
template <bool POW2>
__device__ __forceinline__ void splat_pixels(const float4 r0, const float4 r1, const float4 r2, const float4 r3, const float4 r4,
                                             const float4 r5, int xa, int xb, int ya, int yb, int ystep, int fn,
                                             unsigned long long* __restrict__ kv, int is, float near_, float far_) {
    const float inv_is = 1.0f / (float)is;
    auto center = [&](int i) { return POW2 ? (float)(2 * i + 1 - is) * inv_is : pix_center(i, is); };
    for (int yi = ya; yi <= yb; yi += ystep) {
        const float yp = center(yi);
        for (int xc = xa; xc <= xb; xc += 32) {
            const int xe = min(xb, xc + 31);
            unsigned m = 0u;
            for (int xi = xc; xi <= xe; xi++)
                if (cand_inside(r0, r1, r2, center(xi), yp)) m |= 1u << (xi - xc);
            while (m) {
                const int xi = xc + __builtin_ctz(m);
                m &= m - 1u;
                float zp, w0, w1, w2;
                if (!cand_depth(r3, r4, r5, (float)xi, (float)yi, near_, far_, zp, w0, w1, w2)) continue;
                if (!(zp > 0.0f)) continue;                             // NaN (and anything the bit order cannot rank)
                const unsigned long long key = ((unsigned long long)__builtin_bit_cast(unsigned, zp) << 32) | (unsigned)fn;
                atomicMin(kv + (size_t)yi * is + xi, key);
            }
        }
    }
}

// f / fi: the face's 9 + 9 floats (global memory or the registers of the setup that has just produced them)
template <typename FP>
__device__ __forceinline__ void splat_face(long i, int bn, int fn, const FaceBox b, FP f, FP fi,
                                           unsigned long long* __restrict__ keys, int* __restrict__ tile_count,
                                           int* __restrict__ tile_list, int* __restrict__ wide_count, int* __restrict__ wide_list,
                                           float* __restrict__ wide_rec, int nf, int is, float near_, float far_, int sub = 0,
                                           int nsub = 1) {
    // sub / nsub: this lane is one of nsub that share the face (setup_splat_faces_kernel<., LPF>): it walks the box rows
    // sub, sub + nsub, ...; lane 0 alone bins / lists a face that is not walked
    if (b.xlo != BOX_EXACT && b.xlo > b.xhi) return;                    // empty_box(): culled / off-screen
    const int xa = max((int)b.xlo, 0), xb = b.xhi, ya = max((int)b.ylo, 0), yb = b.yhi;
    if (b.xlo == BOX_EXACT || (xb - xa + 1) * (yb - ya + 1) > SPLAT_MAX_PIX) {
        if (sub != 0) return;
        // too big to walk pixel by pixel.  A trusted box over a few tiles is binned right here (bin_faces_kernel's loop);
        // only untrusted boxes and huge faces take the wide list, whose faces the tile kernel tests against
        // EVERY tile — with every > 256-pixel face on it, a close-up of a coarse mesh cost wide x tiles pair tests
        const int tiles_x = (is + TILE - 1) / TILE;
        const int txa = xa / TILE, txb = xb / TILE, tya = ya / TILE, tyb = yb / TILE;
        if (b.xlo != BOX_EXACT && (txb - txa + 1) * (tyb - tya + 1) <= WIDE_TILES) {
            int* tc = tile_count + (size_t)bn * tiles_x * tiles_x;
            int* tl = tile_list + (size_t)bn * tiles_x * tiles_x * BIN_CAP;
            for (int ty = tya; ty <= tyb; ty++)
                for (int tx = txa; tx <= txb; tx++)
                    if (face_may_touch_tile(f, tx, ty, is)) bin_append(tc, tl, ty * tiles_x + tx, fn);
            return;
        }
        wide_append(wide_count, wide_list, wide_rec, bn, nf, fn, f[0], f[1], f[3], f[4], f[6], f[7]);
        return;
    }
    const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6], y2 = f[7], z2 = f[8];
    const float4 r0 = make_float4(x0, y0, x1, y1);
    const float4 r1 = make_float4(x2, y2, x1 - x0, y1 - y0);
    const float4 r2 = make_float4(x2 - x1, y2 - y1, x0 - x2, y0 - y2);
    const float4 r3 = make_float4(fi[0], fi[1], fi[2], fi[3]);
    const float4 r4 = make_float4(fi[4], fi[5], fi[6], fi[7]);
    const float4 r5 = make_float4(fi[8], z0, z1, z2);
    unsigned long long* kv = keys + (size_t)bn * is * is;
    if ((is & (is - 1)) == 0) splat_pixels<true>(r0, r1, r2, r3, r4, r5, xa, xb, ya + sub, yb, nsub, fn, kv, is, near_, far_);
    else splat_pixels<false>(r0, r1, r2, r3, r4, r5, xa, xb, ya + sub, yb, nsub, fn, kv, is, near_, far_);
}

__global__ void __launch_bounds__(256)
splat_faces_kernel(const float* __restrict__ faces, const float* __restrict__ faces_inv, const FaceBox* __restrict__ boxes,
                   unsigned long long* __restrict__ keys, int* __restrict__ tile_count, int* __restrict__ tile_list,
                   int* __restrict__ wide_count, int* __restrict__ wide_list, float* __restrict__ wide_rec, int batch, int nf,
                   int is, float near_, float far_) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)batch * nf) return;
    splat_face<const float*>(i, (int)(i / nf), (int)(i % nf), boxes[i], faces + i * 9, faces_inv + i * 9, keys, tile_count,
                             tile_list, wide_count, wide_list, wide_rec, nf, is, near_, far_);
}

// face_setup_kernel + splat_faces_kernel in one launch (r04): both are one lane per (view, face) and the splat needs nothing
// but its own face's record, which it takes from the registers of the setup — the same values the setup writes for the tile
// kernel, hence the same bits as the two launches (tests/test_gpu_raster.py).
struct FaceSetupArgs { const float* faces_in; const float* v_uvz; const int32_t* fidx; float* faces_out; int nv; int gather; };
// LPF lanes per face (r06): with a single view in the launch two thirds of the waves hold back faces only and leave at once,
// the others walk ~30 pixels per lane at one wave per SIMD; four lanes per face (each repeats the setup — same instructions,
// lane 0 stores — and walks every fourth box row) spread the live faces over four times the waves: 23.8 -> 15.7 us per 512^2 view.
// Batches that fill the chip anyway keep one lane per face.
template <bool GATHER, int LPF>
__global__ void __launch_bounds__(256)
setup_splat_faces_kernel(const float* __restrict__ faces_in, const float* __restrict__ v_uvz, const int32_t* __restrict__ fidx,
                         float* __restrict__ faces_out, float* __restrict__ faces_inv, FaceBox* __restrict__ boxes,
                         unsigned long long* __restrict__ keys, int* __restrict__ tile_count, int* __restrict__ tile_list,
                         int* __restrict__ wide_count, int* __restrict__ wide_list, float* __restrict__ wide_rec, int batch,
                         int nf, int nv, int is, float near_, float far_) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long i = t / LPF;
    const int sub = (int)(t % LPF);
    if (i >= (long)batch * nf) return;
    float f[9], inv[9];
    FaceBox box;
    if (!face_setup<GATHER>(i, faces_in, v_uvz, fidx, faces_out, faces_inv, boxes, nf, nv, is, f, inv, box, sub == 0)) return;
    splat_face<const float (&)[9]>(i, (int)(i / nf), (int)(i % nf), box, f, inv, keys, tile_count, tile_list, wide_count,
                                   wide_list, wide_rec, nf, is, near_, far_, sub, LPF);
}

// ------------------------------------------------------------------------------------------------
// 2. tile kernel
// ------------------------------------------------------------------------------------------------
struct RasterParams {
    const float* faces;      // [B,nf,9]
    const float* faces_inv;  // [B,nf,9]
    const FaceBox* boxes;    // [B,nf]
    const int* tile_count;   // [B,ntiles]  candidates binned per tile (may exceed BIN_CAP: then the tile rescans)
    const int* tile_list;    // [B,ntiles,BIN_CAP]
    const unsigned long long* keys;   // [B,is,is] winners of the face-parallel path (KEY_EMPTY = none) or NULL
    const int* wide_count;   // [B] faces on the wide list: untrusted or huge boxes, tested by every tile itself
    const int* wide_list;    // [B,nf]
    const float* wide_rec;   // [B,nf,WIDE_REC_FLOATS] what the tiles test of a wide face (wide_append)
    int nf, is;
    float near_, far_;
    int flip;                // 1: write row (is-1-yi)
    // drop-in outputs (MODE 0)
    int32_t* face_index_map;
    float* weight_map;
    float* depth_map;
    float* face_inv_map;     // may be NULL
    // fused outputs (MODE 1)
    rnr_mesh mesh;
    rnr_gbuffer gb;
    const float* pose;       // [B,4,4] or NULL
};

template <int MODE>
__global__ void __launch_bounds__(RTHREADS)
raster_tile_kernel(const RasterParams P) {
    __shared__ int s_queue[QCAP];
    __shared__ __attribute__((aligned(16))) float s_stage[STAGE * STAGE_FLOATS];
    __shared__ int s_qn;
    __shared__ int s_dn;                                // zero-area wide faces queued from the END of s_queue (walk_lines)
    __shared__ unsigned long long s_key[RTHREADS];      // their per-pixel winners: the same (depth bits, face) key as P.keys

    const int is = P.is, nf = P.nf;
    const int tiles_x = (is + TILE - 1) / TILE;
    const int tile = blockIdx.x;
    const int bn = blockIdx.y;
    const int tx0 = (tile % tiles_x) * TILE, ty0 = (tile / tiles_x) * TILE;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int xi = tx0 + (tid & (TILE - 1)), yi = ty0 + (tid >> 4);
    const bool in_img = (xi < is) && (yi < is);
    const int tx1 = min(tx0 + TILE - 1, is - 1), ty1 = min(ty0 + TILE - 1, is - 1);
    const float xp = pix_center(xi, is), yp = pix_center(yi, is);
    const float fxi = (float)xi, fyi = (float)yi;
    const float t_xlo = pix_center(tx0, is), t_xhi = pix_center(tx1, is);
    const float t_ylo = pix_center(ty0, is), t_yhi = pix_center(ty1, is);

    const float* faces = P.faces + (size_t)bn * nf * 9;
    const float* faces_inv = P.faces_inv + (size_t)bn * nf * 9;
    const FaceBox* boxes = P.boxes + (size_t)bn * nf;

    float best_z = P.far_;
    int best = -1;
    float bw0 = 0.f, bw1 = 0.f, bw2 = 0.f;

    // the face-parallel path's winner of this pixel, requested up front: independent of everything the lists below need
    unsigned long long key = KEY_EMPTY;
    if (P.keys && in_img) key = P.keys[((size_t)bn * is + yi) * is + xi];

    if (tid == 0) { s_qn = 0; s_dn = 0; }
    s_key[tid] = KEY_EMPTY;
    __syncthreads();

    // Evaluate the queued candidates.  The queue is NOT in face order (waves append independently), so the
    // reference's "ascending faces, strict <" rule (cu:142-153) is applied as its order-free equivalent:
    // smallest zp wins, equal zp -> smallest face index; a NaN zp never wins either way.
    auto process_queue = [&](const int* ids, int qn) {
        for (int s0 = 0; s0 < qn; s0 += STAGE) {
            const int n = min(STAGE, qn - s0);
            if (tid < n) {
                const int fn = ids[s0 + tid];
                const float* f = faces + (size_t)fn * 9;
                const float* fi = faces_inv + (size_t)fn * 9;
                float4* dst = reinterpret_cast<float4*>(s_stage + tid * STAGE_FLOATS);
                const float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5], x2 = f[6],
                            y2 = f[7], z2 = f[8];
                dst[0] = make_float4(x0, y0, x1, y1);
                dst[1] = make_float4(x2, y2, x1 - x0, y1 - y0);
                dst[2] = make_float4(x2 - x1, y2 - y1, x0 - x2, y0 - y2);
                dst[3] = make_float4(fi[0], fi[1], fi[2], fi[3]);
                dst[4] = make_float4(fi[4], fi[5], fi[6], fi[7]);
                dst[5] = make_float4(fi[8], z0, z1, z2);
            }
            __syncthreads();
            if (in_img) {
                for (int c = 0; c < n; c++) {
                    const float4* rec = reinterpret_cast<const float4*>(s_stage + c * STAGE_FLOATS);
                    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
                    if (!cand_inside(r0, r1, r2, xp, yp)) continue;
                    float zp, w0, w1, w2;
                    if (!cand_depth(rec[3], rec[4], rec[5], fxi, fyi, P.near_, P.far_, zp, w0, w1, w2)) continue;
                    const int fn = ids[s0 + c];
                    if (zp < best_z || (zp == best_z && best >= 0 && fn < best)) {  // cu:142, order-free form
                        best_z = zp;
                        best = fn;
                        bw0 = w0; bw1 = w1; bw2 = w2;
                    }
                }
            }
            __syncthreads();
        }
    };

    // fast path: the candidate list built by the binning kernels; a list that overflowed is ignored and the
    // tile scans every box itself (slow, same result)
    const int ntiles_all = tiles_x * tiles_x;
    const int binned = P.tile_count ? P.tile_count[(size_t)bn * ntiles_all + tile] : BIN_CAP + 1;
    const int nsteps = binned <= BIN_CAP ? 0 : (nf + SCAN_CHUNK - 1) / SCAN_CHUNK;
    if (binned <= BIN_CAP) process_queue(P.tile_list + ((size_t)bn * ntiles_all + tile) * BIN_CAP, binned);
    // the wide list (untrusted boxes, faces over more than WIDE_TILES tiles): every tile tests it against itself — usually
    // empty, and then this costs one scalar load where bin_wide_kernel cost a launch (r04).  A tile that rescans every face
    // anyway (overflowed list) meets the wide faces there.
    // Zero-area wide faces (two coincident vertices: the pole faces of a UV sphere, 520 of the 65 536-face bench mesh).  Their
    // pass region is a line: with the distinct vertices a, b the reference's three edge tests reduce to  P1 >= Q1  and
    // P2 <= Q2, rounded evaluations of the same edge function E(p) = (yp - ya) dx - (xp - xa) dy  (dx = fl(xb - xa), dy alike;
    // the opposite edge has exactly the negated differences), so a pixel can only pass with |E(p)| <= 2.01 u (|yp - ya| |dx| +
    // |xp - xa| |dy|) + 2 u |dx dy|.  With every coordinate in [-4, 4] that puts the pixel centre within 36 u = 2.2e-6 NDC of
    // the line along the axis the line is steeper in — 0.018 pixels at the largest image the library accepts (16384).  So of
    // each pixel row (column, for a shallow line) of the tile only the two pixels next to the crossing can pass; they get the
    // reference's candidate arithmetic (cand_inside / cand_depth: the same bits as everywhere else), 32 tests per (tile, face)
    // instead of 256, sixteen faces at a time, and the result is folded into a per-pixel key like the splat path's.  A tile at
    // a pole of the sphere holds ~260 such candidates: 36 us as ordinary queue entries at one view per call, r06.
    // Faces outside the conditions (three coincident vertices, |coordinate| > 4, |b - a| < 1e-6, near < 0) stay ordinary.
    // what the walk needs of a zero-area face, 12 floats: (x0, y0, x1, y1) (x2, y2, face, steep) and, as doubles, (c0, c1): the
    // crossing of row / column i with the line, in pixel coordinates, is c0 + c1 * pix_center(i)
    constexpr int LINE_FLOATS = 12;
    static_assert(RTHREADS * LINE_FLOATS <= STAGE * STAGE_FLOATS, "a batch of line records fits the staging array");
    auto stage_line = [&](int slot, const float4 ra, const float4 rb) {
        const float x0 = ra.x, y0 = ra.y, x1 = ra.z, y1 = ra.w, x2 = rb.x, y2 = rb.y;
        // the two distinct points: v0 and v1, or v0 and v2 when v0 == v1
        const bool e01 = x0 == x1 && y0 == y1;
        const float xa = x0, ya = y0, xb = e01 ? x2 : x1, yb = e01 ? y2 : y1;
        const float dx = xb - xa, dy = yb - ya;
        const bool steep = fabsf(dy) >= fabsf(dx);
        // steep: x(yp) = xa + (yp - ya) dx / dy, else y(xp) = ya + (xp - xa) dy / dx; NDC t -> pixel coordinate (t * is + is - 1) / 2
        const double sl = steep ? (double)dx / (double)dy : (double)dy / (double)dx;
        const double t0 = steep ? (double)xa - (double)ya * sl : (double)ya - (double)xa * sl;
        const double c0 = (t0 * is + is - 1) * 0.5, c1 = sl * is * 0.5;
        float4* dst = reinterpret_cast<float4*>(s_stage + slot * LINE_FLOATS);
        dst[0] = ra;
        dst[1] = make_float4(x2, y2, rb.z, steep ? 1.0f : 0.0f);
        reinterpret_cast<double2*>(dst)[2] = make_double2(c0, c1);
    };
    const bool pow2 = (is & (is - 1)) == 0;
    const float inv_is = 1.0f / (float)is;
    // pix_center without the division where `is` is a power of two (both forms are exact there)
    auto center = [&](int i) { return pow2 ? (float)(2 * i + 1 - is) * inv_is : pix_center(i, is); };
    // The first RTHREADS records were staged by the lanes that queued them (registers -> LDS, no second fetch); later batches
    // are fetched again by their slot number.
    auto walk_lines = [&](int dn) {
        for (int d0 = 0; d0 < dn; d0 += RTHREADS) {
            const int n = min(RTHREADS, dn - d0);
            if (d0 > 0) {
                if (tid < n) {
                    const int w = s_queue[QCAP - 1 - (d0 + tid)];
                    const float4* r = reinterpret_cast<const float4*>(P.wide_rec + ((size_t)bn * nf + w) * WIDE_REC_FLOATS);
                    stage_line(tid, r[0], r[1]);
                }
                __syncthreads();
            }
            const int sub = tid & 15;
            for (int c = tid >> 4; c < n; c += RTHREADS / 16) {
                const float4* rec = reinterpret_cast<const float4*>(s_stage + c * LINE_FLOATS);
                const float4 ra = rec[0], rb = rec[1];
                const double2 cc = reinterpret_cast<const double2*>(rec)[2];
                const float x0 = ra.x, y0 = ra.y, x1 = ra.z, y1 = ra.w, x2 = rb.x, y2 = rb.y;
                const int fn = __builtin_bit_cast(int, rb.z);
                const bool steep = rb.w != 0.0f;
                // the crossing of this lane's row (steep) or column with the line, in pixel coordinates
                const int fixed = (steep ? ty0 : tx0) + sub;
                const float cf = center(fixed);
                const double tp = cc.x + cc.y * (double)cf;
                if (!(tp > -2.0 && tp < (double)is + 1.0)) continue;
                const int lo = (int)floor(tp);
                const float4 r0 = make_float4(x0, y0, x1, y1);
                const float4 r1 = make_float4(x2, y2, x1 - x0, y1 - y0);
                const float4 r2 = make_float4(x2 - x1, y2 - y1, x0 - x2, y0 - y2);
#pragma unroll
                for (int side = 0; side < 2; side++) {
                    const int pxi = steep ? lo + side : fixed, pyi = steep ? fixed : lo + side;
                    if (pxi < tx0 || pxi > tx1 || pyi < ty0 || pyi > ty1) continue;
                    if (!cand_inside(r0, r1, r2, steep ? center(pxi) : cf, steep ? cf : center(pyi))) continue;
                    const float* f = faces + (size_t)fn * 9;
                    const float* fi = faces_inv + (size_t)fn * 9;
                    float zp, w0, w1, w2;
                    if (!cand_depth(make_float4(fi[0], fi[1], fi[2], fi[3]), make_float4(fi[4], fi[5], fi[6], fi[7]),
                                    make_float4(fi[8], f[2], f[5], f[8]), (float)pxi, (float)pyi, P.near_, P.far_, zp, w0, w1, w2))
                        continue;
                    if (!(zp > 0.0f)) continue;         // NaN (never wins) — and the key order needs zp > 0 (near >= 0 here)
                    atomicMin(&s_key[(pyi - ty0) * TILE + (pxi - tx0)],
                              ((unsigned long long)__builtin_bit_cast(unsigned, zp) << 32) | (unsigned)fn);
                }
            }
            __syncthreads();
        }
    };
    if (P.wide_count && binned <= BIN_CAP) {
        const int wn = P.wide_count[bn];
        const float4* wr = reinterpret_cast<const float4*>(P.wide_rec + (size_t)bn * nf * WIDE_REC_FLOATS);
        // WIDE_ITEMS records per lane and round, all of a round's loads in flight together: one round trip and one barrier
        // pair per 1024 wide faces
        constexpr int WIDE_ITEMS = 4;
        for (int w0 = 0; w0 < wn; w0 += RTHREADS * WIDE_ITEMS) {
            float4 ra[WIDE_ITEMS], rb[WIDE_ITEMS];
#pragma unroll
            for (int j = 0; j < WIDE_ITEMS; j++) {
                const int w = w0 + j * RTHREADS + tid;
                if (w < wn) { ra[j] = wr[2 * w]; rb[j] = wr[2 * w + 1]; }
            }
#pragma unroll
            for (int j = 0; j < WIDE_ITEMS; j++) {
                const int w = w0 + j * RTHREADS + tid;
                if (w0 + j * RTHREADS >= wn) break;                     // block-uniform
                const bool k = w < wn && coords_may_touch_tile(ra[j].x, ra[j].y, ra[j].z, ra[j].w, rb[j].x, rb[j].y, t_xlo, t_xhi,
                                                               t_ylo, t_yhi);
                bool line = false;
                if (k && P.keys) {      // (P.keys: near >= 0)
                    const float x0 = ra[j].x, y0 = ra[j].y, x1 = ra[j].z, y1 = ra[j].w, x2 = rb[j].x, y2 = rb[j].y;
                    const int ne = (x0 == x1 && y0 == y1 ? 1 : 0) + (x1 == x2 && y1 == y2 ? 1 : 0) + (x2 == x0 && y2 == y0 ? 1 : 0);
                    const float amax = fmaxf(fmaxf(fmaxf(fabsf(x0), fabsf(y0)), fmaxf(fabsf(x1), fabsf(y1))), fmaxf(fabsf(x2), fabsf(y2)));
                    const float ex = (x0 == x1 && y0 == y1) ? x2 - x0 : x1 - x0, ey = (x0 == x1 && y0 == y1) ? y2 - y0 : y1 - y0;
                    line = ne == 1 && amax <= 4.0f && fmaxf(fabsf(ex), fabsf(ey)) >= 1e-6f;
                }
                const unsigned long long bal = __ballot(k && !line);
                if (bal) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_qn, __popcll(bal));
                    base = __shfl(base, 0, 64);
                    if (k && !line) s_queue[base + __popcll(bal & ((1ull << lane) - 1ull))] = __builtin_bit_cast(int, rb[j].z);
                }
                const unsigned long long bld = __ballot(line);
                if (bld) {              // the record's slot number, from the end of the queue
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&s_dn, __popcll(bld));
                    base = __shfl(base, 0, 64);
                    if (line) {
                        const int slot = base + __popcll(bld & ((1ull << lane) - 1ull));
                        s_queue[QCAP - 1 - slot] = w;
                        if (slot < RTHREADS) stage_line(slot, ra[j], rb[j]);
                    }
                }
            }
            __syncthreads();
            const int qn = s_qn, dn = s_dn;
            __syncthreads();        // every wave has read the counts before any wave's next append: the flush decision IS block-uniform
            if (qn + dn > QCAP - RTHREADS * WIDE_ITEMS || w0 + RTHREADS * WIDE_ITEMS >= wn) {
                walk_lines(dn);                                    // first: it reads the staging array this round filled; ends with a barrier
                process_queue(s_queue, qn);                        // ends with a barrier
                if (tid == 0) { s_qn = 0; s_dn = 0; }
                __syncthreads();
            }
        }
    }
    FaceBox bx[SCAN_ITEMS];
    auto fetch_boxes = [&](int step) {
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const int fn = step * SCAN_CHUNK + j * RTHREADS + tid;
            FaceBox b = empty_box();
            if (fn < nf) b = boxes[fn];
            bx[j] = b;
        }
    };
    if (nsteps > 0) fetch_boxes(0);
    for (int step = 0; step < nsteps; step++) {
        FaceBox cur[SCAN_ITEMS];
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) cur[j] = bx[j];
        if (step + 1 < nsteps) fetch_boxes(step + 1);          // next step's boxes are in flight during this step
#pragma unroll
        for (int j = 0; j < SCAN_ITEMS; j++) {
            const int fn = step * SCAN_CHUNK + j * RTHREADS + tid;
            const FaceBox b = cur[j];
            bool k = false;
            const bool exact_only = (b.xlo == BOX_EXACT);
            if (fn < nf && (exact_only || (b.xlo <= tx1 && b.xhi >= tx0 && b.ylo <= ty1 && b.yhi >= ty0))) {
                const float* f = faces + (size_t)fn * 9;
                const float x0 = f[0], y0 = f[1], x1 = f[3], y1 = f[4], x2 = f[6], y2 = f[7];
                k = !(edge_rejects_tile(x0, y0, x1, y1, t_xlo, t_xhi, t_ylo, t_yhi) ||
                      edge_rejects_tile(x1, y1, x2, y2, t_xlo, t_xhi, t_ylo, t_yhi) ||
                      edge_rejects_tile(x2, y2, x0, y0, t_xlo, t_xhi, t_ylo, t_yhi));
            }
            const unsigned long long bal = __ballot(k);
            if (bal) {                                          // wave-uniform: most steps add nothing
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_qn, __popcll(bal));
                base = __shfl(base, 0, 64);
                if (k) s_queue[base + __popcll(bal & ((1ull << lane) - 1ull))] = fn;
            }
        }
        if ((step % FLUSH_EVERY) == FLUSH_EVERY - 1 || step == nsteps - 1) {
            __syncthreads();
            const int qn = s_qn;
            __syncthreads();        // as above: no wave appends (next step) before all have read the count
            if (qn > 2 * SCAN_CHUNK || step == nsteps - 1) {
                process_queue(s_queue, qn);                    // ends with a barrier
                if (tid == 0) s_qn = 0;
                __syncthreads();
            }
        }
    }

    if (!in_img) return;
    {                   // winner of the face-parallel paths for this pixel: same rule (smallest zp, then smallest face index)
        key = min(key, s_key[tid]);     // (walk_lines ended with a barrier; the key order IS that rule)
        if (key != KEY_EMPTY) {
            const float kz = __builtin_bit_cast(float, (unsigned)(key >> 32));
            const int kf = (int)(unsigned)(key & 0xffffffffull);
            if (kz < best_z || (kz == best_z && best >= 0 && kf < best)) {
                // recompute this face's weights at this pixel: the same arithmetic on the same inputs, hence the same bits
                const float* f = faces + (size_t)kf * 9;
                const float* fi = faces_inv + (size_t)kf * 9;
                float zp, w0, w1, w2;
                cand_depth(make_float4(fi[0], fi[1], fi[2], fi[3]), make_float4(fi[4], fi[5], fi[6], fi[7]),
                           make_float4(fi[8], f[2], f[5], f[8]), fxi, fyi, P.near_, P.far_, zp, w0, w1, w2);
                best_z = zp; best = kf; bw0 = w0; bw1 = w1; bw2 = w2;
            }
        }
    }
    const int yo = P.flip ? (is - 1 - yi) : yi;
    const size_t pix = ((size_t)bn * is + yo) * is + xi;

    if (MODE == 0) {
        if (best >= 0) {  // uncovered pixels keep the caller's pre-fill (cu:156-168)
            P.depth_map[pix] = best_z;
            P.face_index_map[pix] = best;
            P.weight_map[3 * pix + 0] = bw0;
            P.weight_map[3 * pix + 1] = bw1;
            P.weight_map[3 * pix + 2] = bw2;
            if (P.face_inv_map) {
                const float* fi = faces_inv + (size_t)best * 9;
#pragma unroll
                for (int k = 0; k < 9; k++) P.face_inv_map[9 * pix + k] = fi[k];
            }
        }
        return;
    }

    // ---- MODE 1: network.Rasterizer.forward attribute interpolation (network.py:176-214) ----
    const rnr_gbuffer& G = P.gb;
    const int fa = best >= 0 ? best : nf - 1;  // torch indexing wraps -1 to the last face (weights are 0 there)
    const float* f = faces + (size_t)fa * 9;
    const float depth = best_z;                 // far on background (rasterize.py:52)
    const float wp0 = ((1.0f / f[2]) * bw0) * depth;
    const float wp1 = ((1.0f / f[5]) * bw1) * depth;
    const float wp2 = ((1.0f / f[8]) * bw2) * depth;
    if (G.face_index_map) G.face_index_map[pix] = best;
    if (G.alpha) G.alpha[pix] = best >= 0 ? 1.0f : 0.0f;
    if (G.depth) G.depth[pix] = depth;
    if (G.raw_weight_map) {
        G.raw_weight_map[3 * pix + 0] = bw0; G.raw_weight_map[3 * pix + 1] = bw1; G.raw_weight_map[3 * pix + 2] = bw2;
    }
    if (G.weight_map) {
        G.weight_map[3 * pix + 0] = wp0; G.weight_map[3 * pix + 1] = wp1; G.weight_map[3 * pix + 2] = wp2;
    }
    if (G.uv_map) {
        const int32_t* ti = P.mesh.f_vt_idx + (size_t)fa * 3;
        const float* a = P.mesh.vt + (size_t)ti[0] * 2;
        const float* b = P.mesh.vt + (size_t)ti[1] * 2;
        const float* c = P.mesh.vt + (size_t)ti[2] * 2;
        float u = a[0] * wp0 + b[0] * wp1 + c[0] * wp2;
        float v = a[1] * wp0 + b[1] * wp1 + c[1] * wp2;
        u = u - floorf(u);
        v = v - floorf(v);
        G.uv_map[2 * pix + 0] = u;
        G.uv_map[2 * pix + 1] = v;
    }
    float R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, T[3] = {0, 0, 0};
    if (P.pose) {
        const float* ps = P.pose + (size_t)bn * 16;
#pragma unroll
        for (int r = 0; r < 3; r++) {
            R[3 * r + 0] = ps[4 * r + 0]; R[3 * r + 1] = ps[4 * r + 1]; R[3 * r + 2] = ps[4 * r + 2];
            T[r] = ps[4 * r + 3];
        }
    }
    if (G.normal_map || G.normal_map_cam) {
        const int32_t* ni = P.mesh.f_vn_idx + (size_t)fa * 3;
        const float* a = P.mesh.vn + (size_t)ni[0] * 3;
        const float* b = P.mesh.vn + (size_t)ni[1] * 3;
        const float* c = P.mesh.vn + (size_t)ni[2] * 3;
        float n0 = a[0] * wp0 + b[0] * wp1 + c[0] * wp2;
        float n1 = a[1] * wp0 + b[1] * wp1 + c[1] * wp2;
        float n2 = a[2] * wp0 + b[2] * wp1 + c[2] * wp2;
        float inv = 1.0f / fmaxf(sqrtf(n0 * n0 + n1 * n1 + n2 * n2), 1e-12f);  // F.normalize
        n0 *= inv; n1 *= inv; n2 *= inv;
        if (G.normal_map) {
            G.normal_map[3 * pix + 0] = n0; G.normal_map[3 * pix + 1] = n1; G.normal_map[3 * pix + 2] = n2;
        }
        if (G.normal_map_cam) {
            float c0 = R[0] * n0 + R[1] * n1 + R[2] * n2;
            float c1 = R[3] * n0 + R[4] * n1 + R[5] * n2;
            float c2 = R[6] * n0 + R[7] * n1 + R[8] * n2;
            float ic = 1.0f / fmaxf(sqrtf(c0 * c0 + c1 * c1 + c2 * c2), 1e-12f);
            G.normal_map_cam[3 * pix + 0] = c0 * ic;
            G.normal_map_cam[3 * pix + 1] = c1 * ic;
            G.normal_map_cam[3 * pix + 2] = c2 * ic;
        }
    }
    if (G.position_map || G.position_map_cam) {
        const int32_t* vi = P.mesh.f_v_idx + (size_t)fa * 3;
        const float* a = P.mesh.v + (size_t)vi[0] * 3;
        const float* b = P.mesh.v + (size_t)vi[1] * 3;
        const float* c = P.mesh.v + (size_t)vi[2] * 3;
        const float p0 = a[0] * wp0 + b[0] * wp1 + c[0] * wp2;
        const float p1 = a[1] * wp0 + b[1] * wp1 + c[1] * wp2;
        const float p2 = a[2] * wp0 + b[2] * wp1 + c[2] * wp2;
        if (G.position_map) {
            G.position_map[3 * pix + 0] = p0; G.position_map[3 * pix + 1] = p1; G.position_map[3 * pix + 2] = p2;
        }
        if (G.position_map_cam) {
            G.position_map_cam[3 * pix + 0] = R[0] * p0 + R[1] * p1 + R[2] * p2 + T[0];
            G.position_map_cam[3 * pix + 1] = R[3] * p0 + R[4] * p1 + R[5] * p2 + T[1];
            G.position_map_cam[3 * pix + 2] = R[6] * p0 + R[7] * p1 + R[8] * p2 + T[2];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// per-face texture cube sampling (API completeness; the hot path passes an all-zero texture)
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
texture_sampling_kernel(const float* __restrict__ faces, const float* __restrict__ textures,
                        const int32_t* __restrict__ face_index_map, const float* __restrict__ weight_map,
                        const float* __restrict__ depth_map, float* __restrict__ rgb_map,
                        int32_t* __restrict__ sampling_index_map, float* __restrict__ sampling_weight_map,
                        long npix, int nf, int is, int ts, float eps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const int fidx = face_index_map[i];
    if (fidx < 0) return;
    const int bn = (int)(i / ((long)is * is));
    const float* f = faces + ((size_t)bn * nf + fidx) * 9;
    const float* tex = textures + ((size_t)bn * nf + fidx) * ts * ts * ts * 3;
    const float depth = depth_map[i];
    float t[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {  // cu:208-213
        float v = weight_map[3 * i + k] * (float)(ts - 1) * (depth / f[3 * k + 2]);
        v = fmaxf(v, 0.0f);
        v = fminf(v, (float)(ts - 1) - eps);
        t[k] = v;
    }
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll
    for (int corner = 0; corner < 8; corner++) {  // cu:217-236
        float w = 1.0f;
        int ti[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int b = (int)t[k];
            const float frac = t[k] - (float)b;
            if (((corner >> k) & 1) == 0) { w *= 1.0f - frac; ti[k] = b; }
            else                          { w *= frac;        ti[k] = b + 1; }
        }
        const int isc = ti[0] * ts * ts + ti[1] * ts + ti[2];
        acc0 += w * tex[isc * 3 + 0];
        acc1 += w * tex[isc * 3 + 1];
        acc2 += w * tex[isc * 3 + 2];
        sampling_index_map[8 * i + corner] = isc;
        sampling_weight_map[8 * i + corner] = w;
    }
    rgb_map[3 * i + 0] = acc0; rgb_map[3 * i + 1] = acc1; rgb_map[3 * i + 2] = acc2;
}

static size_t box_bytes(int batch, int nf) { return align_up((size_t)batch * nf * sizeof(FaceBox), 256); }
static size_t face_bytes(int batch, int nf) { return align_up((size_t)batch * nf * 9 * sizeof(float), 256); }
static int num_tiles(int is) { const int t = (is + TILE - 1) / TILE; return t * t; }
// binning scratch: [tile_count B*ntiles | wide_count B] (zeroed every call) | tile_list | wide_list
static size_t bin_counter_bytes(int batch, int is) { return align_up((size_t)batch * (num_tiles(is) + 1) * sizeof(int), 256); }
static size_t key_bytes(int batch, int is) { return align_up((size_t)batch * is * is * sizeof(unsigned long long), 256); }
static size_t wide_rec_bytes(int batch, int nf) { return align_up((size_t)batch * nf * WIDE_REC_FLOATS * sizeof(float), 256); }
// ... | keys | wide records (behind everything earlier rounds laid out: gbuffer_clear_regions' offsets are unchanged)
static size_t bin_bytes(int batch, int nf, int is) {
    return bin_counter_bytes(batch, is) + align_up((size_t)batch * num_tiles(is) * BIN_CAP * sizeof(int), 256) +
           align_up((size_t)batch * nf * sizeof(int), 256) + key_bytes(batch, is) + wide_rec_bytes(batch, nf);
}

// Small trusted faces are resolved face-parallel into P.keys (near >= 0), everything else is binned into the tile lists;
// Clears the bin counters (zeros) and the per-pixel depth keys (all ones) in one launch.  Not hipMemsetAsync: memset
// nodes of a captured HIP graph went stale on replay once any other copy / fill had run in between (GPU write fault,
// scripts/exp_graph2.py raster), kernels replay correctly.
__global__ void __launch_bounds__(256)
raster_clear_kernel(uint4* __restrict__ counters, long counter_vec, uint4* __restrict__ keys, long key_vec) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < counter_vec) counters[i] = make_uint4(0u, 0u, 0u, 0u);
    else if (i - counter_vec < key_vec) keys[i - counter_vec] = make_uint4(~0u, ~0u, ~0u, ~0u);
}

// fills P.tile_count / P.tile_list / P.keys.
// setup != NULL: the per-face setup has not run yet; on the splat path it shares the splat's launch, otherwise it is launched first
static int run_binning(char* ws, const float* faces, const float* faces_inv, const FaceBox* boxes, int batch, int nf, int is,
                       RasterParams* P, hipStream_t st, bool precleared = false, const FaceSetupArgs* setup = nullptr) {
    const int ntiles = num_tiles(is);
    int* tile_count = reinterpret_cast<int*>(ws);
    int* wide_count = tile_count + (size_t)batch * ntiles;
    int* tile_list = reinterpret_cast<int*>(ws + bin_counter_bytes(batch, is));
    int* wide_list = reinterpret_cast<int*>(ws + bin_counter_bytes(batch, is) +
                                            align_up((size_t)batch * ntiles * BIN_CAP * sizeof(int), 256));
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(
        ws + bin_counter_bytes(batch, is) + align_up((size_t)batch * ntiles * BIN_CAP * sizeof(int), 256) +
        align_up((size_t)batch * nf * sizeof(int), 256));
    float* wide_rec = reinterpret_cast<float*>(reinterpret_cast<char*>(keys) + key_bytes(batch, is));
    const long total = (long)batch * nf;
    const bool splat = P->near_ >= 0.0f;            // the key order needs zp > 0, which the near test then guarantees
    if (!precleared) {   // both regions are 256-byte aligned and padded (bin_counter_bytes / key_bytes): whole uint4 stores
        const long cvec = (long)(bin_counter_bytes(batch, is) / sizeof(uint4));
        const long kvec = splat ? (long)(key_bytes(batch, is) / sizeof(uint4)) : 0;
        hipLaunchKernelGGL(raster_clear_kernel, dim3((unsigned)((cvec + kvec + 255) / 256)), dim3(256), 0, st,
                           reinterpret_cast<uint4*>(tile_count), cvec, reinterpret_cast<uint4*>(keys), kvec);
        if (int e = check_launch("raster_clear_kernel")) return e;
    }
    const dim3 fgrid((unsigned)((total + 255) / 256));
    float* faces_inv_w = const_cast<float*>(faces_inv);
    FaceBox* boxes_w = const_cast<FaceBox*>(boxes);
    if (setup && !splat) {
        if (setup->gather) hipLaunchKernelGGL(face_setup_kernel<true>, fgrid, dim3(256), 0, st, setup->faces_in, setup->v_uvz, setup->fidx,
                                              setup->faces_out, faces_inv_w, boxes_w, batch, nf, setup->nv, is);
        else hipLaunchKernelGGL(face_setup_kernel<false>, fgrid, dim3(256), 0, st, setup->faces_in, setup->v_uvz, setup->fidx,
                                setup->faces_out, faces_inv_w, boxes_w, batch, nf, setup->nv, is);
        if (int e = check_launch("face_setup_kernel")) return e;
    }
    if (splat && setup) {
        // lanes per face: four while the launch would otherwise leave most SIMDs with one wave or none (RNR_SPLAT_LPF overrides)
        static const int forced = [] { const char* e = getenv("RNR_SPLAT_LPF"); return e ? atoi(e) : 0; }();
        const int lpf = forced == 1 || forced == 4 ? forced : (total <= 2 * 65536 ? 4 : 1);
        const dim3 sgrid((unsigned)((total * lpf + 255) / 256));
#define RNR_LAUNCH_SETUP_SPLAT(G, L)                                                                                            \
        hipLaunchKernelGGL((setup_splat_faces_kernel<G, L>), sgrid, dim3(256), 0, st, setup->faces_in, setup->v_uvz, setup->fidx, \
                           setup->faces_out, faces_inv_w, boxes_w, keys, tile_count, tile_list, wide_count, wide_list, wide_rec,  \
                           batch, nf, setup->nv, is, P->near_, P->far_)
        if (setup->gather) { if (lpf == 4) RNR_LAUNCH_SETUP_SPLAT(true, 4); else RNR_LAUNCH_SETUP_SPLAT(true, 1); }
        else { if (lpf == 4) RNR_LAUNCH_SETUP_SPLAT(false, 4); else RNR_LAUNCH_SETUP_SPLAT(false, 1); }
#undef RNR_LAUNCH_SETUP_SPLAT
        if (int e = check_launch("setup_splat_faces_kernel")) return e;
        P->keys = keys;
    } else if (splat) {
        hipLaunchKernelGGL(splat_faces_kernel, fgrid, dim3(256), 0, st, faces, faces_inv, boxes,
                           keys, tile_count, tile_list, wide_count, wide_list, wide_rec, batch, nf, is, P->near_, P->far_);
        if (int e = check_launch("splat_faces_kernel")) return e;
        P->keys = keys;
    } else {
        hipLaunchKernelGGL(bin_faces_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, faces, boxes, tile_count,
                           tile_list, wide_count, wide_list, wide_rec, batch, nf, is);
        if (int e = check_launch("bin_faces_kernel")) return e;
    }
    P->wide_count = wide_count;      // tested by the tile kernel (bin_wide_kernel's work, without its launch)
    P->wide_list = wide_list;
    P->wide_rec = wide_rec;
    P->tile_count = tile_count;
    P->tile_list = tile_list;
    return 0;
}

}  // namespace rnr

using namespace rnr;

extern "C" size_t rnr_raster_workspace_bytes(int batch_size, int num_faces, int image_size) {
    return box_bytes(batch_size, num_faces) + bin_bytes(batch_size, num_faces, image_size);
}

extern "C" int rnr_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                                          float* depth_map, float* face_inv_map, float* faces_inv,
                                          int batch_size, int num_faces, int image_size, float near_,
                                          float far_, int return_rgb, int return_alpha, int return_depth,
                                          void* workspace, void* stream) {
    (void)return_rgb; (void)return_alpha;
    RNR_REQUIRE(faces && face_index_map && weight_map && depth_map && faces_inv && workspace,
                "rnr_forward_face_index_map: null pointer argument");
    RNR_REQUIRE(batch_size > 0 && num_faces > 0 && image_size > 0 && image_size <= 16384,
                "rnr_forward_face_index_map: bad sizes B=%d nf=%d is=%d", batch_size, num_faces, image_size);
    RNR_REQUIRE(!return_depth || face_inv_map, "rnr_forward_face_index_map: return_depth needs face_inv_map");
    hipStream_t st = as_stream(stream);
    FaceBox* boxes = reinterpret_cast<FaceBox*>(workspace);
    const FaceSetupArgs setup = {faces, nullptr, nullptr, nullptr, 0, 0};
    RasterParams P = {};
    P.faces = faces; P.faces_inv = faces_inv; P.boxes = boxes; P.nf = num_faces; P.is = image_size;
    P.near_ = near_; P.far_ = far_; P.flip = 0;
    P.face_index_map = face_index_map; P.weight_map = weight_map; P.depth_map = depth_map;
    P.face_inv_map = return_depth ? face_inv_map : nullptr;
    if (int e = run_binning(reinterpret_cast<char*>(workspace) + box_bytes(batch_size, num_faces), faces, faces_inv, boxes,
                            batch_size, num_faces, image_size, &P, st, false, &setup)) return e;
    const int tiles = (image_size + TILE - 1) / TILE;
    hipLaunchKernelGGL(raster_tile_kernel<0>, dim3(tiles * tiles, batch_size), dim3(RTHREADS), 0, st, P);
    return check_launch("raster_tile_kernel<0>");
}

extern "C" int rnr_forward_texture_sampling(const float* faces, const float* textures,
                                            const int32_t* face_index_map, const float* weight_map,
                                            const float* depth_map, float* rgb_map,
                                            int32_t* sampling_index_map, float* sampling_weight_map,
                                            int batch_size, int num_faces, int image_size, int texture_size,
                                            float eps, void* stream) {
    RNR_REQUIRE(faces && textures && face_index_map && weight_map && depth_map && rgb_map &&
                    sampling_index_map && sampling_weight_map,
                "rnr_forward_texture_sampling: null pointer argument");
    RNR_REQUIRE(batch_size > 0 && num_faces > 0 && image_size > 0 && texture_size > 1,
                "rnr_forward_texture_sampling: bad sizes");
    const long npix = (long)batch_size * image_size * image_size;
    hipLaunchKernelGGL(texture_sampling_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0,
                       as_stream(stream), faces, textures, face_index_map, weight_map, depth_map, rgb_map,
                       sampling_index_map, sampling_weight_map, npix, num_faces, image_size, texture_size, eps);
    return check_launch("texture_sampling_kernel");
}

extern "C" size_t rnr_gbuffer_workspace_bytes(int num_views, int num_faces, int image_size) {
    return box_bytes(num_views, num_faces) + 2 * face_bytes(num_views, num_faces) + bin_bytes(num_views, num_faces, image_size);
}

// the two regions of an rnr_rasterize_gbuffer workspace that must be cleared before every call (tile / wide-list counters to
// 0, depth keys to ~0), as whole uint4 vectors: for rnr_frame_prepare (shade.hip), which clears them in its own launch
void rnr::gbuffer_clear_regions(void* workspace, int num_views, int num_faces, int image_size, uint4** counters, long* counter_vec,
                                uint4** keys, long* key_vec) {
    char* ws = reinterpret_cast<char*>(workspace) + box_bytes(num_views, num_faces) + 2 * face_bytes(num_views, num_faces);
    *counters = reinterpret_cast<uint4*>(ws);
    *counter_vec = (long)(bin_counter_bytes(num_views, image_size) / sizeof(uint4));
    *keys = reinterpret_cast<uint4*>(ws + bin_counter_bytes(num_views, image_size) +
                                     align_up((size_t)num_views * num_tiles(image_size) * BIN_CAP * sizeof(int), 256) +
                                     align_up((size_t)num_views * num_faces * sizeof(int), 256));
    *key_vec = (long)(key_bytes(num_views, image_size) / sizeof(uint4));
}

static int rasterize_gbuffer_impl(const rnr_mesh* mesh, const float* v_uvz, const float* pose, int num_views, int image_size,
                                  float near_, float far_, const rnr_gbuffer* out, void* workspace, void* stream, bool precleared);

extern "C" int rnr_rasterize_gbuffer(const rnr_mesh* mesh, const float* v_uvz, const float* pose,
                                     int num_views, int image_size, float near_, float far_,
                                     const rnr_gbuffer* out, void* workspace, void* stream) {
    return rasterize_gbuffer_impl(mesh, v_uvz, pose, num_views, image_size, near_, far_, out, workspace, stream, false);
}

extern "C" int rnr_rasterize_gbuffer_prepared(const rnr_mesh* mesh, const float* v_uvz, const float* pose,
                                              int num_views, int image_size, float near_, float far_,
                                              const rnr_gbuffer* out, void* workspace, void* stream) {
    return rasterize_gbuffer_impl(mesh, v_uvz, pose, num_views, image_size, near_, far_, out, workspace, stream, true);
}

static int rasterize_gbuffer_impl(const rnr_mesh* mesh, const float* v_uvz, const float* pose, int num_views, int image_size,
                                  float near_, float far_, const rnr_gbuffer* out, void* workspace, void* stream, bool precleared) {
    RNR_REQUIRE(mesh && v_uvz && out && workspace, "rnr_rasterize_gbuffer: null pointer argument");
    RNR_REQUIRE(mesh->v && mesh->f_v_idx && mesh->num_faces > 0 && mesh->num_vertices > 0,
                "rnr_rasterize_gbuffer: incomplete mesh");
    RNR_REQUIRE(!(out->uv_map) || (mesh->vt && mesh->f_vt_idx), "rnr_rasterize_gbuffer: uv_map needs vt");
    RNR_REQUIRE(!(out->normal_map || out->normal_map_cam) || (mesh->vn && mesh->f_vn_idx),
                "rnr_rasterize_gbuffer: normal maps need vn");
    RNR_REQUIRE(!(out->normal_map_cam || out->position_map_cam) || pose,
                "rnr_rasterize_gbuffer: camera-space maps need pose");
    RNR_REQUIRE(num_views > 0 && image_size > 0 && image_size <= 16384, "rnr_rasterize_gbuffer: bad sizes");
    hipStream_t st = as_stream(stream);
    const int nf = mesh->num_faces;
    char* ws = reinterpret_cast<char*>(workspace);
    FaceBox* boxes = reinterpret_cast<FaceBox*>(ws);
    float* faces = reinterpret_cast<float*>(ws + box_bytes(num_views, nf));
    float* faces_inv = reinterpret_cast<float*>(ws + box_bytes(num_views, nf) + face_bytes(num_views, nf));
    const FaceSetupArgs setup = {nullptr, v_uvz, mesh->f_v_idx, faces, mesh->num_vertices, 1};
    RasterParams P = {};
    P.faces = faces; P.faces_inv = faces_inv; P.boxes = boxes; P.nf = nf; P.is = image_size;
    P.near_ = near_; P.far_ = far_; P.flip = 1;
    P.mesh = *mesh; P.gb = *out; P.pose = pose;
    if (int e = run_binning(ws + box_bytes(num_views, nf) + 2 * face_bytes(num_views, nf), faces, faces_inv, boxes, num_views,
                            nf, image_size, &P, st, precleared, &setup)) return e;
    const int tiles = (image_size + TILE - 1) / TILE;
    hipLaunchKernelGGL(raster_tile_kernel<1>, dim3(tiles * tiles, num_views), dim3(RTHREADS), 0, st, P);
    return check_launch("raster_tile_kernel<1>");
}
