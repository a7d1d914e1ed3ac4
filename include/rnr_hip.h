/*
 * rnr_hip.h — C ABI of librnr_hip.so: the MI355X (gfx950) implementation of the deferred-render hot path
 * of LansburyCH/relightable-nr (rasterize -> neural-texture sample -> SH basis -> RenderingNet U-Net ->
 * ray render).  Plain pointers and sizes only; no torch / ATen types.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) unless its name ends in `_host`;
 *   - tensors are dense, row-major, float32 / int32, in the layouts written next to each argument;
 *   - the CALLER allocates every output (and pre-fills it where the reference does, see each function);
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream, which is what the
 *     reference's <<<blocks, threads>>> launches use, rasterize_cuda_kernel.cu:615,629,670);
 *   - return value: 0 = success; non-zero = error, message available from rnr_last_error().
 *     Unlike the reference (which only printf()s kernel-launch failures, rasterize_cuda_kernel.cu:623-625)
 *     every launch is checked with hipGetLastError() and reported;
 *   - the device entry points only enqueue kernels on `stream` (no allocation, no host synchronisation, no
 *     hipMemset / hipMemcpy), so a sequence of calls can be captured into a HIP graph and replayed; no kernel uses
 *     scratch memory.
 *
 * Each entry point cites the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef RNR_HIP_H
#define RNR_HIP_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RNR_ABI_VERSION 1

int rnr_abi_version(void);
/* Thread-local, NUL-terminated description of the last failing call on this thread ("" if none). */
const char* rnr_last_error(void);

/* =====================================================================================================
 * 1. neural_renderer.cuda.rasterize — drop-in for the pybind module (rasterize_cuda.cpp:124-191)
 * ===================================================================================================== */

/* Bytes of scratch rnr_forward_face_index_map needs for (batch, num_faces, image_size). */
size_t rnr_raster_workspace_bytes(int batch_size, int num_faces, int image_size);

/*
 * forward_face_index_map (rasterize_cuda.cpp:66-98 -> rasterize_cuda_kernel.cu:24-169, launch 595-650).
 *   faces          [B, nf, 3, 3]  (x_ndc, y_ndc, z_cam) per vertex
 *   face_index_map [B, is, is]    int32, caller pre-fills -1      (rasterize.py:50)
 *   weight_map     [B, is, is, 3] caller pre-fills 0              (rasterize.py:51)
 *   depth_map      [B, is, is]    caller pre-fills `far`          (rasterize.py:52)
 *   face_inv_map   [B, is, is, 3, 3] written when return_depth != 0 (may be NULL otherwise)
 *   faces_inv      [B, nf, 3, 3]  caller pre-fills 0              (rasterize.py:163)
 *   workspace      rnr_raster_workspace_bytes(B, nf, is) bytes of device scratch
 * Rows are in the extension's native order (row 0 = bottom of the image); the vertical flip is done by the
 * Python layer (rasterize.py:307-321).  Only covered pixels are written.  Results are bit-identical to the
 * reference kernels evaluated in IEEE binary32 without FMA contraction.
 */
int rnr_forward_face_index_map(const float* faces, int32_t* face_index_map, float* weight_map,
                               float* depth_map, float* face_inv_map, float* faces_inv,
                               int batch_size, int num_faces, int image_size, float near_, float far_,
                               int return_rgb, int return_alpha, int return_depth,
                               void* workspace, void* stream);

/*
 * forward_texture_sampling (rasterize_cuda.cpp:100-122 -> rasterize_cuda_kernel.cu:171-242, launch 652-691).
 *   textures [B, nf, ts, ts, ts, 3]; rgb_map [B, is, is, 3]; sampling_index_map [B, is, is, 8] int32;
 *   sampling_weight_map [B, is, is, 8].  Only covered pixels are written.
 */
int rnr_forward_texture_sampling(const float* faces, const float* textures, const int32_t* face_index_map,
                                 const float* weight_map, const float* depth_map, float* rgb_map,
                                 int32_t* sampling_index_map, float* sampling_weight_map,
                                 int batch_size, int num_faces, int image_size, int texture_size, float eps,
                                 void* stream);

/*
 * backward_pixel_map (rasterize_cuda.cpp:124-147 -> rasterize_cuda_kernel.cu:244-498, launch 693-732): gradient of
 * the rgb / alpha maps with respect to the NDC x,y of the face vertices (the silhouette sweep of Kato et al.).
 *   faces [B,nf,3,3]; face_index_map [B,is,is]; rgb_map / grad_rgb_map [B,is,is,3] (read when return_rgb);
 *   alpha_map / grad_alpha_map [B,is,is] (read when return_alpha); grad_faces [B,nf,3,3] caller pre-fills 0
 *   (rasterize.py:112): rows of front-facing faces are OVERWRITTEN, rows of back faces are left untouched.
 * Maps are in the extension's native row order.  Each gradient entry is accumulated by one thread in the
 * reference's order, so the result is bit-identical to the reference kernel run without FMA contraction.
 * No-op when both flags are 0 (rasterize.py:203-204).
 */
int rnr_backward_pixel_map(const float* faces, const int32_t* face_index_map, const float* rgb_map,
                           const float* alpha_map, const float* grad_rgb_map, const float* grad_alpha_map,
                           float* grad_faces, int batch_size, int num_faces, int image_size, float eps,
                           int return_rgb, int return_alpha, void* stream);

/*
 * backward_textures (rasterize_cuda.cpp:149-165 -> rasterize_cuda_kernel.cu:500-535, launch 734-763):
 *   grad_textures [B,nf,ts,ts,ts,3] += sampling_weight * grad_rgb at the 8 texels recorded by
 *   rnr_forward_texture_sampling (caller pre-fills 0, rasterize.py:114).  Float atomics: summation order is
 *   unspecified, as in the reference.
 */
int rnr_backward_textures(const int32_t* face_index_map, const float* sampling_weight_map,
                          const int32_t* sampling_index_map, const float* grad_rgb_map, float* grad_textures,
                          int batch_size, int num_faces, int image_size, int texture_size, void* stream);

/*
 * backward_depth_map (rasterize_cuda.cpp:167-189 -> rasterize_cuda_kernel.cu:537-592, launch 765-800): ADDS the
 * gradient of the depth map to grad_faces [B,nf,3,3] (called after rnr_backward_pixel_map, rasterize.py:145-152).
 *   depth_map [B,is,is]; face_inv_map [B,is,is,3,3] and weight_map [B,is,is,3] as written by
 *   rnr_forward_face_index_map(return_depth = 1); grad_depth_map [B,is,is].  Float atomics (pre-reduced per wave).
 */
int rnr_backward_depth_map(const float* faces, const float* depth_map, const int32_t* face_index_map,
                           const float* face_inv_map, const float* weight_map, const float* grad_depth_map,
                           float* grad_faces, int batch_size, int num_faces, int image_size, void* stream);

/*
 * neural_renderer.cuda.load_textures.load_textures (load_textures_cuda.cpp:20-34 -> load_textures_cuda_kernel.cu:6-152):
 * bake a texture image into the per-face texture cubes of the faces flagged in is_update.
 *   image [ih, iw, 3]; faces [nf, 3, 2] uv per face corner, WRAPPED IN PLACE (REPEAT 0 / MIRRORED_REPEAT 1 /
 *   CLAMP_TO_EDGE 2; CLAMP_TO_BORDER 3 leaves them and writes zero cubes); textures [nf, ts, ts, ts, 3] in/out;
 *   is_update [nf] int32.  The reference re-wraps the coordinates from every texel thread (a race when a coordinate is
 *   an exact integer); here each coordinate is wrapped once.
 */
int rnr_load_textures(const float* image, float* faces, float* textures, const int32_t* is_update, int num_faces,
                      int texture_size, int image_height, int image_width, int texture_wrapping, int use_bilinear,
                      void* stream);

/*
 * neural_renderer.cuda.create_texture_image.create_texture_image (create_texture_image_cuda.cpp:17-29 ->
 * create_texture_image_cuda_kernel.cu:8-163): lay the face cubes out as tiles of a texture atlas.
 *   vertices_all [nf, 3, 2] tile triangles in pixels (save_obj.py:10-24); textures [nf, tsi, tsi, tsi, 3];
 *   image [image_height, image_width, 3], tile grid width = int(sqrt(nf - 1)) + 1, tile size = image_width / that.
 *   Tiles beyond the last face are left as the caller filled them (the reference reads out of bounds there).
 */
int rnr_create_texture_image(const float* vertices_all, const float* textures, float* image, int num_faces,
                             int texture_size_in, int image_height, int image_width, float eps, void* stream);

/* =====================================================================================================
 * 2. Fused hot path (one view batch = N camera poses of one mesh)
 * ===================================================================================================== */

/*
 * nr.projection (neural_renderer/projection.py:6-53) for a shared mesh.
 *   vertices [nv, 3] world; K [N,3,3]; R [N,3,3]; t [N,3]; dist_coeffs [N,5] or NULL (= zeros);
 *   offset/scale [N,2] or both NULL; out [N, nv, 3] = (u_ndc, v_ndc, z_cam).
 */
int rnr_project_vertices(const float* vertices, const float* K, const float* R, const float* t,
                         const float* dist_coeffs, const float* offset, const float* scale,
                         float* out, int num_views, int num_vertices, float orig_size, float eps,
                         void* stream);

/* Mesh description shared by the G-buffer kernels (all device pointers, int32 indices 0-based as produced
 * by load_obj.py:176-178). */
typedef struct rnr_mesh {
    const float* v;          /* [nv, 3] world positions (global_RT applied, network.py:127) */
    const float* vt;         /* [nvt, 2] */
    const float* vn;         /* [nvn, 3] (global_RT applied + normalised, network.py:128) */
    const int32_t* f_v_idx;  /* [nf, 3] */
    const int32_t* f_vt_idx; /* [nf, 3] */
    const int32_t* f_vn_idx; /* [nf, 3] */
    int num_vertices, num_texcoords, num_normals, num_faces;
} rnr_mesh;

/* Outputs of network.Rasterizer.forward (network.py:156-216) that are per-pixel maps; any pointer may be
 * NULL to skip that map.  Rows are already flipped (row 0 = top), as rasterize_rgbad returns them. */
typedef struct rnr_gbuffer {
    int32_t* face_index_map; /* [N,S,S]   -1 = background */
    float* alpha;            /* [N,S,S]   {0,1} */
    float* depth;            /* [N,S,S]   far on background */
    float* weight_map;       /* [N,S,S,3] perspective-corrected w'_k = w_k * depth / z_k (network.py:176-180) */
    float* raw_weight_map;   /* [N,S,S,3] the kernel's clamped+renormalised barycentrics */
    float* uv_map;           /* [N,S,S,2] wrapped to [0,1) (network.py:187-190) */
    float* normal_map;       /* [N,S,S,3] */
    float* normal_map_cam;   /* [N,S,S,3] */
    float* position_map;     /* [N,S,S,3] */
    float* position_map_cam; /* [N,S,S,3] */
} rnr_gbuffer;

size_t rnr_gbuffer_workspace_bytes(int num_views, int num_faces, int image_size);

/*
 * projected vertices -> per-face setup -> tiled z-resolve -> attribute interpolation, i.e.
 * renderer.py:244-257 + rasterize.py:255-340 + network.py:156-214 in one pass (no texture kernel: the hot
 * path feeds it an all-zero texture and discards rgb, network.py:140-142,157).
 *   v_uvz [N, nv, 3] from rnr_project_vertices; pose [N,4,4] (for the *_cam maps; may be NULL if unused).
 */
int rnr_rasterize_gbuffer(const rnr_mesh* mesh, const float* v_uvz, const float* pose, int num_views,
                          int image_size, float near_, float far_, const rnr_gbuffer* out,
                          void* workspace, void* stream);

/* rnr_rasterize_gbuffer for a workspace whose per-call regions rnr_frame_prepare has already cleared on the same stream
 * (same arguments, same results; one launch fewer). */
int rnr_rasterize_gbuffer_prepared(const rnr_mesh* mesh, const float* v_uvz, const float* pose, int num_views,
                                   int image_size, float near_, float far_, const rnr_gbuffer* out,
                                   void* workspace, void* stream);

/*
 * The per-call preliminaries of a frame batch in ONE launch (test_rnr.py:283-300 per view: R / t from the pose, nr.projection;
 * render.get_TBN_map's per-face tangents, render.py:135-150; LightingSH.reconstruct_lp, network.py:622-627), plus the clearing
 * of the rnr_rasterize_gbuffer workspace.  Every part is optional (NULL output = skipped) and bit-identical to its
 * stand-alone entry point (rnr_project_vertices with R = pose[:, :3, :3], t = pose[:, :3, 3], no distortion / offset / scale;
 * rnr_face_tangents; rnr_sh_reconstruct):
 *   K [N,3,3], pose [N,4,4] -> v_uvz [N, nv, 3];   tangents [nf, 3];
 *   lp_basis [lp_samples, lp_num_basis], lp_coeff [lp_num_basis, lp_channels] -> light_probe [lp_samples, lp_channels];
 *   gbuffer_workspace: an rnr_gbuffer_workspace_bytes(num_views, mesh->num_faces, image_size) buffer -> cleared for
 *   rnr_rasterize_gbuffer_prepared.
 */
int rnr_frame_prepare(const rnr_mesh* mesh, const float* K, const float* pose, int num_views, int image_size, float eps,
                      float* v_uvz, float* tangents, const float* lp_basis, const float* lp_coeff, float* light_probe,
                      int lp_samples, int lp_num_basis, int lp_channels, void* gbuffer_workspace, void* stream);

/* Per-face unit tangents of render.get_TBN_map (render.py:135-150); static per mesh.  out [nf,3]. */
int rnr_face_tangents(const rnr_mesh* mesh, float* out, void* stream);

/* Ray-sampler constants (network.RaySampler buffers `pivots_dir`, network.py:441-443), HOST pointers. */
typedef struct rnr_rays {
    const float* pivots_spec_host; /* [3, num_spec] */
    const float* pivots_diff_host; /* [3, num_diff] */
    int num_spec, num_diff;        /* <= 32 each */
} rnr_rays;

/*
 * G-buffer -> RenderingNet input, fusing render.get_TBN_map (render.py:152-166), camera.get_view_dir_map
 * (camera.py:5-32), the tangent-space view direction (test_rnr.py:314-315), the lmax=2 SH basis
 * (sph_harm.py:41-71), TextureMapper.forward (network.py:67-91 + misc.py:5-42), both RaySamplers
 * (network.py:445-472) and the channel assembly of test_rnr.py:349-356.
 *   textures[level] [S_l, S_l, C] (level sizes tex_sizes_host[level]); num_levels <= 8
 *   net_in  [N, H, W, Cpad] channel-last; channel order 3*(num_spec+num_diff) ray dirs (ray-major), normal (3),
 *           view_dir (3), neural texture (C); channels >= Cin are zero-filled
 *   rays_uv [N, H, W, 2, num_spec+num_diff] or NULL; neural_img [N, C, H, W] or NULL (API copies)
 *   sh_basis_map [N,H,W,9] or NULL (written when non-NULL)
 */
int rnr_shade_inputs(const int32_t* face_index_map, const float* alpha, const float* uv_map,
                     const float* normal_map, const float* face_tangents, int num_faces,
                     const float* proj_inv, const float* R_inv,
                     const float* const* textures_host, const int* tex_sizes_host, int num_levels,
                     int tex_channels, int sh_start_ch, const rnr_rays* rays,
                     float* net_in, int c_pad, float* rays_uv, float* neural_img, float* sh_basis_map,
                     int num_views, int height, int width, void* stream);

/* ---- U-Net convolution stack (pytorch_prototyping.py:96-277, 370-536), channel-last activations ---- */

enum { RNR_ACT_NONE = 0, RNR_ACT_LRELU02 = 1, RNR_ACT_RELU = 2 };
enum { RNR_CONV3x3_REFLECT = 0, RNR_CONV4x4S2_REFLECT = 1, RNR_CONVT4x4S2 = 2 };

/* One input source of a convolution: a raw (pre-normalisation) channel-last tensor plus the per-view,
 * per-channel affine + activation the producer's BatchNorm/bias/activation implies:
 *     value(n,h,w,c) = act(scale[n,c] * raw[n,h,w,c] + shift[n,c]).
 * scale/shift NULL = identity.  Two sources = torch.cat([src0, src1], 1) (pytorch_prototyping.py:429). */
typedef struct rnr_conv_src {
    const float* data;  /* [N, H, W, channels] */
    const float* scale; /* [N, channels] or NULL */
    const float* shift; /* [N, channels] or NULL */
    int channels;       /* multiple of 4 */
    int act;            /* RNR_ACT_* */
} rnr_conv_src;

/* Static description of one convolution of the U-Net.  Channel counts `*_pad` are the channel strides of the
 * channel-last tensors (multiples of 16; padding channels hold zeros / meet zero weights). */
typedef struct rnr_conv_desc {
    int kind;                 /* RNR_CONV3x3_REFLECT | RNR_CONV4x4S2_REFLECT | RNR_CONVT4x4S2 */
    int c_in0, c_in0_pad;     /* first source: live channels, channel stride */
    int c_in1, c_in1_pad;     /* second source of a skip concat (0, 0 if none) */
    int c_out, c_out_pad;     /* live output channels, channel stride of out_raw (multiple of 16) */
    int flags;                /* RNR_CONV_* bits, 0 by default */
} rnr_conv_desc;
/* `stats` already holds zeros when rnr_conv2d is called (rnr_bn_finalize_reset left them so): skip the memset. */
#define RNR_CONV_STATS_PREZEROED 1
/* fp32 emulation on the bf16 matrix cores: every operand is split exactly into three bf16 terms and the six leading
 * partial products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 (error of the order of fp32's own rounding,
 * 2.7x fewer MFMA cycles).  Must be set both when packing the weights and when convolving; layers it does not cover
 * (maps narrower than 16 pixels or of odd shape) run the fp32 MFMA kernels from the same buffer.  Range: finite operands below
 * 2^127 (the leading bf16 term of a larger value rounds to infinity); residual terms in fp32's subnormal range are
 * flushed, which only costs precision below 1e-38. */
#define RNR_CONV_F32_EMU_BF16X6 2
/* fp32 emulation on the fp16 matrix cores: every operand is split into TWO fp16 terms (22 significand bits, relative
 * representation error <= 2^-23) and the three leading partial products (hh, hl, lh) are accumulated in fp32 by
 * v_mfma_f32_32x32x16_f16: 5.3x fewer MFMA cycles than the exact-fp32 kernel, half of bf16x6.  Measured error against a
 * float64 convolution is below the exact-fp32 kernel's on every U-Net layer shape it covers (fewer accumulator roundings outweigh
 * the two missing significand bits; tests/test_gpu_unet.py).  Weights are pre-scaled per layer by a power of two at pack
 * time (2^k with |k| <= 40, undone exactly in the epilogue; a layer whose max |w| is below 2^-28 keeps fewer fp16 bits), so
 * any finite weights are fine; activations must satisfy
 * |act(scale * x + shift)| < 65504 — true for BatchNorm outputs and bounded network inputs; values below 2^-14 keep an
 * absolute precision of 2^-25 (fp16 subnormals are honoured by the MFMA).  Same packing / fallback rules as BF16X6;
 * the two flags are mutually exclusive. */
#define RNR_CONV_F32_EMU_F16X3 4
#define RNR_CONV_F32_EMU_ANY (RNR_CONV_F32_EMU_BF16X6 | RNR_CONV_F32_EMU_F16X3)
/* Winograd minimal filtering (Lavin & Gray 2016): F(2x2, 3x3) for the 3x3 convolutions — 16 multiplications per 2 x 2
 * output tile instead of 36 — and F(2x2, 2x2) for the two 4x4 stride-2 convolutions, which are sums of 2x2-tap correlations
 * (per input parity phase resp. per output parity class) — 9 instead of 16.  fp32 operands and fp32 accumulation on
 * v_mfma_f32_32x32x2_f32 like the direct kernels; the data transforms only add (F(2x2, 3x3) also halves weights), so the
 * result differs from the direct convolution by rounding of the order of a different summation order (scripts/wino_check.py,
 * tests/test_gpu_unet.py: max error against a float64 convolution <= 1.1e-5 of the output rms on every U-Net layer shape,
 * direct <= 8.6e-6).  Must be set both when packing the weights (the transformed image is stored behind the direct one) and
 * when convolving; calls it does not cover (maps that do not tile into 16 x 8 / 16 x 16 pixels, column counts that are not
 * multiples of 64 / 128, masked launches, too few tiles to fill the chip: rnr_conv_algorithm tells) run the direct kernels
 * from the same buffer.  Not combined with the emulation flags.  Non-finite inputs: the data transforms take differences of
 * neighbouring pixels, so an inf / NaN activation reaches every output of the 2 x 2 tiles whose patch contains it (a direct
 * convolution confines it to the outputs whose window contains it). */
#define RNR_CONV_WINOGRAD 8
/* (with RNR_CONV_WINOGRAD) F(4x4, 3x3) for the 3x3 convolutions whose maps tile into 32 x 16 pixels, whose columns into 64s and
 * whose grid fills the chip: 36 multiplications per 4 x 4 outputs (2.25 per output; F(2x2, 3x3): 4, direct: 9) on the
 * interpolation points (0, +-3/4, +-3/2, inf).  A flag of its own because the larger transforms (coefficients up to 3.375) put the
 * rounding error at ~4.5 x the direct form's (rms; F(2x2, 3x3): 1.3 x) — scripts/experiments/winograd_accuracy_study.py,
 * tests/test_gpu_unet.py: <= 1e-4 of the output peak against a float64 convolution on every U-Net layer shape.  The Python host
 * layer (rnr_amd.unet.UNetPlan) sets it BY DEFAULT since r04 (conv_algo 'winograd4'); a C caller chooses per descriptor.
 * Non-finite inputs: an inf / NaN activation reaches every output of the 4 x 4 tiles whose 6 x 6 input patch contains it
 * (F(2x2, .): 2 x 2 tiles of a 4 x 4 patch; direct: the outputs whose window contains it).  The F(4x4, 3x3) weight image is stored behind the F(2x2, 3x3) one (both flags when packing AND
 * convolving); shapes it does not cover run F(2x2, 3x3) / direct from the same buffer.  rnr_conv_algorithm reports 4. */
#define RNR_CONV_WINOGRAD4 16

/* Floats in the packed weight of `d` ([taps][c_in0_pad + c_in1_pad][c_out_pad], x4 parity classes for convT). */
size_t rnr_packed_weight_floats(const rnr_conv_desc* d);
/* PyTorch layout -> packed.  Conv2d weight [c_out, c_in0 + c_in1, kh, kw] (pytorch_prototyping.py:116, 250-268);
 * ConvTranspose2d weight [c_in0 + c_in1, c_out, 4, 4] (pytorch_prototyping.py:154-159). */
int rnr_pack_conv_weight(const rnr_conv_desc* d, const float* weight, float* packed, void* stream);

/* Which algorithm rnr_conv2d* runs for (desc, N, input H, input W): 0 = direct implicit GEMM, 1 = Winograd F(2x2, 3x3),
 * 3 = the same for the 80-column out layer (16 x 16 x 4 MFMA tiles), 2 = Winograd F(2x2, 2x2) (16 multiplications per 2 x 2
 * outputs instead of 36, resp. 9 instead of 16), 4 = Winograd F(4x4, 3x3) (RNR_CONV_WINOGRAD4); -1 = bad arguments.
 * Non-zero only with RNR_CONV_WINOGRAD in desc->flags.
 * Masked launches (tile_mask != NULL) run 0 unless the plan is 3; rnr_conv2d_ray always runs 0. */
int rnr_conv_algorithm(const rnr_conv_desc* d, int num_views, int in_h, int in_w);

/* Scratch bytes rnr_conv2d may need for (desc, N, input H, input W) (split-K partial slabs). */
size_t rnr_conv_workspace_bytes(const rnr_conv_desc* d, int num_views, int in_h, int in_w);

/*
 * out_raw[n,ho,wo,co] = sum_{taps,ci} value(src)[...] * weight  (no bias: it is folded into the consumer's
 * `shift`, or applied by rnr_ray_render / rnr_nhwc_to_nchw for the last layer).  When stats != NULL the call
 * also zeroes and then accumulates per-view, per-channel sum / sum-of-squares of out_raw into
 * stats [N, c_out_pad, 2] (float64), for the batch-statistics BatchNorm the reference runs at inference
 * (test_rnr.py:229-233).
 *   src0/src1: `channels` must equal c_in0_pad / c_in1_pad; src1 may be NULL when c_in1_pad == 0.
 *   out_raw [N, Ho, Wo, c_out_pad]  (Ho,Wo = H,W | H/2,W/2 | 2H,2W by kind)
 * Reproducibility: out_raw is bit-reproducible run to run (fixed summation order, also across split-K slabs); `stats` is
 * NOT — every workgroup adds its column sums with one float64 atomicAdd per channel, whose order varies, so the last
 * bits of the statistics (and of everything normalised with them) may differ between runs (tests compare at 1e-5).
 */
int rnr_conv2d(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
               const float* weight_packed, float* out_raw, double* stats, int num_views, int in_h, int in_w,
               void* workspace, size_t workspace_bytes, void* stream);

/*
 * Output-tile masking for a convolution whose output is consumed only where a mask is set — the out layer of
 * RenderingNet: the ray renderer zeroes every background pixel (rays_uv = -1 there, network.py:469-470, 497), so the
 * out-layer activations of all-background pixel tiles are never read.
 *   rnr_conv_tile_count   number of pixel tiles rnr_conv2d launches for (desc, N, H, W); 0 if that convolution cannot
 *                         be masked (only 3x3 convolutions on the LDS-halo plan without split-K can);
 *   rnr_conv_active_tiles tile_mask[t] = 1 iff any alpha [N,H,W] > 0 inside tile t;
 *   rnr_conv2d_masked     rnr_conv2d that leaves out_raw of tiles with tile_mask[t] == 0 untouched.  stats must be NULL
 *                         (batch statistics need every pixel).  tile_mask == NULL is rnr_conv2d.
 */
size_t rnr_conv_tile_count(const rnr_conv_desc* d, int num_views, int in_h, int in_w);
int rnr_conv_active_tiles(const rnr_conv_desc* d, const float* alpha, uint8_t* tile_mask, int num_views, int in_h,
                          int in_w, void* stream);
int rnr_conv2d_masked(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                      const float* weight_packed, float* out_raw, double* stats, int num_views, int in_h, int in_w,
                      void* workspace, size_t workspace_bytes, const uint8_t* tile_mask, void* stream);

/*
 * The convolution of the product path (rnr_amd.unet.UNetPlan): rnr_conv2d_masked plus what the reference runs as separate
 * passes behind it — the train-mode BatchNorm2d that follows the convolution (pytorch_prototyping.py:117-120, 250-268,
 * 154-161; per-view batch statistics, test_rnr.py:229-233) — inside ONE launch:
 *   bn != NULL && bn->gamma != NULL: the workgroups add the per-view, per-channel sum / sum of squares of out_raw into
 *     statistics shards inside `sync`; the last workgroup of a view to arrive turns them into
 *       scale[n,c] = gamma[c] / sqrt(var_biased + eps),  shift[n,c] = beta[c] - mean * scale[n,c]   (rnr_bn_finalize)
 *     and leaves the shards at zero.  Channels >= c_out of scale / shift get 0.
 *   split-K (small maps): the slices of a tile meet inside the launch (the last one to arrive adds the accumulator
 *     images in slice order) when there are at most 4 of them; deeper splits keep the reduce kernel, which then also
 *     finalises the BatchNorm.
 * `sync` (rnr_conv_sync_bytes(d, max_views, in_h, in_w) bytes, 256-byte aligned): arrival counters and statistics.  The
 * caller zero-fills it ONCE; every call expects zeros and leaves zeros, also for a different num_views <= max_views.  One
 * buffer per convolution in flight (calls on the same stream may share one).  After a failed launch: zero it again.
 * Run-to-run reproducibility: the statistics are double-precision atomics whose order varies between runs.  What a workgroup
 * adds is a sum of six to twelve float32 lane sums (<= 28 significant bits), so the double additions are EXACT — hence
 * order-independent — as long as the non-zero partial sums of a channel and view span less than ~2^17 in magnitude (53 - 28 - log2(number
 * of workgroups) bits of headroom for 512 workgroups per view); that is the case for activations of any trained network and is why 16 000 frame groups have
 * come out bit-identical run after run (scripts/t_fused_stress.py), but it is a property of the data, not of the code: with a wider
 * spread scale / shift — and everything downstream — may differ in the last bits between runs, exactly like rnr_conv2d +
 * rnr_bn_finalize.  The in-launch split-K combine itself is order-independent by construction.
 */
/* ZERO-INITIALISE this struct (`rnr_conv_bn bn = {0};` / memset) before filling it: it has grown (running_mean, running_var,
 * momentum were added in r05) and carries no size field — a caller compiled against the five-field form that leaves the tail
 * uninitialised hands garbage running_* pointers to rnr_conv2d_fused, i.e. a wild device write when num_views == 1. */
typedef struct rnr_conv_bn {
    const float* gamma; /* [c_out] BatchNorm weight; NULL = no BatchNorm behind this convolution */
    const float* beta;  /* [c_out] */
    float* scale;       /* [N, c_out_pad] out */
    float* shift;       /* [N, c_out_pad] out */
    float eps;
    /* torch.nn.BatchNorm2d's train-mode side effect (pytorch_prototyping.py:109, test_rnr.py:229-233 keep the layers in train mode):
     * running = (1 - momentum) * running + momentum * (batch mean | unbiased batch variance), [c_out] float32, updated in place by
     * the launch that finalises the statistics.  Statistics are per VIEW in this entry point and torch pools the whole batch, so
     * the two agree for num_views == 1 only: non-NULL pointers with num_views > 1 are refused (use rnr_conv2d +
     * rnr_bn_finalize_batch there).  NULL = no update. */
    float* running_mean;
    float* running_var;
    float momentum;
} rnr_conv_bn;
size_t rnr_conv_sync_bytes(const rnr_conv_desc* d, int max_views, int in_h, int in_w);
int rnr_conv2d_fused(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                     const float* weight_packed, float* out_raw, const rnr_conv_bn* bn, int num_views, int in_h, int in_w,
                     void* workspace, size_t workspace_bytes, void* sync, size_t sync_bytes, const uint8_t* tile_mask,
                     void* stream);

/*
 * The ray renderer folded into the out layer (opt-in: RNRPipeline(fuse_ray=True); network.py:253, 481-527, test_rnr.py:357-368).
 *   rnr_ray_weights  the half that does not depend on the U-Net: for pixel p, ray r, colour channel c
 *                        ray_w[p, 3 r + c] = albedo_group(r)[p, c] * env-map colour(direction of ray r)[c] / rays in the group
 *                    from the ray directions / albedo channels of net_in and the light probe (same arithmetic as
 *                    rnr_ray_render); background pixels (alpha = 0) and the padding columns >= 3 * rays get 0.
 *                    ray_w [N, H, W, c_w], c_w = the out layer's c_out_pad.
 *   rnr_conv2d_ray   the out-layer convolution whose epilogue produces the FRAME instead of its 78-channel output:
 *                        image[n, c, y, x] = sum_r (tanh(conv[n, y, x, 3 r + c] + bias[3 r + c]) + 1) * ray_w[n, y, x, 3 r + c]
 *                    straight from the MFMA accumulators (LDS transpose, 26 columns per sum): 12 bytes per pixel leave the
 *                    kernel instead of 320.  Only the exact-fp32 3x3 convolution on the 80-column plan (65 <= c_out <= 80,
 *                    c_out = 3 x rays, map width a multiple of 32, height of 8); anything else returns an error — run
 *                    rnr_conv2d_masked + rnr_ray_render there.  tile_mask as in rnr_conv2d_masked (skipped tiles get 0), laid
 *                    out for the DIRECT plan's 32 x 8-pixel tiles: RNR_CONV_WINOGRAD in d->flags is ignored by this entry
 *                    point, so build the mask with rnr_conv_active_tiles / rnr_conv_tile_count from the descriptor with
 *                    that flag cleared (with it set those describe the Winograd out layer's 16 x 4-pixel tiles).
 * Differs from rnr_ray_render in summation order only (<= 1e-6 on frames in [0, 2]).
 */
int rnr_ray_weights(const float* net_in, int c_pad, const float* alpha, const float* lp, int lp_h, int lp_w, int num_spec,
                    int num_diff, int albedo_diff_ch, int albedo_spec_ch, float* ray_w, int c_w, int num_views, int height,
                    int width, void* stream);
int rnr_conv2d_ray(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1, const float* weight_packed,
                   const float* ray_w, const float* bias, float* image, int num_views, int in_h, int in_w,
                   const uint8_t* tile_mask, void* stream);

/* stats [N,c_pad,2] (sum, sumsq over `count` pixels) + gamma/beta [channels] -> scale/shift [N,c_pad]:
 * scale = gamma / sqrt(var_biased + eps), shift = beta - mean * scale  (BatchNorm2d in train mode: per-view
 * batch statistics, biased variance); channels >= `channels` get scale = shift = 0. */
int rnr_bn_finalize(const double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                    int num_views, int channels, int c_pad, double count, float eps, void* stream);
/* rnr_bn_finalize that also resets stats to zero once consumed, so that the next rnr_conv2d into the same statistics
 * buffer can run with RNR_CONV_STATS_PREZEROED (no memset launch per layer). */
int rnr_bn_finalize_reset(double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                    int num_views, int channels, int c_pad, double count, float eps, void* stream);

/* Whole-batch variant: torch's train-mode BatchNorm2d reduces over (N,H,W) of the call (the reference forces that mode
 * at inference, test_rnr.py:229-233; with its N = 1 per call the two coincide).  The per-view partial sums of
 * rnr_conv2d are added over the `num_views` views (count = num_views * count_per_view), the same scale/shift is
 * written to every view, and — when the pointers are non-NULL — running_mean / running_var [channels] are updated as
 * torch does: running = (1 - momentum) * running + momentum * {mean, UNBIASED variance}.  Resets stats to zero. */
int rnr_bn_finalize_batch(double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                          float* running_mean, float* running_var, float momentum, int num_views, int channels,
                          int c_pad, double count_per_view, float eps, void* stream);

/* Layout helpers for the drop-in RenderingNet.forward (NCHW in / NCHW out). */
int rnr_nchw_to_nhwc(const float* in, float* out, int n, int c, int h, int w, int c_pad, void* stream);
/* out[n,c,h,w] = f(in[n,h,w,c] + bias[c]) with bias optional (NULL) and f = tanhf when apply_tanh != 0 */
int rnr_nhwc_to_nchw(const float* in, float* out, const float* bias, int apply_tanh,
                     int n, int c, int h, int w, int c_pad, void* stream);

/*
 * Last stage of a frame: out-layer bias + tanh (network.py:253), rays_lt = (y*0.5+0.5)*2
 * (test_rnr.py:357-359) and RayRenderer.forward (network.py:481-527) with seperate_albedo=True:
 * equirect uv of every ray from the ray directions stored in net_in, bilinear env-map taps
 * (network.py:497 + misc.py:5-42), light-transport weighted sums / num rays, times the albedos.
 *   unet_raw [N,H,W,c_out_pad] raw output of the out-layer conv (78 live channels, ray-major RGB)
 *   bias [3*(num_spec+num_diff)]; net_in as written by rnr_shade_inputs; alpha [N,H,W]
 *   lp [Hl, Wl, 3] environment map (LightingSH.reconstruct_lp, network.py:622-627)
 *   image [N,3,H,W]
 */
int rnr_ray_render(const float* unet_raw, int c_out_pad, const float* bias, const float* net_in, int c_pad,
                   const float* alpha, const float* lp, int lp_h, int lp_w, int num_spec, int num_diff,
                   int albedo_diff_ch, int albedo_spec_ch, float* image, int num_views, int height,
                   int width, void* stream);

/* ---- spherical harmonics (sph_harm.py:41-102) ---- */

/* Real orthonormal SH without Condon-Shortley phase, columns (l, m=-l..l); dirs [n,3] (need not be unit),
 * out [n, (lmax+1)^2].  lmax <= 16. */
int rnr_sh_basis(const float* dirs, float* out, int n, int lmax, void* stream);
/* out[s, c] = sum_b basis[s,b] * coeff[b,c]  (sph_harm.reconstruct_sh, sph_harm.py:91-102) */
int rnr_sh_reconstruct(const float* basis, const float* coeff, float* out, int num_samples, int num_basis,
                       int num_channels, void* stream);
/* out[b, c] = 4pi/num_samples * sum_s samples[s,c] * basis[s,b]  (sph_harm.fit_sh_coeff, sph_harm.py:74-88) */
int rnr_sh_fit(const float* samples, const float* basis, float* out, int num_samples, int num_basis,
               int num_channels, void* stream);

/* misc.interpolate_bilinear (misc.py:5-42): data [H,W,C], x/y [n] -> out [n,C]; optional tap index output
 * taps [n,4] int32 = (x0,y0,x1,y1) fetch indices (bit-exact integer part of the sampler). */
int rnr_interpolate_bilinear(const float* data, int h, int w, int c, const float* x, const float* y,
                             float* out, int32_t* taps, int n, void* stream);

/* Area-average resize of a channel-last image src [src_h,src_w,C] -> dst [dst_h,dst_w,C]: the
 * cv2.resize(..., interpolation=cv2.INTER_AREA) call of LightingLP.__init__ (network.py:667) restated from OpenCV's
 * published algorithm (fractional-coverage box filter when shrinking on both axes, "area-mode" bilinear otherwise).
 * cv2 is not available in this image: parity with OpenCV is unpinned; the integer-ratio case is the plain box mean. */
int rnr_resize_area(const float* src, float* dst, int src_h, int src_w, int dst_h, int dst_w, int channels, void* stream);

/* =====================================================================================================
 * 3. Stand-alone operators for the drop-in Python API (one reference function each).  The fused entry
 *    points above compute the same quantities without the intermediate HBM round trips.
 * ===================================================================================================== */

/* camera.get_view_dir_map (camera.py:5-32): out_world/out_cam [N,H,W,3] (out_cam may be NULL). */
int rnr_view_dir_map(const float* proj_inv, const float* R_inv, float* out_world, float* out_cam,
                     int num_views, int height, int width, void* stream);

/* render.get_TBN_map (render.py:152-166) given per-face unit tangents: out [N,H,W,3,3], columns (T,B,N). */
int rnr_tbn_map(const float* normal_map, const int32_t* face_index_map, const float* face_tangents,
                int num_faces, float* out, int num_views, int height, int width, void* stream);

/* The per-pixel 3x3 products of the reference's view loop, test_rnr.py:314:
 *   torch.matmul(TBN_map.reshape((-1, 3, 3)).transpose(-2, -1), view_dir_map.reshape((-1, 3, 1)))
 * tbn [P,3,3] (row-major records, as rnr_tbn_map writes them), vec [P,3], out [P,3];  out[p] = tbn[p]^T vec[p] when transposed != 0,
 * tbn[p] vec[p] otherwise.  ((c z + (b y + a x)) with fused multiply-adds: <= 1 ulp per term from the batched-GEMM result.)
 * The drop-in render.get_TBN_map returns a tensor that answers exactly this torch.matmul call with this entry point. */
int rnr_tbn_matvec(const float* tbn, const float* vec, float* out, long num_pixels, int transposed, void* stream);

/* network.RaySampler.forward (network.py:445-472).  reflect != 0: mode 'reflect' (needs view_tangent), else the
 * pivots themselves.  pivots_host [3,R] (HOST).  tbn [P,3,3], view_tangent [P,3], alpha [P];
 * rays_dir [P,3,R], rays_uv [P,2,R], rays_dir_tangent [P,3,R] (reflect mode only; may be NULL). */
int rnr_ray_sampler(int reflect, const float* pivots_host, int num_rays, const float* tbn,
                    const float* view_tangent, const float* alpha, float* rays_dir, float* rays_uv,
                    float* rays_dir_tangent, long num_pixels, void* stream);

/* network.TextureMapper.forward (network.py:67-91) for any channel count: uv_map [N,H,W,2],
 * sh_basis_map [N,H,W,9] or NULL, textures[level] [S_l,S_l,C] -> out [N,C,H,W]. */
int rnr_texture_mapper(const float* uv_map, const float* sh_basis_map, const float* const* textures_host,
                       const int* tex_sizes_host, int num_levels, int tex_channels, int sh_start_ch,
                       float* out, int num_views, int height, int width, void* stream);

/* network.RayRenderer.forward (network.py:481-527) on API-shaped tensors: rays_uv [N,H,W,2,R],
 * rays_lt [N,R,C,H,W], lp [lp_n,Hl,Wl,C] (lp_n = 1 or N), albedos [N,C,H,W] (albedo_diffuse may be NULL).
 * Outputs [N,C,H,W] (out required, others optional) and rays_color [N,R,C,H,W] (optional). */
int rnr_ray_renderer(const float* rays_uv, const float* rays_lt, const float* lp, int lp_n, int lp_h,
                     int lp_w, const float* albedo_specular, const float* albedo_diffuse, int channels,
                     int num_rays, int num_ray_diffuse, int no_albedo, int seperate_albedo,
                     float lp_scale_factor, float* out, float* out_specular, float* out_diffuse,
                     float* ltt_specular, float* ltt_diffuse, float* rays_color, int num_views, int height,
                     int width, void* stream);

/* =====================================================================================================
 * 4. Host-side data front-end: Wavefront OBJ reader behind nr.load_obj(normalization=False, load_texture=False)
 *    (neural_renderer/load_obj.py:108-209: four Python passes over the lines).  No GPU involved.
 *    Two calls: rnr_obj_scan counts the elements, the caller allocates, rnr_obj_parse fills:
 *    v [nv,3], vn [nvn,3], vt [nvt,2] float32 (correctly rounded double -> float32, like float() + astype(float32));
 *    f_v_idx / f_vt_idx / f_vn_idx [nf,3] int32, 0-based.  vt / vn index arrays are filled iff the file holds vt / vn
 *    lines (the reference's has_vt / has_vn, load_obj.py:133-176).  Triangles only; errors via rnr_last_error.
 * ===================================================================================================== */
typedef struct rnr_obj_counts {
    long num_vertices, num_normals, num_texcoords, num_faces;
} rnr_obj_counts;
int rnr_obj_scan(const char* text, size_t len, rnr_obj_counts* counts);
int rnr_obj_parse(const char* text, size_t len, const rnr_obj_counts* counts, float* v, float* vn, float* vt,
                  int32_t* f_v_idx, int32_t* f_vt_idx, int32_t* f_vn_idx);

/* =====================================================================================================
 * 5. Measurement aid (bench.py's `box_calibration`; no counterpart in the reference).
 *    rnr_calibrate_mfma_f32 runs `iters` x 32 register-resident v_mfma_f32_32x32x2_f32 per wave on every SIMD of the
 *    current device (`waves_per_simd` = 1 ... 8 waves per SIMD, full-mantissa operands, no memory traffic), times the launch with HIP events
 *    on `stream`, waits for it, and returns the rate in *tflops (and the duration in *seconds, if not NULL): what this
 *    device sustains NOW on the instruction the U-Net runs on (nominal 157.3 TFLOP/s).  scratch: >= 4 bytes of device
 *    memory (never written in practice).  Blocks the calling thread.
 * ===================================================================================================== */
int rnr_calibrate_mfma_f32(int iters, int waves_per_simd, float* scratch, double* tflops, double* seconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RNR_HIP_H */
