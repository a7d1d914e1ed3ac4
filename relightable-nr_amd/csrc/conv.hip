// RenderingNet U-Net convolutions as implicit GEMMs on the gfx950 matrix cores, exact fp32
// (v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bitwise an fmaf chain; peak 157.3 TFLOP/s).
//
// Covers every conv of pytorch_prototyping.py on the live path:
//   KIND 0  ReflectionPad2d(1) + Conv2d 3x3 s1     Conv2dSame (pytorch_prototyping.py:96-121), DownBlock prep conv (239-246)
//   KIND 1  ReflectionPad2d(1) + Conv2d 4x4 s2     DownBlock (258-264)
//   KIND 2  ConvTranspose2d 4x4 s2 p1              UpBlock (154-159), as four 2x2-tap parity classes
//
// GEMM view: rows = output pixels (n, oy, ox), columns = output channels, K = taps x input channels.
// Activations are channel-last, so a K-chunk of 16 channels of one tap is 64 contiguous bytes per pixel.
// What is fused here instead of being separate passes (the reference runs conv, BatchNorm, activation,
// ReflectionPad and torch.cat as separate kernels with full-tensor round trips):
//   * prologue  : value = act(scale[n,c] * raw + shift[n,c]) — the producer's train-mode BatchNorm (+ bias)
//                 and (Leaky)ReLU applied while the A operand is staged; reflection padding and the skip
//                 concat are pure index arithmetic on the gather;
//   * epilogue  : per-view per-channel sum / sum-of-squares of the raw output (wave shuffles -> LDS ->
//                 one fp64 atomic per channel per workgroup) for the next BatchNorm's batch statistics.
//
// Kernels: conv_halo_kernel (LDS-halo tiles, exact fp32: every map whose width is a multiple of 32, or 16), conv_mfma_kernel (tap-by-tap gather:
// the small maps), conv_halo_emu_kernel (opt-in: fp32 emulated on the 16-bit matrix cores, RNR_CONV_F32_EMU_BF16X6 / _F16X3), and — the
// product path since r03, RNR_CONV_WINOGRAD — the fp32 Winograd kernels of conv_wino.inc (F(2x2, 3x3): conv_wino_kernel),
// conv_wino80.inc (the 80-column out layer) and conv_wino2.inc (F(2x2, 2x2) for the two 4x4 stride-2 convolutions).
// make_plan() picks the kernel, the tile shape (256x64, 256x80, 128x128 or 256x128 rows x columns) and the split-K depth.
#include "rnr_internal.h"

#include <algorithm>
#include <type_traits>
#include <cstdlib>

namespace rnr {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int BK = 16;
constexpr int CTHREADS = 256;
#ifndef RNR_SPLITK_BELOW
#define RNR_SPLITK_BELOW 257        // one workgroup per CU (a single wave per SIMD) is split two ways: 146 vs 156 ... 266 vs 304 us on the
#define RNR_SPLITK_TARGET 512       // 256-tile layers at one view per call
#endif
#ifndef RNR_COMBINE_MAX_BYTES
#define RNR_COMBINE_MAX_BYTES (256 * 1024)      // four 128 x 128 images, sixteen 64 x 64 ones
#endif
#ifndef RNR_SMALL_TILE_BELOW
#define RNR_SMALL_TILE_BELOW 128     // fewer 128 x 128 tiles than this: 64 x 64 tiles (make_plan)
#endif
#ifndef RNR_CFG4_MAX
#define RNR_CFG4_MAX 512             // at most this many 128 x 128 tiles: 128 x 64 tiles instead (make_plan, 3x3 only); 0 = never
#endif
#ifndef RNR_CFG0_SMALL_MAX
#define RNR_CFG0_SMALL_MAX 1024      // at most this many 256 x 64 tiles (one 512^2 view): 128 x 64 tiles, four waves per SIMD, instead
#endif
#ifndef RNR_FUSED_BN_MIN_WGS
#define RNR_FUSED_BN_MIN_WGS 256     // in-kernel BatchNorm finalise for grids larger than this
#endif
#ifndef RNR_NATIVE_BIG_MIN
#define RNR_NATIVE_BIG_MIN 512     // measured: 512 >= 1024, 2048 at 8, 4, 2 views per launch, all equal at 1
#endif
#ifndef RNR_HALO_HDIST
#define RNR_HALO_HDIST 2   // conv_halo_kernel: taps between the fetch of a halo slice and its store to LDS (transposed conv: 1)
#endif
#ifndef RNR_HALO_WAVES
#define RNR_HALO_WAVES 3    // waves per SIMD the halo kernels are register-bounded for (4 would spill and exceed LDS anyway)
#endif

// waves per SIMD a halo-kernel configuration is register-bounded for: 128 x 64 tiles (32 accumulator registers) four, the
// 64-accumulator tiles three (RNR_HALO_WAVES), 256 x 128 and the 80-column remainder configuration two
constexpr int halo_waves(int WM, int WN, int R16) {
    return R16 ? 2 : (WM * WN <= 2 ? 4 : (WM * WN <= 4 ? RNR_HALO_WAVES : (WM * WN <= 8 ? 2 : 1)));
}

struct ConvParams {
    const float* src_data[2];
    const float* src_scale[2];
    const float* src_shift[2];
    int src_c[2];
    int src_act[2];
    const float* weight;
    const void* weight_emu;     // bf16x6 image of the same weights (conv_halo_emu_kernel), or NULL
    const float* weight_wino;   // Winograd-transformed image of the same weights (conv_wino_kernel), or NULL
    float* out;
    double* stats;
    int N, H, W;        // input spatial size
    int Ho, Wo;         // GEMM row space per view (conv: output size; convT: input size, per parity class)
    int OH, OW;         // true output size
    int M;              // GEMM rows (per parity class)
    int c_out, c_out_pad;
    int wstride;        // floats per packed weight row: c_out_pad rounded up to 128 (tiles never read past a row)
    int chunks0, chunks_per_tap, kt_total;
    int splitk;
    long slab_stride;   // floats between split-K slabs
    int mtiles, ntiles, zdim;   // logical grid; the launch is 1-D and remapped per XCD (see tile_coords)
    const uint8_t* tile_mask;   // optional [mtiles]: 0 = nobody reads this pixel tile's output, skip it (halo kernels)
    // ---- rnr_conv2d_fused: producer-side BatchNorm and in-launch split-K combine (all zero on the legacy entry points) ----
    long stats_shard;           // doubles between the statistics shards (a workgroup adds into shard blockIdx % n_shards)
    int n_shards;               // 1 (legacy) or STAT_SHARDS
    unsigned* arrive;           // arrival counters of the statistics: [N] (arrive_per_view) or [1]; NULL = no in-kernel finalise
    int arrive_per_view;
    unsigned n_arrive;          // arrivals that complete a counter
    const float* gamma; const float* beta; float* scale; float* shift;
    float eps; double count;    // pixels per view behind one statistic
    float* running_mean; float* running_var; float momentum;     // torch's train-mode side effect, one-view calls only (rnr_conv_bn)
    unsigned* tile_arrive;      // [par * mtiles * ntiles] split-K slices of a tile that have published their slab; NULL = legacy slabs + reduce kernel
    // ---- rnr_conv2d_ray: the U-Net-dependent half of the ray renderer in the out layer's epilogue (80-column configuration) ----
    const float* ray_w;         // [N*OH*OW][c_out_pad] ray weights (rnr_ray_weights); NULL = ordinary epilogue
    const float* ray_bias;      // [c_out_pad] out-layer bias
    float* ray_image;           // [N,3,OH,OW]
    int par_inner;              // tile order of the transposed conv: parity class inside the pixel tile (see tile_coords)
    float* slabs;               // in-launch combine: [splitk][tile][wave][i][j][4 quads][64 lanes][4] accumulator images
};
constexpr int STAT_SHARDS = 8;  // one per XCD: at one view per call every workgroup of a layer hits the same 2 * c_out words

// XCD-aware tile order.  Workgroup b is dispatched to XCD b % 8 (observed; used for speed only), and each XCD has a
// private 4 MiB L2.  Give every XCD a contiguous run of logical tiles so that vertically adjacent pixel tiles (which
// share their 3x3 halo rows) and the column tiles of one pixel tile (which share the whole A operand) meet in the same
// L2.  Bijective for any tile count (cdna_hip_programming.md, "XCD swizzle must be bijective").
__device__ __forceinline__ void tile_coords(const ConvParams& P, int& mt, int& nt, int& z) {
    const int total = P.mtiles * P.ntiles * P.zdim;
    const int b = blockIdx.x;
    const int q = total >> 3, r = total & 7;
    const int xcd = b & 7, idx = b >> 3;
    const int id = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    nt = id % P.ntiles;
    const int t = id / P.ntiles;
    if (P.par_inner) {      // transposed conv: the four output-parity classes of a pixel tile are neighbours (they stage the SAME input halo)
        const int t2 = t >> 2;
        mt = t2 % P.mtiles;
        z = (t2 / P.mtiles) * 4 + (t & 3);
    } else {
        mt = t % P.mtiles;
        z = t / P.mtiles;
    }
}

__device__ __forceinline__ int reflect1(int i, int n) {   // ReflectionPad2d(1)
    i = i < 0 ? -i : i;
    return i >= n ? 2 * n - 2 - i : i;
}

// act(v) = max(v, slope * v) with slope = 1 (none), 0.2 (LeakyReLU 0.2), 0 (ReLU): branch-free, wave-uniform slope
__device__ __forceinline__ float act_slope(int act) {
    return act == RNR_ACT_LRELU02 ? 0.2f : (act == RNR_ACT_RELU ? 0.0f : 1.0f);
}
__device__ __forceinline__ float apply_act(float v, int act) {
    return fmaxf(v, act_slope(act) * v);
}


// ------------------------------------------------------------------------------------------------
// Producer-side BatchNorm and in-launch split-K combine (rnr_conv2d_fused).
//
// One view per call (the reference's mode, test_rnr.py:265) makes a convolution a 40 - 350 us kernel; the 17
// bn_finalize launches and 9 split-K reduce launches behind them cost 6 % of the U-Net there.  Both are folded
// into the convolution's epilogue with arrival counters:
//   * batch statistics are added into one of STAT_SHARDS copies (workgroup b -> shard b % 8, i.e. its XCD), the
//     workgroup waits for the acknowledgement of its atomics, and draws a ticket; the workgroup that draws the
//     last ticket of a view sums the shards, writes scale / shift and leaves statistics and counter at zero.
//     Statistics and counters are touched by agent-scope atomics only (add / load / store meet at the memory
//     side, MI355X_MICROARCH.md "8-B agent atomics both sides"), so no fence is involved; scale / shift are plain
//     stores read by the next kernel of the stream;
//   * a split-K slice stores its accumulators as a write-through (sc1) register image — 16 B per lane, the order
//     the registers have —, drains, and draws a ticket of its tile; the last slice adds all images IN SLICE ORDER
//     (its own included: the sum does not depend on who arrives last) with sc1 loads and runs the normal epilogue.
// Counters and statistics are zero before and after every call; the host zeroes them once.
// ------------------------------------------------------------------------------------------------
// REQUIRED ASSUMPTIONS of the fence-free hand-off (checked below at compile time as far as they can be; tested by
// tests/test_gpu_unet.py::test_conv_fused_equals_separate_launches and scripts/t_fused_stress.py on the hardware):
//   (1) gfx950 acknowledges a write-through (sc1) buffer store and a no-return agent-scope atomic — i.e. decrements vmcnt —
//       only once it has reached the coherence point (the memory-side L2 / MALL path all XCDs share), so "s_waitcnt
//       vmcnt(0); s_barrier; ticket" orders every lane's publication before the ticket without a release fence;
//   (2) bit 4 (value 16) of the aux operand of the raw-buffer intrinsics is sc1 on this target: loads bypass the
//       non-coherent caches, stores write through, which stands in for the acquire side;
//   (3) the compiler keeps the buffer intrinsics on their side of the `asm volatile(... "memory")` + __syncthreads pair.
// None of this is portable to another target or to a partition mode in which the XCDs do not share a coherence point; the
// file therefore refuses to compile device code for anything but gfx950.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "conv.hip: the producer-side BatchNorm / split-K hand-off relies on gfx950 memory semantics (see above); gfx950 only"
#endif
#define RNR_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT
typedef unsigned int uintx4_t __attribute__((__vector_size__(4 * sizeof(unsigned int))));
constexpr int AUX_SC1 = 16;         // write-through / L1-bypassing buffer access

__device__ __forceinline__ double* stat_slot(const ConvParams& P, int n, int col) {
    return P.stats + (size_t)(blockIdx.x & (unsigned)(P.n_shards - 1)) * P.stats_shard + ((size_t)n * P.c_out_pad + col) * 2;
}

// scale / shift of views [n_first, n_first + n_views) from the summed shards (the arithmetic of bn_finalize_kernel).
// The (sum, sum of squares) pair of a channel is one 16-byte sc1 load per shard — L1-bypassing like the 8-byte agent-scope
// atomic load, but all STAT_SHARDS of them are in flight at once (a chain of __hip_atomic_load is issued one at a time:
// 16 round trips, 20 us at the end of a 150 us kernel) — and one 16-byte sc1 store of zeros.
__device__ __forceinline__ void bn_finalize_views(const ConvParams& P, int n_first, int n_views, int tid, int i_first = 0,
                                                  int i_stride = 0) {
    const int total = n_views * P.c_out_pad;
    const int stride = i_stride ? i_stride : (int)blockDim.x;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(P.stats, 0, 0x7fffffff, 0x27000);
    const uintx4_t zero4 = {0u, 0u, 0u, 0u};
    typedef double doublex2 __attribute__((ext_vector_type(2)));
    for (int i = i_first + tid; i < total; i += stride) {
        const int idx = n_first * P.c_out_pad + i;
        const int c = i % P.c_out_pad;
        doublex2 part[STAT_SHARDS];
#pragma unroll
        for (int sh = 0; sh < STAT_SHARDS; sh++) {
            part[sh] = doublex2{0.0, 0.0};
            if (sh < P.n_shards)
                part[sh] = __builtin_bit_cast(doublex2, __builtin_amdgcn_raw_buffer_load_b128(
                                                            rsrc, idx * 16, (int)(sh * P.stats_shard * 8), AUX_SC1));
        }
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sh = 0; sh < STAT_SHARDS; sh++) {
            s1 += part[sh][0];
            s2 += part[sh][1];
            if (sh < P.n_shards)
                __builtin_amdgcn_raw_buffer_store_b128(zero4, rsrc, idx * 16, (int)(sh * P.stats_shard * 8), AUX_SC1);
        }
        float sc = 0.f, sf = 0.f;
        if (c < P.c_out) {
            const double mean = s1 / P.count;
            double var = s2 / P.count - mean * mean;    // biased variance
            var = var < 0.0 ? 0.0 : var;
            const double g = (double)P.gamma[c] / sqrt(var + (double)P.eps);
            sc = (float)g;
            sf = (float)((double)P.beta[c] - mean * g);
            // torch.nn.BatchNorm2d in train mode also moves its running statistics (momentum, unbiased variance): the arithmetic
            // of bn_finalize_batch_kernel; the host passes the buffers for one-view calls only, where per-view = whole batch
            if (P.running_mean) P.running_mean[c] = (float)((1.0 - (double)P.momentum) * (double)P.running_mean[c] + (double)P.momentum * mean);
            if (P.running_var) {
                const double unb = P.count > 1.0 ? var * P.count / (P.count - 1.0) : var;
                P.running_var[c] = (float)((1.0 - (double)P.momentum) * (double)P.running_var[c] + (double)P.momentum * unb);
            }
        }
        P.scale[idx] = sc;
        P.shift[idx] = sf;
    }
}

// Every thread has issued what it publishes (statistics atomics, slab stores).  Returns the ticket in thread 0.
__device__ __forceinline__ unsigned publish_and_draw(unsigned* counter, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // acknowledged by the memory side
    __syncthreads();
    unsigned t = 0;
    if (tid == 0) t = __hip_atomic_fetch_add(counter, 1u, RNR_RLX_AGENT);
    return t;
}
// thread 0 holds the ticket; true in every thread of the workgroup that drew the last one (which resets the counter)
__device__ __forceinline__ bool drew_last(unsigned* counter, unsigned ticket, unsigned expected, int tid, int* flag) {
    if (tid == 0) {
        const int last = ticket == expected - 1u ? 1 : 0;
        *flag = last;
        if (last) __hip_atomic_store(counter, 0u, RNR_RLX_AGENT);
    }
    __syncthreads();
    return *flag != 0;
}

// statistics published -> ticket -> (caller's stores overlap the round trip) -> finalise.  n: the view of this workgroup
// (halo kernels) or -1 (one counter for the whole launch).
struct BnArrival { unsigned* counter; unsigned ticket; };
__device__ __forceinline__ BnArrival bn_arrive(const ConvParams& P, int n, int tid) {
    BnArrival a;
    a.counter = P.arrive + (P.arrive_per_view && n >= 0 ? n : 0);
    a.ticket = publish_and_draw(a.counter, tid);
    return a;
}
__device__ __forceinline__ void bn_complete(const ConvParams& P, const BnArrival& a, int n, int tid, int* flag) {
    if (drew_last(a.counter, a.ticket, P.n_arrive, tid, flag)) {
        if (P.arrive_per_view && n >= 0) bn_finalize_views(P, n, 1, tid);
        else bn_finalize_views(P, 0, P.N, tid);
    }
}

// accumulator images of a wave: [i][j][quad q][lane][4 floats]
template <int WM, int WN>
__device__ __forceinline__ void slab_store(float* base, const floatx16 (&acc)[WM][WN], int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x27000);
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const floatx4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uintx4_t, v), rsrc, lane * 16,
                                                       ((i * WN + j) * 4 + q) * 1024, AUX_SC1);
            }
}
template <int WM, int WN>
__device__ __forceinline__ void slab_add(const float* base, floatx16 (&acc)[WM][WN], int lane) {
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 0x7fffffff, 0x27000);
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const floatx4 v = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                  rsrc, lane * 16, ((i * WN + j) * 4 + q) * 1024, AUX_SC1));
                acc[i][j][4 * q] += v[0]; acc[i][j][4 * q + 1] += v[1]; acc[i][j][4 * q + 2] += v[2]; acc[i][j][4 * q + 3] += v[3];
            }
}
// The split-K slices of a tile meet here.  Returns false in the slices that are done (their slab is published), true
// in the last one, whose accumulators then hold the sum over all slices.
template <int WM, int WN>
__device__ __forceinline__ bool splitk_combine(const ConvParams& P, floatx16 (&acc)[WM][WN], int tile, int split, int wave,
                                               int lane, int tid, int* flag) {
    constexpr size_t WAVE_FLOATS = (size_t)WM * WN * 1024, TILE_FLOATS = 4 * WAVE_FLOATS;
    const size_t ntiles_all = (size_t)P.mtiles * P.ntiles * (P.zdim / P.splitk);
    float* mine = P.slabs + ((size_t)split * ntiles_all + tile) * TILE_FLOATS + wave * WAVE_FLOATS;
    slab_store<WM, WN>(mine, acc, lane);
    const unsigned t = publish_and_draw(P.tile_arrive + tile, tid);
    if (!drew_last(P.tile_arrive + tile, t, (unsigned)P.splitk, tid, flag)) return false;
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;
    for (int s = 0; s < P.splitk; s++)
        slab_add<WM, WN>(P.slabs + ((size_t)s * ntiles_all + tile) * TILE_FLOATS + wave * WAVE_FLOATS, acc, lane);
    return true;
}

template <int KIND, int WAVES_M, int WAVES_N, int WM, int WN>
__global__ void __launch_bounds__(CTHREADS)
conv_mfma_kernel(const ConvParams P) {
    constexpr int BM = WAVES_M * WM * 32, BN = WAVES_N * WN * 32;
    // LDS images as in conv_halo_kernel: [4 planes g][rows | columns][4 floats e], channel k -> g = 2*(k&1) + (k>>3),
    // e = (k>>1)&3: an MFMA lane reads its 8 channels of a row / column as two lane-consecutive float4s
    constexpr int ACH = 16 * BM, BCH = 16 * BN;
    constexpr int RPT = BM / 64;
    constexpr int BQ = 4 * BN;                            // float4 slots of a B tile: (plane, column)
    constexpr int BPT = (BQ + CTHREADS - 1) / CTHREADS;
    constexpr int TAPS = KIND == 0 ? 9 : (KIND == 1 ? 16 : 4);
    static_assert(WAVES_M * WAVES_N == 4, "four waves per workgroup");

    __shared__ __attribute__((aligned(16))) float As[2][ACH];
    __shared__ __attribute__((aligned(16))) float Bs[2][BCH];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int wm0 = wave_m * WM * 32, wn0 = wave_n * WN * 32;
    int mt_, nt_, z_;
    tile_coords(P, mt_, nt_, z_);
    const int m0 = mt_ * BM, n0 = nt_ * BN;
    const int par = (KIND == 2) ? (z_ & 3) : 0;
    const int split = (KIND == 2) ? (z_ >> 2) : z_;
    const int py = par >> 1, px = par & 1;
    const int hw_rows = P.Ho * P.Wo;

    // ---- the rows this thread stages (fixed for the whole K loop) ----
    const int q = tid & 3;
    int rn[RPT], roy[RPT], rox[RPT];
    bool rvalid[RPT];
#pragma unroll
    for (int p = 0; p < RPT; p++) {
        const int m = m0 + (tid >> 2) + 64 * p;
        rvalid[p] = m < P.M;
        const int mm = rvalid[p] ? m : 0;
        rn[p] = mm / hw_rows;
        const int rem = mm - rn[p] * hw_rows;
        roy[p] = rem / P.Wo;
        rox[p] = rem - roy[p] * P.Wo;
    }
    const int m_last = min(m0 + BM, P.M) - 1;
    const bool single_view = (m0 / hw_rows) == (m_last / hw_rows);
    const int n_tile = m0 / hw_rows;

    const int per_split = (P.kt_total + P.splitk - 1) / P.splitk;
    const int kt0 = split * per_split;
    const int kt1 = min(P.kt_total, kt0 + per_split);

    float4 areg[RPT];
    float4 breg[BPT];

    auto load_regs = [&](int kt) {
        const int tp = kt / P.chunks_per_tap;
        const int ch = kt - tp * P.chunks_per_tap;
        const int s = ch < P.chunks0 ? 0 : 1;
        const int cc = (ch - (s ? P.chunks0 : 0)) * BK + 4 * q;
        const int C = P.src_c[s];
        const float* data = P.src_data[s];
        const float* scp = P.src_scale[s];
        const float* shp = P.src_shift[s];
        const int act = P.src_act[s];
        float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (single_view) {
            if (scp) sc = *reinterpret_cast<const float4*>(scp + (size_t)n_tile * C + cc);
            if (shp) sh = *reinterpret_cast<const float4*>(shp + (size_t)n_tile * C + cc);
        }
#pragma unroll
        for (int p = 0; p < RPT; p++) {
            int iy, ix;
            bool ok = rvalid[p];
            if (KIND == 0) {
                iy = reflect1(roy[p] + tp / 3 - 1, P.H);
                ix = reflect1(rox[p] + tp % 3 - 1, P.W);
            } else if (KIND == 1) {
                iy = reflect1(2 * roy[p] + (tp >> 2) - 1, P.H);
                ix = reflect1(2 * rox[p] + (tp & 3) - 1, P.W);
            } else {
                // out(2a+py) = sum over ky with iy = (oy + 1 - ky)/2: py=0 -> (ky=1, iy=a), (ky=3, iy=a-1);
                //                                                     py=1 -> (ky=0, iy=a+1), (ky=2, iy=a)
                const int ty = tp >> 1, tx = tp & 1;
                iy = roy[p] + (py == 0 ? (ty == 0 ? 0 : -1) : (ty == 0 ? 1 : 0));
                ix = rox[p] + (px == 0 ? (tx == 0 ? 0 : -1) : (tx == 0 ? 1 : 0));
                ok = ok && iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            }
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const size_t pidx = ((size_t)rn[p] * P.H + iy) * P.W + ix;
                v = *reinterpret_cast<const float4*>(data + pidx * C + cc);
                if (!single_view) {
                    if (scp) sc = *reinterpret_cast<const float4*>(scp + (size_t)rn[p] * C + cc);
                    if (shp) sh = *reinterpret_cast<const float4*>(shp + (size_t)rn[p] * C + cc);
                }
                v.x = apply_act(v.x * sc.x + sh.x, act);
                v.y = apply_act(v.y * sc.y + sh.y, act);
                v.z = apply_act(v.z * sc.z + sh.z, act);
                v.w = apply_act(v.w * sc.w + sh.w, act);
            }
            areg[p] = v;
        }
        // packed weights are plane images per chunk, [4 g][wstride][4 e]: a (plane, column) float4 is copied as it is
        const float* wchunk = P.weight + ((size_t)(par * TAPS + tp) * P.chunks_per_tap + ch) * (16 * (size_t)P.wstride);
#pragma unroll
        for (int b = 0; b < BPT; b++) {
            const int idx = tid + CTHREADS * b;
            float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < BQ) {
                const int g = idx / BN, c = idx - g * BN;
                if (n0 + c < P.c_out_pad) w = *reinterpret_cast<const float4*>(wchunk + ((size_t)g * P.wstride + n0 + c) * 4);
            }
            breg[b] = w;
        }
    };

    auto store_lds = [&](int buf) {
#pragma unroll
        for (int p = 0; p < RPT; p++) {
            const int r = (tid >> 2) + 64 * p;
            float* a = &As[buf][((q >> 1) * BM + r) * 4 + 2 * (q & 1)];
            *reinterpret_cast<float2*>(a) = make_float2(areg[p].x, areg[p].z);                 // channels 4q, 4q+2
            *reinterpret_cast<float2*>(a + 2 * BM * 4) = make_float2(areg[p].y, areg[p].w);    // channels 4q+1, 4q+3
        }
#pragma unroll
        for (int b = 0; b < BPT; b++) {
            const int idx = tid + CTHREADS * b;
            if (idx < BQ) *reinterpret_cast<float4*>(&Bs[buf][idx * 4]) = breg[b];
        }
    };

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    if (kt0 < kt1) {
        load_regs(kt0);
        store_lds(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; kt++) {
        const int buf = (kt - kt0) & 1;
        const bool more = kt + 1 < kt1;
        if (more) load_regs(kt + 1);
        const float* a_s = &As[buf][((2 * h) * BM + wm0 + l31) * 4];
        const float* b_s = &Bs[buf][((2 * h) * BN + wn0 + l31) * 4];
#pragma unroll
        for (int sg = 0; sg < 2; sg++) {
            floatx4 a4[WM], b4[WN];
#pragma unroll
            for (int i = 0; i < WM; i++) a4[i] = *reinterpret_cast<const floatx4*>(a_s + (sg * BM + 32 * i) * 4);
#pragma unroll
            for (int j = 0; j < WN; j++) b4[j] = *reinterpret_cast<const floatx4*>(b_s + (sg * BN + 32 * j) * 4);
#pragma unroll
            for (int e = 0; e < 4; e++)
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int j = 0; j < WN; j++)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][e], b4[j][e], acc[i][j], 0, 0, 0);
        }
        if (more) store_lds(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: raw output (or split-K slab) ----
    float* out = P.out + (size_t)split * P.slab_stride;
#pragma unroll
    for (int i = 0; i < WM; i++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int row = wm0 + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * h;
            const int m = m0 + row;
            if (m < P.M) {
                size_t off;
                if (KIND == 2) {
                    const int n = m / hw_rows;
                    const int rem = m - n * hw_rows;
                    const int a = rem / P.Wo, b = rem - a * P.Wo;
                    off = (((size_t)n * P.OH + 2 * a + py) * P.OW + 2 * b + px) * P.c_out_pad;
                } else {
                    off = (size_t)m * P.c_out_pad;
                }
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int col = n0 + wn0 + 32 * j + l31;
                    if (col < P.c_out_pad) out[off + col] = acc[i][j][g];
                }
            }
        }
    }

    // ---- epilogue: batch statistics of the raw output ----
    if (P.stats && P.splitk == 1) {
        if (single_view) {
            float* red = &As[0][0];   // [WAVES_M][BN][2], LDS is free after the last barrier
#pragma unroll
            for (int j = 0; j < WN; j++) {
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int i = 0; i < WM; i++)
#pragma unroll
                    for (int g = 0; g < 16; g++) {
                        const float v = acc[i][j][g];
                        s1 += v;
                        s2 += v * v;
                    }
                s1 += __shfl_xor(s1, 32, 64);
                s2 += __shfl_xor(s2, 32, 64);
                if (h == 0) {
                    const int col = wn0 + 32 * j + l31;
                    red[(wave_m * BN + col) * 2 + 0] = s1;
                    red[(wave_m * BN + col) * 2 + 1] = s2;
                }
            }
            __syncthreads();
            if (tid < BN) {
                const int col = n0 + tid;
                if (col < P.c_out) {
                    double s1 = 0.0, s2 = 0.0;
#pragma unroll
                    for (int w = 0; w < WAVES_M; w++) {
                        s1 += (double)red[(w * BN + tid) * 2 + 0];
                        s2 += (double)red[(w * BN + tid) * 2 + 1];
                    }
                    double* st = stat_slot(P, n_tile, col);
                    atomicAdd(st + 0, s1);
                    atomicAdd(st + 1, s2);
                }
            }
        } else {   // tiles straddling views only occur for maps smaller than a tile (tiny layers)
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const int m = m0 + wm0 + 32 * i + (g & 3) + 8 * (g >> 2) + 4 * h;
                    if (m < P.M) {
                        const int n = m / hw_rows;
#pragma unroll
                        for (int j = 0; j < WN; j++) {
                            const int col = n0 + wn0 + 32 * j + l31;
                            if (col < P.c_out) {
                                const float v = acc[i][j][g];
                                double* st = stat_slot(P, n, col);
                                atomicAdd(st + 0, (double)v);
                                atomicAdd(st + 1, (double)v * (double)v);
                            }
                        }
                    }
                }
        }
        if (P.arrive) {      // producer-side BatchNorm: one counter for the launch (tiles may straddle views here)
            const BnArrival a = bn_arrive(P, -1, tid);
            bn_complete(P, a, -1, tid, reinterpret_cast<int*>(&Bs[0][0]));
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-halo kernels (maps whose output width is a multiple of 32 pixels, or 16 pixels wide).
//
// A workgroup owns a 32 x TH tile of output pixels (TH = 8 or 4).  For every 16-channel chunk it stages the input
// HALO of that tile once in LDS — reflection padding (or the zero border of the transposed conv), the producer's
// BatchNorm/bias/activation and the skip concat are applied on the way in — and runs ALL taps of the chunk from it.
// For a tap the MFMA A operand of output row y, lane x is one halo element at a lane-consecutive LDS address:
//   KIND 0  3x3 s1          halo (TH+2) x 34,   element (y+ky, x+kx)                       9 taps
//   KIND 1  4x4 s2          per input parity phase: halo (TH+1) x 33 of that phase, element (y+a, x+b)   4 x 4 taps
//   KIND 2  convT 4x4 s2    halo (TH+2) x 34 of the INPUT, zero outside; one output parity class per workgroup,
//                           element (y+oy(ty), x+ox(tx))                                   4 taps
// Versus the tap-by-tap gather of conv_mfma_kernel the L2->LDS traffic, the global-load / ds_write instruction count
// and the prologue math of the A operand drop by taps*256/halo = 6.8x / 3.1x / 3.0x; the MFMA work is identical.
// Weights do not pass through LDS: a tap's B operand is 2*WN coalesced buffer loads per wave straight into the MFMA
// operand registers, requested one tap ahead.  A slice of the next chunk's halo is fetched before a tap's MFMAs and
// stored to the alternate LDS buffer after them; ONE barrier per 16-channel chunk (round 1 staged weight tiles by
// LDS-DMA and needed a barrier per tap).
// ------------------------------------------------------------------------------------------------
// Epilogue of the halo kernels: the wave's accumulator tiles go to the NHWC output as buffer stores.  The address of
// element (row block i, register g, column block j) splits into a workgroup-uniform 64-bit base (the resource), a
// wave-uniform 32-bit offset per (i, g) (SGPR) and ONE per-lane 32-bit offset per column block, computed once: a store
// is one instruction (no 64-bit VALU address arithmetic, no branch); lanes of padding columns beyond c_out_pad get an
// out-of-range offset and the hardware drops their store.
// Accumulator layout of v_mfma_f32_32x32x*: lane (l31, h), register g holds row (g & 3) + 8 (g >> 2) + 4 h, column l31.
template <int KIND, int WM, int WN, int TW, bool SCALED = false>
__device__ __forceinline__ void store_acc_tiles(const ConvParams& P, float* out, const floatx16 (&acc)[WM][WN], int n, int y0,
                                                int x0, int py, int px, int wave_m, int n0, int wn0, int l31, int h,
                                                float scale = 1.0f) {
    constexpr int RPB = 32 / TW, XM = KIND == 2 ? 2 : 1;       // transposed conv: this parity class writes every other pixel
    const int wm = __builtin_amdgcn_readfirstlane(wave_m);
    const int Y00 = XM * y0 + (KIND == 2 ? py : 0), X00 = XM * x0 + (KIND == 2 ? px : 0);
    float* base = out + (((size_t)n * P.OH + Y00) * P.OW + X00) * P.c_out_pad + n0;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7fffffff, 0x27000);
    const unsigned cp4 = (unsigned)P.c_out_pad * 4u;
    unsigned voff[WN];
#pragma unroll
    for (int j = 0; j < WN; j++) {
        const int colw = wn0 + 32 * j + l31;
        voff[j] = (n0 + colw < P.c_out_pad) ? (unsigned)(XM * 4 * h) * cp4 + (unsigned)colw * 4u : 0x7fffffffu;
    }
#pragma unroll
    for (int i = 0; i < WM; i++) {
#pragma unroll
        for (int g = 0; g < 16; g++) {
            const int pb0 = (g & 3) + 8 * (g >> 2);             // + 4 h: lane part (never crosses an image row of the tile)
            const int yrel = (wm * WM + i) * RPB + pb0 / TW, xrel = pb0 % TW;
            const unsigned soff = (unsigned)(XM * (yrel * P.OW + xrel)) * cp4;
#pragma unroll
            for (int j = 0; j < WN; j++) {
                // (bit_cast straight from the vector element stores element 0: compiler bug — go through a scalar)
                const float v = SCALED ? acc[i][j][g] * scale : acc[i][j][g];
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsrc, (int)voff[j], (int)soff, 0);
            }
        }
    }
}

template <int KIND, int WAVES_M, int WAVES_N, int WM, int WN, int R16, int TW>
__global__ void __launch_bounds__(CTHREADS, halo_waves(WM, WN, R16))
conv_halo_kernel(const ConvParams P) {
    // TW = 32: a 32-row MFMA block is one image row of the tile; TW = 16 (maps 16 pixels wide): two image rows
    static_assert(TW == 32 || (TW == 16 && !R16), "tile width");
    constexpr int RPB = 32 / TW;                           // image rows per MFMA row block
    constexpr int TH = WAVES_M * WM * RPB;
    // R16 = 1 adds a 16-column remainder tile per wave on v_mfma_f32_16x16x4_f32 (same FLOP rate): Cout = 78 runs as
    // 64 + 16 = 80 columns instead of 96.
    constexpr int WCOLS = WN * 32 + R16 * 16;
    constexpr int BN = WAVES_N * WCOLS;
    // KIND 1 (4x4 stride 2) runs as FOUR stride-1 2x2-tap convolutions, one per input parity phase (py, px): input row
    // 2y + ky - 1 = 2(y + ty) + py with (ky; ty, py) = (0; -1, 1), (1; 0, 0), (2; 0, 1), (3; 1, 0).  A K step is a
    // (16-channel chunk, phase) pair whose halo is the (TH+1) x (TW+1) pixels of that phase only (297 pixels for a 32 x 8
    // tile; the interleaved halo of all 16 taps takes 1188 and fitted LDS only single-buffered with 4-row tiles, r01).
    constexpr int NPH = KIND == 1 ? 4 : 1;                 // K steps per 16-channel chunk
    constexpr int TAPS = KIND == 0 ? 9 : 4;                // taps per K step
    constexpr int HWD = KIND == 1 ? TW + 1 : TW + 2;       // halo width
    constexpr int HHT = KIND == 1 ? TH + 1 : TH + 2;       // halo height
    constexpr int HP = HWD * HHT;
    constexpr int ASLOTS = HP * 4;                         // float4 slots of one halo chunk (pixel x channel quad)
    constexpr int APT = (ASLOTS + CTHREADS - 1) / CTHREADS;
    constexpr int APS = (APT + TAPS - 1) / TAPS;           // halo float4 fetched per pipeline step
    // two taps of MFMAs between a halo slice's fetch and its store cover the HBM latency (+0.4 %); the transposed conv at
    // three waves per SIMD has no registers for the second slice (it would spill to scratch)
    constexpr int HDIST = KIND == 2 ? 1 : RNR_HALO_HDIST;
    // LDS image of one 16-channel chunk, for the halo (X = HP pixels) and for the weight tile (X = BN columns):
    //   [4 planes g][X][4 floats e],  channel k of the chunk -> g = 2*(k&1) + (k>>3), e = (k>>1)&3.
    // An MFMA lane (x, h) needs channels k = 2s+h, s = 0..7, of ONE pixel / column: that is planes 2h and 2h+1 at
    // lane-consecutive float4s -> two conflict-free ds_read_b128 per operand row per tap, every address an immediate
    // offset from one per-lane base (the old [k][X] image took 16 ds_read_b32 and a VALU add per pair).
    constexpr int ACH = 16 * HP;                           // floats per halo chunk image
    static_assert(WAVES_M * WAVES_N == 4, "four waves per workgroup");
    static_assert(BK == 16, "plane mapping assumes 16-channel chunks");

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                       // [2][ACH]: the halo is double-buffered in LDS

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int wn0 = wave_n * WCOLS;
    const int l15 = lane & 15, kq = lane >> 4;
    int mt_, nt_, z_;
    tile_coords(P, mt_, nt_, z_);
    const int par = (KIND == 2) ? (z_ & 3) : 0;
    const int split = (KIND == 2) ? (z_ >> 2) : z_;
    const int py = par >> 1, px = par & 1;
    const int n0 = nt_ * BN;
    // tile grid lives in the GEMM row space: output pixels (KIND 0/1) or input pixels of a parity class (KIND 2)
    const int tiles_x = P.Wo / TW, tiles_y = P.Ho / TH;
    const int n = mt_ / (tiles_x * tiles_y);
    const int trem = mt_ - n * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;
    if (P.tile_mask && P.tile_mask[mt_] == 0) {             // workgroup-uniform, before any barrier
        if (R16 && KIND == 0 && P.ray_w) {                  // nobody computes this tile: its pixels are background, the frame is 0 there
            const int hw = P.OH * P.OW;
            for (int it = tid; it < 3 * TH * TW; it += CTHREADS) {
                const int c = it / (TH * TW), pl = it - c * (TH * TW);
                P.ray_image[((size_t)n * 3 + c) * hw + (size_t)(y0 + pl / TW) * P.OW + x0 + pl % TW] = 0.0f;
            }
        }
        return;
    }

    // halo slots of this thread: fixed source pixels for the whole K loop.  Slots past the halo fetch a valid address
    // (an earlier slot's) and are never stored; the zero border of the transposed conv is a 0/1 factor.
    const int q = tid & 3;
    unsigned spix[KIND == 1 ? 1 : APT];      // pixel index inside the view (KIND 1: recomputed per phase from siy / six)
    short siy[KIND == 1 ? APT : 1], six[KIND == 1 ? APT : 1];      // KIND 1: 2 (y0 + hy), 2 (x0 + hx)
    int sdst[APT];           // float index of the (x, z) pair inside a chunk image; the (y, w) pair is 2 planes on
    float smask[KIND == 2 ? APT : 1];
#pragma unroll
    for (int j = 0; j < APT; j++) {
        int s = tid + CTHREADS * j;
        if (s >= ASLOTS) s -= ASLOTS;
        const int hp = s >> 2;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        int iy = 0, ix = 0;
        const int col = hx;
        if (KIND == 0) { iy = reflect1(y0 - 1 + hy, P.H); ix = reflect1(x0 - 1 + hx, P.W); }
        else if (KIND == 1) { siy[j] = (short)(2 * (y0 + hy)); six[j] = (short)(2 * (x0 + hx)); }
        else {
            iy = y0 - 1 + hy; ix = x0 - 1 + hx;
            const bool inside = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            smask[j] = inside ? 1.f : 0.f;                          // exactly 0 outside, not act(shift)
            iy = min(max(iy, 0), P.H - 1); ix = min(max(ix, 0), P.W - 1);
        }
        sdst[j] = ((q >> 1) * HP + hy * HWD + col) * 4 + 2 * (q & 1);
        if (KIND != 1) spix[j] = (unsigned)(iy * P.W + ix);
    }

    const int nchunks = P.chunks_per_tap;
    const int per_split = (nchunks + P.splitk - 1) / P.splitk;
    const int c_begin = split * per_split;
    const int c_end = min(nchunks, c_begin + per_split);

    // view base + channel offset are wave-uniform (SGPR pair); the per-lane part is a 32-bit element offset
    // (make_plan keeps H*W*C below 2^30), so a halo fetch is one global_load_dwordx4 v, voff, s[base] and one VALU mad.
    // (buffer loads: resource + scalar offset + one 32-bit lane offset; flat 64-bit addresses make the unrolled tap loop
    // keep a strength-reduced pointer pair per (tap, plane) alive across the chunk loop — registers the kernel lacks)
    // K steps: step = chunk * NPH + phase
    struct ChunkSrc { __amdgpu_buffer_rsrc_t rsrc; unsigned C; unsigned soff; int act; int phy, phx; float4 sc, sh; };
    auto chunk_src = [&](int step) {
        ChunkSrc cs;
        const int c = step / NPH;
        cs.phy = (step % NPH) >> 1; cs.phx = (step % NPH) & 1;
        const int s = c < P.chunks0 ? 0 : 1;
        const int cc = (c - (s ? P.chunks0 : 0)) * BK;
        cs.C = (unsigned)P.src_c[s];
        cs.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.src_data[s] + (size_t)n * P.H * P.W * cs.C), 0,
                                                    0x7fffffff, 0x27000);
        cs.soff = (unsigned)cc * 4u;
        cs.act = P.src_act[s];
        cs.sc = make_float4(1.f, 1.f, 1.f, 1.f);
        cs.sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.src_scale[s]) cs.sc = *reinterpret_cast<const float4*>(P.src_scale[s] + (size_t)n * cs.C + cc + 4 * q);
        if (P.src_shift[s]) cs.sh = *reinterpret_cast<const float4*>(P.src_shift[s] + (size_t)n * cs.C + cc + 4 * q);
        return cs;
    };
    auto load_a = [&](const ChunkSrc& cs, int j) {
        unsigned pixel;
        if (KIND == 1) pixel = (unsigned)(reflect1(siy[j] - cs.phy, P.H) * P.W + reflect1(six[j] - cs.phx, P.W));
        else pixel = spix[j];
        const unsigned voff = (pixel * cs.C + 4u * (unsigned)q) * 4u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cs.rsrc, (int)voff, (int)cs.soff, 0));
    };
    auto store_a = [&](const ChunkSrc& cs, float4 v, int j, int buf) {
        float x = apply_act(v.x * cs.sc.x + cs.sh.x, cs.act);
        float y = apply_act(v.y * cs.sc.y + cs.sh.y, cs.act);
        float z = apply_act(v.z * cs.sc.z + cs.sh.z, cs.act);
        float w = apply_act(v.w * cs.sc.w + cs.sh.w, cs.act);
        if (KIND == 2) { x *= smask[j]; y *= smask[j]; z *= smask[j]; w *= smask[j]; }
        float* a = As + buf * ACH + sdst[j];
        *reinterpret_cast<float2*>(a) = make_float2(x, z);                  // channels 4q, 4q+2   (h = 0 planes)
        *reinterpret_cast<float2*>(a + 2 * HP * 4) = make_float2(y, w);     // channels 4q+1, 4q+3 (h = 1 planes)
    };
    // Weights of one (chunk, tap): the packed layout (pack_weight_kernel) is the plane image [4 planes][wstride][4 floats]
    // per chunk, and an MFMA lane (column, k parity h) needs exactly the float4s of planes 2h and 2h+1 at its column:
    // lanes of a wave are consecutive columns, so a tap's B operand is 2*WN coalesced buffer_load_dwordx4 per wave
    // straight into the operand registers, requested one tap ahead.  No LDS weight tile, no LDS-DMA, and therefore no
    // barrier per tap: the only LDS hazard left is the halo swap (one barrier per chunk).
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.weight), 0, 0x7fffffff, 0x27000);
    const unsigned tile_bytes = 64u * (unsigned)P.wstride;                  // one (chunk, tap) block: 16 channels x wstride floats
    unsigned bvoff[2];
#pragma unroll
    for (int sg = 0; sg < 2; sg++)
        bvoff[sg] = ((unsigned)(2 * h + sg) * (unsigned)P.wstride + (unsigned)(n0 + wn0 + l31)) * 16u;
    const int g16 = (kq & 1) * 2 + (kq >> 1);
    const unsigned bvoff16 = ((unsigned)g16 * (unsigned)P.wstride + (unsigned)(n0 + wn0 + WN * 32 + l15)) * 16u;
    struct BRegs { floatx4 b[2][WN]; floatx4 b16; };
    auto load_b = [&](BRegs& dst, int step, int t) {
        const int c = step / NPH;
        int tap = par * TAPS + t;
        if (KIND == 1) {    // tap (a, b) of phase (py, px) is kernel element ky = py ? 2a : 1 + 2a (same for kx)
            const int phy = (step % NPH) >> 1, phx = (step % NPH) & 1, ta = t >> 1, tb = t & 1;
            tap = (phy ? 2 * ta : 1 + 2 * ta) * 4 + (phx ? 2 * tb : 1 + 2 * tb);
        }
        const unsigned soff = (unsigned)(tap * nchunks + c) * tile_bytes;      // wave-uniform
#pragma unroll
        for (int sg = 0; sg < 2; sg++)
#pragma unroll
            for (int j = 0; j < WN; j++)
                dst.b[sg][j] = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(bvoff[sg] + 512u * j),
                                                                                             (int)soff, 0));
        if (R16) dst.b16 = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)bvoff16, (int)soff, 0));
    };

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    floatx4 acc16[R16 ? 2 * WM : 1];
#pragma unroll
    for (int i = 0; i < (R16 ? 2 * WM : 1); i++) acc16[i] = floatx4{0.f, 0.f, 0.f, 0.f};

    // per-lane LDS bases (floats): plane pair of this lane's k parity, its pixel / column, the wave's rows
    const int wrow = (wave_m * WM * RPB + l31 / TW) * HWD + (KIND == 2 ? py * HWD + px : 0);
    const float* a_lane = As + ((2 * h) * HP + wrow + l31 % TW) * 4;
    const float* a16_lane = As + (g16 * HP + wave_m * WM * HWD + (KIND == 2 ? py * HWD + px : 0) + l15) * 4;   // R16: TW = 32 only

    if (c_begin < c_end) {
        const ChunkSrc cs = chunk_src(c_begin * NPH);
#pragma unroll
        for (int j = 0; j < APT; j++)
            if (tid + CTHREADS * j < ASLOTS) store_a(cs, load_a(cs, j), j, 0);
    }
    const int s_begin = c_begin * NPH, s_end = c_end * NPH;
    BRegs breg[2];
    if (s_begin < s_end) load_b(breg[0], s_begin, 0);
    __syncthreads();
    for (int c = s_begin; c < s_end; c++) {                     // c = K step (chunk, phase)
        const int abuf = (c - s_begin) & 1;
        const bool next_chunk = c + 1 < s_end;
        ChunkSrc csn = chunk_src(next_chunk ? c + 1 : c);
        float4 avr[HDIST < 2 ? 2 : HDIST][APS];     // halo slices in flight: fetched during tap t, stored after tap t + HDIST - 1
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            // the next tap's (or the next chunk's first) weights are requested before this tap's MFMAs
            if (t < TAPS - 1) load_b(breg[(t + 1) & 1], c, t + 1);
            else if (next_chunk) load_b(breg[TAPS & 1], c + 1, 0);
            float4 (&av)[APS] = avr[t % (HDIST < 2 ? 2 : HDIST)];
#pragma unroll
            for (int u = 0; u < APS; u++) {
                const int j = t * APS + u;
                av[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (next_chunk && j < APT) av[u] = load_a(csn, j);
            }
            // halo pixel of output row (wave_m*WM + i), lane x for this tap: compile-time part here, the parity shift
            // of the transposed conv and the wave's row block are in a_lane
            int aoff;
            if (KIND == 0) aoff = (t / 3) * HWD + (t % 3);
            else if (KIND == 1) aoff = (t >> 1) * HWD + (t & 1);
            else aoff = ((t >> 1) == 0 ? 1 : 0) * HWD + ((t & 1) == 0 ? 1 : 0);
            constexpr int ROWSTEP = RPB * HWD;      // halo pixels between consecutive MFMA row blocks
            const float* a_s = a_lane + abuf * ACH + aoff * 4;
            const BRegs& bt = breg[t & 1];
#pragma unroll
            for (int sg = 0; sg < 2; sg++) {
                floatx4 a4[WM];
#pragma unroll
                for (int i = 0; i < WM; i++) a4[i] = *reinterpret_cast<const floatx4*>(a_s + (sg * HP + i * ROWSTEP) * 4);
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int i = 0; i < WM; i++)
#pragma unroll
                        for (int j = 0; j < WN; j++)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][e], bt.b[sg][j][e], acc[i][j], 0, 0, 0);
            }
            if (R16) {      // 16-column remainder: lane (row/col = l&15, kq = l>>4) takes plane (kq&1)*2 + (kq>>1): 4 k per MFMA
                const floatx4 bv = bt.b16;
#pragma unroll
                for (int sb = 0; sb < 2 * WM; sb++) {
                    const floatx4 av16 = *reinterpret_cast<const floatx4*>(
                        a16_lane + abuf * ACH + (aoff + (sb >> 1) * ROWSTEP + (sb & 1) * 16) * 4);
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        acc16[sb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av16[e], bv[e], acc16[sb], 0, 0, 0);
                }
            }
            // the slice fetched HDIST - 1 taps ago goes to LDS now (HDIST = 2: two taps of MFMAs cover the HBM latency)
            const int ts = t - (HDIST - 1);
            if (ts >= 0) {
#pragma unroll
                for (int u = 0; u < APS; u++) {
                    const int j = ts * APS + u;
                    if (next_chunk && j < APT && tid + CTHREADS * j < ASLOTS)
                        store_a(csn, avr[ts % (HDIST < 2 ? 2 : HDIST)][u], j, abuf ^ 1);
                }
            }
        }
        // slices fetched during the last HDIST - 1 taps
#pragma unroll
        for (int ts = TAPS - (HDIST - 1); ts < TAPS; ts++) {
            if (ts < 0 || ts * APS >= APT) continue;
#pragma unroll
            for (int u = 0; u < APS; u++) {
                const int j = ts * APS + u;
                if (next_chunk && j < APT && tid + CTHREADS * j < ASLOTS)
                    store_a(csn, avr[ts % (HDIST < 2 ? 2 : HDIST)][u], j, abuf ^ 1);
            }
        }
        if (TAPS & 1) breg[0] = breg[1];
        __syncthreads();                    // everybody is done reading this step's halo; the next one is complete
    }

    // ---- epilogue of rnr_conv2d_ray: frame = sum over the 26 rays of (tanh(y + b) + 1) * W, straight from the accumulators ----
    // Two passes (one image row per wave each): every lane scales the elements it holds — lanes of a wave are consecutive
    // columns, so the weights arrive as coalesced 128-byte rows — and parks them in LDS as [pixel][column]; after a barrier
    // 128 pixels x 3 colour channels are summed over their 26 columns (stride 3) and written as three coalesced row pieces.
    if (R16 && KIND == 0 && P.ray_w) {
        static_assert(!R16 || TW == 32, "remainder configuration");
        float* tbuf = As;                                   // [WAVES_M * 32 pixels][c_out_pad]: 40 KB of the 43 KB halo buffers
        const int cw = P.c_out_pad, hw = P.OH * P.OW;
        float bj[WN];
#pragma unroll
        for (int j = 0; j < WN; j++) bj[j] = P.ray_bias[n0 + wn0 + 32 * j + l31];
        const float b16 = P.ray_bias[n0 + wn0 + WN * 32 + l15];
#pragma unroll
        for (int i = 0; i < WM; i++) {
            const int y = y0 + wave_m * WM + i;
            const float* wrow = P.ray_w + (((size_t)n * P.OH + y) * P.OW + x0) * cw;
#pragma unroll
            for (int g = 0; g < 16; g++) {
                const int xx = (g & 3) + 8 * (g >> 2) + 4 * h;
#pragma unroll
                for (int j = 0; j < WN; j++) {
                    const int col = wn0 + 32 * j + l31;
                    const float w = wrow[xx * cw + col];
                    const float v = fast_tanh_plus1f(acc[i][j][g] + bj[j]) * w;
                    tbuf[(wave_m * 32 + xx) * cw + col] = (w == 0.0f) ? 0.0f : v;       // background (and padding) columns: exactly 0
                }
            }
#pragma unroll
            for (int half16 = 0; half16 < 2; half16++)
#pragma unroll
                for (int g4 = 0; g4 < 4; g4++) {
                    const int xx = half16 * 16 + kq * 4 + g4, col = wn0 + WN * 32 + l15;
                    const float w = wrow[xx * cw + col];
                    const float v = fast_tanh_plus1f(acc16[2 * i + half16][g4] + b16) * w;
                    tbuf[(wave_m * 32 + xx) * cw + col] = (w == 0.0f) ? 0.0f : v;
                }
            __syncthreads();
            const int n_cols = P.c_out / 3;                 // rays
            for (int it = tid; it < 3 * WAVES_M * 32; it += CTHREADS) {
                const int c = it / (WAVES_M * 32), pl = it - c * (WAVES_M * 32);
                const float* tp = tbuf + pl * cw + c;
                float sum = 0.0f;
                for (int r = 0; r < n_cols; r++) sum += tp[3 * r];
                const int yy = y0 + (pl >> 5) * WM + i;
                P.ray_image[((size_t)n * 3 + c) * hw + (size_t)yy * P.OW + x0 + (pl & 31)] = sum;
            }
            __syncthreads();                                // the next pass overwrites tbuf
        }
        return;
    }

    // ---- epilogue ----
    int* flag = reinterpret_cast<int*>(As + WAVES_M * BN * 2);      // behind the statistics scratch; LDS is free after the last barrier
    // in-launch split-K combine (128 x 128 tiles and smaller: the plans that split K): only the last slice of a tile goes
    // on, holding the sum
    if (!R16 && WM * WN <= 4 && P.tile_arrive) {
        if (!splitk_combine<WM, WN>(P, acc, (par * P.mtiles + mt_) * P.ntiles + nt_, split, wave, lane, tid, flag)) return;
    }
    float* out = P.out + (size_t)split * P.slab_stride;
    if (R16) {   // C layout of the 16x16 tiles: col = lane & 15, row = (lane >> 4) * 4 + reg
        const int col = n0 + wn0 + WN * 32 + l15;
#pragma unroll
        for (int sb = 0; sb < 2 * WM; sb++) {
            const int y = y0 + wave_m * WM + (sb >> 1);
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int x = x0 + (sb & 1) * 16 + kq * 4 + g;
                const size_t off = (KIND == 2)
                    ? (((size_t)n * P.OH + 2 * y + py) * P.OW + 2 * x + px) * P.c_out_pad
                    : (((size_t)n * P.OH + y) * P.OW + x) * P.c_out_pad;
                if (col < P.c_out_pad) out[off + col] = acc16[sb][g];
            }
        }
    }
    const bool with_stats = P.stats && (P.splitk == 1 || P.tile_arrive);
    if (with_stats) {
        float* red = As;   // [WAVES_M][BN][2]; LDS is free after the last barrier
#pragma unroll
        for (int j = 0; j < WN; j++) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const float v = acc[i][j][g];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                const int col = wn0 + 32 * j + l31;
                red[(wave_m * BN + col) * 2 + 0] = s1;
                red[(wave_m * BN + col) * 2 + 1] = s2;
            }
        }
        if (R16) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int sb = 0; sb < 2 * WM; sb++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const float v = acc16[sb][g];
                    s1 += v;
                    s2 += v * v;
                }
            s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
            s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
            if (kq == 0) {
                const int col = wn0 + WN * 32 + l15;
                red[(wave_m * BN + col) * 2 + 0] = s1;
                red[(wave_m * BN + col) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n0 + tid;
            if (col < P.c_out) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int w = 0; w < WAVES_M; w++) {
                    s1 += (double)red[(w * BN + tid) * 2 + 0];
                    s2 += (double)red[(w * BN + tid) * 2 + 1];
                }
                double* st = stat_slot(P, n, col);
                atomicAdd(st + 0, s1);
                atomicAdd(st + 1, s2);
            }
        }
    }
    if (with_stats && P.arrive) {       // producer-side BatchNorm: the output stores overlap the ticket's round trip
        const BnArrival a = bn_arrive(P, n, tid);
        store_acc_tiles<KIND, WM, WN, TW>(P, out, acc, n, y0, x0, py, px, wave_m, n0, wn0, l31, h);
        bn_complete(P, a, n, tid, flag);
    } else {
        store_acc_tiles<KIND, WM, WN, TW>(P, out, acc, n, y0, x0, py, px, wave_m, n0, wn0, l31, h);
    }
}

// (r03 experiment, measured and not adopted: conv_halo_kernel as a persistent tile loop with next-tile prefetch — -1.4 % at 8
// views per launch, DESIGN.md §3.3, profiles/archive/r03_layer_time_persist_ab_*.  The source is kept outside the product tree:
// scripts/experiments/conv_persist_experiment.inc, with the three inclusion points it needs.)

// ------------------------------------------------------------------------------------------------
// fp32 emulation on the 16-bit matrix cores (opt-in; gfx950's fp32-input MFMA runs at 1/16 of the 16-bit rate).
//
// Every conv operand is split into a few 16-bit terms whose sum is (nearly) the fp32 value, the partial products of
// significant weight are accumulated in fp32 by v_mfma_f32_32x32x16_{bf16,f16}; every partial product of two 16-bit
// terms is exact in fp32.  Two formats:
//   FMT 0 "bf16x6": x = h + m + l EXACTLY, three bf16 terms (8 + 8 + 8 significand bits, each the round-to-nearest bf16
//          of the running remainder); six products of weight >= 2^-16: hh, hm, mh, hl, lh, mm (dropped: <= 2^-24 |ab|).
//          6 x 32 = 192 MFMA cycles per 16-channel chunk instead of 8 x 64 = 512.
//   FMT 1 "f16x3":  x ~= h + l, two fp16 terms (11 + 11 significand bits: relative representation error <= 2^-23, a
//          quarter of an fp32 ulp... of a 22-bit significand); three products hh, hl, lh (dropped ll <= 2^-22 |ab|).
//          3 x 32 = 96 MFMA cycles per chunk.  Fewer accumulator roundings (3 per 16 channels instead of 8 per 16 in the
//          exact-fp32 chain) more than pay for the two dropped significand bits: measured error vs float64 is BELOW the
//          exact-fp32 kernel's on every layer shape of the U-Net (tests/test_gpu_unet.py).  fp16's narrow
//          exponent is handled as follows: the MFMA honours fp16 subnormals (scripts/micro/mfma_f16_denorm.hip), so a
//          residual term below 2^-14 keeps an absolute precision of 2^-25; weights (typically 1e-2, whose residuals
//          would all be subnormal) are pre-multiplied at pack time by a per-layer power of two that brings max|w| into
//          [2^12, 2^13) and the accumulators are multiplied by its inverse in the epilogue (exact); activations are
//          used as they are: valid for |act(scale*x+shift)| < 65504 (a BatchNorm output or a bounded input).
// LDS image of a 16-channel halo chunk: [NT terms][2 k-halves][pixels][8 x 16 bit] — an MFMA lane (pixel, k-half) reads
// its 8 channels of one term as one lane-consecutive ds_read_b128; the packed weights hold the same image per
// (parity, tap, chunk): [NT terms][2 k-halves][wstride columns][8 x 16 bit].
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// x = h + m + l exactly (for finite x away from the subnormal range); pairs of values -> packed bf16x2 words
__device__ __forceinline__ void split_bf16x3(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const floatx2 v = {x0, x1};
    const bf16x2 hb = __builtin_convertvector(v, bf16x2);
    const floatx2 r1 = v - __builtin_convertvector(hb, floatx2);
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const floatx2 r2 = r1 - __builtin_convertvector(mb, floatx2);
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}

typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));

// x ~= h + l, two fp16 terms (round-to-nearest; subnormal residuals keep 2^-25 absolute precision)
__device__ __forceinline__ void split_f16x2(float x0, float x1, unsigned& h, unsigned& l) {
    const floatx2 v = {x0, x1};
    const halfx2 hb = __builtin_convertvector(v, halfx2);
    const floatx2 r1 = v - __builtin_convertvector(hb, floatx2);
    const halfx2 lb = __builtin_convertvector(r1, halfx2);
    h = __builtin_bit_cast(unsigned, hb); l = __builtin_bit_cast(unsigned, lb);
}

template <int FMT> struct EmuFmt;
template <> struct EmuFmt<0> {
    static constexpr int NT = 3;
    typedef bf16x8 vec8;
    static __device__ __forceinline__ floatx16 mfma(vec8 a, vec8 b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&t)[3]) { split_bf16x3(x0, x1, t[0], t[1], t[2]); }
};
template <> struct EmuFmt<1> {
    static constexpr int NT = 2;
    typedef halfx8 vec8;
    static __device__ __forceinline__ floatx16 mfma(vec8 a, vec8 b, floatx16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ void split(float x0, float x1, unsigned (&t)[2]) { split_f16x2(x0, x1, t[0], t[1]); }
};
constexpr int EMU_HEADER_BYTES = 64;     // packed emulation image: [0] float 2^-k (accumulator rescale), [1] float k, [2] uint bits of max|w|

// ------------------------------------------------------------------------------------------------
// conv_halo_emu_kernel<FMT, ...>: tiling, fused BatchNorm prologue, halo image and statistics epilogue as in
// conv_halo_kernel (the exact-fp32 kernel); how the operands travel:
//   * weights never touch LDS: an MFMA lane's B operand (8 bf16 of one column, one k-half, one term) is 16 contiguous
//     bytes of the packed image, lanes of a wave are consecutive columns, so a tap's weights are 3*WN coalesced
//     global_load_dwordx4 per wave straight into the operand registers, prefetched one tap ahead (L2 / L1 serve the
//     re-reads of the other row-waves and workgroups).  No LDS-DMA issue cost, no weight tile to guard;
//   * hence no barrier per tap: the only LDS hazard left is the halo swap, and the halo is double-buffered in LDS
//     (the space the weight tiles used to take), so ONE barrier per 16-channel chunk (every TAPS * 6 * WM * WN MFMAs per
//     wave) suffices (the first generation of this kernel staged weight tiles by LDS-DMA with a barrier per tap: 5 %
//     slower, profiles/README.md);
//   * the next chunk's halo is fetched, normalised, split into its three bf16 terms and written to the other buffer
//     in slices spread over the taps (a slice is loaded during tap t and converted during tap t+1), i.e. the VALU work
//     of the split sits in the issue slots between MFMAs instead of in a burst between two barriers.
// Two workgroups (2 x 4 waves) per CU: two waves per SIMD cover each other's operand waits.
// ------------------------------------------------------------------------------------------------
template <int FMT, int KIND, int WAVES_M, int WAVES_N, int WM, int WN, int TW>
__global__ void __launch_bounds__(CTHREADS, 2)
conv_halo_emu_kernel(const ConvParams P) {
    typedef EmuFmt<FMT> F;
    typedef typename F::vec8 vec8;
    constexpr int NT = F::NT;
    static_assert(WAVES_M * WAVES_N == 4 && BK == 16, "four waves, 16-channel chunks");
    // TW = 32: a 32-row MFMA block is one image row of the tile; TW = 16 (maps 16 pixels wide): two image rows
    static_assert(TW == 32 || TW == 16, "tile width");
    constexpr int RPB = 32 / TW;                              // image rows per MFMA row block
    constexpr int TH = WAVES_M * WM * RPB;
    constexpr int BN = WAVES_N * WN * 32;
    // KIND 1 (4x4 stride 2) runs as FOUR stride-1 2x2-tap convolutions, one per input parity phase (py, px): input
    // row 2y + ky - 1 = 2(y + ty) + py with (ky; ty, py) = (0; -1, 1), (1; 0, 0), (2; 0, 1), (3; 1, 0).  A K step is a
    // (16-channel chunk, phase) pair whose halo is the (TH+1) x 33 pixels of THAT phase only (297 pixels for a 32 x 8
    // tile, where the interleaved 66-wide halo of all 16 taps takes 1188): small enough to double-buffer 256-row tiles
    // at two workgroups per CU, 4 taps between barriers.
    constexpr int NPH = KIND == 1 ? 4 : 1;                    // K steps per 16-channel chunk
    constexpr int TAPS = KIND == 0 ? 9 : 4;                   // taps per K step
    constexpr int HWD = KIND == 1 ? TW + 1 : TW + 2;
    constexpr int HHT = KIND == 1 ? TH + 1 : TH + 2;
    constexpr int HP = HWD * HHT;
    constexpr int ASLOTS = HP * 4;
    constexpr int APT = (ASLOTS + CTHREADS - 1) / CTHREADS;
    constexpr int SPT = (APT + TAPS - 1) / TAPS;              // halo slots a thread converts per tap
    constexpr int NGROUPS = (APT + SPT - 1) / SPT;            // <= TAPS
    constexpr int APL = 2 * HP * 16;                          // bytes per term plane (two k-halves)
    constexpr int ACHB = NT * APL;                            // bytes per halo image
    constexpr int ROWSTEP = RPB * HWD;                        // halo pixels between consecutive MFMA row blocks

    extern __shared__ __attribute__((aligned(16))) char smemb[];
    char* As = smemb;                   // [2][ACHB]

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wave_m = wave / WAVES_N, wave_n = wave % WAVES_N;
    const int wn0 = wave_n * WN * 32;
    int mt_, nt_, z_;
    tile_coords(P, mt_, nt_, z_);
    if (P.tile_mask && P.tile_mask[mt_] == 0) return;
    const int par = (KIND == 2) ? (z_ & 3) : 0;
    const int split = (KIND == 2) ? (z_ >> 2) : z_;
    const int py = par >> 1, px = par & 1;
    const int n0 = nt_ * BN;
    const int tiles_x = P.Wo / TW, tiles_y = P.Ho / TH;
    const int n = mt_ / (tiles_x * tiles_y);
    const int trem = mt_ - n * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * TH, x0 = (trem % tiles_x) * TW;

    const int q = tid & 3;
    unsigned spix[KIND == 1 ? 1 : APT];      // source pixel of a slot (KIND 1: recomputed per phase from siy / six)
    short siy[KIND == 1 ? APT : 1], six[KIND == 1 ? APT : 1];      // KIND 1: 2 (y0 + hy), 2 (x0 + hx)
    int sdst[APT];           // byte offset of this slot's 4 bf16 inside a term plane
    float smask[KIND == 2 ? APT : 1];
#pragma unroll
    for (int j = 0; j < APT; j++) {
        // slots past the halo only fetch (a valid address); they are never stored (every LDS slot is written once)
        int s = tid + CTHREADS * j;
        if (s >= ASLOTS) s -= ASLOTS;
        const int hp = s >> 2;
        const int hy = hp / HWD, hx = hp - hy * HWD;
        int iy = 0, ix = 0;
        const int col = hx;
        if (KIND == 0) { iy = reflect1(y0 - 1 + hy, P.H); ix = reflect1(x0 - 1 + hx, P.W); }
        else if (KIND == 1) { siy[j] = (short)(2 * (y0 + hy)); six[j] = (short)(2 * (x0 + hx)); }
        else {
            iy = y0 - 1 + hy; ix = x0 - 1 + hx;
            const bool inside = iy >= 0 && iy < P.H && ix >= 0 && ix < P.W;
            smask[j] = inside ? 1.f : 0.f;
            iy = min(max(iy, 0), P.H - 1); ix = min(max(ix, 0), P.W - 1);
        }
        sdst[j] = ((q >> 1) * HP + hy * HWD + col) * 16 + (q & 1) * 8;      // k-half = channels 8*(q>>1) .., 4 bf16 at (q&1)*4
        if (KIND != 1) spix[j] = (unsigned)(iy * P.W + ix);
    }

    const int nchunks = P.chunks_per_tap;
    const int per_split = (nchunks + P.splitk - 1) / P.splitk;
    const int c_begin = split * per_split;
    const int c_end = min(nchunks, c_begin + per_split);

    // Global operands go through buffer loads: a wave-uniform base (resource + scalar offset) plus ONE 32-bit per-lane
    // offset.  With flat 64-bit addresses the unrolled tap loop keeps a strength-reduced pointer pair per (tap, term)
    // alive across the chunk loop and spills.
    // K steps: step = chunk * NPH + phase
    struct ChunkSrc { __amdgpu_buffer_rsrc_t rsrc; unsigned C; unsigned soff; int act; int phy, phx; float4 sc, sh; };
    auto chunk_src = [&](int step) {
        ChunkSrc cs;
        const int c = step / NPH;
        cs.phy = (step % NPH) >> 1; cs.phx = (step % NPH) & 1;
        const int s = c < P.chunks0 ? 0 : 1;
        const int cc = (c - (s ? P.chunks0 : 0)) * BK;
        cs.C = (unsigned)P.src_c[s];
        cs.rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(P.src_data[s] + (size_t)n * P.H * P.W * cs.C), 0,
                                                    0x7fffffff, 0x27000);
        cs.soff = (unsigned)cc * 4u;
        cs.act = P.src_act[s];
        cs.sc = make_float4(1.f, 1.f, 1.f, 1.f);
        cs.sh = make_float4(0.f, 0.f, 0.f, 0.f);
        if (P.src_scale[s]) cs.sc = *reinterpret_cast<const float4*>(P.src_scale[s] + (size_t)n * cs.C + cc + 4 * q);
        if (P.src_shift[s]) cs.sh = *reinterpret_cast<const float4*>(P.src_shift[s] + (size_t)n * cs.C + cc + 4 * q);
        return cs;
    };
    auto load_a = [&](const ChunkSrc& cs, int j) {
        unsigned pixel;
        if (KIND == 1) pixel = (unsigned)(reflect1(siy[j] - cs.phy, P.H) * P.W + reflect1(six[j] - cs.phx, P.W));
        else pixel = spix[j];
        const unsigned voff = (pixel * cs.C + 4u * (unsigned)q) * 4u;
        return __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(cs.rsrc, (int)voff, (int)cs.soff, 0));
    };
    auto store_a = [&](const ChunkSrc& cs, float4 v, int j, char* img) {
        float x = apply_act(v.x * cs.sc.x + cs.sh.x, cs.act);
        float y = apply_act(v.y * cs.sc.y + cs.sh.y, cs.act);
        float z = apply_act(v.z * cs.sc.z + cs.sh.z, cs.act);
        float w = apply_act(v.w * cs.sc.w + cs.sh.w, cs.act);
        if (KIND == 2) { x *= smask[j]; y *= smask[j]; z *= smask[j]; w *= smask[j]; }
        unsigned t0[NT], t1[NT];
        F::split(x, y, t0);
        F::split(z, w, t1);
        char* a = img + sdst[j];
#pragma unroll
        for (int term = 0; term < NT; term++) *reinterpret_cast<uint2*>(a + term * APL) = make_uint2(t0[term], t1[term]);
    };
    // weights: packed image per (parity, tap, chunk) = [NT terms][2 k-halves][wstride][8 x 16 bit] behind a 64-byte header
    const char* wimg = reinterpret_cast<const char*>(P.weight_emu) + EMU_HEADER_BYTES;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wimg), 0, 0x7fffffff, 0x27000);
    const unsigned tile_bytes = (unsigned)(NT * 32) * (unsigned)P.wstride;
    unsigned bvoff[NT];
#pragma unroll
    for (int term = 0; term < NT; term++)
        bvoff[term] = ((unsigned)(term * 2 + h) * (unsigned)P.wstride + (unsigned)(n0 + wn0 + l31)) * 16u;
    auto load_b = [&](vec8 (&dst)[NT][WN], int step, int t) {
        const int c = step / NPH;
        int tap = par * TAPS + t;
        if (KIND == 1) {    // tap (a, b) of phase (py, px) is kernel element ky = py ? 2a : 1 + 2a (same for kx)
            const int phy = (step % NPH) >> 1, phx = (step % NPH) & 1, ta = t >> 1, tb = t & 1;
            tap = (phy ? 2 * ta : 1 + 2 * ta) * 4 + (phx ? 2 * tb : 1 + 2 * tb);
        }
        const unsigned soff = (unsigned)(tap * nchunks + c) * tile_bytes;      // wave-uniform
#pragma unroll
        for (int term = 0; term < NT; term++)
#pragma unroll
            for (int j = 0; j < WN; j++)
                dst[term][j] = __builtin_bit_cast(vec8, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, (int)(bvoff[term] + 512u * j),
                                                                                           (int)soff, 0));
    };

    floatx16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; i++)
#pragma unroll
        for (int j = 0; j < WN; j++)
#pragma unroll
            for (int g = 0; g < 16; g++) acc[i][j][g] = 0.0f;

    const int wrow = wave_m * WM * RPB * HWD + (KIND == 2 ? py * HWD + px : 0);
    const int a_lane_off = (h * HP + wrow + (l31 / TW) * HWD + (l31 % TW)) * 16;

    const int s_begin = c_begin * NPH, s_end = c_end * NPH;
    vec8 b[2][NT][WN];
    if (s_begin < s_end) {
        const ChunkSrc cs = chunk_src(s_begin);
        load_b(b[0], s_begin, 0);
#pragma unroll
        for (int j = 0; j < APT; j++)
            if (tid + CTHREADS * j < ASLOTS) store_a(cs, load_a(cs, j), j, As);
    }
    __syncthreads();
    int cur = 0;
    for (int c = s_begin; c < s_end; c++, cur ^= 1) {            // c = K step (chunk, phase)
        const bool next_chunk = c + 1 < s_end;
        const ChunkSrc csn = chunk_src(next_chunk ? c + 1 : c);
        const char* a_rd = As + cur * ACHB + a_lane_off;
        char* a_wr = As + (cur ^ 1) * ACHB;
        float4 av[2][SPT];
#pragma unroll
        for (int t = 0; t < TAPS; t++) {
            // ---- operands of the NEXT tap / chunk are requested before this tap's MFMAs ----
            if (t < TAPS - 1) load_b(b[(t + 1) & 1], c, t + 1);
            else if (next_chunk) load_b(b[TAPS & 1], c + 1, 0);
            if (next_chunk) {
                if (t < NGROUPS) {
#pragma unroll
                    for (int u = 0; u < SPT; u++)
                        if (t * SPT + u < APT) av[t & 1][u] = load_a(csn, t * SPT + u);
                }
                if (t >= 1 && t - 1 < NGROUPS) {
#pragma unroll
                    for (int u = 0; u < SPT; u++) {
                        const int j = (t - 1) * SPT + u;
                        if (j < APT && tid + CTHREADS * j < ASLOTS) store_a(csn, av[(t - 1) & 1][u], j, a_wr);
                    }
                }
            }
            int aoff;
            if (KIND == 0) aoff = (t / 3) * HWD + (t % 3);
            else if (KIND == 1) aoff = (t >> 1) * HWD + (t & 1);
            else aoff = ((t >> 1) == 0 ? 1 : 0) * HWD + ((t & 1) == 0 ? 1 : 0);
            const char* a_s = a_rd + aoff * 16;
            // halo terms are fetched one at a time, smallest first; term ta pairs with the weight terms tb <= NT-1-ta
            // (bf16x6: a_l b_h; a_m b_m, a_m b_h; a_h b_l, a_h b_m, a_h b_h.  f16x3: a_l b_h; a_h b_l, a_h b_h)
#pragma unroll
            for (int ta = NT - 1; ta >= 0; ta--) {
                vec8 a[WM];
#pragma unroll
                for (int i = 0; i < WM; i++) a[i] = *reinterpret_cast<const vec8*>(a_s + ta * APL + i * ROWSTEP * 16);
#pragma unroll
                for (int tb = NT - 1 - ta; tb >= 0; tb--)
#pragma unroll
                    for (int i = 0; i < WM; i++)
#pragma unroll
                        for (int j = 0; j < WN; j++)
                            acc[i][j] = F::mfma(a[i], b[t & 1][tb][j], acc[i][j]);
            }
        }
        if (next_chunk && NGROUPS == TAPS) {        // the slice loaded during the last tap
#pragma unroll
            for (int u = 0; u < SPT; u++) {
                const int j = (TAPS - 1) * SPT + u;
                if (j < APT && tid + CTHREADS * j < ASLOTS) store_a(csn, av[(TAPS - 1) & 1][u], j, a_wr);
            }
        }
        if (TAPS & 1) {
#pragma unroll
            for (int term = 0; term < NT; term++)
#pragma unroll
                for (int j = 0; j < WN; j++) b[0][term][j] = b[1][term][j];
        }
        __syncthreads();        // next halo complete and visible; everybody is done reading the current one
    }

    // ---- epilogue (identical to conv_halo_kernel: same accumulator layout) ----
    // f16x3: undo the power-of-two weight scale — at the store, and on the column sums of the statistics (a power of
    // two commutes with every fp32 rounding involved, so this equals scaling the accumulators first; scaling all 128 of
    // them in place made the compiler keep both copies and spill)
    const float winv = FMT == 1 ? *reinterpret_cast<const float*>(P.weight_emu) : 1.0f;
    int* flag = reinterpret_cast<int*>(As) + WAVES_M * BN * 2;      // behind the statistics scratch
    if (WM * WN <= 4 && P.tile_arrive) {    // in-launch split-K combine (images hold the unscaled accumulators)
        if (!splitk_combine<WM, WN>(P, acc, (par * P.mtiles + mt_) * P.ntiles + nt_, split, wave, lane, tid, flag)) return;
    }
    float* out = P.out + (size_t)split * P.slab_stride;
    const bool with_stats = P.stats && (P.splitk == 1 || P.tile_arrive);
    if (with_stats) {
        float* red = reinterpret_cast<float*>(As);   // [WAVES_M][BN][2]; LDS is free after the last barrier
#pragma unroll
        for (int j = 0; j < WN; j++) {
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int i = 0; i < WM; i++)
#pragma unroll
                for (int g = 0; g < 16; g++) {
                    const float v = acc[i][j][g];
                    s1 += v;
                    s2 += v * v;
                }
            if (FMT == 1) { s1 *= winv; s2 = (s2 * winv) * winv; }
            s1 += __shfl_xor(s1, 32, 64);
            s2 += __shfl_xor(s2, 32, 64);
            if (h == 0) {
                const int col = wn0 + 32 * j + l31;
                red[(wave_m * BN + col) * 2 + 0] = s1;
                red[(wave_m * BN + col) * 2 + 1] = s2;
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int col = n0 + tid;
            if (col < P.c_out) {
                double s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int w = 0; w < WAVES_M; w++) {
                    s1 += (double)red[(w * BN + tid) * 2 + 0];
                    s2 += (double)red[(w * BN + tid) * 2 + 1];
                }
                double* st = stat_slot(P, n, col);
                atomicAdd(st + 0, s1);
                atomicAdd(st + 1, s2);
            }
        }
    }
    if (with_stats && P.arrive) {       // producer-side BatchNorm
        const BnArrival a = bn_arrive(P, n, tid);
        store_acc_tiles<KIND, WM, WN, TW, FMT == 1>(P, out, acc, n, y0, x0, py, px, wave_m, n0, wn0, l31, h, winv);
        bn_complete(P, a, n, tid, flag);
    } else {
        store_acc_tiles<KIND, WM, WN, TW, FMT == 1>(P, out, acc, n, y0, x0, py, px, wave_m, n0, wn0, l31, h, winv);
    }
}

template <int FMT, int KIND, int WAVES_M, int WAVES_N, int WM, int WN, int TW = 32>
static void launch_halo_emu_cfg(const dim3 grid, const ConvParams& P, hipStream_t st) {
    constexpr int TH = WAVES_M * WM * (32 / TW);
    constexpr int HP = (KIND == 1 ? TW + 1 : TW + 2) * (KIND == 1 ? TH + 1 : TH + 2);
    constexpr size_t lds_halo = (size_t)(2 * EmuFmt<FMT>::NT * 32 * HP);
    constexpr size_t lds_red = (size_t)(WAVES_M * WAVES_N * WN * 32 * 2) * sizeof(float);
    constexpr size_t lds = lds_halo > lds_red ? lds_halo : lds_red;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_emu_kernel<FMT, KIND, WAVES_M, WAVES_N, WM, WN, TW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv_halo_emu_kernel<FMT, KIND, WAVES_M, WAVES_N, WM, WN, TW>), grid, dim3(CTHREADS), lds, st, P);
}

// Wave quantisation of small grids.  With n tiles per CU and s co-resident workgroups per CU the grid runs in
// ceil(n / s) rounds; a last round of ONE workgroup per CU leaves a single wave per SIMD, which cannot keep the matrix
// pipe busy (measured on the 64-column layers at one view per call: 4 tiles per CU on 3 slots = 3 + 1 -> 60 % of the
// peak, 2 + 2 on two slots -> 80 %).  Model: a round of r co-resident workgroups costs r / eff(r) tile-times with
// eff = 0.55 / 0.86 / 0.90 for r = 1 / 2 / 3; pick the slot count (<= what registers and LDS allow) with the smallest
// total.  Large grids (>= 4 rounds) keep the maximum.  RNR_HALO_SLOTS=k forces k (experiments).
static int balanced_slots(long tiles, int max_slots) {
    static const int forced = [] { const char* e = getenv("RNR_HALO_SLOTS"); return e ? atoi(e) : 0; }();
    if (forced > 0) return forced < max_slots ? forced : max_slots;
    const long n = (tiles + 255) / 256;            // tiles per CU (256 CUs)
    if (n >= 4L * max_slots || max_slots <= 1) return max_slots;
    static const double eff[4] = {0.0, 0.55, 0.86, 0.90};
    int best = max_slots;
    double best_t = 1e30;
    for (int s = max_slots; s >= 1; s--) {
        const long full = n / s, rem = n % s;
        double t = (double)full * s / eff[s < 3 ? s : 3];
        if (rem) t += (double)rem / eff[rem < 3 ? rem : 3];
        if (t < best_t - 1e-9) { best_t = t; best = s; }
    }
    return best;
}

template <int KIND, int WAVES_M, int WAVES_N, int WM, int WN, int R16, int TW = 32>
static void launch_halo_cfg(const dim3 grid, const ConvParams& P, hipStream_t st) {
    constexpr int TH = WAVES_M * WM * (32 / TW), BN = WAVES_N * (WN * 32 + R16 * 16);
    constexpr int HP = (KIND == 1 ? TW + 1 : TW + 2) * (KIND == 1 ? TH + 1 : TH + 2);
    constexpr size_t lds_halo = (size_t)(2 * BK * HP) * sizeof(float);
    constexpr size_t lds_red = (size_t)(WAVES_M * BN * 2) * sizeof(float);         // statistics reduction of the epilogue
    constexpr size_t lds_min = lds_halo > lds_red ? lds_halo : lds_red;
    // workgroups per CU the registers allow (the kernel's __launch_bounds__) and LDS allows
    constexpr int nat = halo_waves(WM, WN, R16);
    constexpr int lds_slots = (int)((160 * 1024) / lds_min);
    const int slots = balanced_slots((long)grid.x, nat < lds_slots ? nat : lds_slots);
    // fewer co-resident workgroups are requested by padding the dynamic LDS allocation
    const size_t lds = slots < lds_slots ? (size_t)(160 * 1024 / slots) & ~(size_t)255 : lds_min;
    static size_t attr_set = 0;
    if (attr_set < lds) {    // > 64 KiB of dynamic LDS needs the opt-in
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_halo_kernel<KIND, WAVES_M, WAVES_N, WM, WN, R16, TW>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = lds;
    }
    hipLaunchKernelGGL((conv_halo_kernel<KIND, WAVES_M, WAVES_N, WM, WN, R16, TW>), grid, dim3(CTHREADS), lds, st, P);
}


// out[m,c] = sum_s slab[s][m,c]; statistics per view.  One float4 of an output row per thread and pass: 16 rows x 64 columns
// per pass of a 256-thread workgroup, rpw / 16 passes (the host picks rpw = 16 ... 128 rows per workgroup so that big maps do
// not pay one float64 atomic per 16 rows and column, while even the 16 x 16 maps spread over >= 128 workgroups).
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int splitk, float* __restrict__ out,
                     long rows, int rows_per_view, int rpw, const ConvParams P) {
    __shared__ double red[16][64][2];
    double* const stats = P.stats;
    const int c_out = P.c_out, c_out_pad = P.c_out_pad;
    const int tid = threadIdx.x;
    const int cq = tid & 15, ry = tid >> 4;
    const int col = blockIdx.y * 64 + cq * 4;
    const long r0 = (long)blockIdx.x * rpw;
    const bool single_view = (r0 / rows_per_view) == ((min(r0 + rpw, rows) - 1) / rows_per_view);
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (int rr = 0; rr < rpw; rr += 16) {
        const long m = r0 + rr + ry;
        const bool live = m < rows && col < c_out_pad;
        if (!live) continue;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float* p = slabs + (size_t)m * c_out_pad + col;
        for (int s = 0; s < splitk; s++) {
            const float4 t = *reinterpret_cast<const float4*>(p + (size_t)s * slab_stride);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        *reinterpret_cast<float4*>(out + (size_t)m * c_out_pad + col) = v;
        if (!stats) continue;
        const float vv[4] = {v.x, v.y, v.z, v.w};
        if (!single_view) {     // rows of two views in one workgroup (maps smaller than 16 pixels; rpw = 16 there)
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (col + k < c_out) {
                    double* st = stat_slot(P, (int)(m / rows_per_view), col + k);
                    atomicAdd(st + 0, (double)vv[k]);
                    atomicAdd(st + 1, (double)vv[k] * (double)vv[k]);
                }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) { s1[k] += (double)vv[k]; s2[k] += (double)vv[k] * (double)vv[k]; }
        }
    }
    if (!stats) return;
    if (single_view) {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            red[ry][cq * 4 + k][0] = s1[k];
            red[ry][cq * 4 + k][1] = s2[k];
        }
        __syncthreads();
        if (tid < 64) {
            const int c = blockIdx.y * 64 + tid;
            if (c < c_out) {
                double a = 0.0, b2 = 0.0;
#pragma unroll
                for (int r = 0; r < 16; r++) { a += red[r][tid][0]; b2 += red[r][tid][1]; }
                double* st = stat_slot(P, (int)(r0 / rows_per_view), c);
                atomicAdd(st + 0, a);
                atomicAdd(st + 1, b2);
            }
        }
    }
    if (P.arrive) {      // producer-side BatchNorm: one counter for the launch
        const BnArrival a = bn_arrive(P, -1, tid);
        bn_complete(P, a, -1, tid, reinterpret_cast<int*>(&red[0][0][0]));
    }
}

// rnr_conv2d_fused's BatchNorm as its own launch: convolutions whose workgroups all finish together (one workgroup per CU,
// the split-K reduce kernel) gain nothing from drawing tickets — three dependent round trips at the end of the kernel cost
// more than the 4.7 us of this launch (measured at one view per call: +8 ... +15 us on the 256-workgroup layers).
// One workgroup per (view, 256 channels): every thread has ONE channel, its eight shard loads in flight together (a single
// workgroup per view walked 512 channels in two dependent rounds: 1.5 us of the 4.7, r06).
__global__ void __launch_bounds__(CTHREADS)
bn_finalize_shards_kernel(const ConvParams P) {
    bn_finalize_views(P, blockIdx.x, 1, threadIdx.x, (int)(blockIdx.y * blockDim.x), (int)(gridDim.y * blockDim.x));
}

template <bool RESET>
__global__ void __launch_bounds__(256)
bn_finalize_kernel(double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                   float* __restrict__ scale, float* __restrict__ shift, int nviews, int channels, int c_pad,
                   double count, float eps) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nviews * c_pad) return;
    const int c = i % c_pad;
    float sc = 0.f, sh = 0.f;
    if (c < channels) {
        const double mean = stats[2 * (size_t)i + 0] / count;
        double var = stats[2 * (size_t)i + 1] / count - mean * mean;    // biased variance
        var = var < 0.0 ? 0.0 : var;
        const double s = (double)gamma[c] / sqrt(var + (double)eps);
        sc = (float)s;
        sh = (float)((double)beta[c] - mean * s);
    }
    scale[i] = sc;
    shift[i] = sh;
    if (RESET) { stats[2 * (size_t)i + 0] = 0.0; stats[2 * (size_t)i + 1] = 0.0; }
}

// Whole-batch statistics (torch's train-mode BatchNorm2d over (N,H,W), pytorch_prototyping.py via nn.BatchNorm2d):
// one lane per channel sums the per-view partial sums, writes the same scale/shift to every view and optionally
// updates running_mean / running_var (momentum m, UNBIASED variance, as torch does).  Always resets stats.
__global__ void __launch_bounds__(256)
bn_finalize_batch_kernel(double* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta,
                         float* __restrict__ scale, float* __restrict__ shift, float* __restrict__ running_mean,
                         float* __restrict__ running_var, float momentum, int nviews, int channels, int c_pad,
                         double count_per_view, float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= c_pad) return;
    double s1 = 0.0, s2 = 0.0;
    for (int n = 0; n < nviews; n++) {
        const size_t i = (size_t)n * c_pad + c;
        s1 += stats[2 * i + 0];
        s2 += stats[2 * i + 1];
        stats[2 * i + 0] = 0.0;
        stats[2 * i + 1] = 0.0;
    }
    float sc = 0.f, sh = 0.f;
    if (c < channels) {
        const double count = count_per_view * (double)nviews;
        const double mean = s1 / count;
        double var = s2 / count - mean * mean;
        var = var < 0.0 ? 0.0 : var;
        const double s = (double)gamma[c] / sqrt(var + (double)eps);
        sc = (float)s;
        sh = (float)((double)beta[c] - mean * s);
        if (running_mean) running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        if (running_var) {
            const double unb = count > 1.0 ? var * count / (count - 1.0) : var;
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
        }
    }
    for (int n = 0; n < nviews; n++) {
        scale[(size_t)n * c_pad + c] = sc;
        shift[(size_t)n * c_pad + c] = sh;
    }
}

__host__ __device__ __forceinline__ int weight_row_stride(int c_out_pad) { return (c_out_pad + 127) / 128 * 128; }

// value of the (parity, tap, padded input channel c, output column co) entry of the implicit-GEMM weight matrix
__device__ __forceinline__ float gemm_weight(const rnr_conv_desc& d, const float* __restrict__ w, int par, int tp, int c, int co) {
    int ci = -1;
    if (c < d.c_in0_pad) { if (c < d.c_in0) ci = c; }
    else { const int c1 = c - d.c_in0_pad; if (c1 < d.c_in1) ci = d.c_in0 + c1; }
    if (ci < 0 || co >= d.c_out) return 0.0f;
    const int cin = d.c_in0 + d.c_in1;
    if (d.kind == RNR_CONV3x3_REFLECT) return w[((size_t)co * cin + ci) * 9 + tp];
    if (d.kind == RNR_CONV4x4S2_REFLECT) return w[((size_t)co * cin + ci) * 16 + tp];
    const int py = par >> 1, px = par & 1, ty = tp >> 1, tx = tp & 1;
    const int ky = py == 0 ? (ty == 0 ? 1 : 3) : (ty == 0 ? 0 : 2);
    const int kx = px == 0 ? (tx == 0 ? 1 : 3) : (tx == 0 ? 0 : 2);
    return w[((size_t)ci * d.c_out + co) * 16 + ky * 4 + kx];
}

// i enumerates [par][tap][c][co < wstride]
__device__ __forceinline__ void gemm_weight_index(const rnr_conv_desc& d, long i, int& par, int& tp, int& c, int& co,
                                                  int& taps, int& ctot, int& wstride) {
    taps = d.kind == RNR_CONV3x3_REFLECT ? 9 : (d.kind == RNR_CONV4x4S2_REFLECT ? 16 : 4);
    ctot = d.c_in0_pad + d.c_in1_pad;
    wstride = weight_row_stride(d.c_out_pad);
    co = (int)(i % wstride);
    long r = i / wstride;
    c = (int)(r % ctot);
    r /= ctot;
    tp = (int)(r % taps);
    par = (int)(r / taps);
}

__global__ void __launch_bounds__(256)
pack_weight_kernel(rnr_conv_desc d, const float* __restrict__ w, float* __restrict__ packed, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    int par, tp, c, co, taps, ctot, wstride;
    gemm_weight_index(d, i, par, tp, c, co, taps, ctot, wstride);
    const float v = gemm_weight(d, w, par, tp, c, co);
    // destination: chunk-major plane image [par][tap][chunk][4 g][wstride][4 e], k = c % 16 -> g = 2*(k&1) + (k>>3),
    // e = (k>>1)&3  (the LDS image conv_halo_kernel DMAs verbatim)
    const int k = c & 15;
    const long chunk = ((long)(par * taps + tp) * (ctot / 16) + (c >> 4));
    packed[(chunk * 4 + (2 * (k & 1) + (k >> 3))) * ((long)wstride * 4) + (long)co * 4 + ((k >> 1) & 3)] = v;
}

// max |w| over the PyTorch weight tensor, as the bit pattern of a non-negative float (orders like an unsigned)
__global__ void __launch_bounds__(256) weight_amax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ amax_bits) {
    unsigned m = 0;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float v = fabsf(w[i]);
        if (v == v && v < __builtin_inff()) m = max(m, __builtin_bit_cast(unsigned, v));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
    if ((threadIdx.x & 63) == 0 && m) atomicMax(amax_bits, m);
}

// emulation image: 64-byte header, then [par][tap][chunk][NT terms][2 k-halves][wstride][8 x 16 bit]  (conv_halo_emu_kernel).
// FMT 1 (f16x3): weights are multiplied by 2^k, k = 12 - floor(log2 max|w|), before the split; header[0] = 2^-k.
template <int FMT>
__global__ void __launch_bounds__(256)
pack_weight_emu_kernel(rnr_conv_desc d, const float* __restrict__ w, char* __restrict__ image, long total) {
    constexpr int NT = EmuFmt<FMT>::NT;
    float* header = reinterpret_cast<float*>(image);
    unsigned short* packed = reinterpret_cast<unsigned short*>(image + EMU_HEADER_BYTES);
    float wscale = 1.0f;
    int kexp = 0;
    if (FMT == 1) {
        const float amax = __builtin_bit_cast(float, reinterpret_cast<const unsigned*>(image)[2]);
        if (amax > 0.0f) {
            kexp = 12 - ilogbf(amax);
            // +-40: the epilogue's column statistics square the SCALED accumulators before the scale is undone, and 2^80
            // times a squared activation sum still fits fp32; weights below 2^-28 simply keep fewer fp16 bits
            kexp = min(max(kexp, -40), 40);
        }
        wscale = ldexpf(1.0f, kexp);
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { header[0] = ldexpf(1.0f, -kexp); header[1] = (float)kexp; }
    if (i >= total) return;
    int par, tp, c, co, taps, ctot, wstride;
    gemm_weight_index(d, i, par, tp, c, co, taps, ctot, wstride);
    const float v = gemm_weight(d, w, par, tp, c, co) * wscale;
    unsigned term[NT];
    EmuFmt<FMT>::split(v, 0.0f, term);
    const int k = c & 15;
    const long chunk = ((long)(par * taps + tp) * (ctot / 16) + (c >> 4));
#pragma unroll
    for (int t = 0; t < NT; t++)
        packed[(((chunk * NT + t) * 2 + (k >> 3)) * (long)wstride + co) * 8 + (k & 7)] = (unsigned short)(term[t] & 0xffffu);
}

#ifndef WINO_OUT_AUX
#define WINO_OUT_AUX 0           // cache policy of the Winograd kernels' output stores (aux operand: 0 default, 2 = nt / streaming)
#endif
#include "conv_wino.inc"
#include "conv_wino80.inc"
#include "conv_wino2.inc"
#ifndef W2_PAIRS
#define W2_PAIRS 2            // bit 1: conv_wino2p_kernel (K-step pairs, r05) runs the TRANSPOSED 4x4 stride-2 layers (-4.3 ... -4.7 %), bit 0:
                              // ... the stride-2 convolutions too (measured + 8 ... + 23 %: the kernel needs scratch there and its 12-MFMA weight
                              // look-ahead stalls behind the two HBM halo loads of every phase block; profiles/r05_wino2_pairs_ab.txt); the
                              // others run conv_wino2_kernel (r03 / r04)
#endif
#define W2_PAIRS_KIND(kind) ((kind) == RNR_CONVT4x4S2 ? ((W2_PAIRS) & 2) != 0 : ((W2_PAIRS) & 1) != 0)
#include "conv_wino2p.inc"
#include "conv_wino4.inc"

// mask[tile] = any(alpha > 0) over the tw x th output pixels of the tile (tile order = the halo kernels' mt index)
__global__ void __launch_bounds__(256) active_tile_kernel(const float* __restrict__ alpha, uint8_t* __restrict__ mask, int H,
                                                          int W, int th, int tw) {
    const int tiles_x = W / tw, tiles_y = H / th;
    const int mt = blockIdx.x;
    const int n = mt / (tiles_x * tiles_y);
    const int trem = mt - n * (tiles_x * tiles_y);
    const int y0 = (trem / tiles_x) * th, x0 = (trem % tiles_x) * tw;
    const int tx = threadIdx.x % tw, ty = threadIdx.x / tw;
    int any = 0;
    for (int y = ty; y < th; y += 256 / tw) any |= alpha[((size_t)n * H + y0 + y) * W + x0 + tx] > 0.f ? 1 : 0;
    any = __syncthreads_or(any);
    if (threadIdx.x == 0) mask[mt] = any ? 1 : 0;
}

struct ConvPlan {
    int halo;       // 1: conv3x3_halo_kernel (2-D pixel tiles), 0: conv_mfma_kernel (linear pixel tiles)
    int wino;       // 4: conv_wino4_kernel (F(4x4, 3x3), 32 x 16 pixel tiles x 64 columns), 3: conv_wino80_kernel (F(2x2, 3x3) for the 80-column out layer, 16 x 4 pixel tiles),
                    // 1: conv_wino_kernel (Winograd F(2x2, 3x3), 16 x 8 pixel tiles x 64 columns), 2: conv_wino2_kernel (F(2x2, 2x2),
                    // the 4x4 stride-2 convolutions: 16 x 8 tiles of the GEMM row space x 128 (conv) / 64 (transposed) columns)
    int cfg;        // column config 0: 64, 1: 96 (gather) / 80 (halo) / 96 (emulation), 2: 128; rows = bm (64 ... 256)
    int bm, bn, mtiles, ntiles, par, splitk;
    int tw;         // pixel-tile width of the halo plan: 32, or 16 (maps 16 pixels wide)
    int Ho, Wo, OH, OW, M;
    int taps, chunks_per_tap, kt_total;
};

__global__ void __launch_bounds__(256) zero_f64_kernel(double* __restrict__ p, long n) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

#ifndef RNR_WINO_SPLIT_MIN_CHUNKS
#define RNR_WINO_SPLIT_MIN_CHUNKS 4     // 16-channel chunks per split-K slice of a Winograd kernel, at least
#define RNR_WINO_SPLIT_MIN_WGS 192      // workgroups a split Winograd grid must reach (128: the 64^2 stride-2 and 16^2 transposed layers at one view lose 7 - 14 us)
#endif
#ifndef RNR_WINO2_MIN_WGS
#define RNR_WINO2_MIN_WGS 200        // fewer workgroups (one per CU) than this: the direct kernels
#endif
#ifndef RNR_WINO4_SPLIT_MIN_CHUNKS
#define RNR_WINO4_SPLIT_MIN_CHUNKS 4    // 16-channel chunks per split-K slice of conv_wino4_kernel, at least
#endif
#ifndef RNR_WINO4_MIN_WGS
#define RNR_WINO4_MIN_WGS 256        // fewer 32 x 16 pixel x 64 column tiles than this (one 12-wave workgroup per CU): F(2x2, 3x3)
#endif
#ifndef RNR_WINO_MIN_WGS
#define RNR_WINO_MIN_WGS 256         // fewer 16 x 8 pixel x 64 column tiles than this: the direct kernels (they split K)
#endif

// split depth of a Winograd grid of `wgs` workgroups (0: too small even when split — the direct kernels).  Below `min_wgs`
// the K loop is cut into slices whose partial outputs splitk_reduce_kernel adds: at least 4 chunks per slice, at most 8
// slices, and the split grid must reach half of `min_wgs`.  RNR_WINO_SPLITK=0 in the environment disables the split.
static int wino_splitk(long wgs, int min_wgs, int chunks) {
    if (wgs >= min_wgs) return 1;
    static const int enabled = [] { const char* e = getenv("RNR_WINO_SPLITK"); return e ? atoi(e) : 1; }();
    if (!enabled || wgs <= 0) return 0;
    int sk = (int)((min_wgs + wgs - 1) / wgs);
    const int max_sk = chunks / RNR_WINO_SPLIT_MIN_CHUNKS < 8 ? chunks / RNR_WINO_SPLIT_MIN_CHUNKS : 8;
    if (sk > max_sk) sk = max_sk;
    if (sk >= 2) {
        // no empty trailing slice: the kernels give every slice ceil(chunks / sk) chunks, so e.g. 29 chunks cut 7 ways (5 per
        // slice) would leave slice 6 starting at chunk 30 — nothing to add, but its weight look-ahead would read past the
        // column tile's image.  Keep the slice length, drop the empty slices.
        const int per = (chunks + sk - 1) / sk;
        sk = (chunks + per - 1) / per;
    }
    if (sk < 2 || wgs * sk < RNR_WINO_SPLIT_MIN_WGS) return 0;
    return sk;
}

static int make_plan(const rnr_conv_desc* d, int N, int H, int W, ConvPlan* p) {
    p->wino = 0;
    p->taps = d->kind == RNR_CONV3x3_REFLECT ? 9 : (d->kind == RNR_CONV4x4S2_REFLECT ? 16 : 4);
    p->par = d->kind == RNR_CONVT4x4S2 ? 4 : 1;
    if (d->kind == RNR_CONV3x3_REFLECT) { p->Ho = H; p->Wo = W; p->OH = H; p->OW = W; }
    else if (d->kind == RNR_CONV4x4S2_REFLECT) { p->Ho = H / 2; p->Wo = W / 2; p->OH = H / 2; p->OW = W / 2; }
    else { p->Ho = H; p->Wo = W; p->OH = 2 * H; p->OW = 2 * W; }
    p->M = N * p->Ho * p->Wo;
    p->chunks_per_tap = (d->c_in0_pad + d->c_in1_pad) / BK;
    p->kt_total = p->taps * p->chunks_per_tap;
    // cfg 1: 256 x 96 columns in the gather kernel, 256 x 80 (64 + a 16-column remainder tile) in the halo kernel
    if (d->c_out_pad <= 64) { p->cfg = 0; p->bm = 256; p->bn = 64; }
    else if (d->c_out_pad <= 80) { p->cfg = 1; p->bm = 256; p->bn = 96; }
    else { p->cfg = 2; p->bm = 128; p->bn = 128; }
    p->mtiles = (p->M + p->bm - 1) / p->bm;
    p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn;
    // halo kernels: 32 x th tiles of the GEMM row space, all tile pixels inside the map.  The 4x4-s2 convolution runs
    // on the 128-column configuration.
    if (d->kind == RNR_CONV4x4S2_REFLECT && p->cfg != 2 && p->Wo % 32 == 0 && p->Ho % 4 == 0) {
        p->cfg = 2; p->bm = 128; p->bn = 128;
        p->mtiles = (p->M + p->bm - 1) / p->bm;
        p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn;
    }
    // bf16x6 emulation: 256 x 128 tiles (32 x 8 pixels, two waves per SIMD) halve the weight traffic and barriers per MFMA
    if ((d->flags & RNR_CONV_F32_EMU_ANY) && p->cfg == 2 && d->kind != RNR_CONV4x4S2_REFLECT && p->Wo % 32 == 0 &&
        p->Ho % 8 == 0) {
        p->bm = 256;
        p->mtiles = (p->M + p->bm - 1) / p->bm;
    }
    // exact-fp32 kernels: 256 x 128 tiles (two waves per SIMD, half the weight traffic and barriers per MFMA) once there
    // are enough of them to fill the 256 CUs twice over; below that the 128 x 128 tiles keep more CUs busy
    if (!(d->flags & RNR_CONV_F32_EMU_ANY) && p->cfg == 2 && p->Wo % 32 == 0 &&
        p->Ho % 8 == 0 && (long)(p->M / 256) * p->ntiles * p->par >= RNR_NATIVE_BIG_MIN) {
        p->bm = 256;
        p->mtiles = (p->M + p->bm - 1) / p->bm;
    }
    // ... the emulated 4x4-s2 convolution runs per input parity phase (conv_halo_emu_kernel): 256-, 128- or 64-row tiles
    if ((d->flags & RNR_CONV_F32_EMU_ANY) && d->kind == RNR_CONV4x4S2_REFLECT && p->cfg == 2 && p->Wo % 32 == 0 &&
        p->Ho % 2 == 0) {
        p->bm = p->Ho % 8 == 0 ? 256 : (p->Ho % 4 == 0 ? 128 : 64);
        p->mtiles = (p->M + p->bm - 1) / p->bm;
    }
    int th = p->bm / 32;
    p->tw = 32;
    p->halo = (p->Wo % 32 == 0 && p->Ho % th == 0 && p->Ho >= th) ? 1 : 0;
    // maps 16 pixels wide: 16 x 8 pixel tiles (two image rows per 32-row MFMA block), 128 columns
    if (!p->halo && p->Wo % 32 != 0 && p->Wo % 16 == 0 && p->Ho % 8 == 0) {
        p->tw = 16; p->cfg = 2; p->bm = 128; p->bn = 128; th = 8;
        p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn;
        p->halo = 1;
    }
    // exact-fp32 kernels, small maps (the 32^2 / 16^2 layers at one view per call): when 128 x 128 tiles would have to split
    // K more than four ways to fill the chip, 64 x 64 tiles (32 x 2 or 16 x 4 pixels, one MFMA block per wave) give four
    // times as many tiles: a shallow split whose slices meet inside the launch instead of 16 - 32 slabs and a reduce kernel
    if (p->halo && !(d->flags & RNR_CONV_F32_EMU_ANY) && p->cfg == 2 && p->bm == 128) {
        const long t128 = (long)N * (p->Ho / th) * (p->Wo / p->tw) * p->ntiles * p->par;
        const int th64 = p->tw == 32 ? 2 : 4;
        // (the 4x4-s2 convolution has a barrier every 4 taps: with 8 MFMAs per tap the halo fetch of the next K step is not
        // covered any more — 64^2 -> 32^2 at one view: 95 us on 64 x 64 tiles against 88 on 128 x 128 x 16 slices — so
        // it takes the small tiles only where the alternative is a 32-way split)
        const int below = d->kind == RNR_CONV4x4S2_REFLECT ? RNR_SMALL_TILE_BELOW / 4 : RNR_SMALL_TILE_BELOW;
        if (t128 < below && p->Ho % th64 == 0) {
            p->cfg = 3; p->bm = 64; p->bn = 64; th = th64;
            p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn;
        }
    }
    // ... and mid-size maps (one view per call: 256 - 512 tiles of 128 x 128, i.e. one or two workgroups = waves per SIMD):
    // 128 x 64 tiles (32 x 4 pixels, 64 columns; 32 accumulator registers, four waves per SIMD) double the tile count
    static const int cfg4_max = [] { const char* e = getenv("RNR_CFG4_MAX"); return e ? atoi(e) : RNR_CFG4_MAX; }();
    // (3x3 only: measured at one view per call 154 -> 149 us at 128^2 and 159 -> 154 at 64^2, but the stride-2 and transposed
    // convolutions — four taps per barrier — lose 3 - 18 us with half the MFMAs per tap)
    if (p->halo && d->kind == RNR_CONV3x3_REFLECT && !(d->flags & RNR_CONV_F32_EMU_ANY) && p->cfg == 2 && p->bm == 128 && p->tw == 32) {
        const long t128 = (long)N * (p->Ho / th) * (p->Wo / p->tw) * p->ntiles * p->par;
        if (t128 <= cfg4_max && t128 >= RNR_SMALL_TILE_BELOW) {
            p->cfg = 4; p->bm = 128; p->bn = 64;
            p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn;
        }
    }
    // ... and the 64-column 3x3 layers (256 x 64 tiles, three waves per SIMD) when one 512^2 view is all there is: 152 -> 148 us
    // per layer on 128 x 64 tiles; no gain from two views on (RNR_CFG0_SMALL_MAX in the environment overrides)
    static const int cfg0_small = [] { const char* e = getenv("RNR_CFG0_SMALL_MAX"); return e ? atoi(e) : RNR_CFG0_SMALL_MAX; }();
    // (the 64-column transposed conv too: 271 -> 268 us)
    if (p->halo && d->kind != RNR_CONV4x4S2_REFLECT && !(d->flags & RNR_CONV_F32_EMU_ANY) && p->cfg == 0 && p->tw == 32 && p->Ho % 4 == 0) {
        const long t256 = (long)N * (p->Ho / th) * (p->Wo / p->tw) * p->ntiles * p->par;
        if (t256 <= cfg0_small) {
            p->cfg = 4; p->bm = 128; p->bn = 64; th = 4;
        }
    }
    // (the 80-column out layer was tried on 128 x 80 tiles at one view per call: 364 vs 351 us, not kept)
    // the halo kernels address a view with 32-bit element offsets
    const long view_elems = (long)H * W * (d->c_in0_pad > d->c_in1_pad ? d->c_in0_pad : d->c_in1_pad);
    if (view_elems >= (1L << 30)) p->halo = 0;
    if (p->halo) {
        p->mtiles = N * (p->Ho / th) * (p->Wo / p->tw);
        if (p->cfg == 1) { p->bn = 80; p->ntiles = (d->c_out_pad + p->bn - 1) / p->bn; }
    }
    const long tiles = (long)p->mtiles * p->ntiles * p->par;
    int sk = 1;
    // (experiments: RNR_SPLITK_BELOW / RNR_SPLITK_TARGET in the environment override the compiled-in thresholds)
    static const int sk_below = [] { const char* e = getenv("RNR_SPLITK_BELOW"); return e ? atoi(e) : RNR_SPLITK_BELOW; }();
    static const int sk_target = [] { const char* e = getenv("RNR_SPLITK_TARGET"); return e ? atoi(e) : RNR_SPLITK_TARGET; }();
    if (tiles < sk_below) {      // fewer workgroups than ~2-3 per CU: split K so the 256 CUs stay filled
        sk = (int)((sk_target + tiles - 1) / tiles);
        // split granularity: K-chunks x taps for the gather kernel, K-chunks (all nine taps) for the halo kernel
        const int units = p->halo ? p->chunks_per_tap : p->kt_total / 4;
        const int max_sk = units > 0 ? units : 1;
        if (sk > max_sk) sk = max_sk;
        if (sk > 64) sk = 64;
        if (sk < 1) sk = 1;
    }
    p->splitk = sk;
    // Winograd F(2x2, 3x3): every 3x3 layer whose map tiles into 16 x 8 pixels and whose columns into 64s, when there are
    // enough tiles to give every CU two (RNR_WINO_MIN_WGS in the environment overrides)
    static const int min_wgs = [] { const char* e = getenv("RNR_WINO_MIN_WGS"); return e ? atoi(e) : RNR_WINO_MIN_WGS; }();
    if ((d->flags & RNR_CONV_WINOGRAD) && d->kind == RNR_CONV3x3_REFLECT && H % WINO_PH == 0 && W % WINO_PW == 0 &&
        d->c_out_pad % WINO_BN == 0 && view_elems < (1L << 30)) {
        const long wgs = (long)N * (H / WINO_PH) * (W / WINO_PW) * (d->c_out_pad / WINO_BN);
        const int sk = wino_splitk(wgs, min_wgs, p->chunks_per_tap);
        if (sk > 0) {
            p->wino = 1; p->halo = 1; p->cfg = 0; p->tw = WINO_PW; p->bm = WINO_PW * WINO_PH; p->bn = WINO_BN;
            p->mtiles = N * (H / WINO_PH) * (W / WINO_PW);
            p->ntiles = d->c_out_pad / WINO_BN;
            p->splitk = sk;
        }
    }
    // F(4x4, 3x3) (opt-in, RNR_CONV_WINOGRAD4): 32 x 16 pixel tiles x 64 columns, one 12-wave workgroup per CU — when the grid
    // gives every CU a workgroup (no split-K form)
    static const int min_wgs4 = [] { const char* e = getenv("RNR_WINO4_MIN_WGS"); return e ? atoi(e) : RNR_WINO4_MIN_WGS; }();
    if ((d->flags & RNR_CONV_WINOGRAD) && (d->flags & RNR_CONV_WINOGRAD4) && d->kind == RNR_CONV3x3_REFLECT && H % W4_PH == 0 &&
        W % W4_PW == 0 && d->c_out_pad % W4_BN == 0 && view_elems < (1L << 30) &&
        (!W4_BN_LDS || d->c_in0_pad + d->c_in1_pad <= W4_BN_MAXC)) {      // the kernel's LDS table of BatchNorm scale / shift holds that many channels
        const long wgs = (long)N * (H / W4_PH) * (W / W4_PW) * (d->c_out_pad / W4_BN);
        // small grids are cut over K like the F(2x2, .) ones — slices of >= RNR_WINO4_SPLIT_MIN_CHUNKS chunks, at most 8, and the
        // split grid must give every CU its workgroup again (this kernel runs one workgroup per CU)
        int sk = wgs >= min_wgs4 ? 1 : 0;
        if (!sk && wgs > 0) {
            static const int enabled = [] { const char* e = getenv("RNR_WINO_SPLITK"); return e ? atoi(e) : 1; }();
            int want = (int)((min_wgs4 + wgs - 1) / wgs);
            const int max_sk = p->chunks_per_tap / RNR_WINO4_SPLIT_MIN_CHUNKS < 8 ? p->chunks_per_tap / RNR_WINO4_SPLIT_MIN_CHUNKS : 8;
            if (want > max_sk) want = max_sk;
            if (want >= 2) {
                const int per = (p->chunks_per_tap + want - 1) / want;
                want = (p->chunks_per_tap + per - 1) / per;         // no empty trailing slice
            }
            if (enabled && want >= 2 && wgs * want >= min_wgs4) sk = want;
        }
        if (sk) {
            p->wino = 4; p->halo = 1; p->cfg = 0; p->tw = W4_PW; p->bm = W4_PW * W4_PH; p->bn = W4_BN;
            p->mtiles = N * (H / W4_PH) * (W / W4_PW);
            p->ntiles = d->c_out_pad / W4_BN;
            p->splitk = sk;
        }
    }
    // ... and the 80-column out layer on the 16 x 16 x 4 instruction: 16 x 4 pixel tiles x all 80 columns (conv_wino80_kernel)
    if ((d->flags & RNR_CONV_WINOGRAD) && d->kind == RNR_CONV3x3_REFLECT && d->c_out_pad == 80 && H % W80_PH == 0 &&
        W % W80_PW == 0 && view_elems < (1L << 30)) {
        const long wgs = (long)N * (H / W80_PH) * (W / W80_PW);
        if (wgs >= min_wgs) {
            p->wino = 3; p->halo = 1; p->cfg = 1; p->tw = W80_PW; p->bm = W80_PW * W80_PH; p->bn = 80;
            p->mtiles = N * (H / W80_PH) * (W / W80_PW);
            p->ntiles = 1;
            p->splitk = 1;
        }
    }
    // Winograd F(2x2, 2x2) for the 4x4 stride-2 convolutions: 16 x 16 output pixels x 128 columns (convolution) or the four
    // parity classes of 16 x 8 input pixels x 64 columns (transposed) per workgroup
    if ((d->flags & RNR_CONV_WINOGRAD) && d->kind != RNR_CONV3x3_REFLECT && p->Wo % WINO_PW == 0 &&
        p->Ho % (d->kind == RNR_CONVT4x4S2 ? WINO_PH : 16) == 0 && d->c_out_pad % (d->kind == RNR_CONVT4x4S2 ? 64 : 128) == 0 &&
        view_elems < (1L << 30)) {
        const int bnw = d->kind == RNR_CONVT4x4S2 ? 64 : 128, tph = d->kind == RNR_CONVT4x4S2 ? WINO_PH : 16;
        const long wgs = (long)N * (p->Ho / tph) * (p->Wo / WINO_PW) * (d->c_out_pad / bnw);
        static const int min_wgs2 = [] { const char* e = getenv("RNR_WINO2_MIN_WGS"); return e ? atoi(e) : RNR_WINO2_MIN_WGS; }();
        const int sk = wino_splitk(wgs, min_wgs2, p->chunks_per_tap);
        if (sk > 0) {
            p->wino = 2; p->halo = 1; p->cfg = 0; p->tw = WINO_PW; p->bm = WINO_PW * tph; p->bn = bnw;
            p->mtiles = N * (p->Ho / tph) * (p->Wo / WINO_PW);
            p->ntiles = d->c_out_pad / bnw;
            p->par = 1;
            p->splitk = sk;
        }
    }
    return 0;
}

template <int FMT, int KIND>
static void launch_halo_emu(const ConvPlan& pl, const dim3 grid, const ConvParams& P, hipStream_t st) {
    if (pl.tw == 16) launch_halo_emu_cfg<FMT, KIND, 2, 2, 2, 2, 16>(grid, P, st);       // 16 x 8 pixel tiles, 128 columns
    else if (KIND == 1) {        // make_plan forces the 128-column config; rows per tile by what divides the map
        if (pl.bm == 256) launch_halo_emu_cfg<FMT, 1, 2, 2, 4, 2>(grid, P, st);
        else if (pl.bm == 128) launch_halo_emu_cfg<FMT, 1, 2, 2, 2, 2>(grid, P, st);
        else launch_halo_emu_cfg<FMT, 1, 2, 2, 1, 2>(grid, P, st);
    }
    else if (pl.cfg == 0) launch_halo_emu_cfg<FMT, KIND == 1 ? 0 : KIND, 4, 1, 2, 2>(grid, P, st);         // 256 x 64
    else if (pl.cfg == 1) launch_halo_emu_cfg<FMT, KIND == 1 ? 0 : KIND, 4, 1, 2, 3>(grid, P, st);         // 256 x 96 (Cout 78)
    else if (pl.bm == 256) launch_halo_emu_cfg<FMT, KIND == 1 ? 0 : KIND, 2, 2, 4, 2>(grid, P, st);        // 256 x 128
    else launch_halo_emu_cfg<FMT, KIND == 1 ? 0 : KIND, 2, 2, 2, 2>(grid, P, st);                          // 128 x 128
}

template <int KIND>
static void launch_halo(const ConvPlan& pl, const ConvParams& P, hipStream_t st) {
    const dim3 grid((unsigned)(pl.mtiles * pl.ntiles * pl.splitk * pl.par));
    if (pl.cfg == 4) launch_halo_cfg<KIND, 4, 1, 1, 2, 0>(grid, P, st);                         // 32 x 4 pixel tiles, 64 columns
    else if (pl.cfg == 3 && pl.tw == 16) launch_halo_cfg<KIND, 2, 2, 1, 1, 0, 16>(grid, P, st);      // 16 x 4 pixel tiles, 64 columns
    else if (pl.cfg == 3) launch_halo_cfg<KIND, 2, 2, 1, 1, 0>(grid, P, st);                    // 32 x 2 pixel tiles, 64 columns
    else if (pl.tw == 16) launch_halo_cfg<KIND, 2, 2, 2, 2, 0, 16>(grid, P, st);       // 16 x 8 pixel tiles, 128 columns
    else if (pl.cfg == 0) launch_halo_cfg<KIND, 4, 1, 2, 2, 0>(grid, P, st);
    else if (pl.cfg == 1) launch_halo_cfg<KIND, 4, 1, 2, 2, 1>(grid, P, st);      // 256 x 80
    else if (pl.bm == 256) launch_halo_cfg<KIND, 2, 2, 4, 2, 0>(grid, P, st);
    else launch_halo_cfg<KIND, 2, 2, 2, 2, 0>(grid, P, st);
}

template <int KIND>
static void launch_kind(const ConvPlan& pl, const ConvParams& P, hipStream_t st) {
    const dim3 grid((unsigned)(pl.mtiles * pl.ntiles * pl.splitk * pl.par));
    if (pl.cfg == 0) hipLaunchKernelGGL((conv_mfma_kernel<KIND, 4, 1, 2, 2>), grid, dim3(CTHREADS), 0, st, P);
    else if (pl.cfg == 1) hipLaunchKernelGGL((conv_mfma_kernel<KIND, 4, 1, 2, 3>), grid, dim3(CTHREADS), 0, st, P);
    else hipLaunchKernelGGL((conv_mfma_kernel<KIND, 2, 2, 2, 2>), grid, dim3(CTHREADS), 0, st, P);
}

static int check_desc(const rnr_conv_desc* d, const char* who) {
    RNR_REQUIRE(d, "%s: null descriptor", who);
    RNR_REQUIRE(d->kind >= 0 && d->kind <= 2, "%s: unknown kind %d", who, d->kind);
    RNR_REQUIRE(d->c_in0 > 0 && d->c_in0_pad >= d->c_in0 && d->c_in0_pad % BK == 0,
                "%s: c_in0 %d / pad %d (pad must be a multiple of %d)", who, d->c_in0, d->c_in0_pad, BK);
    RNR_REQUIRE(d->c_in1 >= 0 && d->c_in1_pad >= d->c_in1 && d->c_in1_pad % BK == 0,
                "%s: c_in1 %d / pad %d", who, d->c_in1, d->c_in1_pad);
    RNR_REQUIRE(d->c_out > 0 && d->c_out_pad >= d->c_out && d->c_out_pad % BK == 0,
                "%s: c_out %d / pad %d", who, d->c_out, d->c_out_pad);
    RNR_REQUIRE((d->flags & ~(RNR_CONV_STATS_PREZEROED | RNR_CONV_F32_EMU_ANY | RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4)) == 0,
                "%s: unknown flags 0x%x", who, d->flags);
    RNR_REQUIRE(!(d->flags & RNR_CONV_WINOGRAD4) || (d->flags & RNR_CONV_WINOGRAD),
                "%s: RNR_CONV_WINOGRAD4 goes with RNR_CONV_WINOGRAD (its fallback for the shapes it does not cover)", who);
    RNR_REQUIRE((d->flags & RNR_CONV_F32_EMU_ANY) != RNR_CONV_F32_EMU_ANY, "%s: choose ONE emulation format", who);
    RNR_REQUIRE(!(d->flags & RNR_CONV_WINOGRAD) || !(d->flags & RNR_CONV_F32_EMU_ANY),
                "%s: RNR_CONV_WINOGRAD is an exact-fp32 algorithm, not combined with the emulation formats", who);
    return 0;
}

}  // namespace rnr

using namespace rnr;

static size_t wino_weight_floats(const rnr_conv_desc* d) {       // 0: this convolution has no Winograd image
    if (!(d->flags & RNR_CONV_WINOGRAD)) return 0;
    const size_t npairs = (size_t)(d->c_in0_pad + d->c_in1_pad) / 2;       // K steps per tap set
    if (d->kind == RNR_CONV3x3_REFLECT && d->c_out_pad == 80) return (npairs / 2 + W80_BDIST) * W80_STEP_FLOATS;
    if (d->kind == RNR_CONV3x3_REFLECT)
        return d->c_out_pad % WINO_BN ? 0 : (size_t)(d->c_out_pad / WINO_BN) * (npairs + WINO_BDIST) * WINO_STEP_FLOATS;
    // pair layout (conv_wino2p.inc): npairs K steps = npairs / 2 pair-steps per phase, + the look-ahead padding
    if (d->kind == RNR_CONVT4x4S2) {
        if (d->c_out_pad % 64) return 0;
        return W2_PAIRS_KIND(RNR_CONVT4x4S2) ? (size_t)(d->c_out_pad / 64) * (npairs / 2 + W2P_PAD_PAIRS) * w2p_pair_floats<2>()
                                             : (size_t)(d->c_out_pad / 64) * (npairs + W2_BDIST) * w2_step_floats<2>();
    }
    if (d->c_out_pad % 128) return 0;
    return W2_PAIRS_KIND(RNR_CONV4x4S2_REFLECT) ? (size_t)(d->c_out_pad / 128) * (4 * (npairs / 2) + W2P_PAD_PAIRS) * w2p_pair_floats<1>()
                                                : (size_t)(d->c_out_pad / 128) * (4 * npairs + W2_BDIST) * w2_step_floats<1>();     // four phases
}
static size_t wino4_weight_floats(const rnr_conv_desc* d) {      // 0: this convolution has no F(4x4, 3x3) image
    if (!(d->flags & RNR_CONV_WINOGRAD) || !(d->flags & RNR_CONV_WINOGRAD4) || d->kind != RNR_CONV3x3_REFLECT ||
        d->c_out_pad % W4_BN || d->c_out_pad == 80)
        return 0;
    const size_t npairs = (size_t)(d->c_in0_pad + d->c_in1_pad) / 2;
    return (size_t)(d->c_out_pad / W4_BN) * (npairs + W4_BDIST) * W4_STEP_FLOATS;
}
static size_t packed_f32_floats(const rnr_conv_desc* d) {
    const size_t taps = d->kind == RNR_CONV3x3_REFLECT ? 9 : 16;          // 16 = 4x4 taps, or 4 parity classes x 4 taps
    return taps * (size_t)(d->c_in0_pad + d->c_in1_pad) * (size_t)weight_row_stride(d->c_out_pad);
}

extern "C" size_t rnr_packed_weight_floats(const rnr_conv_desc* d) {
    if (!d) return 0;
    const size_t f32 = packed_f32_floats(d);
    // emulation image behind the fp32 image: 64-byte header + 3 bf16 terms (6 bytes) or 2 fp16 terms (4 bytes) per weight
    if (d->flags & RNR_CONV_F32_EMU_BF16X6) return f32 + EMU_HEADER_BYTES / 4 + (f32 * 6 + 3) / 4;
    if (d->flags & RNR_CONV_F32_EMU_F16X3) return f32 + EMU_HEADER_BYTES / 4 + f32;
    // Winograd image behind the fp32 image: 16 planes instead of 9 taps (and, with RNR_CONV_WINOGRAD4, the 36-plane image behind it)
    return f32 + wino_weight_floats(d) + wino4_weight_floats(d);
}

extern "C" int rnr_pack_conv_weight(const rnr_conv_desc* d, const float* weight, float* packed, void* stream) {
    if (int e = check_desc(d, "rnr_pack_conv_weight")) return e;
    RNR_REQUIRE(weight && packed, "rnr_pack_conv_weight: null pointer argument");
    const long total = (long)packed_f32_floats(d);
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, as_stream(stream),
                       *d, weight, packed, total);
    if (int e = check_launch("pack_weight_kernel")) return e;
    if (d->flags & RNR_CONV_F32_EMU_ANY) {
        char* image = reinterpret_cast<char*>(packed + total);
        static_assert(EMU_HEADER_BYTES % sizeof(double) == 0, "header cleared as doubles");
        hipLaunchKernelGGL(zero_f64_kernel, dim3(1), dim3(256), 0, as_stream(stream), reinterpret_cast<double*>(image),
                           (long)(EMU_HEADER_BYTES / sizeof(double)));
        const dim3 grid((unsigned)((total + 255) / 256));
        if (d->flags & RNR_CONV_F32_EMU_F16X3) {
            const long nw = (long)(d->c_in0 + d->c_in1) * d->c_out * (d->kind == RNR_CONV3x3_REFLECT ? 9 : 16);
            hipLaunchKernelGGL(weight_amax_kernel, dim3((unsigned)std::min<long>((nw + 255) / 256, 1024)), dim3(256), 0,
                               as_stream(stream), weight, nw, reinterpret_cast<unsigned*>(image) + 2);
            hipLaunchKernelGGL(pack_weight_emu_kernel<1>, grid, dim3(256), 0, as_stream(stream), *d, weight, image, total);
        } else {
            hipLaunchKernelGGL(pack_weight_emu_kernel<0>, grid, dim3(256), 0, as_stream(stream), *d, weight, image, total);
        }
        return check_launch("pack_weight_emu_kernel");
    }
    if (wino_weight_floats(d)) {
        const long nw = (long)wino_weight_floats(d);
        if (d->kind == RNR_CONV3x3_REFLECT && d->c_out_pad == 80)
            hipLaunchKernelGGL(pack_weight_wino80_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, as_stream(stream), *d,
                               weight, packed + total, nw);
        else if (d->kind == RNR_CONV3x3_REFLECT)
            hipLaunchKernelGGL(pack_weight_wino_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, as_stream(stream), *d,
                               weight, packed + total, nw);
        else
            hipLaunchKernelGGL(W2_PAIRS_KIND(d->kind) ? pack_weight_wino2p_kernel : pack_weight_wino2_kernel,
                               dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, as_stream(stream), *d, weight, packed + total, nw);
        if (int e = check_launch("pack_weight_wino_kernel")) return e;
        if (const long nw4 = (long)wino4_weight_floats(d)) {
            hipLaunchKernelGGL(pack_weight_wino4_kernel, dim3((unsigned)((nw4 + 255) / 256)), dim3(256), 0, as_stream(stream), *d,
                               weight, packed + total + nw, nw4);
            return check_launch("pack_weight_wino4_kernel");
        }
        return 0;
    }
    return 0;
}

// split-K slices of a tile can meet inside the launch (splitk_combine) when there are few of them: the last slice reads
// splitk accumulator images, a serial tail that a chip-wide reduce kernel beats for deep splits
// OFF since r06 (RNR_CONV_COMBINE=1 in the environment turns it on): re-measured at one view per call, slabs + reduce kernel
// beat the in-launch form on every layer that took it (L11 46.9 -> 33.2 us, L12 31.4 -> 29.8, frame 449.0 -> 452.2 frames/s;
// equal from three views on) — the last slices of all tiles fetch their images past the L2 at the same moment behind two fabric
// round trips, a chip-wide reduce reads the same bytes at 4 - 5 TB/s (profiles/r06_one_view_frontend.txt (3), (5)).
static bool combines_in_launch(const ConvPlan& pl) {
    static const int enabled = [] { const char* e = getenv("RNR_CONV_COMBINE"); return e ? atoi(e) : 0; }();
    return enabled && pl.halo && !pl.wino && pl.splitk > 1 && pl.cfg != 1 && pl.bm * pl.bn <= 128 * 128 &&
           (long)pl.splitk * pl.bm * pl.bn * (long)sizeof(float) <= RNR_COMBINE_MAX_BYTES;
}
static size_t combine_slab_floats(const ConvPlan& pl) {     // one accumulator image per (slice, tile): bm x bn floats
    return (size_t)pl.splitk * pl.par * pl.mtiles * pl.ntiles * pl.bm * pl.bn;
}

extern "C" size_t rnr_conv_workspace_bytes(const rnr_conv_desc* d, int num_views, int in_h, int in_w) {
    if (!d || num_views <= 0) return 0;
    ConvPlan pl;
    make_plan(d, num_views, in_h, in_w, &pl);
    if (pl.splitk <= 1) return 256;
    const size_t legacy = (size_t)pl.splitk * num_views * pl.OH * pl.OW * d->c_out_pad * sizeof(float) + 256;
    const size_t fused = combines_in_launch(pl) ? combine_slab_floats(pl) * sizeof(float) + 256 : 0;
    return legacy > fused ? legacy : fused;
}

// rnr_conv2d_fused's sync buffer for one call: [arrival counters | tile counters | statistics shards]
struct SyncLayout { size_t arrive, tiles, stats, total; };
static SyncLayout sync_layout(const rnr_conv_desc* d, const ConvPlan& pl, int num_views) {
    SyncLayout L;
    L.arrive = 0;
    L.tiles = align_up(sizeof(unsigned) * (size_t)(num_views + 1), 256);
    L.stats = L.tiles + align_up(sizeof(unsigned) * (size_t)pl.par * pl.mtiles * pl.ntiles, 256);
    L.total = L.stats + sizeof(double) * 2 * (size_t)STAT_SHARDS * num_views * d->c_out_pad;
    return L;
}

extern "C" size_t rnr_conv_sync_bytes(const rnr_conv_desc* d, int max_views, int in_h, int in_w) {
    if (!d || max_views <= 0) return 0;
    size_t need = 0;
    for (int n = 1; n <= max_views; n++) {      // the plan (tile shape, split depth) depends on the number of views
        ConvPlan pl;
        make_plan(d, n, in_h, in_w, &pl);
        const size_t t = sync_layout(d, pl, n).total;
        need = t > need ? t : need;
    }
    return need;
}

extern "C" int rnr_conv_algorithm(const rnr_conv_desc* d, int num_views, int in_h, int in_w) {
    if (!d || num_views <= 0 || d->kind < 0 || d->kind > 2) return -1;
    ConvPlan pl;
    make_plan(d, num_views, in_h, in_w, &pl);
    return pl.wino;
}

// the plan a MASKED launch runs: the out layer's own Winograd kernel takes a mask, the other Winograd kernels do not — those
// calls run the direct kernels, on the direct kernels' tiles
static void mask_plan(const rnr_conv_desc* d, int num_views, int in_h, int in_w, ConvPlan* pl) {
    make_plan(d, num_views, in_h, in_w, pl);
    if (pl->wino && pl->wino != 3) {
        rnr_conv_desc dd = *d;
        dd.flags &= ~(RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4);
        make_plan(&dd, num_views, in_h, in_w, pl);
    }
}

extern "C" size_t rnr_conv_tile_count(const rnr_conv_desc* d, int num_views, int in_h, int in_w) {
    if (!d || num_views <= 0 || d->kind != RNR_CONV3x3_REFLECT) return 0;
    ConvPlan pl;
    mask_plan(d, num_views, in_h, in_w, &pl);
    return (pl.halo && (pl.wino == 3 || (!pl.wino && pl.tw == 32)) && pl.splitk == 1) ? (size_t)pl.mtiles : 0;
}

extern "C" int rnr_conv_active_tiles(const rnr_conv_desc* d, const float* alpha, uint8_t* tile_mask, int num_views,
                                     int in_h, int in_w, void* stream) {
    if (int e = check_desc(d, "rnr_conv_active_tiles")) return e;
    RNR_REQUIRE(alpha && tile_mask, "rnr_conv_active_tiles: null pointer argument");
    RNR_REQUIRE(rnr_conv_tile_count(d, num_views, in_h, in_w) > 0,
                "rnr_conv_active_tiles: this convolution does not run on maskable pixel tiles (3x3 halo plan without split-K)");
    ConvPlan pl;
    mask_plan(d, num_views, in_h, in_w, &pl);
    hipLaunchKernelGGL(active_tile_kernel, dim3((unsigned)pl.mtiles), dim3(256), 0, as_stream(stream), alpha, tile_mask,
                       in_h, in_w, pl.bm / pl.tw, pl.tw);
    return check_launch("active_tile_kernel");
}

struct RayEpilogue { const float* w; const float* bias; float* image; };
static thread_local RayEpilogue g_ray = {nullptr, nullptr, nullptr};      // set by rnr_conv2d_ray around conv2d_run

static int conv2d_run(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1, const float* weight_packed,
                      float* out_raw, double* stats, const rnr_conv_bn* bn, void* sync, size_t sync_bytes, int num_views,
                      int in_h, int in_w, void* workspace, size_t workspace_bytes, const uint8_t* tile_mask, void* stream) {
    if (int e = check_desc(d, "rnr_conv2d")) return e;
    RNR_REQUIRE(src0 && src0->data && weight_packed && out_raw, "rnr_conv2d: null pointer argument");
    RNR_REQUIRE(src0->channels == d->c_in0_pad, "rnr_conv2d: src0 has %d channels, descriptor says %d",
                src0->channels, d->c_in0_pad);
    RNR_REQUIRE(d->c_in1_pad == 0 || (src1 && src1->data && src1->channels == d->c_in1_pad),
                "rnr_conv2d: second source missing or channel mismatch");
    const int min_hw = d->kind == RNR_CONVT4x4S2 ? 1 : 2;   // ReflectionPad2d(1) needs >= 2 pixels
    RNR_REQUIRE(num_views > 0 && in_h >= min_hw && in_w >= min_hw, "rnr_conv2d: bad sizes N=%d H=%d W=%d", num_views,
                in_h, in_w);
    RNR_REQUIRE(d->kind != RNR_CONV4x4S2_REFLECT || (in_h % 2 == 0 && in_w % 2 == 0),
                "rnr_conv2d: stride-2 conv needs even input size");
    const bool fused = sync != nullptr;          // rnr_conv2d_fused
    const bool with_bn = fused && bn && bn->gamma;
    if (with_bn) RNR_REQUIRE(bn->beta && bn->scale && bn->shift, "rnr_conv2d_fused: null BatchNorm pointer");
    if (with_bn) RNR_REQUIRE(!(bn->running_mean || bn->running_var) || num_views == 1,
                             "rnr_conv2d_fused: running statistics are per-view here; torch pools the batch — pass them for num_views == 1 only (got %d)",
                             num_views);
    hipStream_t st = as_stream(stream);
    ConvPlan pl;
    make_plan(d, num_views, in_h, in_w, &pl);
    if (pl.wino && (g_ray.w || (tile_mask && pl.wino != 3))) {    // ray-epilogue / masked launches run on the direct kernels' tiles (the out layer's own Winograd kernel takes a mask)
        rnr_conv_desc dd = *d;
        dd.flags &= ~(RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4);
        make_plan(&dd, num_views, in_h, in_w, &pl);
    }
    if (g_ray.w) pl.splitk = 1;         // the ray-renderer epilogue needs the whole K sum in one workgroup (small maps would split)
    ConvParams P = {};
    P.src_data[0] = src0->data; P.src_scale[0] = src0->scale; P.src_shift[0] = src0->shift;
    P.src_c[0] = src0->channels; P.src_act[0] = src0->act;
    if (d->c_in1_pad) {
        P.src_data[1] = src1->data; P.src_scale[1] = src1->scale; P.src_shift[1] = src1->shift;
        P.src_c[1] = src1->channels; P.src_act[1] = src1->act;
    }
    P.weight = weight_packed; P.stats = stats; P.n_shards = 1;
    P.N = num_views; P.H = in_h; P.W = in_w; P.Ho = pl.Ho; P.Wo = pl.Wo; P.OH = pl.OH; P.OW = pl.OW; P.M = pl.M;
    P.c_out = d->c_out; P.c_out_pad = d->c_out_pad; P.wstride = weight_row_stride(d->c_out_pad);
    P.chunks0 = d->c_in0_pad / BK; P.chunks_per_tap = pl.chunks_per_tap; P.kt_total = pl.kt_total;
    P.splitk = pl.splitk;
    P.mtiles = pl.mtiles; P.ntiles = pl.ntiles; P.zdim = pl.splitk * pl.par;
    P.ray_w = g_ray.w; P.ray_bias = g_ray.bias; P.ray_image = g_ray.image;
    if (d->kind == RNR_CONVT4x4S2) {
        // Each parity class is its own workgroup and stages the same input halo.  With the class as the slowest tile index the
        // input is streamed from HBM four times (r02 PMC: 2.9x the compulsory bytes on the 64-column transposed conv); as
        // neighbours the four share one L2.  The price is four weight sets in flight per XCD instead of one, so the order is
        // chosen by which operand is bigger.  RNR_PAR_INNER=0/1 forces it (experiments).
        static const int forced = [] { const char* e = getenv("RNR_PAR_INNER"); return e ? atoi(e) : -1; }();
        const size_t w_bytes = 16 * (size_t)(d->c_in0_pad + d->c_in1_pad) * d->c_out_pad * sizeof(float);
        const size_t in_bytes = (size_t)in_h * in_w * (d->c_in0_pad + d->c_in1_pad) * sizeof(float);     // per view
        P.par_inner = forced >= 0 ? forced : (in_bytes >= w_bytes ? 1 : 0);
    }
    if (tile_mask) {
        RNR_REQUIRE(!stats && !with_bn, "rnr_conv2d_masked: skipped tiles would falsify the batch statistics (no statistics / BatchNorm with a mask)");
        RNR_REQUIRE(rnr_conv_tile_count(d, num_views, in_h, in_w) > 0,
                    "rnr_conv2d_masked: this convolution does not run on maskable pixel tiles");
        P.tile_mask = tile_mask;
    }
    const size_t out_floats = (size_t)num_views * pl.OH * pl.OW * d->c_out_pad;
    const long grid_wgs = (long)pl.mtiles * pl.ntiles * pl.splitk * pl.par;
    const bool combine = fused && combines_in_launch(pl);
    bool in_kernel_bn = false;
    if (fused) {
        const SyncLayout L = sync_layout(d, pl, num_views);
        RNR_REQUIRE(sync_bytes >= L.total, "rnr_conv2d_fused: sync buffer too small (%zu < %zu, see rnr_conv_sync_bytes)",
                    sync_bytes, L.total);
        char* sb = reinterpret_cast<char*>(sync);
        if (combine) P.tile_arrive = reinterpret_cast<unsigned*>(sb + L.tiles);
        if (with_bn) {
            P.stats = reinterpret_cast<double*>(sb + L.stats);
            P.n_shards = STAT_SHARDS;
            P.stats_shard = 2L * num_views * d->c_out_pad;
            P.gamma = bn->gamma; P.beta = bn->beta; P.scale = bn->scale; P.shift = bn->shift;
            P.eps = bn->eps; P.count = (double)pl.OH * pl.OW;
            P.running_mean = bn->running_mean; P.running_var = bn->running_var; P.momentum = bn->momentum;
            // tickets only where workgroups finish at different times: more than one workgroup per CU (see
            // bn_finalize_shards_kernel); the others get the finalise as a launch of its own
            // (a separate finalise launch for the big Winograd grids too was measured: -0.5 % on seven layers at 8 views — the
            // ticket is not what the short-K layers lose)
            // (conv_wino4_kernel, one 12-wave workgroup per CU: tickets vs a separate finalise launch measured equal, r04)
            in_kernel_bn = (pl.splitk == 1 || combine) && grid_wgs > RNR_FUSED_BN_MIN_WGS;
            if (in_kernel_bn) {
                P.arrive = reinterpret_cast<unsigned*>(sb + L.arrive);
                if (pl.halo) {      // the halo kernels' tiles lie inside one view
                    P.arrive_per_view = 1;
                    P.n_arrive = (unsigned)((long)pl.mtiles * pl.ntiles * pl.par / num_views);
                } else {
                    P.arrive_per_view = 0;
                    P.n_arrive = (unsigned)grid_wgs;
                }
            }
        }
    }
    if (stats && !(d->flags & RNR_CONV_STATS_PREZEROED)) {
        // (a kernel, not hipMemsetAsync: memset nodes of a captured HIP graph went stale on replay, raster.hip)
        const long n = (long)num_views * d->c_out_pad * 2;
        hipLaunchKernelGGL(zero_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, stats, n);
    }
    if (combine) {
        RNR_REQUIRE(workspace && workspace_bytes >= combine_slab_floats(pl) * sizeof(float),
                    "rnr_conv2d: workspace too small (%zu < %zu)", workspace_bytes, combine_slab_floats(pl) * sizeof(float));
        P.slabs = reinterpret_cast<float*>(workspace);
        P.out = out_raw;
        P.slab_stride = 0;
    } else if (pl.splitk > 1) {
        RNR_REQUIRE(workspace && workspace_bytes >= (size_t)pl.splitk * out_floats * sizeof(float),
                    "rnr_conv2d: workspace too small (%zu < %zu)", workspace_bytes,
                    (size_t)pl.splitk * out_floats * sizeof(float));
        P.out = reinterpret_cast<float*>(workspace);
        P.slab_stride = (long)out_floats;
    } else {
        P.out = out_raw;
        P.slab_stride = 0;
    }
    // fp32 emulation on the 16-bit matrix cores: every convolution on the halo plan
    const bool emu = (d->flags & RNR_CONV_F32_EMU_ANY) && pl.halo && (d->kind != RNR_CONV4x4S2_REFLECT || pl.cfg == 2);
    if (emu) {
        P.weight_emu = weight_packed + packed_f32_floats(d);
        const dim3 grid((unsigned)grid_wgs);
        const bool f16 = (d->flags & RNR_CONV_F32_EMU_F16X3) != 0;
        if (d->kind == RNR_CONV4x4S2_REFLECT) { if (f16) launch_halo_emu<1, 1>(pl, grid, P, st); else launch_halo_emu<0, 1>(pl, grid, P, st); }
        else if (d->kind == RNR_CONV3x3_REFLECT) { if (f16) launch_halo_emu<1, 0>(pl, grid, P, st); else launch_halo_emu<0, 0>(pl, grid, P, st); }
        else { if (f16) launch_halo_emu<1, 2>(pl, grid, P, st); else launch_halo_emu<0, 2>(pl, grid, P, st); }
    }
    else if (pl.wino) {
        // the Winograd kernels run their last chunk unconditionally and give every slice ceil(chunks / splitk) chunks: an empty
        // trailing slice would stage the wrong range and look ahead past the weight image (make_plan never produces one)
        RNR_REQUIRE(pl.splitk <= 1 || (long)(pl.splitk - 1) * ((pl.chunks_per_tap + pl.splitk - 1) / pl.splitk) < pl.chunks_per_tap,
                    "rnr_conv2d: split-K plan with an empty slice (%d chunks cut %d ways)", pl.chunks_per_tap, pl.splitk);
        P.weight_wino = weight_packed + packed_f32_floats(d);
        P.par_inner = 0;
        if (pl.wino == 4) {
            P.weight_wino = weight_packed + packed_f32_floats(d) + wino_weight_floats(d);
            launch_wino4(dim3((unsigned)grid_wgs), P, st);
        }
        else if (pl.wino == 1) launch_wino(dim3((unsigned)grid_wgs), P, st);
        else if (pl.wino == 3) launch_wino80(dim3((unsigned)grid_wgs), P, st);
        // (preprocessor, not `if`: only the kernel that runs is instantiated into the library)
        else if (d->kind == RNR_CONV4x4S2_REFLECT) {
#if (W2_PAIRS) & 1
            launch_wino2p<1>(dim3((unsigned)grid_wgs), P, st);
#else
            launch_wino2<1>(dim3((unsigned)grid_wgs), P, st);
#endif
        } else {
#if (W2_PAIRS) & 2
            launch_wino2p<2>(dim3((unsigned)grid_wgs), P, st);
#else
            launch_wino2<2>(dim3((unsigned)grid_wgs), P, st);
#endif
        }
    }
    else if (pl.halo && d->kind == RNR_CONV3x3_REFLECT) launch_halo<0>(pl, P, st);
    else if (pl.halo && d->kind == RNR_CONV4x4S2_REFLECT) launch_halo<1>(pl, P, st);
    else if (pl.halo) launch_halo<2>(pl, P, st);
    else if (d->kind == RNR_CONV3x3_REFLECT) launch_kind<0>(pl, P, st);
    else if (d->kind == RNR_CONV4x4S2_REFLECT) launch_kind<1>(pl, P, st);
    else launch_kind<2>(pl, P, st);
    if (int e = check_launch("conv_mfma_kernel")) return e;
    if (pl.splitk > 1 && !combine) {
        const long rows = (long)num_views * pl.OH * pl.OW;
        const int col_tiles = (d->c_out_pad + 63) / 64, rows_per_view = pl.OH * pl.OW;
        int rpw = 128;          // rows per workgroup: as many as leave >= 512 workgroups and stay inside one view
        while (rpw > 16 && (rows_per_view % rpw != 0 || (rows / rpw) * col_tiles < 512)) rpw >>= 1;
        const dim3 grid((unsigned)((rows + rpw - 1) / rpw), (unsigned)col_tiles);
        hipLaunchKernelGGL(splitk_reduce_kernel, grid, dim3(256), 0, st, reinterpret_cast<const float*>(workspace),
                           (long)out_floats, pl.splitk, out_raw, rows, rows_per_view, rpw, P);
        if (int e = check_launch("splitk_reduce_kernel")) return e;
    }
    if (with_bn && !in_kernel_bn) {
        hipLaunchKernelGGL(bn_finalize_shards_kernel, dim3((unsigned)num_views, (unsigned)((d->c_out_pad + CTHREADS - 1) / CTHREADS)),
                           dim3(CTHREADS), 0, st, P);
        if (int e = check_launch("bn_finalize_shards_kernel")) return e;
    }
    return 0;
}

extern "C" int rnr_conv2d(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                          const float* weight_packed, float* out_raw, double* stats, int num_views, int in_h,
                          int in_w, void* workspace, size_t workspace_bytes, void* stream) {
    return conv2d_run(d, src0, src1, weight_packed, out_raw, stats, nullptr, nullptr, 0, num_views, in_h, in_w, workspace,
                      workspace_bytes, nullptr, stream);
}

extern "C" int rnr_conv2d_masked(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                                 const float* weight_packed, float* out_raw, double* stats, int num_views, int in_h,
                                 int in_w, void* workspace, size_t workspace_bytes, const uint8_t* tile_mask,
                                 void* stream) {
    return conv2d_run(d, src0, src1, weight_packed, out_raw, stats, nullptr, nullptr, 0, num_views, in_h, in_w, workspace,
                      workspace_bytes, tile_mask, stream);
}

extern "C" int rnr_conv2d_ray(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                              const float* weight_packed, const float* ray_w, const float* bias, float* image, int num_views,
                              int in_h, int in_w, const uint8_t* tile_mask, void* stream) {
    RNR_REQUIRE(ray_w && bias && image, "rnr_conv2d_ray: null pointer argument");
    if (int e = check_desc(d, "rnr_conv2d_ray")) return e;
    ConvPlan pl;
    rnr_conv_desc dd = *d;
    dd.flags &= ~(RNR_CONV_WINOGRAD | RNR_CONV_WINOGRAD4);         // the ray-renderer epilogue lives in the direct 80-column kernel
    make_plan(&dd, num_views, in_h, in_w, &pl);
    RNR_REQUIRE(d->kind == RNR_CONV3x3_REFLECT && !(d->flags & RNR_CONV_F32_EMU_ANY) && pl.halo && pl.cfg == 1 && pl.tw == 32 &&
                    d->c_out % 3 == 0 && d->c_out_pad == 80,
                "rnr_conv2d_ray: only the exact-fp32 3x3 out layer on the 80-column plan (65 <= c_out <= 80, c_out = 3 x rays, map "
                "width a multiple of 32, height of 8) has the ray-renderer epilogue; run rnr_conv2d_masked + rnr_ray_render otherwise");
    g_ray = {ray_w, bias, image};
    // the stripped descriptor goes down: the plan, the tile count and therefore the layout tile_mask is read in are those of
    // the direct 32 x 8-pixel tiles whatever flags the caller's descriptor carries
    const int rc = conv2d_run(&dd, src0, src1, weight_packed, image /* never written as out_raw */, nullptr, nullptr, nullptr, 0,
                              num_views, in_h, in_w, nullptr, 0, tile_mask, stream);
    g_ray = {nullptr, nullptr, nullptr};
    return rc;
}

extern "C" int rnr_conv2d_fused(const rnr_conv_desc* d, const rnr_conv_src* src0, const rnr_conv_src* src1,
                                const float* weight_packed, float* out_raw, const rnr_conv_bn* bn, int num_views, int in_h,
                                int in_w, void* workspace, size_t workspace_bytes, void* sync, size_t sync_bytes,
                                const uint8_t* tile_mask, void* stream) {
    RNR_REQUIRE(sync, "rnr_conv2d_fused: null sync buffer");
    return conv2d_run(d, src0, src1, weight_packed, out_raw, nullptr, bn, sync, sync_bytes, num_views, in_h, in_w, workspace,
                      workspace_bytes, tile_mask, stream);
}

static int bn_finalize_impl(double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                            int num_views, int channels, int c_pad, double count, float eps, bool reset, void* stream) {
    RNR_REQUIRE(stats && gamma && beta && scale && shift, "rnr_bn_finalize: null pointer argument");
    RNR_REQUIRE(num_views > 0 && channels > 0 && c_pad >= channels && count > 0, "rnr_bn_finalize: bad sizes");
    const int total = num_views * c_pad;
    if (reset)
        hipLaunchKernelGGL(bn_finalize_kernel<true>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), stats, gamma,
                           beta, scale, shift, num_views, channels, c_pad, count, eps);
    else
        hipLaunchKernelGGL(bn_finalize_kernel<false>, dim3((total + 255) / 256), dim3(256), 0, as_stream(stream), stats, gamma,
                           beta, scale, shift, num_views, channels, c_pad, count, eps);
    return check_launch("bn_finalize_kernel");
}

extern "C" int rnr_bn_finalize(const double* stats, const float* gamma, const float* beta, float* scale,
                               float* shift, int num_views, int channels, int c_pad, double count, float eps,
                               void* stream) {
    return bn_finalize_impl(const_cast<double*>(stats), gamma, beta, scale, shift, num_views, channels, c_pad, count, eps,
                            false, stream);
}

extern "C" int rnr_bn_finalize_reset(double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                                     int num_views, int channels, int c_pad, double count, float eps, void* stream) {
    return bn_finalize_impl(stats, gamma, beta, scale, shift, num_views, channels, c_pad, count, eps, true, stream);
}

extern "C" int rnr_bn_finalize_batch(double* stats, const float* gamma, const float* beta, float* scale, float* shift,
                                     float* running_mean, float* running_var, float momentum, int num_views,
                                     int channels, int c_pad, double count_per_view, float eps, void* stream) {
    RNR_REQUIRE(stats && gamma && beta && scale && shift, "rnr_bn_finalize_batch: null pointer argument");
    RNR_REQUIRE(num_views > 0 && channels > 0 && c_pad >= channels && count_per_view > 0, "rnr_bn_finalize_batch: bad sizes");
    hipLaunchKernelGGL(bn_finalize_batch_kernel, dim3((c_pad + 255) / 256), dim3(256), 0, as_stream(stream), stats, gamma,
                       beta, scale, shift, running_mean, running_var, momentum, num_views, channels, c_pad, count_per_view,
                       eps);
    return check_launch("bn_finalize_batch_kernel");
}
