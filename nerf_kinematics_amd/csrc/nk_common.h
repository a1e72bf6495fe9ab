// Shared declarations of the gfx950 NeRF render path: error reporting, the counter-based RNG,
// MFMA operand types and the network plan (how an nk_arch maps onto fragment streams).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/nerf_kinematics.h"

#define NK_HD __host__ __device__ __forceinline__
#define NK_D __device__ __forceinline__

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define NK_LDS __attribute__((address_space(3)))
#define NK_GLB __attribute__((address_space(1)))

// ---- error reporting (host) ---------------------------------------------------------------
void nk_set_error(const char* fmt, ...);
#define NK_FAIL(code, ...) do { nk_set_error(__VA_ARGS__); return (code); } while (0)
#define NK_CHECK_HIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    nk_set_error("%s failed: %s", #expr, hipGetErrorString(e_)); return NK_ERR_HIP; } } while (0)
#define NK_CHECK_LAUNCH(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { \
    nk_set_error("launch of %s failed: %s", name, hipGetErrorString(e_)); return NK_ERR_HIP; } } while (0)

// ---- counter-based RNG: 32 bits for element idx of a stream under a seed --------------------
enum { NK_STREAM_STRATIFIED = 0, NK_STREAM_PDF = 1, NK_STREAM_NOISE_COARSE = 2,
       NK_STREAM_NOISE_FINE = 3, NK_STREAM_PIXELS = 4 };

NK_HD uint32_t nk_lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
NK_HD uint32_t nk_rand_u32(uint32_t seed, uint32_t stream, uint64_t idx) {
    const uint32_t key = nk_lowbias32(seed ^ (stream * 0x85EBCA6Bu));
    const uint32_t lo = (uint32_t)idx, hi = (uint32_t)(idx >> 32);
    uint32_t x = lo * 0x9E3779B9u + key;
    return nk_lowbias32(x ^ (hi * 0xC2B2AE35u));
}
NK_HD float nk_rand_uniform(uint32_t seed, uint32_t stream, uint64_t idx) {
    return (float)(nk_rand_u32(seed, stream, idx) >> 8) * (1.0f / 16777216.0f);
}
NK_D float nk_rand_normal(uint32_t seed, uint32_t stream, uint64_t idx) {
    const uint32_t a = nk_rand_u32(seed, stream, 2 * idx), b = nk_rand_u32(seed, stream, 2 * idx + 1);
    const float u1 = ((float)(a >> 8) + 1.0f) * (1.0f / 16777216.0f);
    const float u2 = (float)(b >> 8) * (1.0f / 16777216.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// torch.linspace(0, 1, steps)[i] in float32: forward from 0 for the first half, backward from 1
// for the second (the symmetric form torch uses), so coarse depths match the reference bitwise.
NK_HD float nk_linspace01(int i, int steps) {
    if (steps <= 1) return 0.0f;
    const float step = 1.0f / (float)(steps - 1);
    return (i < steps / 2) ? step * (float)i : 1.0f - step * (float)(steps - 1 - i);
}

// ---- positional-encoding slot maps --------------------------------------------------------
// The MLP kernels hold gamma(x) as 64 k-slots and gamma(d) as 32 k-slots of MFMA B operands.
// Slot sigma = 16*s + 8*h + j is element j of lane-half h in k-step s.  These give the column of
// the reference encoding [x, sin(2^0 x), cos(2^0 x), ...] a slot carries, or -1 for padding.
NK_HD int nk_pe_xyz_col(int q /* 8*s + j */, int h, int L) {
    if (q < 30) { const int l = 5 * h + q / 6; return l < L ? 3 + 6 * l + q % 6 : -1; }
    if (q == 30) return h == 0 ? 0 : 2;
    return h == 0 ? 1 : -1;
}
NK_HD int nk_pe_dir_col(int q /* 8*s + j */, int h, int L) {
    if (q < 12) { const int l = 2 * h + q / 6; return l < L ? 3 + 6 * l + q % 6 : -1; }
    if (q == 12) return h == 0 ? 0 : 2;
    if (q == 13) return h == 0 ? 1 : -1;
    return -1;
}

// Column of an encoding WITHOUT the raw input (NK_ENC_NO_INPUT_*) for a column of the full one: the
// three raw coordinates have no column, every sin / cos moves down by three.
NK_HD int nk_enc_col(int col, int with_input) { return (with_input || col < 0) ? col : (col < 3 ? -1 : col - 3); }

// Frequency l of an encoding of L frequencies: 2^l, or element l of torch.linspace(1, 2^(L-1), L) in
// float32 (forward from the start for the first half, backward from the end for the second, as torch
// computes it; `step` and `end` come from the plan).  Slots beyond L carry no weight: any finite value.
NK_HD float nk_enc_freq(int linear, float step, float end, int L, int l, float pow2) {
    if (!linear) return pow2;
    return (l < L / 2) ? 1.0f + step * (float)l : end - step * (float)(L - 1 - l);
}

// ---- network plan ---------------------------------------------------------------------------
// Layers in forward order: trunk 0..D-1, then ALPHA, FEAT, VIEWS, RGB.  A network without view
// directions (NkPlan::no_dirs) keeps the sequence: its output layer sits in ALPHA's place (rows
// rgb_raw, sigma_raw), the forward kernels leave after that stage, and FEAT, VIEWS and RGB are
// parameterless (all-zero blocks in the forward streams, which are never consumed; absent from the
// transposed stream: the backward chain starts at FEAT's stage with dOut as its only operand).
enum { NK_SEG_ACT = 0, NK_SEG_PEX = 1, NK_SEG_PED = 2, NK_SEG_DOUT = 3 };
enum { NK_MAX_LAYERS = 12 };

struct NkLayer {
    int w_off, b_off;       // offsets of weight / bias in the flat fp32 parameter array
    int has_params;         // 0: the layer is a placeholder of a network without view directions
    int out_dim, in_dim;    // shape of the reference weight matrix [out_dim, in_dim]
    int n_tiles;            // 32-row output tiles
    int n_seg;              // forward K segments, in k-step order
    int seg_type[3], seg_steps[3], seg_col[3];
    int nk;                 // forward k-steps (16 input features each)
    int fwd_block0;         // first 1-KiB block of this layer in the forward stream;
                            // each tile takes nk weight blocks + 1 bias block
    // transposed stream (dH_in = W^T dZ): rows = the ACT input features of this layer
    int bwd_tiles;          // 32-row tiles of the ACT input (0: layer has no transposed pass)
    int bwd_nk;             // k-steps over the output features (+1 dOut step for FEAT)
    int bwd_act_col;        // first column of the ACT input inside the reference matrix
    int bwd_block0;         // first block in the transposed stream
};

struct NkPlan {
    int D, W, WD, skip, Lx, Ld, relu_first, relu_feat, lineage, no_dirs;
    int inc_x, inc_d;       // the encodings start with the raw 3-vector (include_input_*)
    int lin_x, lin_d;       // linearly spaced frequencies (log_sampling_*: False) ...
    float step_x, end_x, step_d, end_d;     // ... torch.linspace(1, end, L): its step and last element
    int NL;                 // D + 4
    NkLayer layer[NK_MAX_LAYERS];
    int fwd_blocks, bwd_blocks;   // stream lengths in 1-KiB blocks
    int n_params;
};

// Activation regions saved by the forward pass.  Two element formats (eb = bytes per element):
//   eb = 2  [G][width/8][32][8] bf16: the encodings in slot order, the layer outputs with every
//           32-feature tile in nk_tile_feature() order;
//   eb = 1  [G][width/16][32][16] OCP e4m3 (NK_ACT_FP8*): the encodings in slot order, layer outputs
//           with every 32-feature tile in nk_tile_feature8() order.
// index: 0 = gamma(x) slots (64), 1 = gamma(d) slots (32), 2+i = h_i (W), 2+D = feat (W), 3+D = g (WD)
NK_HD int nk_act_width(const NkPlan& p, int idx) {
    if (idx == 0) return 64;
    if (idx == 1) return 32;
    if (idx < 2 + p.D + 1) return p.W;
    return p.WD;
}
NK_HD int64_t nk_act_offset(const NkPlan& p, int idx, int64_t G, int eb = 2) {   // bytes
    int64_t feats = 0;
    for (int i = 0; i < idx; ++i) feats += nk_act_width(p, i);
    return feats * G * 32 * eb;      // 32 points * eb bytes
}
NK_HD int nk_act_count(const NkPlan& p) { return p.D + 4; }
// element bytes of the saved activations / of dZ under an NK_ACT_* format
NK_HD int nk_act_elem_bytes(int act_format) { return act_format == NK_ACT_BF16 ? 2 : 1; }
NK_HD int nk_dz_elem_bytes(int act_format) { return act_format == NK_ACT_FP8_BF8 ? 1 : 2; }

// Feature (row of the 32-row tile) held at position x = 8 * octet + e of a saved activation / dZ
// tile: the chain kernels store a converted accumulator fragment as it lies in registers (one
// 16-byte store per fragment), which orders a tile's features this way rather than ascending.
NK_HD int nk_tile_feature(int x) {
    const int o = x >> 3, e = x & 7;
    return 16 * (o >> 1) + 8 * (e >> 2) + 4 * (o & 1) + (e & 3);
}
// The same for the 8-bit formats: a lane stores its 16 accumulator registers of a tile as ONE
// 16-byte piece, position x = 16 * h + e holds register e of lane-half h.
NK_HD int nk_tile_feature8(int x) {
    const int h = x >> 4, e = x & 15;
    return 8 * (e >> 2) + 4 * h + (e & 3);
}

// ReLU masks, one 16-byte word per lane per layer ([G][64 lanes][4 dwords]), stored after the
// activation regions: idx 0..D-1 = h_i, D = feat, D+1 = g.  Bit 31 - (16*(t&1) + i) of dword t>>1
// is set when accumulator register i of output tile t passes its gradient (sign bit of the
// pre-activation clear, or the layer has no ReLU).
NK_HD int nk_mask_count(const NkPlan& p) { return p.D + 2; }
NK_HD int64_t nk_mask_offset(const NkPlan& p, int idx, int64_t G, int eb = 2) {
    return nk_act_offset(p, nk_act_count(p), G, eb) + (int64_t)idx * G * 1024;
}

// dZ regions written by the backward chain: 0 = dOut (32: dr,dg,db,dsigma), 1 = dZ_views (WD),
// 2 = dZ_feat (W), 3+i = dZ of trunk layer D-1-i (W)
NK_HD int nk_dz_width(const NkPlan& p, int idx) {
    if (idx == 0) return 32;
    if (idx == 1) return p.WD;
    return p.W;
}
NK_HD int64_t nk_dz_offset(const NkPlan& p, int idx, int64_t G, int eb = 2) {   // eb = 1: OCP e5m2, laid out like 8-bit activations
    int64_t feats = 0;
    for (int i = 0; i < idx; ++i) feats += nk_dz_width(p, i);
    return feats * G * 32 * eb;
}
NK_HD int nk_dz_count(const NkPlan& p) { return p.D + 3; }

// groups of 32 points, padded to whole 256-point workgroups
static inline int64_t nk_groups(int64_t P) { return ((P + 255) / 256) * 8; }

int nk_build_plan(const nk_arch* arch, NkPlan* plan);   // host; returns nk_status

static inline hipStream_t nk_stream(void* s) { return (hipStream_t)s; }

// Wavefronts per workgroup of the MLP chain kernels (4: two workgroups share a CU and run out of
// phase; 8: one workgroup per CU streams the weights once for 256 points), for the forward
// (backward = 0) or the backward chain.  NK_WAVES / NK_WAVES_FWD / NK_WAVES_BWD override.
#define NK_WAVES_BWD_DEFAULT 8
int nk_waves_per_wg(int backward);

// The two halves of nk_mlp_backward, for callers that run the chains of two networks first and
// then all weight-gradient GEMMs together (nk_render_rays_bwd on small batches).
struct NkBwdJob {
    const nk_arch* arch; const void* packed; const void* act; const float* d_raw;
    int64_t R; int32_t n; void* dz; float* grad;
    int act_format;         // NK_ACT_*: format of `act` (as the forward call saved it) and of `dz`
    float dz_scale;         // NK_ACT_FP8_BF8: power of two dZ is scaled by before it is rounded to e5m2
};
int nk_mlp_backward_chain(const NkBwdJob& job, hipStream_t stream);
int nk_mlp_backward_gemms(const NkBwdJob* jobs, int count, hipStream_t stream);
// nk_mlp_forward / nk_act_bytes / nk_dz_bytes with the saved-tensor format named
int nk_mlp_forward_fmt(const nk_arch* arch, const void* packed, const float* ray_batch, const float* z, int64_t R,
                       int32_t n, float* raw, void* act_save, int act_format, const int32_t* live_groups,
                       const int32_t* n_live, void* stream);
// Occupancy skipping (nk_render_opts.skip_empty_eps): lists in live[] the groups of 32 consecutive
// fine samples (points of z_all, [R, N]) that have a sample whose four nearest coarse samples reach
// compositing weight eps, counts them in *n_live, and writes zero rows (rgb_raw = 0, sigma_raw = 0)
// to raw_f for the others.
int nk_mark_live_groups(const float* z_coarse, const float* w_coarse, const float* z_all, int64_t R, int32_t Nc,
                        int32_t N, float eps, int32_t* live, int32_t* n_live, float* raw_f, void* stream);
int64_t nk_act_bytes_fmt(const nk_arch* arch, int64_t R, int32_t n, int act_format);
int64_t nk_dz_bytes_fmt(const nk_arch* arch, int64_t R, int32_t n, int act_format);
