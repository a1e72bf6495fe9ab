// Embedder + NeRF MLP forward for gfx950, fused into one kernel.
//
// Orientation: every layer is computed transposed, Y^T[out, points] = W[out, in] . X^T[in, points],
// on v_mfma_f32_32x32x16_bf16.  A wavefront owns 32 points (one per lane column); the 32x32 f32
// result tile keeps output features in its 16 registers and points on lanes, which is exactly
// the B-operand shape of the next layer's MFMA (it sums over features = the tile's rows).  So an
// activation never leaves registers between layers: ReLU, convert to bf16, feed the next MFMA.
// The k order a converted accumulator presents is permuted (element j of lane-half h of k-step s
// is row 16s + 8(j>>2) + 4h + (j&3)); the weights are packed once per optimiser step into 1-KiB
// A-operand fragments with the matching permutation (k_pack_weights), in the exact order the
// kernel consumes them, so the weight side is a linear stream: global -> LDS by
// global_load_lds_dwordx4 (one fragment per wave-instruction), LDS -> VGPR by ds_read_b128.
// A workgroup is 4 wavefronts = 128 points sharing one pass over the stream (two workgroups per
// CU run out of phase); the stream goes through a three-slot LDS ring, one stage per output tile
// (several tiles for the thin first layer), one barrier per stage.
#include "nk_common.h"
#include "nk_mfma.h"

// --------------------------------------------------------------------------------------------
// weight packing: flat fp32 parameters -> bf16 fragment streams
// --------------------------------------------------------------------------------------------
// forward stream element: layer li, output tile t, k-step ks, lane (r = lane&31, h = lane>>5), j
__device__ float fwd_weight_elem(const NkPlan& p, const float* __restrict__ prm, int li, int t, int ks, int r, int h, int j) {
    const NkLayer& l = p.layer[li];
    const int row = 32 * t + r;
    if (row >= l.out_dim || !l.has_params) return 0.0f;
    int s = ks, seg = 0;
    while (seg < l.n_seg - 1 && s >= l.seg_steps[seg]) { s -= l.seg_steps[seg]; ++seg; }
    int col;
    if (l.seg_type[seg] == NK_SEG_ACT) {
        col = 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    } else if (l.seg_type[seg] == NK_SEG_PEX) {
        col = nk_enc_col(nk_pe_xyz_col(8 * s + j, h, p.Lx), p.inc_x);
    } else {
        col = nk_enc_col(nk_pe_dir_col(8 * s + j, h, p.Ld), p.inc_d);
    }
    if (col < 0) return 0.0f;
    col += l.seg_col[seg];
    if (col >= l.in_dim) return 0.0f;
    return prm[l.w_off + row * l.in_dim + col];
}

// transposed stream element: rows = ACT input features of layer li (tile ti), k = its outputs
__device__ float bwd_weight_elem(const NkPlan& p, const float* __restrict__ prm, int li, int ti, int ks, int r, int h, int j) {
    const NkLayer& l = p.layer[li];
    const int L_ALPHA = p.D, L_FEAT = p.D + 1, L_RGB = p.D + 3;
    const int in_col = l.bwd_act_col + 32 * ti + r;
    if (li == L_FEAT && ks == l.bwd_nk - 1) { // extra dOut k-step: k = 3 is d sigma_raw -> alpha row
        const NkLayer& a = p.layer[L_ALPHA];
        if (p.no_dirs) {                      // ... k = 0..3 -> the four rows of the output layer
            if (h != 0 || j > 3) return 0.0f;
            return prm[a.w_off + j * a.in_dim + 32 * ti + r];
        }
        if (h != 0 || j != 3) return 0.0f;
        return prm[a.w_off + 32 * ti + r];
    }
    if (!l.has_params) return 0.0f;
    if (li == L_RGB) {                       // single dOut k-step: k = 0..2 are d rgb_raw
        if (h != 0 || j >= 3) return 0.0f;
        return prm[l.w_off + j * l.in_dim + in_col];
    }
    const int m = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    if (m >= l.out_dim) return 0.0f;
    return prm[l.w_off + m * l.in_dim + in_col];
}

// Encoding slot -> column of the reference encoding for the 16x16x32 inference stream (the slot
// order pe16_xyz_frags / pe16_dir_frags produce): slot 32 s + 8 g + j, j < 6 = value j (sin x, y, z,
// cos x, y, z) of frequency 4 s + g; j = 6, 7 = left-over slot e = 8 s + 2 g + (j - 6): frequencies 8
// and 9 (e < 12), the raw input (12..14), padding.  Directions: one k-step, frequency g, left-over
// slots 2 g + (j - 6) = raw input (0..2), padding.
__device__ int nk16_pe_xyz_col(int slot, int L) {
    const int s = slot >> 5, g = (slot >> 3) & 3, j = slot & 7;
    if (j < 6) return 4 * s + g < L ? 3 + 6 * (4 * s + g) + j : -1;
    const int e = 8 * s + 2 * g + (j - 6);
    if (e < 12) return 8 + e / 6 < L ? 51 + e : -1;
    return e < 15 ? e - 12 : -1;
}
__device__ int nk16_pe_dir_col(int slot, int L) {
    const int g = (slot >> 3) & 3, j = slot & 7;
    if (j < 6) return g < L ? 3 + 6 * g + j : -1;
    const int e = 2 * g + (j - 6);
    return e < 3 ? e : -1;
}

// 16x16x32 inference stream element: layer li, 32-row output tile t, block b of the tile (k-step
// b >> 1 of 32 inputs, row tile b & 1), lane (row = lane & 15, k quarter kq = lane >> 4), element j
__device__ float fwd16_weight_elem(const NkPlan& p, const float* __restrict__ prm, int li, int t, int b, int lane, int j) {
    const NkLayer& l = p.layer[li];
    const int row = 32 * t + 16 * (b & 1) + (lane & 15), kq = lane >> 4;
    if (row >= l.out_dim || !l.has_params) return 0.0f;
    int s = b >> 1, seg = 0;
    while (seg < l.n_seg - 1 && s >= l.seg_steps[seg] / 2) { s -= l.seg_steps[seg] / 2; ++seg; }
    int col;
    if (l.seg_type[seg] == NK_SEG_ACT) col = 32 * s + 16 * (j >> 2) + 4 * kq + (j & 3);
    else if (l.seg_type[seg] == NK_SEG_PEX) col = nk_enc_col(nk16_pe_xyz_col(32 * s + 8 * kq + j, p.Lx), p.inc_x);
    else col = nk_enc_col(nk16_pe_dir_col(32 * s + 8 * kq + j, p.Ld), p.inc_d);
    if (col < 0) return 0.0f;
    col += l.seg_col[seg];
    if (col >= l.in_dim) return 0.0f;
    return prm[l.w_off + row * l.in_dim + col];
}

__global__ __launch_bounds__(64) void k_pack_weights(NkPlan p, const float* __restrict__ prm, char* __restrict__ packed) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    char* dst = packed + (int64_t)blk * 1024;
    if (blk >= p.fwd_blocks + p.bwd_blocks) {
        // third stream: the forward stream's block structure in the 16x16x32 form (inference kernel)
        const int fb = blk - p.fwd_blocks - p.bwd_blocks;
        int li = 0;
        while (li + 1 < p.NL && fb >= p.layer[li + 1].fwd_block0) ++li;
        const NkLayer& l = p.layer[li];
        const int rel = fb - l.fwd_block0, t = rel / (l.nk + 1), b = rel % (l.nk + 1);
        if (b == l.nk) {
            // bias block: float (rt * 4 + g) * 4 + i = bias[32 t + 16 rt + 4 g + i]; rest zero
            float4 v = make_float4(0, 0, 0, 0);
            if (lane < 8) {
                float e[4];
                for (int k = 0; k < 4; ++k) {
                    const int row = 32 * t + 16 * (lane >> 2) + 4 * (lane & 3) + k;
                    e[k] = (row < l.out_dim && l.has_params) ? prm[l.b_off + row] : 0.0f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            *(float4*)(dst + lane * 16) = v;
            return;
        }
        bf16x8 f;
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)fwd16_weight_elem(p, prm, li, t, b, lane, j);
        *(bf16x8*)(dst + lane * 16) = f;
        return;
    }
    if (blk < p.fwd_blocks) {
        int li = 0;
        while (li + 1 < p.NL && blk >= p.layer[li + 1].fwd_block0) ++li;
        const NkLayer& l = p.layer[li];
        const int rel = blk - l.fwd_block0, t = rel / (l.nk + 1), ks = rel % (l.nk + 1);
        if (ks == l.nk) {
            // bias block: float[2][16], [h][i] = bias[32t + (i&3) + 8(i>>2) + 4h]; rest zero
            float4 v = make_float4(0, 0, 0, 0);
            if (lane < 8) {
                const int hh = lane >> 2, q = lane & 3;
                float e[4];
                for (int k = 0; k < 4; ++k) {
                    const int row = 32 * t + 8 * q + 4 * hh + k;
                    e[k] = (row < l.out_dim && l.has_params) ? prm[l.b_off + row] : 0.0f;
                }
                v = make_float4(e[0], e[1], e[2], e[3]);
            }
            *(float4*)(dst + lane * 16) = v;
            return;
        }
        bf16x8 f;
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)fwd_weight_elem(p, prm, li, t, ks, r, h, j);
        *(bf16x8*)(dst + lane * 16) = f;
    } else {
        const int b = blk - p.fwd_blocks;
        // transposed stream order: RGB, VIEWS, FEAT, trunk D-1 .. 1
        int order[NK_MAX_LAYERS], n = 0;
        if (!p.no_dirs) { order[n++] = p.D + 3; order[n++] = p.D + 2; }
        order[n++] = p.D + 1;
        for (int i = p.D - 1; i >= 1; --i) order[n++] = i;
        int k = 0;
        while (k + 1 < n && b >= p.layer[order[k + 1]].bwd_block0) ++k;
        const int li = order[k];
        const NkLayer& l = p.layer[li];
        const int rel = b - l.bwd_block0, ti = rel / l.bwd_nk, ks = rel % l.bwd_nk;
        bf16x8 f;
        for (int j = 0; j < 8; ++j) f[j] = (__bf16)bwd_weight_elem(p, prm, li, ti, ks, r, h, j);
        *(bf16x8*)(dst + lane * 16) = f;
    }
}

extern "C" int nk_pack_weights(const nk_arch* arch, const float* params, void* packed, void* stream) {
    if (!params || !packed) NK_FAIL(NK_ERR_ARG, "nk_pack_weights: null pointer");
    if (((uintptr_t)packed & 15) != 0) NK_FAIL(NK_ERR_ARG, "nk_pack_weights: packed buffer must be 16-byte aligned");
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    hipLaunchKernelGGL(k_pack_weights, dim3(2 * p.fwd_blocks + p.bwd_blocks), dim3(64), 0, nk_stream(stream), p, params,
                       (char*)packed);
    NK_CHECK_LAUNCH("k_pack_weights");
    return NK_OK;
}

// --------------------------------------------------------------------------------------------
// fused forward
// --------------------------------------------------------------------------------------------
struct FwdArgs {
    NkPlan plan;
    const char* packed;
    const char* packed16;  // the 16x16x32 inference stream
    const float* rays;     // [R,11]
    const float* z;        // [P]
    float* raw;            // [P,4]
    char* act;             // saved activations or null
    int64_t P, G;
    int n;                 // samples per ray
    const int32_t* live;   // inference only: compacted list of the groups to evaluate, or null (all of them)
    const int32_t* n_live; // its length, in device memory
    int stagger;           // upper half of each workgroup's waves run the staggered stage order
#ifdef NK_STAMPS
    long long* stamps;     // debug build: per-wave cycle stamps of one workgroup
#endif
};

#include "nk_fwd_chain.inc"
#include "nk_fwd16_chain.inc"


int64_t nk_act_bytes_fmt(const nk_arch* arch, int64_t R, int32_t n, int act_format) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    if (R <= 0 || n <= 0) { nk_set_error("nk_act_bytes: non-positive size"); return NK_ERR_ARG; }
    if (act_format < NK_ACT_BF16 || act_format > NK_ACT_FP8_BF8) { nk_set_error("unknown act_format %d", act_format); return NK_ERR_ARG; }
    return nk_mask_offset(p, nk_mask_count(p), nk_groups(R * (int64_t)n), nk_act_elem_bytes(act_format));
}

extern "C" int64_t nk_act_bytes(const nk_arch* arch, int64_t R, int32_t n) { return nk_act_bytes_fmt(arch, R, n, NK_ACT_BF16); }

// live_groups / n_live (device, inference only): evaluate only the listed groups of 32 points
int nk_mlp_forward_fmt(const nk_arch* arch, const void* packed, const float* ray_batch, const float* z, int64_t R,
                       int32_t n, float* raw, void* act_save, int act_format, const int32_t* live_groups,
                       const int32_t* n_live, void* stream) {
    if (!packed || !ray_batch || !z || !raw || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: null pointer or non-positive size");
    if (((uintptr_t)packed & 15) || ((uintptr_t)raw & 15) || ((uintptr_t)act_save & 15))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: packed, raw and act_save must be 16-byte aligned");
    if (act_format < NK_ACT_BF16 || act_format > NK_ACT_FP8_BF8) NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: unknown act_format %d", act_format);
    if (live_groups && (act_save || !n_live)) NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: a group list is for inference and needs its length");
    FwdArgs a;
    int rc = nk_build_plan(arch, &a.plan);
    if (rc < 0) return rc;
    a.packed = (const char*)packed; a.rays = ray_batch; a.z = z; a.raw = raw; a.act = (char*)act_save;
    a.packed16 = (const char*)packed + (int64_t)(a.plan.fwd_blocks + a.plan.bwd_blocks) * 1024;
    a.P = R * (int64_t)n; a.G = nk_groups(a.P); a.n = n;
    a.live = live_groups; a.n_live = n_live;
#ifdef NK_STAMPS
    { const char* e = getenv("NK_STAMP_PTR"); a.stamps = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    const int nw = nk_waves_per_wg(0);
    static const int env_stagger = []() { const char* e = getenv("NK_STAGGER_FWD"); return e ? atoi(e) : 0; }();
    a.stagger = env_stagger;
    const dim3 grid((unsigned)(a.G / nw)), block(nw * 64);
    const int fmt = !act_save ? 0 : (act_format == NK_ACT_BF16 ? 1 : 2);
    // inference runs the 16x16x32 form (NK_FWD16=0: the 32x32x16 form the training kernels use)
    static const int env_fwd16 = []() { const char* e = getenv("NK_FWD16"); return e ? atoi(e) : 1; }();
#define NK_FWD_LAUNCH(W_, WD_, NW_, LDS_) do { \
        if (fmt == 0 && env_fwd16) hipLaunchKernelGGL((k_mlp_fwd16<W_, WD_, NW_>), grid, block, LDS_, nk_stream(stream), a); \
        else if (fmt == 1) hipLaunchKernelGGL((k_mlp_fwd<W_, WD_, NW_, 1>), grid, block, LDS_, nk_stream(stream), a); \
        else if (fmt == 2) hipLaunchKernelGGL((k_mlp_fwd<W_, WD_, NW_, 2>), grid, block, LDS_, nk_stream(stream), a); \
        else hipLaunchKernelGGL((k_mlp_fwd<W_, WD_, NW_, 0>), grid, block, LDS_, nk_stream(stream), a); } while (0)
    if (a.plan.W == 256) {
        constexpr int lds_bytes = 3 * (16 + 4 + 1) * 1024;     // three ring slots
        if (nw == 8) NK_FWD_LAUNCH(256, 128, 8, lds_bytes); else NK_FWD_LAUNCH(256, 128, 4, lds_bytes);
    } else {
        constexpr int lds_bytes = 3 * (8 + 4 + 1) * 1024;
        if (nw == 8) NK_FWD_LAUNCH(128, 64, 8, lds_bytes); else NK_FWD_LAUNCH(128, 64, 4, lds_bytes);
    }
#undef NK_FWD_LAUNCH
    NK_CHECK_LAUNCH("k_mlp_fwd");
    return NK_OK;
}

extern "C" int nk_mlp_forward(const nk_arch* arch, const void* packed, const float* ray_batch, const float* z,
                              int64_t R, int32_t n, float* raw, void* act_save, void* stream) {
    return nk_mlp_forward_fmt(arch, packed, ray_batch, z, R, n, raw, act_save, NK_ACT_BF16, nullptr, nullptr, stream);
}
