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
// A workgroup is 8 wavefronts = 256 points sharing one pass over the stream; the stream is
// double-buffered per output tile (stage), one barrier per stage.
#include "nk_common.h"
#include "nk_mfma.h"

// --------------------------------------------------------------------------------------------
// weight packing: flat fp32 parameters -> bf16 fragment streams
// --------------------------------------------------------------------------------------------
// forward stream element: layer li, output tile t, k-step ks, lane (r = lane&31, h = lane>>5), j
__device__ float fwd_weight_elem(const NkPlan& p, const float* __restrict__ prm, int li, int t, int ks, int r, int h, int j) {
    const NkLayer& l = p.layer[li];
    const int row = 32 * t + r;
    if (row >= l.out_dim) return 0.0f;
    int s = ks, seg = 0;
    while (seg < l.n_seg - 1 && s >= l.seg_steps[seg]) { s -= l.seg_steps[seg]; ++seg; }
    int col;
    if (l.seg_type[seg] == NK_SEG_ACT) {
        col = 32 * (s >> 1) + 16 * (s & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    } else if (l.seg_type[seg] == NK_SEG_PEX) {
        col = nk_pe_xyz_col(8 * s + j, h, p.Lx);
    } else {
        col = nk_pe_dir_col(8 * s + j, h, p.Ld);
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
    if (li == L_RGB) {                       // single dOut k-step: k = 0..2 are d rgb_raw
        if (h != 0 || j >= 3) return 0.0f;
        return prm[l.w_off + j * l.in_dim + in_col];
    }
    if (li == L_FEAT && ks == l.bwd_nk - 1) { // extra dOut k-step: k = 3 is d sigma_raw -> alpha row
        if (h != 0 || j != 3) return 0.0f;
        const NkLayer& a = p.layer[L_ALPHA];
        return prm[a.w_off + 32 * ti + r];
    }
    const int m = 32 * (ks >> 1) + 16 * (ks & 1) + 8 * (j >> 2) + 4 * h + (j & 3);
    if (m >= l.out_dim) return 0.0f;
    return prm[l.w_off + m * l.in_dim + in_col];
}

__global__ __launch_bounds__(64) void k_pack_weights(NkPlan p, const float* __restrict__ prm, char* __restrict__ packed) {
    const int blk = blockIdx.x, lane = threadIdx.x;
    const int r = lane & 31, h = lane >> 5;
    char* dst = packed + (int64_t)blk * 1024;
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
                    e[k] = row < l.out_dim ? prm[l.b_off + row] : 0.0f;
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
        order[n++] = p.D + 3; order[n++] = p.D + 2; order[n++] = p.D + 1;
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
    hipLaunchKernelGGL(k_pack_weights, dim3(p.fwd_blocks + p.bwd_blocks), dim3(64), 0, nk_stream(stream), p, params,
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
    const float* rays;     // [R,11]
    const float* z;        // [P]
    float* raw;            // [P,4]
    char* act;             // saved activations or null
    int64_t P, G;
    int n;                 // samples per ray
};

template <int W, int WD>
__global__ __launch_bounds__(512, 2) void k_mlp_fwd(FwdArgs a) {
    constexpr int NT = W / 32, NKT = W / 16, NTD = WD / 32, NKD = WD / 16;
    constexpr int HALF = (NKT + 4 + 1) * 1024;          // largest stage: skip layer
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const NkPlan& p = a.plan;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;
    const int64_t grp = (int64_t)blockIdx.x * 8 + wave;
    const int64_t pt = grp * 32 + c;
    const bool valid = pt < a.P;
    const int64_t ptc = valid ? pt : a.P - 1;
    const int64_t ray = ptc / a.n;
    const float* rb = a.rays + ray * 11;
    const float zv = a.z[ptc];
    const float px = rb[0] + rb[3] * zv, py = rb[1] + rb[4] * zv, pz = rb[2] + rb[5] * zv;
    const float vx = rb[8], vy = rb[9], vz = rb[10];
    const int D = p.D;
    const int NOW = W / 8;

    WStream ws;
    ws.g = a.packed; ws.ring = lds; ws.half_bytes = HALF; ws.cur = 1; ws.wave = wave; ws.lane = lane;
    ws.issue(4 + 1);     // stage 0 = tile 0 of trunk layer 0 -> half 0
    ws.flip();

    f32x16 acc[NT];
    Frag hb[2 * NT];

    // ---- trunk layer 0: gamma(x) -> W ----------------------------------------------------
    {
        Frag pex[4];
        pe_xyz_frags(px, py, pz, h, pex);
        if (a.act) {
            char* dst = a.act + nk_act_offset(p, 0, a.G);
#pragma unroll
            for (int s = 0; s < 4; ++s) *(u32x4*)(dst + ((grp * 8 + 2 * s + h) * 32 + c) * 16) = pex[s].u;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            // next stage: another tile of this layer, or the first tile of layer 1
            const int next_nb = (t + 1 < NT) ? 5 : ((p.skip == 0 ? NKT + 4 : NKT) + 1);
            ws.issue(next_nb);
            const char* st = ws.stage();
            acc[t] = tile_mfma<4>(lds_bias(st, 4, h), st, 0, pex, lane);
            ws.flip();
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc_to_frags(acc[t], p.relu_first != 0, hb[2 * t], hb[2 * t + 1]);
            if (a.act) save_tile(a.act + nk_act_offset(p, 2, a.G), grp, NOW, t, c, h, hb[2 * t], hb[2 * t + 1]);
        }
    }
    // ---- trunk layers 1..D-1 -------------------------------------------------------------
    for (int L = 1; L < D; ++L) {
        const bool is_skip = (p.skip >= 0 && L == p.skip + 1);
        const int nk_next_layer = (L + 1 < D) ? (((p.skip >= 0 && L + 1 == p.skip + 1) ? NKT + 4 : NKT) + 1) : (NKT + 1);  // next trunk layer or ALPHA
        if (is_skip) {
            Frag pex[4];
            pe_xyz_frags(px, py, pz, h, pex);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ws.issue((t + 1 < NT) ? NKT + 4 + 1 : nk_next_layer);
                const char* st = ws.stage();
                f32x16 v = tile_mfma<4>(lds_bias(st, NKT + 4, h), st, 0, pex, lane);
                acc[t] = tile_mfma<NKT>(v, st, 4, hb, lane);
                ws.flip();
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                ws.issue((t + 1 < NT) ? NKT + 1 : nk_next_layer);
                const char* st = ws.stage();
                acc[t] = tile_mfma<NKT>(lds_bias(st, NKT, h), st, 0, hb, lane);
                ws.flip();
            }
        }
        char* dst = a.act ? a.act + nk_act_offset(p, 2 + L, a.G) : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc_to_frags(acc[t], true, hb[2 * t], hb[2 * t + 1]);
            if (dst) save_tile(dst, grp, NOW, t, c, h, hb[2 * t], hb[2 * t + 1]);
        }
    }
    // ---- ALPHA: 1 tile, row 0 = sigma_raw ------------------------------------------------
    float sigma;
    {
        ws.issue(NKT + 1);                     // first FEAT tile
        const char* st = ws.stage();
        const f32x16 v = tile_mfma<NKT>(lds_bias(st, NKT, h), st, 0, hb, lane);
        sigma = v[0];
        ws.flip();
    }
    // ---- FEAT ----------------------------------------------------------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        ws.issue((t + 1 < NT) ? NKT + 1 : NKT + 2 + 1);   // next: FEAT tile or first VIEWS tile
        const char* st = ws.stage();
        acc[t] = tile_mfma<NKT>(lds_bias(st, NKT, h), st, 0, hb, lane);
        ws.flip();
    }
    {
        char* dst = a.act ? a.act + nk_act_offset(p, 2 + D, a.G) : nullptr;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            acc_to_frags(acc[t], p.relu_feat != 0, hb[2 * t], hb[2 * t + 1]);
            if (dst) save_tile(dst, grp, NOW, t, c, h, hb[2 * t], hb[2 * t + 1]);
        }
    }
    // ---- VIEWS: cat[feat, gamma(d)] -> WD, ReLU -------------------------------------------
    Frag gb[2 * NTD];
    {
        Frag ped[2];
        pe_dir_frags(vx, vy, vz, h, ped);
        if (a.act) {
            char* dst = a.act + nk_act_offset(p, 1, a.G);
#pragma unroll
            for (int s = 0; s < 2; ++s) *(u32x4*)(dst + ((grp * 4 + 2 * s + h) * 32 + c) * 16) = ped[s].u;
        }
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
            ws.issue((t + 1 < NTD) ? NKT + 2 + 1 : NKD + 1);   // next: VIEWS tile or the RGB tile
            const char* st = ws.stage();
            f32x16 v = tile_mfma<NKT>(lds_bias(st, NKT + 2, h), st, 0, hb, lane);
            acc[t] = tile_mfma<2>(v, st, NKT, ped, lane);
            ws.flip();
        }
        char* dst = a.act ? a.act + nk_act_offset(p, 3 + D, a.G) : nullptr;
#pragma unroll
        for (int t = 0; t < NTD; ++t) {
            acc_to_frags(acc[t], true, gb[2 * t], gb[2 * t + 1]);
            if (dst) save_tile(dst, grp, WD / 8, t, c, h, gb[2 * t], gb[2 * t + 1]);
        }
    }
    // ---- RGB: 1 tile, rows 0..2 ------------------------------------------------------------
    {
        const char* st = ws.stage();
        const f32x16 v = tile_mfma<NKD>(lds_bias(st, NKD, h), st, 0, gb, lane);
        if (valid && h == 0) *(float4*)(a.raw + pt * 4) = make_float4(v[0], v[1], v[2], sigma);
    }
}

extern "C" int64_t nk_act_bytes(const nk_arch* arch, int64_t R, int32_t n) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    if (R <= 0 || n <= 0) { nk_set_error("nk_act_bytes: non-positive size"); return NK_ERR_ARG; }
    return nk_act_offset(p, nk_act_count(p), nk_groups(R * (int64_t)n));
}

extern "C" int nk_mlp_forward(const nk_arch* arch, const void* packed, const float* ray_batch, const float* z,
                              int64_t R, int32_t n, float* raw, void* act_save, void* stream) {
    if (!packed || !ray_batch || !z || !raw || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: null pointer or non-positive size");
    if (((uintptr_t)packed & 15) || ((uintptr_t)raw & 15) || ((uintptr_t)act_save & 15))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_forward: packed, raw and act_save must be 16-byte aligned");
    FwdArgs a;
    int rc = nk_build_plan(arch, &a.plan);
    if (rc < 0) return rc;
    a.packed = (const char*)packed; a.rays = ray_batch; a.z = z; a.raw = raw; a.act = (char*)act_save;
    a.P = R * (int64_t)n; a.G = nk_groups(a.P); a.n = n;
    const dim3 grid((unsigned)(a.G / 8)), block(512);
    if (a.plan.W == 256) {
        constexpr int lds_bytes = 2 * (16 + 4 + 1) * 1024;
        hipLaunchKernelGGL((k_mlp_fwd<256, 128>), grid, block, lds_bytes, nk_stream(stream), a);
    } else {
        constexpr int lds_bytes = 2 * (8 + 4 + 1) * 1024;
        hipLaunchKernelGGL((k_mlp_fwd<128, 64>), grid, block, lds_bytes, nk_stream(stream), a);
    }
    NK_CHECK_LAUNCH("k_mlp_fwd");
    return NK_OK;
}
