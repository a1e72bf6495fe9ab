// NeRF MLP backward for gfx950: a fused backward chain (dL/d activation through every layer,
// activations masked and kept in registers exactly as in the forward kernel, transposed weight
// stream through LDS) that writes each layer's dZ as bf16, and a split-K MFMA GEMM that turns
// (dZ, saved layer input) pairs into weight and bias gradients.
//
// Saved tensors all use one layout, [group of 32 points][feature octet][point][8 features] bf16,
// which is what a wavefront's converted accumulators store with 8-byte pieces and what
// ds_read_b64_tr_b16 turns into MFMA operands whose k index runs over points.
#include "nk_common.h"
#include "nk_mfma.h"

// --------------------------------------------------------------------------------------------
// backward chain
// --------------------------------------------------------------------------------------------
struct BwdArgs {
    NkPlan plan;
    const char* packed_t;   // transposed weight stream
    const char* act;        // saved activations
    const float* d_raw;     // [P,4]
    char* dz;               // dZ regions
    int64_t P, G;
};

#include "nk_bwd_chain.inc"


// --------------------------------------------------------------------------------------------
// dW GEMM: C[m, n] += sum_p A[p, m] * B[p, n]  (A = dZ region, B = saved layer-input region)
// --------------------------------------------------------------------------------------------
enum { NK_COL_IDENT = 0, NK_COL_PEX = 1, NK_COL_PED = 2 };

struct GemmArgs {
    const char* A; const char* B;   // regions [G][MT*4][32][8], [G][NT*4][32][8] bf16
    int MT, NT;                     // 32-wide tiles of A features / B features
    int WM, WN;                     // wave grid (WM*WN <= 8); wave tile = (MT/WM) x (NT/WN) tiles
    int64_t n_stages;               // G / 2 (a stage = 2 groups = 64 points)
    int64_t stages_per_wg;
    float* grad;                    // flat gradient
    int w_off, ld;                  // destination matrix origin and leading dimension
    int row_first, row_count, row_dst0;
    int colmode, col_base, col_count, nfreq;
    int b_off;                      // bias gradient origin, or -1
};

NK_D bf16x8 tr_frag(const char* piece, int tile, int ks, int lane) {
    // MFMA operand whose 32 "rows" are features 32*tile.. and whose k index is the point
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int o = 4 * tile + 2 * (g4 & 1) + (pp >> 1);
    const int sw = (o & 3) << 2;
    const int c0 = 16 * ks + 8 * (g4 >> 1) + q;
    const char* base = piece + o * 512 + (pp & 1) * 8;
    const bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NK_LDS bf16x4*)(base + ((c0) ^ sw) * 16));
    const bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((NK_LDS bf16x4*)(base + ((c0 + 4) ^ sw) * 16));
    bf16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
}

template <int TM, int TN>
__global__ __launch_bounds__(512, 2) void k_dw_gemm(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int MT = a.MT, NT = a.NT;
    const int grp_bytes = (MT + NT) * 2048;         // one group's A piece then B piece
    const int stage_bytes = 2 * grp_bytes;
    const int blocks_per_stage = 2 * (MT + NT) * 2;
    const bool active = wave < a.WM * a.WN;
    const int wm = active ? wave / a.WN : 0, wn = active ? wave % a.WN : 0;
    const int64_t s0 = (int64_t)blockIdx.x * a.stages_per_wg;
    int64_t s1 = s0 + a.stages_per_wg;
    if (s1 > a.n_stages) s1 = a.n_stages;
    if (s0 >= s1) return;

    auto issue = [&](int64_t stage, int half) {
        char* dst = lds + half * stage_bytes;
        for (int b = wave; b < blocks_per_stage; b += 8) {
            const int gi = b / ((MT + NT) * 2);               // group inside the stage
            const int rel = b - gi * (MT + NT) * 2;           // 1-KiB block inside the group's A|B image
            const bool isA = rel < MT * 2;
            const int pb = isA ? rel : rel - MT * 2;          // block inside the piece
            const int64_t grp = stage * 2 + gi;
            const char* src = isA ? a.A + grp * MT * 2048 : a.B + grp * NT * 2048;
            const int o = 2 * pb + (lane >> 5), cpos = lane & 31;
            const int csrc = cpos ^ ((o & 3) << 2);           // swizzle on the source side
            __builtin_amdgcn_global_load_lds((const NK_GLB void*)(src + (o * 32 + csrc) * 16),
                                             (NK_LDS void*)(dst + gi * grp_bytes + rel * 1024), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    float bsum[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) bsum[i] = 0.0f;
    bf16x2 ones; ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f;
    const bool want_bias = a.b_off >= 0;

    issue(s0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    int cur = 0;
    for (int64_t s = s0; s < s1; ++s) {
        if (s + 1 < s1) issue(s + 1, cur ^ 1);
        if (active) {
            const char* st = lds + cur * stage_bytes;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi) {
                const char* pa = st + gi * grp_bytes;
                const char* pb = pa + MT * 2048;
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    bf16x8 af[TM], bf[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) af[i] = tr_frag(pa, wm * TM + i, ks, lane);
#pragma unroll
                    for (int j = 0; j < TN; ++j) bf[j] = tr_frag(pb, wn * TN + j, ks, lane);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
                    // bias gradient: the WN waves of a row share it, each takes its k-steps
                    if (want_bias && ((2 * gi + ks) % a.WN) == wn) {
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
#pragma unroll
                            for (int w = 0; w < 4; ++w) {
                                bf16x2 pr; pr[0] = af[i][2 * w]; pr[1] = af[i][2 * w + 1];
                                bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
                            }
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        cur ^= 1;
    }
    if (!active) return;
    // epilogue: atomically add this K-slice's tiles into the flat gradient
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = 32 * (wn * TN + j) + r;
        int dcol = -1;
        if (a.colmode == NK_COL_IDENT) dcol = n < a.col_count ? a.col_base + n : -1;
        else {
            const int s = n >> 4, hh = (n >> 3) & 1, jj = n & 7;
            const int cc = a.colmode == NK_COL_PEX ? (s < 4 ? nk_pe_xyz_col(8 * s + jj, hh, a.nfreq) : -1)
                                                   : (s < 2 ? nk_pe_dir_col(8 * s + jj, hh, a.nfreq) : -1);
            dcol = cc >= 0 ? a.col_base + cc : -1;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = 32 * (wm * TM + i) + (e & 3) + 8 * (e >> 2) + 4 * h;
                const int mr = m - a.row_first;
                if (dcol >= 0 && mr >= 0 && mr < a.row_count)
                    atomicAdd(a.grad + a.w_off + (int64_t)(a.row_dst0 + mr) * a.ld + dcol, acc[i][j][e]);
            }
        }
    }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            const int m = 32 * (wm * TM + i) + r, mr = m - a.row_first;
            if (h == 0 && mr >= 0 && mr < a.row_count) atomicAdd(a.grad + a.b_off + a.row_dst0 + mr, tot);
        }
    }
}

static int launch_gemm(const NkPlan& p, const char* A, int MA, const char* B, int NB, int64_t G, float* grad,
                       int w_off, int ld, int row_first, int row_count, int row_dst0, int colmode, int col_base,
                       int col_count, int nfreq, int b_off, hipStream_t stream) {
    GemmArgs g;
    g.A = A; g.B = B; g.MT = MA / 32; g.NT = NB / 32;
    g.n_stages = G / 2;
    g.stages_per_wg = (g.n_stages + 1023) / 1024;
    if (g.stages_per_wg < 1) g.stages_per_wg = 1;
    g.grad = grad; g.w_off = w_off; g.ld = ld; g.row_first = row_first; g.row_count = row_count; g.row_dst0 = row_dst0;
    g.colmode = colmode; g.col_base = col_base; g.col_count = col_count; g.nfreq = nfreq; g.b_off = b_off;
    const unsigned nwg = (unsigned)((g.n_stages + g.stages_per_wg - 1) / g.stages_per_wg);
    const int lds_bytes = 2 * 2 * (g.MT + g.NT) * 2048;
    int TM, TN;
    // wave tilings: (MT, NT) -> (TM, TN, WM, WN)
    if (g.MT == 8 && g.NT == 8) { TM = 4; TN = 2; g.WM = 2; g.WN = 4; }
    else if (g.MT == 8 && g.NT == 2) { TM = 2; TN = 1; g.WM = 4; g.WN = 2; }
    else if (g.MT == 4 && g.NT == 8) { TM = 2; TN = 2; g.WM = 2; g.WN = 4; }
    else if (g.MT == 4 && g.NT == 4) { TM = 2; TN = 1; g.WM = 2; g.WN = 4; }
    else if (g.MT * g.NT <= 8) { TM = 1; TN = 1; g.WM = g.MT; g.WN = g.NT; }
    else NK_FAIL(NK_ERR_UNSUPPORTED, "dW GEMM: no wave tiling for %d x %d tiles", g.MT, g.NT);
    const dim3 grid(nwg), block(512);
#define NK_GEMM_LAUNCH(A_, B_) do { \
        NK_CHECK_HIP(hipFuncSetAttribute((const void*)k_dw_gemm<A_, B_>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes)); \
        hipLaunchKernelGGL((k_dw_gemm<A_, B_>), grid, block, lds_bytes, stream, g); } while (0)
    if (TM == 4 && TN == 2) NK_GEMM_LAUNCH(4, 2);
    else if (TM == 2 && TN == 2) NK_GEMM_LAUNCH(2, 2);
    else if (TM == 2 && TN == 1) NK_GEMM_LAUNCH(2, 1);
    else NK_GEMM_LAUNCH(1, 1);
#undef NK_GEMM_LAUNCH
    NK_CHECK_LAUNCH("k_dw_gemm");
    (void)p;
    return NK_OK;
}

extern "C" int64_t nk_dz_bytes(const nk_arch* arch, int64_t R, int32_t n) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    if (R <= 0 || n <= 0) { nk_set_error("nk_dz_bytes: non-positive size"); return NK_ERR_ARG; }
    return nk_dz_offset(p, nk_dz_count(p), nk_groups(R * (int64_t)n));
}

extern "C" int nk_mlp_backward(const nk_arch* arch, const void* packed, const void* act_save, const float* d_raw,
                               int64_t R, int32_t n, void* dz_scratch, float* grad, void* stream) {
    if (!packed || !act_save || !d_raw || !dz_scratch || !grad || R <= 0 || n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: null pointer or non-positive size");
    if (((uintptr_t)packed & 15) || ((uintptr_t)act_save & 15) || ((uintptr_t)dz_scratch & 15) || ((uintptr_t)d_raw & 15))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: buffers must be 16-byte aligned");
    BwdArgs a;
    int rc = nk_build_plan(arch, &a.plan);
    if (rc < 0) return rc;
    const NkPlan& p = a.plan;
    a.packed_t = (const char*)packed + (int64_t)p.fwd_blocks * 1024;
    a.act = (const char*)act_save; a.d_raw = d_raw; a.dz = (char*)dz_scratch;
    a.P = R * (int64_t)n; a.G = nk_groups(a.P);
    hipStream_t st = nk_stream(stream);
    const int nw = nk_waves_per_wg();
    const dim3 grid((unsigned)(a.G / nw)), block(nw * 64);
    if (p.W == 256) {
        if (nw == 8) hipLaunchKernelGGL((k_mlp_bwd<256, 128, 8>), grid, block, 2 * (16 + 1) * 1024, st, a);
        else hipLaunchKernelGGL((k_mlp_bwd<256, 128, 4>), grid, block, 2 * (16 + 1) * 1024, st, a);
    } else {
        if (nw == 8) hipLaunchKernelGGL((k_mlp_bwd<128, 64, 8>), grid, block, 2 * (8 + 1) * 1024, st, a);
        else hipLaunchKernelGGL((k_mlp_bwd<128, 64, 4>), grid, block, 2 * (8 + 1) * 1024, st, a);
    }
    NK_CHECK_LAUNCH("k_mlp_bwd");

    // weight / bias gradients, one GEMM per (dZ, layer input) pair
    const int D = p.D, W = p.W, WD = p.WD;
    const int64_t G = a.G;
    auto act = [&](int idx) { return a.act + nk_act_offset(p, idx, G); };
    auto dz = [&](int idx) { return (const char*)a.dz + nk_dz_offset(p, idx, G); };
    const NkLayer& lrgb = p.layer[D + 3];
    const NkLayer& lviews = p.layer[D + 2];
    const NkLayer& lfeat = p.layer[D + 1];
    const NkLayer& lalpha = p.layer[D];
    // rgb: rows 0..2 of dOut against g
    rc = launch_gemm(p, dz(0), 32, act(3 + D), WD, G, grad, lrgb.w_off, lrgb.in_dim, 0, 3, 0, NK_COL_IDENT, 0, WD, 0,
                     lrgb.b_off, st);
    if (rc < 0) return rc;
    // alpha: row 3 of dOut against h_{D-1}
    rc = launch_gemm(p, dz(0), 32, act(2 + D - 1), W, G, grad, lalpha.w_off, lalpha.in_dim, 3, 1, 0, NK_COL_IDENT, 0, W,
                     0, lalpha.b_off, st);
    if (rc < 0) return rc;
    // views: dZ_views against cat[feat, gamma(d)]
    rc = launch_gemm(p, dz(1), WD, act(2 + D), W, G, grad, lviews.w_off, lviews.in_dim, 0, WD, 0, NK_COL_IDENT, 0, W, 0,
                     lviews.b_off, st);
    if (rc < 0) return rc;
    rc = launch_gemm(p, dz(1), WD, act(1), 32, G, grad, lviews.w_off, lviews.in_dim, 0, WD, 0, NK_COL_PED, W, 32, p.Ld,
                     -1, st);
    if (rc < 0) return rc;
    // feat: dZ_feat against h_{D-1}
    rc = launch_gemm(p, dz(2), W, act(2 + D - 1), W, G, grad, lfeat.w_off, lfeat.in_dim, 0, W, 0, NK_COL_IDENT, 0, W, 0,
                     lfeat.b_off, st);
    if (rc < 0) return rc;
    // trunk
    for (int L = D - 1; L >= 0; --L) {
        const NkLayer& l = p.layer[L];
        const char* dzl = dz(3 + D - 1 - L);
        if (L == 0) {
            rc = launch_gemm(p, dzl, W, act(0), 64, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_PEX, 0, 64, p.Lx, l.b_off, st);
        } else if (p.skip >= 0 && L == p.skip + 1) {
            rc = launch_gemm(p, dzl, W, act(0), 64, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_PEX, 0, 64, p.Lx, -1, st);
            if (rc < 0) return rc;
            rc = launch_gemm(p, dzl, W, act(2 + L - 1), W, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_IDENT,
                             3 + 6 * p.Lx, W, 0, l.b_off, st);
        } else {
            rc = launch_gemm(p, dzl, W, act(2 + L - 1), W, G, grad, l.w_off, l.in_dim, 0, W, 0, NK_COL_IDENT, 0, W, 0,
                             l.b_off, st);
        }
        if (rc < 0) return rc;
    }
    return NK_OK;
}
