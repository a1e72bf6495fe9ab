// NeRF MLP backward for gfx950: a fused backward chain (dL/d activation through every layer,
// activations masked and kept in registers exactly as in the forward kernel, transposed weight
// stream through LDS) that writes each layer's dZ as bf16, and a split-K MFMA GEMM that turns
// (dZ, saved layer input) pairs into weight and bias gradients.
//
// Saved tensors all use one layout, [group of 32 points][feature octet][point][8 features] bf16,
// which is what a wavefront's converted accumulators store with 8-byte pieces and what
// ds_read_b64_tr_b16 turns into MFMA operands whose k index runs over points.
#include <stdio.h>
#include <string.h>
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
    int stagger;            // upper half of each workgroup's waves run the staggered stage order
    int act_eb;             // bytes per saved-activation element (places the ReLU bit words behind them)
    float dz_scale;         // e5m2 dZ only: power of two applied to d_raw on the way in
};

#include "nk_bwd_chain.inc"


// --------------------------------------------------------------------------------------------
// dW GEMM: C[m, n] += sum_p A[p, m] * B[p, n]  (A = dZ region, B = saved layer-input region)
// --------------------------------------------------------------------------------------------
enum { NK_COL_IDENT = 0, NK_COL_PEX = 1, NK_COL_PED = 2 };

// Where the product of one A segment (rows) and one B segment (columns) is accumulated.
struct GemmDst {
    int enabled;
    int w_off, ld;                          // destination matrix origin and leading dimension
    int row_first, row_count, row_dst0;     // segment rows [row_first, +row_count) -> matrix rows row_dst0..
    int colmode, col_base, col_count, nfreq;
    int with_input;                         // encodings: the reference matrix has columns for the raw 3-vector
    int row_tiled, col_tiled;               // rows / identity columns arrive in nk_tile_feature() order
};

// A and B are each one or two regions ([G][tiles*4][32][8] bf16) laid side by side, so that GEMMs
// sharing an operand (a skip layer's two inputs, the view layer's two inputs) read it from HBM
// once.  What the product kernel needs of one product:
struct GemmSrc {
    const char* A[2];               // A tiles [0, MA0) come from A[0], the rest from A[1]
    const char* B[2];
    int64_t n_stages;               // G (a stage = 1 group = 32 points)
    int64_t partial_off;            // first float of this product's slices in the partial-sum scratch
    int stages_per_wg;              // K-slice length of this product in this launch
    int wg_begin;                   // first workgroup of this product in the launch's 1-D grid
    // (32-bit fields only: a 16-bit one is fetched with a VECTOR load, and the wait for it inside
    // the stage loop would drain the whole load pipeline)
    int MA0, NB0;
    int shape;                      // index into NK_GEMM_SHAPES
    int want_bias;
};
// ... and what the reduction needs: where the sums go.
struct GemmOut {
    GemmDst dst[2][2];              // [A segment][B segment]
    int b_off[2];                   // bias gradient origin per A segment (rows as dst[seg][0]), or -1
    int MA0, NB0;
    int mt, nt, wn;                 // tile counts of the product and waves across a tile row
    int n_slices;
    int row8, col8;                 // A / B regions are 8-bit: their tiles keep features in nk_tile_feature8() order
    float scale;                    // the sums are multiplied by this (1, or 1 / the dZ scale of e5m2 dZ)
    int64_t partial_off;
    float* grad;                    // flat gradient
};

// Transposed LDS reads are issued from inline asm: hipcc drains vmcnt before a
// ds_read_b64_tr_b16 INTRINSIC that follows an LDS-DMA load (it does not for plain LDS loads),
// which would serialise every stage's loads with its MFMAs.  The price is that the waits for the
// read results are written by hand too (tr_wait).
struct TrFrag { u32x2 lo, hi; };

// issue the two 8-byte transposed reads of one MFMA operand: 32 "rows" = features 32*tile..,
// k index = point 16*ks..; `piece` is the LDS byte address of the [octet][point][8] image
NK_D void tr_issue(TrFrag& f, uint32_t piece, int tile, int ks, int lane) {
    const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int o = 4 * tile + 2 * (g4 & 1) + (pp >> 1);
    const int sw = (o & 3) << 2;
    const int c0 = 16 * ks + 8 * (g4 >> 1) + q;
    const uint32_t base = piece + o * 512 + (pp & 1) * 8;
    const uint32_t a0 = base + ((c0) ^ sw) * 16, a1 = base + ((c0 + 4) ^ sw) * 16;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.lo) : "v"(a0));
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(f.hi) : "v"(a1));
}
template <int N>
NK_D void tr_wait(TrFrag (&f)[N]) {           // results of every read issued so far have landed
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].lo), "+v"(f[i].hi));
}
NK_D bf16x8 tr_value(const TrFrag& f) {
    const u32x4 u = {f.lo[0], f.lo[1], f.hi[0], f.hi[1]};
    return __builtin_bit_cast(bf16x8, u);
}

// The same for 8-bit regions ([16-feature piece][point][16] bytes, one tile = pieces 2*tile and
// 2*tile + 1): ONE ds_read_b64_tr_b8 is a whole MFMA operand.  Per group of 16 lanes the
// instruction reads a block of 8 rows (points) x 16 columns (features) of bytes -- lane 2q + p of the
// group supplies the address of row q, columns 8p .. 8p+7 -- and hands lane i column i of the eight
// rows.  The points of an odd piece sit swizzled by 8 (the load side writes them that way), so the two
// pieces a 32-lane half reads fall into different halves of the 64 LDS banks.
struct TrFrag8 { u32x2 v; };
NK_D void tr_issue(TrFrag8& f, uint32_t piece, int tile, int ks, int lane) {
    const int g4 = lane >> 4, i = lane & 15;
    const int x = 2 * tile + (g4 & 1);
    const int row = (16 * ks + 8 * (g4 >> 1) + (i >> 1)) ^ ((x & 1) << 3);
    const uint32_t a0 = piece + x * 512 + row * 16 + (i & 1) * 8;
    asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(f.v) : "v"(a0));
}
template <int N>
NK_D void tr_wait(TrFrag8 (&f)[N]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(f[i].v));
}
NK_D long tr_value8(const TrFrag8& f) { return __builtin_bit_cast(long, f.v); }
// eight e4m3 bytes -> the bf16 operand of the same values (exact: e4m3 is a subset of bf16)
NK_D bf16x8 fp8_to_bf16x8(const TrFrag8& f) {
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    u32x4 u;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const f32x2 p = (w & 1) ? __builtin_amdgcn_cvt_pk_f32_fp8((int)f.v[w >> 1], true)
                                : __builtin_amdgcn_cvt_pk_f32_fp8((int)f.v[w >> 1], false);
        u[w] = pack_bf16(p[0], p[1]);
    }
    return __builtin_bit_cast(bf16x8, u);
}
NK_D float bf8_sum8(const TrFrag8& f) {       // sum of the eight e5m2 values
    typedef __attribute__((ext_vector_type(2))) float f32x2;
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        const f32x2 p = (w & 1) ? __builtin_amdgcn_cvt_pk_f32_bf8((int)f.v[w >> 1], true)
                                : __builtin_amdgcn_cvt_pk_f32_bf8((int)f.v[w >> 1], false);
        s += p[0] + p[1];
    }
    return s;
}
template <int F> struct TrOperand { typedef TrFrag type; };
template <> struct TrOperand<1> { typedef TrFrag8 type; };

// A stage is one group of 32 points: A piece (MT*2 KiB) then B piece (NT*2 KiB).  Stages go
// through a ring of NK_GEMM_RING LDS slots with three stages in flight, because one CU needs
// ~50 KB of loads outstanding at all times to keep its share of HBM bandwidth busy.  Order per
// stage s: counted wait (stage s landed; the loads of s+1, s+2 stay in flight) -> barrier (also
// retires every wave's reads of slot (s-1) % RING) -> issue stage s+3 into that slot -> MFMAs.
// (The bf16 x e4m3 form, whose stages are a quarter fewer bytes, was also run with a ring of five and
// four stages in flight: 12 % slower per stage, so every form keeps the ring of four.)
enum { NK_GEMM_RING = 4, NK_GEMM_RING_MIXED = 4 };
NK_HD int gemm_ring(int fa, int fb) { return (fa == 2 && fb == 1) ? NK_GEMM_RING_MIXED : NK_GEMM_RING; }

template <int N>
NK_D void gemm_wait_barrier() {
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}

// ONE launch runs every weight-gradient product of a network (of both networks when the batch
// is small), whatever their shapes: the launch is one round of workgroups over the chip, the
// points of each product are cut into K-slices of about equal predicted TIME (the ns-per-stage
// column of NK_GEMM_SHAPES), one per workgroup (product k
// owns workgroups [wg_begin_k, wg_begin_{k+1})), and a workgroup branches once into the body
// compiled for its product's tile shape.  A workgroup does not add its K-slice into the gradient
// (device-scope fp32 atomics on one 256-KiB matrix run at ~0.6 TB/s chip-wide): it stores its
// accumulator tiles to its own slice of a scratch region with plain coalesced stores, and
// k_dw_reduce sums the slices of each product in a fixed order and adds the result to the
// gradient, which also makes the gradient bit-reproducible from run to run.
//   slice layout (floats): [MT*NT tiles][16 accumulator registers][64 lanes], then the bias
//   partial sums [MT][WN][32].
enum { NK_GEMM_MAX_PRODUCTS = 24, NK_REDUCE_BATCH = 12 };
struct GemmBatch { GemmSrc g[NK_GEMM_MAX_PRODUCTS]; float* partial; int count; };
struct ReduceBatch { GemmOut g[NK_REDUCE_BATCH]; const float* partial; };

// (A tiles, B tiles, waves down, waves across, ns per stage in the bf16 / bf16 x e4m3 / e5m2 x e4m3
// form) of every product the supported networks produce.  The last columns are what the K-slices of
// a launch are balanced with: the measured time of one stage (32 points; 64 in the all-8-bit form)
// when every CU runs that shape (NK_GEMM_PER_SHAPE=1 launches the shapes one by one,
// tools/gemm_costs.py turns the kernel trace into this column; NK_GEMM_COSTS="id:ns,..." overrides
// entries).  The thinnest product (1 x 2 tiles, 6 KB per stage) is latency-bound and slows down next
// to bandwidth-bound neighbours: it is entered a third above its stand-alone 713 ns.  The 8-bit
// columns of the width-128 shapes are the bf16 figure times the ratio measured on the width-256 shape
// of the same layer.
//   width 256: trunk, trunk 0, rgb, feature + alpha, skip (h + gamma(x)), views (feat + gamma(d))
//   width 128: the same six
//   width 256 without view directions: the output layer (dOut against h_{D-1}); width 128 uses 1 x 4
#define NK_GEMM_SHAPES(X) \
    X(0, 8, 8, 2, 4, 1455, 1397, 1833) X(1, 8, 2, 4, 2, 921, 820, 976) X(2, 1, 4, 1, 4, 571, 474, 634) \
    X(3, 9, 8, 1, 8, 1589, 1510, 2082) X(4, 8, 10, 4, 2, 1669, 1519, 2070) X(5, 4, 9, 2, 3, 1148, 1079, 1469) \
    X(6, 4, 4, 2, 4, 694, 666, 874) X(7, 4, 2, 4, 2, 601, 535, 637) X(8, 1, 2, 1, 2, 950, 789, 1055) \
    X(9, 5, 4, 1, 4, 831, 789, 1089) X(10, 4, 6, 2, 2, 900, 819, 1116) X(11, 2, 5, 2, 1, 673, 633, 861) \
    X(12, 1, 8, 1, 8, 853, 637, 871)

NK_HD int64_t gemm_slice_floats(int MT, int NT, int WN) { return (int64_t)MT * NT * 1024 + (int64_t)MT * WN * 32; }

// Flat gradient index that element (row m of A tile mt, column n-in-tile r of B tile nt) of a
// product accumulates into, or -1 (padding rows / columns, disabled segment pairs).
NK_D int64_t gemm_dst_index(const GemmOut& a, int mt, int nt, int m_in_tile, int r) {
    const int sa = mt >= a.MA0, sb = nt >= a.NB0;
    const GemmDst& d = a.dst[sa][sb];
    if (!d.enabled) return -1;
    if (d.col_tiled && d.colmode == NK_COL_IDENT) r = a.col8 ? nk_tile_feature8(r) : nk_tile_feature(r);
    if (d.row_tiled) m_in_tile = a.row8 ? nk_tile_feature8(m_in_tile) : nk_tile_feature(m_in_tile);
    const int n = 32 * (nt - (sb ? a.NB0 : 0)) + r;         // column inside the B segment
    int dcol;
    if (d.colmode == NK_COL_IDENT) dcol = n < d.col_count ? d.col_base + n : -1;
    else {
        const int s = n >> 4, hh = (n >> 3) & 1, jj = n & 7;
        const int cc = d.colmode == NK_COL_PEX ? (s < 4 ? nk_pe_xyz_col(8 * s + jj, hh, d.nfreq) : -1)
                                               : (s < 2 ? nk_pe_dir_col(8 * s + jj, hh, d.nfreq) : -1);
        const int ec = nk_enc_col(cc, d.with_input);
        dcol = ec >= 0 ? d.col_base + ec : -1;
    }
    if (dcol < 0) return -1;
    const int mr = 32 * (mt - (sa ? a.MA0 : 0)) + m_in_tile - d.row_first;
    if (mr < 0 || mr >= d.row_count) return -1;
    return d.w_off + (int64_t)(d.row_dst0 + mr) * d.ld + dcol;
}

// FA / FB: bytes per element of the A regions (dZ: 2 = bf16, 1 = e5m2) and of the B regions (saved
// layer inputs: 2 = bf16, 1 = e4m3).  bf16 x bf16 and e5m2 x e4m3 go to the MFMA as they are read;
// bf16 x e4m3 (NK_ACT_FP8) converts the B operand to bf16 in registers.  With both sides 8-bit a
// stage is TWO groups (64 points), so that a stage keeps the bytes, the load count per wave and the
// counted waits of the bf16 form.
template <int MT, int NT, int WM, int WN, int FA, int FB>
NK_D void gemm_body(const GemmSrc& a, float* slice, char* lds) {
    static_assert(FA == 2 || FB == 1, "8-bit dZ comes with 8-bit activations");
    constexpr int TM = MT / WM, TN = NT / WN;       // wave tile, in 32x32 tiles
    static_assert(TM * WM == MT && TN * WN == NT && WM * WN <= 8, "wave grid must tile the output");
    constexpr int NG = (FA == 1 && FB == 1) ? 2 : 1;            // groups of 32 points per stage
    constexpr int a_blocks = MT * FA, b_blocks = NT * FB;       // 1-KiB blocks of one group's A / B piece
    constexpr int a_bytes = NG * a_blocks * 1024;               // the stage's A pieces, then its B pieces
    constexpr int stage_bytes = NG * (a_blocks + b_blocks) * 1024;
    constexpr int blocks_per_stage = NG * (a_blocks + b_blocks);
    constexpr int min_pieces = blocks_per_stage / 8;    // every wave issues at least this many loads per stage
    constexpr int RING = (FA == 2 && FB == 1) ? NK_GEMM_RING_MIXED : NK_GEMM_RING;
    typedef typename TrOperand<FA>::type OpA;
    typedef typename TrOperand<FB>::type OpB;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool active = wave < WM * WN;
    const int wm = active ? wave / WN : 0, wn = active ? wave % WN : 0;
    const int64_t s0 = (int64_t)((int)blockIdx.x - a.wg_begin) * a.stages_per_wg;
    int64_t s1 = s0 + a.stages_per_wg;
    if (s1 > a.n_stages) s1 = a.n_stages;
    if (s0 >= s1) return;

    auto issue = [&](int64_t stage) {
        char* dst = lds + (int)(stage % RING) * stage_bytes;
#pragma unroll
        for (int i = 0; i < (blocks_per_stage + 7) / 8; ++i) {      // compile-time trip count: the compiler
            const int b = wave + 8 * i;                             // would drain vmcnt after a real loop
            if (b >= blocks_per_stage) break;
            const bool isA = b < NG * a_blocks;
            const int bb = isA ? b : b - NG * a_blocks;
            const int per_g = isA ? a_blocks : b_blocks, F = isA ? FA : FB;
            const int g = NG == 1 ? 0 : (bb >= per_g ? 1 : 0);      // group inside the stage
            const int pb = bb - g * per_g;                          // 1-KiB block inside the group's piece
            const int t0 = isA ? a.MA0 : a.NB0;                     // tiles of the piece's first segment
            const int seg = pb >= F * t0;
            const int seg_tiles = seg ? (isA ? MT : NT) - t0 : t0;
            const char* src = (isA ? a.A[seg] : a.B[seg]) + (stage * NG + g) * seg_tiles * 1024 * F;
            // 512-byte piece inside the segment (bf16: a feature octet, 8-bit: 16 features) and the
            // point this lane fetches: swizzled on the source side so that the transposed reads are
            // conflict-free (see tr_issue)
            const int o = 2 * (pb - (seg ? F * t0 : 0)) + (lane >> 5), cpos = lane & 31;
            const int csrc = cpos ^ (F == 2 ? ((o & 3) << 2) : ((o & 1) << 3));
            __builtin_amdgcn_global_load_lds((const NK_GLB void*)(src + (o * 32 + csrc) * 16),
                                             (NK_LDS void*)(dst + b * 1024), 16, 0, 0);
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
    const bool want_bias = a.want_bias != 0;

    const uint32_t lds0 = (uint32_t)(uintptr_t)(NK_LDS char*)lds;
    // Stage loop: publish stage s (counted wait + barrier, refill of the slot behind it), then its
    // k-steps, each read under the MFMAs of the one before.  (Publishing stage s+1 in front of the
    // LAST k-step of stage s, so that its first operands are read under that k-step's MFMAs, was
    // measured: bf16 3.80 -> 4.22 ms per launch, e5m2 x e4m3 2.51 -> 2.58 -- the launch is paced by
    // the arrival of the loads, and an earlier wait only takes MFMAs away from under it.)
    constexpr int mp = min_pieces > 0 ? min_pieces : 1;
    // (fewer blocks than waves -- thin products in the 8-bit formats: a wave issues one load per stage
    // or none at all; counting one keeps the loads of the later stages in flight, and a wave without
    // loads has nothing outstanding, so any count lets it through)
    auto publish = [&](int64_t stage) {         // stage has landed for every wave; refill the slot behind it
        const int64_t left = s1 - 1 - stage;
        if (RING > 4 && left >= 3) gemm_wait_barrier<3 * mp>();
        else if (left >= 2) gemm_wait_barrier<2 * mp>();
        else if (left == 1) gemm_wait_barrier<mp>();
        else gemm_wait_barrier<0>();
        if (stage + RING - 1 < s1) issue(stage + RING - 1);
    };
    constexpr int NKS = 2 * NG;                 // k-steps of a stage (two per group)
    auto read_step = [&](OpA (&ra)[TM], OpB (&rb)[TN], int64_t stage, int kk) {
        const uint32_t st0 = lds0 + (uint32_t)(stage % RING) * stage_bytes;
        const uint32_t pa = st0 + (kk >> 1) * a_blocks * 1024;
        const uint32_t pb = st0 + a_bytes + (kk >> 1) * b_blocks * 1024;
#pragma unroll
        for (int i = 0; i < TM; ++i) tr_issue(ra[i], pa, wm * TM + i, kk & 1, lane);
#pragma unroll
        for (int j = 0; j < TN; ++j) tr_issue(rb[j], pb, wn * TN + j, kk & 1, lane);
    };
    // MFMAs (and this wave's share of the bias sums) of k-step kk of stage s on landed operands
    auto mma_step = [&](const OpA (&ra)[TM], const OpB (&rb)[TN], int64_t s, int kk) {
        const bool bias_step = want_bias && (((int)(s & 1) * NKS + kk) % WN) == wn;   // the WN waves of a row take turns
        if constexpr (FA == 2) {
            bf16x8 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = tr_value(ra[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if constexpr (FB == 2) bf[j] = tr_value(rb[j]);
                else bf[j] = fp8_to_bf16x8(rb[j]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf[j], acc[i][j], 0, 0, 0);
            if (bias_step) {        // bias gradient: the sum of the A operand over the k-step's points
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int w = 0; w < 4; ++w) {
                        bf16x2 pr; pr[0] = af[i][2 * w]; pr[1] = af[i][2 * w + 1];
                        bsum[i] = __builtin_amdgcn_fdot2_f32_bf16(pr, ones, bsum[i], false);
                    }
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf8_fp8(tr_value8(ra[i]), tr_value8(rb[j]), acc[i][j], 0, 0, 0);
            if (bias_step) {
#pragma unroll
                for (int i = 0; i < TM; ++i) bsum[i] += bf8_sum8(ra[i]);
            }
        }
    };
    for (int64_t s = s0; s < s0 + RING - 1 && s < s1; ++s) issue(s);
    for (int64_t s = s0; s < s1; ++s) {
        publish(s);
        if (active) {
            OpA fa[2][TM];
            OpB fb[2][TN];
            read_step(fa[0], fb[0], s, 0);
#pragma unroll
            for (int kk = 0; kk < NKS; ++kk) {
                const int cur = kk & 1;
                tr_wait(fa[cur]); tr_wait(fb[cur]);
                if (kk + 1 < NKS) read_step(fa[cur ^ 1], fb[cur ^ 1], s, kk + 1);
                mma_step(fa[cur], fb[cur], s, kk);
            }
        }
    }
    if (!active) return;
    // epilogue: this K-slice's tiles to the workgroup's scratch slice (256 B per store instruction)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            float* t = slice + ((wm * TM + i) * NT + (wn * TN + j)) * 1024 + lane;
#pragma unroll
            for (int e = 0; e < 16; ++e) t[e * 64] = acc[i][j][e];
        }
    if (want_bias) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const float tot = bsum[i] + __shfl_xor(bsum[i], 32, 64);
            if (lane < 32) slice[MT * NT * 1024 + ((wm * TM + i) * WN + wn) * 32 + lane] = tot;
        }
    }
}

template <int FA, int FB>
__global__ __launch_bounds__(512, 1) void k_dw_gemm(GemmBatch batch) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int prod = 0;
    while (prod + 1 < batch.count && (int)blockIdx.x >= batch.g[prod + 1].wg_begin) ++prod;
    const GemmSrc& a = batch.g[prod];
    const int wg = (int)blockIdx.x - a.wg_begin;
    switch (a.shape) {
#define NK_GEMM_CASE(ID, MT, NT, WM, WN, NS, NSM, NS8) \
    case ID: gemm_body<MT, NT, WM, WN, FA, FB>(a, batch.partial + a.partial_off + (int64_t)wg * gemm_slice_floats(MT, NT, WN), lds); break;
        NK_GEMM_SHAPES(NK_GEMM_CASE)
#undef NK_GEMM_CASE
    default: break;
    }
}

// Second stage: sums every output element of a product over the product's slices and adds it to
// the flat gradient.  A workgroup owns 64 consecutive elements (one accumulator register of one
// tile); its four wavefronts take the slices w = 0, 1, 2, 3 (mod 4) with eight loads in flight
// each and meet in LDS, so the order of the additions is fixed.  blockIdx.y = product (the grid
// is as wide as the largest one needs); the last workgroups of a product do the bias rows.
__global__ __launch_bounds__(256) void k_dw_reduce(ReduceBatch batch) {
    __shared__ float part[4][64];
    const GemmOut& a = batch.g[blockIdx.y];
    const int MT = a.mt, NT = a.nt, WN = a.wn;
    const int lane = threadIdx.x & 63, sg = threadIdx.x >> 6;
    const int n_rows = MT * NT * 16;                     // 64-element rows of the weight part
    if ((int)blockIdx.x >= n_rows + (MT * 32 + 63) / 64) return;        // (uniform: before the barrier)
    const int64_t stride = gemm_slice_floats(MT, NT, WN);
    const int n_slices = a.n_slices;
    const float* p0 = batch.partial + a.partial_off;
    int64_t dst = -1;
    const float* p = nullptr;
    int terms = 1;                                       // values per slice that fold into the element
    if ((int)blockIdx.x < n_rows) {
        const int tile = blockIdx.x >> 4, e = blockIdx.x & 15;
        dst = gemm_dst_index(a, tile / NT, tile % NT, (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5), lane & 31);
        p = p0 + (int64_t)blockIdx.x * 64 + lane;
    } else {
        const int b = ((int)blockIdx.x - n_rows) * 64 + lane;       // bias element: tile row mt, row r
        if (b < MT * 32) {
            const int mt = b >> 5, r = b & 31;
            const int sa = mt >= a.MA0;
            const GemmDst& d = a.dst[sa][0];
            const int mr = 32 * (mt - (sa ? a.MA0 : 0)) +
                           (d.row_tiled ? (a.row8 ? nk_tile_feature8(r) : nk_tile_feature(r)) : r) - d.row_first;
            if (a.b_off[sa] >= 0 && mr >= 0 && mr < d.row_count) dst = a.b_off[sa] + d.row_dst0 + mr;
            p = p0 + (int64_t)n_rows * 64 + mt * WN * 32 + r;
            terms = WN;
        }
    }
    float sum = 0.0f;
    if (dst >= 0) {
        for (int k = 0; k < terms; ++k) {
            const float* q = p + k * 32;
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            int w = sg;
            for (; w + 28 < n_slices; w += 32) {
#pragma unroll
                for (int u = 0; u < 8; ++u) s[u] += q[(int64_t)(w + 4 * u) * stride];
            }
            for (; w < n_slices; w += 4) s[0] += q[(int64_t)w * stride];
            sum += ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
        }
    }
    part[sg][lane] = sum;
    __syncthreads();
    if (sg == 0 && dst >= 0) a.grad[dst] += ((part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane])) * a.scale;
}

static GemmDst gemm_dst(int w_off, int ld, int row_first, int row_count, int colmode, int col_base, int col_count,
                        int nfreq, int with_input = 1) {
    GemmDst d;
    d.with_input = with_input;
    d.enabled = 1; d.w_off = w_off; d.ld = ld; d.row_first = row_first; d.row_count = row_count; d.row_dst0 = 0;
    d.colmode = colmode; d.col_base = col_base; d.col_count = col_count; d.nfreq = nfreq;
    d.row_tiled = 1; d.col_tiled = 1;       // dZ / activation tiles as the chain kernels store them
    return d;
}

// Products queued by nk_mlp_backward_gemms for one launch.
struct GemmQueue {
    GemmSrc src[NK_GEMM_MAX_PRODUCTS];
    GemmOut out[NK_GEMM_MAX_PRODUCTS];
    int n = 0;
    int fa = 2, fb = 2;     // bytes per element of the A (dZ) and B (layer input) regions, all products alike
    float scale = 1.0f;     // what the reduction multiplies the sums by
};

// A0 (+A1) against B0 (+B1); widths in features (multiples of 32, 0 = segment absent)
static void queue_gemm(GemmQueue& q, const char* A0, int wA0, const char* A1, int wA1, const char* B0, int wB0,
                       const char* B1, int wB1, int64_t G, float* grad, const GemmDst (&dst)[2][2], int b_off0,
                       int b_off1) {
    GemmSrc& g = q.src[q.n];
    GemmOut& o = q.out[q.n];
    g.A[0] = A0; g.A[1] = A1 ? A1 : A0; g.MA0 = wA0 / 32;
    g.B[0] = B0; g.B[1] = B1 ? B1 : B0; g.NB0 = wB0 / 32;
    g.n_stages = (q.fa == 1 && q.fb == 1) ? G / 2 : G;     // all-8-bit stages hold two groups (G is a multiple of 8)
    g.partial_off = 0; g.stages_per_wg = 0; g.wg_begin = 0; g.shape = -1;     // set per launch (flush_gemms)
    g.want_bias = (b_off0 >= 0 || b_off1 >= 0) ? 1 : 0;
    o.mt = (wA0 + wA1) / 32; o.nt = (wB0 + wB1) / 32; o.wn = 0;
    o.MA0 = g.MA0; o.NB0 = g.NB0;
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2; ++j) o.dst[i][j] = dst[i][j];
    o.b_off[0] = b_off0; o.b_off[1] = b_off1;
    o.n_slices = 0; o.partial_off = 0;
    o.row8 = q.fa == 1; o.col8 = q.fb == 1; o.scale = q.scale;
    o.grad = grad;
    ++q.n;
}

// Workgroups per launch: one per CU of the current device (NK_GEMM_WGS overrides).
enum { NK_GEMM_MIN_STAGES = 6, NK_MAX_GPUS = 64 };
static int gemm_wg_limit() {
    static const int env_wgs = []() { const char* e = getenv("NK_GEMM_WGS"); return e ? atoi(e) : 0; }();
    if (env_wgs > 0) return env_wgs < NK_GEMM_MAX_PRODUCTS ? NK_GEMM_MAX_PRODUCTS : env_wgs;
    static int cus[NK_MAX_GPUS];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NK_MAX_GPUS) return 256;
    if (!cus[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev] = n < NK_GEMM_MAX_PRODUCTS ? NK_GEMM_MAX_PRODUCTS : n;
    }
    return cus[dev];
}

// Scratch for the K-slice partial sums of one launch: one slice per workgroup, so at most one
// slice of the largest product shape of the network width per workgroup.
static int64_t gemm_partial_bytes(int W) {
    // (MT, NT, WN) of the largest slice: skip layer (8,10,2) for width 256, (4,6,2) for width 128
    const int64_t f = W == 256 ? gemm_slice_floats(8, 10, 2) : gemm_slice_floats(4, 6, 2);
    const int64_t g = W == 256 ? gemm_slice_floats(9, 8, 8) : gemm_slice_floats(5, 4, 4);
    return (f > g ? f : g) * 4 * gemm_wg_limit();
}

static int flush_gemms(GemmQueue& q, float* partial, int64_t partial_bytes, hipStream_t stream) {
    static const int shapes[][7] = {
#define NK_GEMM_ROW(ID, MT, NT, WM, WN, NS, NSM, NS8) {MT, NT, WM, WN, NS, NSM, NS8},
        NK_GEMM_SHAPES(NK_GEMM_ROW)
#undef NK_GEMM_ROW
    };
    constexpr int n_shapes = (int)(sizeof(shapes) / sizeof(shapes[0]));
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= NK_MAX_GPUS) dev = 0;
    if (q.n < 1 || q.n > NK_GEMM_MAX_PRODUCTS) NK_FAIL(NK_ERR_ARG, "dW GEMM: %d products queued", q.n);
    int64_t stage_ns[NK_GEMM_MAX_PRODUCTS];
    int64_t total = 0;
    int lds_bytes = 0;
    for (int k = 0; k < q.n; ++k) {
        GemmOut& o = q.out[k];
        int id = 0;
        while (id < n_shapes && !(shapes[id][0] == o.mt && shapes[id][1] == o.nt)) ++id;
        if (id == n_shapes) NK_FAIL(NK_ERR_UNSUPPORTED, "dW GEMM: no kernel for %d x %d tiles", o.mt, o.nt);
        q.src[k].shape = id;
        o.wn = shapes[id][3];
        // 1-KiB blocks of a stage (a stage of the all-8-bit form is two groups)
        const int blocks = (q.fa == 1 && q.fb == 1) ? 2 * (o.mt + o.nt) : q.fa * o.mt + q.fb * o.nt;
        stage_ns[k] = shapes[id][q.fa == 2 ? (q.fb == 2 ? 4 : 5) : 6];
        static const char* const cost_env = getenv("NK_GEMM_COSTS");      // (tuning) "id:ns,id:ns,..."
        if (cost_env) {
            for (const char* c = cost_env; c && *c; ) {
                int i = 0, ns = 0;
                if (sscanf(c, "%d:%d", &i, &ns) == 2 && i == id) stage_ns[k] = ns;
                c = strchr(c, ',');
                if (c) ++c;
            }
        }
        total += q.src[k].n_stages * stage_ns[k];
        const int need = gemm_ring(q.fa, q.fb) * blocks * 1024;
        if (need > lds_bytes) lds_bytes = need;
    }
    // K-slice lengths: every workgroup gets about the same time (at least a few stages, so the
    // load pipeline fills), one workgroup per CU at most
    const int limit = gemm_wg_limit();
    int64_t per_wg = (total + limit - 1) / limit;
    int64_t nwg = 0;
    for (;;) {
        nwg = 0;
        for (int k = 0; k < q.n; ++k) {
            int64_t spw = per_wg / stage_ns[k];
            if (spw < NK_GEMM_MIN_STAGES) spw = NK_GEMM_MIN_STAGES;
            if (spw > 0x7fffffff) NK_FAIL(NK_ERR_ARG, "dW GEMM: batch too large");
            q.src[k].stages_per_wg = (int)spw;
            nwg += (q.src[k].n_stages + spw - 1) / spw;
        }
        if (nwg <= limit) break;
        per_wg += per_wg / 64 + 1;
    }
    GemmBatch batch;
    int64_t off = 0;
    int wg = 0;
    for (int k = 0; k < q.n; ++k) {
        GemmSrc& g = q.src[k];
        GemmOut& o = q.out[k];
        const int slices = (int)((g.n_stages + g.stages_per_wg - 1) / g.stages_per_wg);
        g.wg_begin = wg; g.partial_off = off;
        o.n_slices = slices; o.partial_off = off;
        wg += slices;
        off += (int64_t)slices * gemm_slice_floats(o.mt, o.nt, o.wn);
        batch.g[k] = g;
    }
    for (int k = q.n; k < NK_GEMM_MAX_PRODUCTS; ++k) batch.g[k] = batch.g[0];
    batch.count = q.n;
    batch.partial = partial;
    if (off * 4 > partial_bytes)
        NK_FAIL(NK_ERR_WORKSPACE, "dW GEMM: partial-sum scratch too small (%lld > %lld bytes)", (long long)(off * 4), (long long)partial_bytes);
    const int fmt = q.fa == 2 ? (q.fb == 2 ? 0 : 1) : 2;
    const void* kern = fmt == 0 ? (const void*)k_dw_gemm<2, 2> : fmt == 1 ? (const void*)k_dw_gemm<2, 1> : (const void*)k_dw_gemm<1, 1>;
    static bool lds_allowed[NK_MAX_GPUS][3];
    if (!lds_allowed[dev][fmt]) {   // the most any shape asks for: four stages of the skip layer's product
        NK_CHECK_HIP(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         fmt == 1 ? NK_GEMM_RING_MIXED * (2 * 8 + 10) * 1024 : NK_GEMM_RING * (8 + 10) * 2048));
        lds_allowed[dev][fmt] = true;
    }
    if (fmt == 0) hipLaunchKernelGGL((k_dw_gemm<2, 2>), dim3((unsigned)wg), dim3(512), lds_bytes, stream, batch);
    else if (fmt == 1) hipLaunchKernelGGL((k_dw_gemm<2, 1>), dim3((unsigned)wg), dim3(512), lds_bytes, stream, batch);
    else hipLaunchKernelGGL((k_dw_gemm<1, 1>), dim3((unsigned)wg), dim3(512), lds_bytes, stream, batch);
    NK_CHECK_LAUNCH("k_dw_gemm");
    for (int first = 0; first < q.n; first += NK_REDUCE_BATCH) {
        const int count = q.n - first < NK_REDUCE_BATCH ? q.n - first : NK_REDUCE_BATCH;
        ReduceBatch rb;
        unsigned rows = 0;
        for (int k = 0; k < NK_REDUCE_BATCH; ++k) {
            rb.g[k] = q.out[first + (k < count ? k : 0)];
            const unsigned r = (unsigned)(rb.g[k].mt * rb.g[k].nt * 16 + (rb.g[k].mt * 32 + 63) / 64);
            if (r > rows) rows = r;
        }
        rb.partial = partial;
        hipLaunchKernelGGL(k_dw_reduce, dim3(rows, (unsigned)count), dim3(256), 0, stream, rb);
        NK_CHECK_LAUNCH("k_dw_reduce");
    }
    return NK_OK;
}

static int64_t dz_region_bytes(const NkPlan& p, int64_t R, int32_t n, int deb) {
    return (nk_dz_offset(p, nk_dz_count(p), nk_groups(R * (int64_t)n), deb) + 255) & ~(int64_t)255;
}

int64_t nk_dz_bytes_fmt(const nk_arch* arch, int64_t R, int32_t n, int act_format) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    if (rc < 0) return rc;
    if (R <= 0 || n <= 0) { nk_set_error("nk_dz_bytes: non-positive size"); return NK_ERR_ARG; }
    if (act_format < NK_ACT_BF16 || act_format > NK_ACT_FP8_BF8) { nk_set_error("unknown act_format %d", act_format); return NK_ERR_ARG; }
    // dZ regions, then the partial-sum scratch of the weight-gradient launches
    return dz_region_bytes(p, R, n, nk_dz_elem_bytes(act_format)) + gemm_partial_bytes(p.W);
}

extern "C" int64_t nk_dz_bytes(const nk_arch* arch, int64_t R, int32_t n) { return nk_dz_bytes_fmt(arch, R, n, NK_ACT_BF16); }

// dZ chain of one network: d_raw -> dZ of every layer (bf16, into job.dz)
int nk_mlp_backward_chain(const NkBwdJob& j, hipStream_t st) {
    if (!j.packed || !j.act || !j.d_raw || !j.dz || !j.grad || j.R <= 0 || j.n <= 0)
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: null pointer or non-positive size");
    if (((uintptr_t)j.packed & 15) || ((uintptr_t)j.act & 15) || ((uintptr_t)j.dz & 15) || ((uintptr_t)j.d_raw & 15))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: buffers must be 16-byte aligned");
    BwdArgs a;
    int rc = nk_build_plan(j.arch, &a.plan);
    if (rc < 0) return rc;
    const NkPlan& p = a.plan;
    a.packed_t = (const char*)j.packed + (int64_t)p.fwd_blocks * 1024;
    a.act = (const char*)j.act; a.d_raw = j.d_raw; a.dz = (char*)j.dz;
    a.P = j.R * (int64_t)j.n; a.G = nk_groups(a.P);
    if (j.act_format < NK_ACT_BF16 || j.act_format > NK_ACT_FP8_BF8) NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: unknown act_format %d", j.act_format);
    a.act_eb = nk_act_elem_bytes(j.act_format);
    const bool d8 = nk_dz_elem_bytes(j.act_format) == 1;
    a.dz_scale = d8 ? j.dz_scale : 1.0f;
    if (d8 && !(j.dz_scale > 0.0f)) NK_FAIL(NK_ERR_ARG, "nk_mlp_backward: e5m2 dZ needs a positive dz_scale");
    const int nw = nk_waves_per_wg(1);
    static const int env_stagger = []() { const char* e = getenv("NK_STAGGER_BWD"); return e ? atoi(e) : 1; }();
    a.stagger = env_stagger;
    const dim3 grid((unsigned)(a.G / nw)), block(nw * 64);
#define NK_BWD_LAUNCH(W_, WD_, NW_, LDS_) do { \
        if (d8) hipLaunchKernelGGL((k_mlp_bwd<W_, WD_, NW_, true>), grid, block, LDS_, st, a); \
        else hipLaunchKernelGGL((k_mlp_bwd<W_, WD_, NW_, false>), grid, block, LDS_, st, a); } while (0)
    if (p.W == 256) {
        if (nw == 8) NK_BWD_LAUNCH(256, 128, 8, 3 * (16 + 1) * 1024); else NK_BWD_LAUNCH(256, 128, 4, 3 * (16 + 1) * 1024);
    } else {
        if (nw == 8) NK_BWD_LAUNCH(128, 64, 8, 3 * (8 + 1) * 1024); else NK_BWD_LAUNCH(128, 64, 4, 3 * (8 + 1) * 1024);
    }
#undef NK_BWD_LAUNCH
    NK_CHECK_LAUNCH("k_mlp_bwd");
    return NK_OK;
}

// weight / bias gradients of one network: one GEMM per dZ tensor, with operands that two layers
// share read once
static int queue_network_gemms(GemmQueue& q, const NkBwdJob& j) {
    NkPlan p;
    int rc = nk_build_plan(j.arch, &p);
    if (rc < 0) return rc;
    const int D = p.D, W = p.W, WD = p.WD;
    const int64_t G = nk_groups(j.R * (int64_t)j.n);
    float* grad = j.grad;
    auto act = [&](int idx) { return (const char*)j.act + nk_act_offset(p, idx, G, q.fb); };
    auto dz = [&](int idx) { return (const char*)j.dz + nk_dz_offset(p, idx, G, q.fa); };
    const NkLayer& lrgb = p.layer[D + 3];
    const NkLayer& lviews = p.layer[D + 2];
    const NkLayer& lfeat = p.layer[D + 1];
    const NkLayer& lalpha = p.layer[D];
    const int pex0 = 3 * p.inc_x + 6 * p.Lx;          // width of gamma(x) inside a skip layer's input
    if (p.no_dirs) {
        // no view directions: the only product above the trunk is the output layer's, rows 0..3 of dOut
        // against h_{D-1}
        GemmDst d[2][2] = {};
        d[0][0] = gemm_dst(lalpha.w_off, lalpha.in_dim, 0, 4, NK_COL_IDENT, 0, W, 0);
        d[0][0].row_tiled = 0;              // the dOut tile keeps its rows in ascending order
        queue_gemm(q, dz(0), 32, nullptr, 0, act(2 + D - 1), W, nullptr, 0, G, grad, d, lalpha.b_off, -1);
    } else {
    {   // rgb: rows 0..2 of dOut against g
        GemmDst d[2][2] = {};
        d[0][0] = gemm_dst(lrgb.w_off, lrgb.in_dim, 0, 3, NK_COL_IDENT, 0, WD, 0);
        d[0][0].row_tiled = 0;              // the dOut tile keeps its rows in ascending order
        queue_gemm(q, dz(0), 32, nullptr, 0, act(3 + D), WD, nullptr, 0, G, grad, d, lrgb.b_off, -1);
    }
    {   // views: dZ_views against cat[feat, gamma(d)]
        GemmDst d[2][2] = {};
        d[0][0] = gemm_dst(lviews.w_off, lviews.in_dim, 0, WD, NK_COL_IDENT, 0, W, 0);
        d[0][1] = gemm_dst(lviews.w_off, lviews.in_dim, 0, WD, NK_COL_PED, W, 32, p.Ld, p.inc_d);
        queue_gemm(q, dz(1), WD, nullptr, 0, act(2 + D), W, act(1), 32, G, grad, d, lviews.b_off, -1);
    }
    {   // feature layer and alpha: dZ_feat and row 3 of dOut, side by side, against h_{D-1} (read once)
        GemmDst d[2][2] = {};
        d[0][0] = gemm_dst(lfeat.w_off, lfeat.in_dim, 0, W, NK_COL_IDENT, 0, W, 0);
        d[1][0] = gemm_dst(lalpha.w_off, lalpha.in_dim, 3, 1, NK_COL_IDENT, 0, W, 0);
        d[1][0].row_tiled = 0;              // (dOut tile)
        queue_gemm(q, dz(2), W, dz(0), 32, act(2 + D - 1), W, nullptr, 0, G, grad, d, lfeat.b_off, lalpha.b_off);
    }
    }
    for (int L = D - 1; L >= 0; --L) {     // trunk
        const NkLayer& l = p.layer[L];
        const char* dzl = dz(3 + D - 1 - L);
        GemmDst d[2][2] = {};
        if (L == 0) {
            d[0][0] = gemm_dst(l.w_off, l.in_dim, 0, W, NK_COL_PEX, 0, 64, p.Lx, p.inc_x);
            queue_gemm(q, dzl, W, nullptr, 0, act(0), 64, nullptr, 0, G, grad, d, l.b_off, -1);
        } else if (p.skip >= 0 && L == p.skip + 1) {
            d[0][0] = gemm_dst(l.w_off, l.in_dim, 0, W, NK_COL_IDENT, pex0, W, 0);
            d[0][1] = gemm_dst(l.w_off, l.in_dim, 0, W, NK_COL_PEX, 0, 64, p.Lx, p.inc_x);
            queue_gemm(q, dzl, W, nullptr, 0, act(2 + L - 1), W, act(0), 64, G, grad, d, l.b_off, -1);
        } else {
            d[0][0] = gemm_dst(l.w_off, l.in_dim, 0, W, NK_COL_IDENT, 0, W, 0);
            queue_gemm(q, dzl, W, nullptr, 0, act(2 + L - 1), W, nullptr, 0, G, grad, d, l.b_off, -1);
        }
    }
    return NK_OK;
}

// GEMMs of one or two networks whose chains have run: one launch for all of them (two networks
// in one launch is what fills the chip when a training batch is small)
int nk_mlp_backward_gemms(const NkBwdJob* jobs, int count, hipStream_t st) {
    if (count < 1 || count > 2) NK_FAIL(NK_ERR_ARG, "nk_mlp_backward_gemms: %d networks", count);
    GemmQueue q;
    if (count == 2 && (jobs[0].act_format != jobs[1].act_format || jobs[0].dz_scale != jobs[1].dz_scale))
        NK_FAIL(NK_ERR_ARG, "nk_mlp_backward_gemms: networks sharing a launch must share the saved-tensor format");
    q.fa = nk_dz_elem_bytes(jobs[0].act_format); q.fb = nk_act_elem_bytes(jobs[0].act_format);
    q.scale = q.fa == 1 ? 1.0f / jobs[0].dz_scale : 1.0f;
    for (int k = 0; k < count; ++k) {
        int rc = queue_network_gemms(q, jobs[k]);
        if (rc < 0) return rc;
    }
    // the partial-sum scratch sits behind the first network's dZ regions (see nk_dz_bytes)
    NkPlan p0;
    int rc = nk_build_plan(jobs[0].arch, &p0);
    if (rc < 0) return rc;
    float* partial = (float*)((char*)jobs[0].dz + dz_region_bytes(p0, jobs[0].R, jobs[0].n, q.fa));
    int64_t partial_bytes = gemm_partial_bytes(p0.W);
    if (count > 1) {        // products of both networks share launches: slices sized for the wider one
        NkPlan p1;
        if ((rc = nk_build_plan(jobs[1].arch, &p1)) < 0) return rc;
        if (p1.W > p0.W) {
            partial = (float*)((char*)jobs[1].dz + dz_region_bytes(p1, jobs[1].R, jobs[1].n, q.fa));
            partial_bytes = gemm_partial_bytes(p1.W);
        }
    }
    static const int per_shape = []() { const char* e = getenv("NK_GEMM_PER_SHAPE"); return e ? atoi(e) : 0; }();
    if (per_shape) {            // one launch per tile shape: how the cost column of NK_GEMM_SHAPES is measured
        bool done[NK_GEMM_MAX_PRODUCTS] = {};
        for (int first = 0; first < q.n; ++first) {
            if (done[first]) continue;
            GemmQueue sub;
            sub.fa = q.fa; sub.fb = q.fb; sub.scale = q.scale;
            for (int k = first; k < q.n; ++k)
                if (!done[k] && q.out[k].mt == q.out[first].mt && q.out[k].nt == q.out[first].nt) {
                    sub.src[sub.n] = q.src[k]; sub.out[sub.n] = q.out[k]; ++sub.n; done[k] = true;
                }
            if ((rc = flush_gemms(sub, partial, partial_bytes, st)) < 0) return rc;
        }
        return NK_OK;
    }
    return flush_gemms(q, partial, partial_bytes, st);
}

extern "C" int nk_mlp_backward(const nk_arch* arch, const void* packed, const void* act_save, const float* d_raw,
                               int64_t R, int32_t n, void* dz_scratch, float* grad, void* stream) {
    const NkBwdJob j = {arch, packed, act_save, d_raw, R, n, dz_scratch, grad, NK_ACT_BF16, 1.0f};
    int rc = nk_mlp_backward_chain(j, nk_stream(stream));
    if (rc < 0) return rc;
    return nk_mlp_backward_gemms(&j, 1, nk_stream(stream));
}
