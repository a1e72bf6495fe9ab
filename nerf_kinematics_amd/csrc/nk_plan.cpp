// Host side of the network plan: turns an nk_arch into parameter offsets, k-segment lists and
// fragment-stream block ranges.  Parameter order follows the state-dict order of the two
// nerf-pytorch lineages (key names as stored in
// /root/reference/trained_models/nerf-pytorch/lego-lowres3/checkpoint110000.ckpt for FLEXIBLE).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "nk_common.h"

static thread_local char g_err[512] = "";

void nk_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int nk_waves_per_wg(int backward) {
    static int nw[2] = {0, 0};
    if (nw[0] == 0) {
        // NK_WAVES=4|8 sets both kernels, NK_WAVES_FWD / NK_WAVES_BWD one of them
        const char* e = getenv("NK_WAVES");
        const char* ef = getenv("NK_WAVES_FWD");
        const char* eb = getenv("NK_WAVES_BWD");
        const int all = e ? atoi(e) : 0;
        const int f = ef ? atoi(ef) : all, b = eb ? atoi(eb) : all;
        nw[1] = (b == 4) ? 4 : ((b == 8) ? 8 : NK_WAVES_BWD_DEFAULT);
        nw[0] = (f == 8) ? 8 : 4;
    }
    return nw[backward ? 1 : 0];
}

extern "C" const char* nk_last_error(void) { return g_err; }
extern "C" int nk_version(void) { return 103; }

int nk_build_plan(const nk_arch* a, NkPlan* p) {
    if (!a || !p) NK_FAIL(NK_ERR_ARG, "nk_build_plan: null argument");
    const int W = a->width, D = a->depth, WD = a->dir_width;
    if (!((W == 256 && WD == 128) || (W == 128 && WD == 64)))
        NK_FAIL(NK_ERR_UNSUPPORTED, "unsupported widths: width=%d dir_width=%d (compiled: 256/128, 128/64)", W, WD);
    if (D < 2 || D > 8) NK_FAIL(NK_ERR_UNSUPPORTED, "unsupported depth %d (2..8)", D);
    if (a->skip < -1 || a->skip > D - 2) NK_FAIL(NK_ERR_UNSUPPORTED, "skip %d outside [-1, depth-2]", a->skip);
    if (a->n_freq_xyz < 0 || a->n_freq_xyz > 10 || a->n_freq_dir < 0 || a->n_freq_dir > 4)
        NK_FAIL(NK_ERR_UNSUPPORTED, "encoding frequencies xyz=%d dir=%d outside 0..10 / 0..4", a->n_freq_xyz, a->n_freq_dir);
    if (a->lineage != NK_LINEAGE_RUN_NERF && a->lineage != NK_LINEAGE_FLEXIBLE)
        NK_FAIL(NK_ERR_ARG, "unknown lineage %d", a->lineage);
    if (a->lineage == NK_LINEAGE_FLEXIBLE && a->skip != -1)
        NK_FAIL(NK_ERR_UNSUPPORTED, "FLEXIBLE lineage is only evidenced without a skip connection");

    memset(p, 0, sizeof(*p));
    p->D = D; p->W = W; p->WD = WD; p->skip = a->skip; p->Lx = a->n_freq_xyz; p->Ld = a->n_freq_dir;
    p->relu_first = a->relu_first != 0; p->relu_feat = a->relu_feat != 0; p->lineage = a->lineage;
    p->no_dirs = a->no_viewdirs != 0;
    p->NL = D + 4;
    if (a->enc_flags & ~(NK_ENC_NO_INPUT_XYZ | NK_ENC_NO_INPUT_DIR | NK_ENC_LINEAR_XYZ | NK_ENC_LINEAR_DIR))
        NK_FAIL(NK_ERR_ARG, "unknown enc_flags 0x%x", a->enc_flags);
    p->inc_x = !(a->enc_flags & NK_ENC_NO_INPUT_XYZ); p->inc_d = !(a->enc_flags & NK_ENC_NO_INPUT_DIR);
    p->lin_x = (a->enc_flags & NK_ENC_LINEAR_XYZ) != 0; p->lin_d = (a->enc_flags & NK_ENC_LINEAR_DIR) != 0;
    if ((!p->inc_x && p->Lx == 0) || (!p->inc_d && p->Ld == 0 && !p->no_dirs))
        NK_FAIL(NK_ERR_UNSUPPORTED, "an encoding without the raw input needs at least one frequency");
    // torch.linspace(1, 2^(L-1), L): step in float32 as torch computes it (a single element is the start)
    p->end_x = p->Lx > 0 ? (float)(1 << (p->Lx - 1)) : 1.0f; p->step_x = p->Lx > 1 ? (p->end_x - 1.0f) / (float)(p->Lx - 1) : 0.0f;
    p->end_d = p->Ld > 0 ? (float)(1 << (p->Ld - 1)) : 1.0f; p->step_d = p->Ld > 1 ? (p->end_d - 1.0f) / (float)(p->Ld - 1) : 0.0f;
    const int in_xyz = 3 * p->inc_x + 6 * p->Lx, in_dir = 3 * p->inc_d + 6 * p->Ld;
    const int L_ALPHA = D, L_FEAT = D + 1, L_VIEWS = D + 2, L_RGB = D + 3;

    // shapes and forward segments
    for (int i = 0; i < D; ++i) {
        NkLayer& l = p->layer[i];
        l.out_dim = W;
        if (i == 0) {
            l.in_dim = in_xyz; l.n_seg = 1;
            l.seg_type[0] = NK_SEG_PEX; l.seg_steps[0] = 4; l.seg_col[0] = 0;
        } else if (a->skip >= 0 && i == a->skip + 1) {
            l.in_dim = in_xyz + W; l.n_seg = 2;
            l.seg_type[0] = NK_SEG_PEX; l.seg_steps[0] = 4; l.seg_col[0] = 0;
            l.seg_type[1] = NK_SEG_ACT; l.seg_steps[1] = W / 16; l.seg_col[1] = in_xyz;
        } else {
            l.in_dim = W; l.n_seg = 1;
            l.seg_type[0] = NK_SEG_ACT; l.seg_steps[0] = W / 16; l.seg_col[0] = 0;
        }
    }
    {   // (without view directions: the output layer, rows rgb_raw, sigma_raw)
        NkLayer& l = p->layer[L_ALPHA]; l.out_dim = p->no_dirs ? 4 : 1; l.in_dim = W; l.n_seg = 1;
        l.seg_type[0] = NK_SEG_ACT; l.seg_steps[0] = W / 16; l.seg_col[0] = 0;
    }
    {
        NkLayer& l = p->layer[L_FEAT]; l.out_dim = W; l.in_dim = W; l.n_seg = 1;
        l.seg_type[0] = NK_SEG_ACT; l.seg_steps[0] = W / 16; l.seg_col[0] = 0;
    }
    {
        NkLayer& l = p->layer[L_VIEWS]; l.out_dim = WD; l.in_dim = W + in_dir; l.n_seg = 2;
        l.seg_type[0] = NK_SEG_ACT; l.seg_steps[0] = W / 16; l.seg_col[0] = 0;
        l.seg_type[1] = NK_SEG_PED; l.seg_steps[1] = 2; l.seg_col[1] = W;
    }
    {
        NkLayer& l = p->layer[L_RGB]; l.out_dim = 3; l.in_dim = WD; l.n_seg = 1;
        l.seg_type[0] = NK_SEG_ACT; l.seg_steps[0] = WD / 16; l.seg_col[0] = 0;
    }

    // parameter offsets in state-dict order.  Without view directions the three layers of the
    // direction branch keep their place in the streams (same stage sequence, all-zero blocks) but
    // own no parameters.
    for (int i = 0; i < p->NL; ++i) p->layer[i].has_params = 1;
    int order[NK_MAX_LAYERS], n = 0;
    for (int i = 0; i < D; ++i) order[n++] = i;
    if (p->no_dirs) {
        order[n++] = L_ALPHA;
        p->layer[L_FEAT].has_params = p->layer[L_VIEWS].has_params = p->layer[L_RGB].has_params = 0;
        p->layer[L_FEAT].w_off = p->layer[L_VIEWS].w_off = p->layer[L_RGB].w_off = -1;
        p->layer[L_FEAT].b_off = p->layer[L_VIEWS].b_off = p->layer[L_RGB].b_off = -1;
    } else if (a->lineage == NK_LINEAGE_RUN_NERF) {
        order[n++] = L_VIEWS; order[n++] = L_FEAT; order[n++] = L_ALPHA; order[n++] = L_RGB;
    } else {
        order[n++] = L_VIEWS; order[n++] = L_ALPHA; order[n++] = L_RGB; order[n++] = L_FEAT;
    }
    int off = 0;
    for (int k = 0; k < n; ++k) {
        NkLayer& l = p->layer[order[k]];
        l.w_off = off; off += l.out_dim * l.in_dim;
        l.b_off = off; off += l.out_dim;
    }
    p->n_params = off;

    // forward stream: layer by layer, tile by tile; a tile = nk weight blocks + 1 bias block
    int blk = 0;
    for (int i = 0; i < p->NL; ++i) {
        NkLayer& l = p->layer[i];
        l.n_tiles = (l.out_dim + 31) / 32;
        l.nk = 0;
        for (int s = 0; s < l.n_seg; ++s) l.nk += l.seg_steps[s];
        l.fwd_block0 = blk;
        blk += l.n_tiles * (l.nk + 1);
    }
    p->fwd_blocks = blk;

    // transposed stream, in the order the backward chain consumes it
    blk = 0;
    auto add_bwd = [&](int li, int tiles, int nk, int act_col) {
        NkLayer& l = p->layer[li];
        l.bwd_tiles = tiles; l.bwd_nk = nk; l.bwd_act_col = act_col; l.bwd_block0 = blk;
        blk += tiles * nk;
    };
    if (!p->no_dirs) {      // (without view directions the chain starts at the stage below, on dOut alone)
        add_bwd(L_RGB, WD / 32, 1, 0);
        add_bwd(L_VIEWS, W / 32, WD / 16, 0);
    }
    add_bwd(L_FEAT, W / 32, W / 16 + 1, 0);
    for (int i = D - 1; i >= 1; --i)
        add_bwd(i, W / 32, W / 16, (a->skip >= 0 && i == a->skip + 1) ? in_xyz : 0);
    p->bwd_blocks = blk;

    return NK_OK;
}

extern "C" int64_t nk_num_params(const nk_arch* arch) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    return rc < 0 ? rc : p.n_params;
}

extern "C" int64_t nk_packed_bytes(const nk_arch* arch) {
    NkPlan p;
    int rc = nk_build_plan(arch, &p);
    // forward stream, transposed stream, and the forward stream again in the 16x16x32 form of the inference kernel
    return rc < 0 ? rc : (int64_t)(2 * p.fwd_blocks + p.bwd_blocks) * 1024;
}
