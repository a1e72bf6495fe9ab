/*
 * nerf_kinematics.h -- C ABI of the MI355X (gfx950) NeRF volume-rendering path.
 *
 * The reference project (tomeupm/nerf-kinematics) has no FFI or plugin interface for this
 * path: its renderer is Python code in an un-vendored nerf-pytorch submodule
 * (/root/reference/.gitmodules:10-12) that calls PyTorch tensor ops.  The boundary replaced here
 * is therefore the function contract of that code's `render_rays` (README.md:90-93 shows the
 * only call site, `python run_nerf.py --config ...`), parameterised by the `nerf.train` /
 * `nerf.validation` blocks of the experiment files
 * (/root/reference/trained_models/nerf-pytorch/lego-lowres3/config.yml:46-63) and the `models`
 * block (config.yml:18-42).  Each entry point below names the stage of that code it replaces.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer borrowed for the call unless the comment says "host";
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream), allocates nothing and keeps no global state besides an error string;
 *   - return value: NK_OK or a negative nk_status; nk_last_error() gives the text (per thread);
 *   - parameters of a network are ONE flat fp32 array in the state-dict order of the lineage
 *     (see nk_arch.lineage); gradients use the same layout.
 */
#ifndef NERF_KINEMATICS_H
#define NERF_KINEMATICS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum nk_status {
    NK_OK = 0,
    NK_ERR_ARG = -1,          /* null pointer, non-positive size, misaligned buffer */
    NK_ERR_UNSUPPORTED = -2,  /* network shape / sample count outside the compiled kernels */
    NK_ERR_HIP = -3,          /* a HIP call or launch failed */
    NK_ERR_WORKSPACE = -4     /* workspace smaller than nk_render_workspace_bytes() */
} nk_status;

/* Network lineages (parameter order and activation placement):
 *   NK_LINEAGE_RUN_NERF  pts_linears.0..D-1 (ReLU each; layer skip+1 takes cat[gamma(x), h]),
 *                        views_linears.0, feature_linear, alpha_linear, rgb_linear
 *   NK_LINEAGE_FLEXIBLE  layer1, layers_xyz.0..D-2, layers_dir.0, fc_alpha, fc_rgb, fc_feat
 *                        (the key order of checkpoint110000.ckpt in
 *                        /root/reference/trained_models/nerf-pytorch/lego-lowres3/)            */
enum { NK_LINEAGE_RUN_NERF = 0, NK_LINEAGE_FLEXIBLE = 1 };
/* Without view directions (nk_arch.no_viewdirs, `use_viewdirs: False` of config.yml:55) both
 * lineages end in ONE linear layer on the last trunk activation whose four rows are
 * (rgb_raw, sigma_raw): pts_linears.0..D-1, output_linear / layer1, layers_xyz.0..D-2, fc_out.
 * dir_width and n_freq_dir are then ignored apart from the width pairing the kernels are
 * compiled for, and columns 8..10 of a ray-batch row are never read.                          */

/* Variants of the positional encoding (include_input_xyz / _dir, log_sampling_xyz / _dir of
 * config.yml:21-24; both reference files set all four True = no bit set):
 *   NK_ENC_NO_INPUT_*  gamma() does not start with the raw 3-vector (input width 6 L instead of 3 + 6 L)
 *   NK_ENC_LINEAR_*    frequencies torch.linspace(1, 2^(L-1), L) instead of 2^0 .. 2^(L-1)             */
enum { NK_ENC_NO_INPUT_XYZ = 1, NK_ENC_NO_INPUT_DIR = 2, NK_ENC_LINEAR_XYZ = 4, NK_ENC_LINEAR_DIR = 8 };

/* models.coarse / models.fine of config.yml:18-42 */
typedef struct nk_arch {
    int32_t depth;       /* trunk layers D, first layer included            (8 | 4)   */
    int32_t width;       /* trunk width W                                   (256 | 128) */
    int32_t skip;        /* trunk layer skip+1 takes cat[gamma(x), h]; -1 = none (4 | -1) */
    int32_t dir_width;   /* hidden width of the view-direction branch       (128 | 64) */
    int32_t n_freq_xyz;  /* num_encoding_fn_xyz, <= 10                                 */
    int32_t n_freq_dir;  /* num_encoding_fn_dir, <= 4                                  */
    int32_t relu_first;  /* ReLU after the first trunk layer                           */
    int32_t relu_feat;   /* ReLU after the feature layer                               */
    int32_t lineage;     /* NK_LINEAGE_*                                               */
    int32_t no_viewdirs; /* 1: no view-direction branch (use_viewdirs: False)                  */
    int32_t enc_flags;   /* NK_ENC_* bits; 0 = include_input_*: True, log_sampling_*: True     */
} nk_arch;

/* Per-iteration scalars of a training step kept in DEVICE memory, so that a step whose launches
 * were captured into a HIP graph can be replayed with new values (kernel arguments are frozen at
 * capture; these are read by the kernels when they run).  The caller writes one of these before
 * each replay (a 32-byte copy on the same stream).  No counterpart in the reference.           */
typedef struct nk_step_state {
    uint32_t render_seed;         /* what nk_render_opts.seed would be this iteration      */
    uint32_t pixel_seed;          /* seed of nk_sample_training_rays                       */
    uint64_t pixel_counter;       /* counter of nk_sample_training_rays                    */
    int32_t  view;                /* index into the camera table / image stack             */
    float    adam_step_size;      /* (float)(lr / (1 - beta1^step))                        */
    float    adam_bc2_sqrt;       /* (float)sqrt(1 - beta2^step)                           */
    int32_t  reserved;
} nk_step_state;

/* Formats of the tensors a training forward saves for backward (nk_render_opts.act_format).
 * No counterpart in the reference (torch autograd keeps fp32 activations).  NK_ACT_BF16 is the
 * default; the two 8-bit forms are experiments that halve the bytes the training step moves
 * through HBM at the price of a coarser weight gradient:
 *   NK_ACT_FP8      layer inputs saved as OCP e4m3 (rounded once from the fp32 accumulator; the forward
 *                   chain itself still runs on bf16 operands, so outputs do not change), dZ as bf16
 *   NK_ACT_FP8_BF8  the same, and dZ saved as OCP e5m2 scaled by a power of two that the weight-gradient
 *                   reduction divides out again (the backward chain itself still runs on bf16)      */
enum { NK_ACT_BF16 = 0, NK_ACT_FP8 = 1, NK_ACT_FP8_BF8 = 2 };

/* nerf.train / nerf.validation of config.yml:46-63 */
typedef struct nk_render_opts {
    int32_t num_coarse;
    int32_t num_fine;             /* 0: coarse pass only                                  */
    int32_t perturb;              /* stratified jitter + random u in sample_pdf           */
    int32_t lindisp;
    int32_t white_background;
    float   radiance_field_noise_std;
    uint32_t seed;                /* seed of the counter-based RNG streams                */
    int32_t  act_format;          /* NK_ACT_*: training calls only; forward and backward must agree */
    const nk_step_state* state;   /* device pointer or NULL; when set, the kernels take the seed
                                     from state->render_seed when they run and `seed` is ignored */
    float    skip_empty_eps;      /* inference calls only (ignored when training): > 0 skips the fine
                                     network for every group of 32 consecutive fine samples whose
                                     neighbouring coarse samples all have compositing weight below this
                                     value; their rows are emitted with zero density.  0 = evaluate
                                     every sample (the reference's behaviour).  Experiment.          */
    int32_t  reserved;
} nk_render_opts;

/* Camera for nk_get_rays: intrinsics as in transforms.json (fl_x, fl_y, cx, cy) and a 3x4
 * camera-to-world matrix, row major.  Blender/OpenGL convention.                            */
typedef struct nk_camera {
    int32_t H, W;
    float fx, fy, cx, cy;
    float c2w[12];
    float near, far;
    int32_t ndc;                  /* forward-facing NDC warp (dataset.no_ndc: false)      */
    float ndc_near;               /* near plane of the warp, 1.0 in nerf-pytorch          */
} nk_camera;

/* Outputs of render_rays.  rgb: [R,3]; the others [R].  *_coarse may be NULL when unwanted;
 * *_fine must be NULL when num_fine == 0.                                                  */
typedef struct nk_render_out {
    float *rgb_fine, *disp_fine, *acc_fine, *depth_fine;
    float *rgb_coarse, *disp_coarse, *acc_coarse, *depth_coarse;
} nk_render_out;

/* Version of this header: the structs above changed layout at 101 (nk_render_opts), 102 and 103 (nk_arch). */
int nk_version(void);
const char* nk_last_error(void);

/* ---- model ------------------------------------------------------------------------------ */
/* Number of fp32 parameters of `arch` (weights and biases), or a negative nk_status.        */
int64_t nk_num_params(const nk_arch* arch);
/* Bytes of the bf16 MFMA-fragment image of one network (forward and transposed streams of the
 * training kernels, and the forward stream in the operand order of the inference kernel).  */
int64_t nk_packed_bytes(const nk_arch* arch);
/* fp32 master parameters -> bf16 fragment streams consumed by the MLP kernels.
 * Replaces nothing in the reference (torch keeps fp32 nn.Linear weights); call after every
 * optimiser step.                                                                           */
int nk_pack_weights(const nk_arch* arch, const float* params, void* packed, void* stream);

/* ---- stages (each replaces one function of the nerf-pytorch render path) ------------------ */
/* get_rays + ndc_rays + ray-batch assembly: rows [o(3) d(3) near far viewdir(3)] (11 floats).
 * pixel_idx: R flat pixel indices (row*W+col) or NULL for pixels first_pixel..first_pixel+R-1 */
int nk_get_rays(const nk_camera* cam /*host*/, const int32_t* pixel_idx, int64_t first_pixel,
                int64_t R, float* ray_batch, void* stream);
/* Ray selection of one training iteration on one view (the train loop's random pixel choice,
 * get_rays and target gather; config.yml:51 num_random_rays): R pixels drawn with replacement
 * from RNG stream 4 at element `counter` + r, their ray-batch rows, and their colours from the
 * 8-bit H x W x channels image (channels 3 | 4; with 4 and white_background the colour is
 * composited onto white, otherwise alpha is ignored).  pixel_idx (optional) receives the draw. */
int nk_sample_training_rays(const nk_camera* cam /*host*/, const uint8_t* image, int32_t channels,
                            int32_t white_background, uint32_t seed, uint64_t counter, int64_t R,
                            float* ray_batch, float* target, int32_t* pixel_idx, void* stream);
/* The same with everything that changes per iteration read on the device: cams = table of
 * nk_camera in device memory (all H x W equal), images = stack of 8-bit images image_stride bytes
 * apart; view, seed and counter come from *state when the kernel runs (graph replay).          */
int nk_sample_training_rays_dev(const nk_camera* cams /*device*/, int32_t n_views, const uint8_t* images,
                                int64_t image_stride, int32_t channels, int32_t white_background,
                                const nk_step_state* state /*device*/, int64_t R, float* ray_batch,
                                float* target, int32_t* pixel_idx, void* stream);
/* Coarse depths: linspace in depth or disparity + stratified jitter (render_rays, first part) */
int nk_stratified_z(const float* ray_batch, int64_t R, const nk_render_opts* opts, float* z,
                    void* stream);
/* Embedder + NeRF.forward for n samples per ray: raw[R,n,4] = (rgb_raw, sigma_raw).
 * act_save: NULL (inference) or nk_act_bytes() bytes receiving the activations for backward. */
int64_t nk_act_bytes(const nk_arch* arch, int64_t R, int32_t n);
int nk_mlp_forward(const nk_arch* arch, const void* packed, const float* ray_batch,
                   const float* z, int64_t R, int32_t n, float* raw, void* act_save,
                   void* stream);
/* raw2outputs.  noise_stream: RNG stream id of the sigma noise (2 coarse, 3 fine).
 * Any of disp/acc/depth/weights may be NULL.                                                */
int nk_raw2outputs(const float* raw, const float* z, const float* ray_batch, int64_t R,
                   int32_t n, const nk_render_opts* opts, int32_t noise_stream, float* rgb,
                   float* disp, float* acc, float* depth, float* weights, void* stream);
/* sample_pdf on the interior coarse weights + sort(cat[z, z_samples]) -> z_all[R,Nc+Nf].
 * z_samples (optional) receives the unsorted new samples [R,Nf].  With z_coarse ascending along
 * the ray, as nk_stratified_z (and the reference's depth sampling) produces it, only the new
 * samples are sorted and the two lists are merged; a ray whose coarse depths are not ascending
 * (or hold a NaN) is detected and sorted whole, like the reference's torch.sort of the
 * concatenation (NaNs then end up in unspecified slots).                                      */
int nk_sample_pdf(const float* z_coarse, const float* weights, int64_t R,
                  const nk_render_opts* opts, float* z_all, float* z_samples, void* stream);
/* Backward of raw2outputs w.r.t. raw, given d(loss)/d(rgb_map) [R,3] (d_acc, d_depth optional,
 * NULL = zero).  The sample_pdf inputs are detached, as in the reference.                   */
int nk_raw2outputs_bwd(const float* raw, const float* z, const float* ray_batch, int64_t R,
                       int32_t n, const nk_render_opts* opts, int32_t noise_stream,
                       const float* d_rgb, const float* d_acc, const float* d_depth,
                       float* d_raw, void* stream);
/* Backward of the MLP: grad (flat fp32, same layout as params) += dL/dparams.
 * dz_scratch: nk_dz_bytes() bytes of scratch.                                               */
int64_t nk_dz_bytes(const nk_arch* arch, int64_t R, int32_t n);
int nk_mlp_backward(const nk_arch* arch, const void* packed, const void* act_save,
                    const float* d_raw, int64_t R, int32_t n, void* dz_scratch, float* grad,
                    void* stream);
/* img2mse on both maps: loss[0] += mean((rgb-target)^2) for each non-NULL rgb; d_rgb = its
 * gradient times grad_scale.                                                                */
int nk_mse_loss(const float* rgb_coarse, const float* rgb_fine, const float* target, int64_t R,
                float grad_scale, float* loss, float* d_rgb_coarse, float* d_rgb_fine,
                void* stream);
/* torch.optim.Adam update (the optimiser stored in the reference checkpoints) on flat buffers */
int nk_adam_step(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                 int64_t step, double lr, double beta1, double beta2, double eps, float grad_scale,
                 void* stream);

/* *state_dev = *state_host, stream-ordered: the struct travels as a kernel argument (copied at the
 * call), so the host copy may be overwritten as soon as the call returns.                      */
int nk_set_step_state(const nk_step_state* state_host, nk_step_state* state_dev, void* stream);
/* nk_adam_step with the two step-dependent scalars read from *state on the device (graph replay). */
int nk_adam_step_dev(float* params, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n,
                     const nk_step_state* state /*device*/, double beta1, double beta2, double eps,
                     float grad_scale, void* stream);

/* ---- render_rays -------------------------------------------------------------------------- */
/* Workspace for R rays; `training` != 0 adds the activation and gradient scratch.           */
int64_t nk_render_workspace_bytes(const nk_arch* coarse, const nk_arch* fine,
                                  const nk_render_opts* opts, int64_t R, int training);
/* render_rays forward: stratified z -> coarse MLP -> raw2outputs -> sample_pdf -> fine MLP ->
 * raw2outputs.  packed_fine / arch_fine may be NULL when opts->num_fine == 0.               */
int nk_render_rays_fwd(const nk_arch* arch_coarse, const void* packed_coarse,
                       const nk_arch* arch_fine, const void* packed_fine,
                       const float* ray_batch, int64_t R, const nk_render_opts* opts,
                       int training, void* workspace, int64_t workspace_bytes,
                       const nk_render_out* out, void* stream);
/* render_rays backward from d(loss)/d(rgb_fine), d(loss)/d(rgb_coarse) (either may be NULL).
 * Must follow nk_render_rays_fwd(training=1) on the same workspace.  grad_* accumulate.     */
int nk_render_rays_bwd(const nk_arch* arch_coarse, const void* packed_coarse,
                       const nk_arch* arch_fine, const void* packed_fine,
                       const float* ray_batch, int64_t R, const nk_render_opts* opts,
                       void* workspace, int64_t workspace_bytes, const float* d_rgb_fine,
                       const float* d_rgb_coarse, float* grad_coarse, float* grad_fine,
                       void* stream);

/* ---- profiling hook ----------------------------------------------------------------------- */
/* No counterpart in the reference.  events: `count` hipEvent_t handles created by the caller
 * (host array, copied; entries may be NULL).  Until it is called again with count 0, every
 * nk_render_rays_fwd / _bwd call of THIS thread records events[i] on its stream at stage
 * boundary i, so a caller can time the MLP launches of the shipped path without re-issuing the
 * stages one by one.  With shared weight-gradient launches (small batches) the fine network's
 * GEMMs run after the coarse chain: NK_EV_BWD_FINE_GEMMS_END then coincides with
 * NK_EV_BWD_FINE_CHAIN_END.                                                                  */
enum {
    NK_EV_FWD_COARSE_BEGIN = 0, NK_EV_FWD_COARSE_END = 1,     /* coarse Embedder + MLP forward   */
    NK_EV_FWD_FINE_BEGIN = 2, NK_EV_FWD_FINE_END = 3,         /* fine Embedder + MLP forward     */
    NK_EV_BWD_FINE_CHAIN_BEGIN = 4, NK_EV_BWD_FINE_CHAIN_END = 5,
    NK_EV_BWD_FINE_GEMMS_END = 6,
    NK_EV_BWD_COARSE_CHAIN_BEGIN = 7, NK_EV_BWD_COARSE_CHAIN_END = 8,
    NK_EV_BWD_END = 9,
    NK_STAGE_EVENT_COUNT = 10
};
int nk_set_stage_events(void** events, int count);

/* ---- render_rays for callers that cannot allocate ------------------------------------------- */
/* No counterpart in the reference.  For callers that cannot call hipMalloc or pass a workspace:
 * code under HIP graph capture, tight serving loops, harnesses that forbid allocation.  Same
 * result as nk_render_rays_fwd(training = 0): the stage buffers (depths, raw network outputs,
 * weights; NK_NOALLOC_SCRATCH_BYTES at most) live in a device array that is part of the library
 * image, one copy per GPU, held until the process ends.  The device the stream belongs to must be
 * current.  Fails with NK_ERR_WORKSPACE when R rays at opts' sample counts need more than that
 * (about 5.4 KB per ray at 64 + 128 samples: 12 000 rays); allocates and frees nothing.  All calls
 * on one GPU share the array: calls that overlap on one GPU (two streams, two threads) race.
 * The first call on a device looks the array's address up (hipGetSymbolAddress, which may load the
 * code object): make one call on each device OUTSIDE stream capture before capturing calls into a
 * HIP graph; later calls only launch kernels.                                                   */
#define NK_NOALLOC_SCRATCH_BYTES (64u << 20)
int nk_render_rays_fwd_noalloc(const nk_arch* arch_coarse, const void* packed_coarse,
                               const nk_arch* arch_fine, const void* packed_fine,
                               const float* ray_batch, int64_t R, const nk_render_opts* opts,
                               const nk_render_out* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NERF_KINEMATICS_H */
