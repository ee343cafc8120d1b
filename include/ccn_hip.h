/*
 * ccn_hip.h -- C ABI of libccn_hip.so: the MI355X (gfx950) DDIM reconstruction path of
 * clip-feature-codec (CLIPCondUNet epsilon-prediction forward + DDIM update).
 *
 * The reference (lionl1106/Clip-Neural-image-conpression) is pure Python on torch.nn and has no
 * FFI of its own; every entry point below names the reference interface it stands in for
 * (paths relative to src/clip_feature_codec/).  A reference maintainer binds them with ctypes
 * (see INTEGRATION.md); nothing here carries a torch type.
 *
 * Conventions
 *   - every function returns 0 on success, a CCN_E* code otherwise; ccn_last_error() returns a
 *     thread-local message for the last failure.  No C++ exception crosses the boundary.
 *   - pointers named *_dev are device (HBM) pointers owned by the caller; *_host are host pointers.
 *   - `stream` is a hipStream_t passed as void* (0 = the null stream).  Calls enqueue work and return;
 *     they never synchronise the device, except where stated.
 *   - image tensors at the boundary are NCHW fp32 contiguous, exactly what the reference passes to
 *     CLIPCondUNet.forward; NHWC and (in bf16 mode) bf16 are internal.
 *   - one handle per process per device; handles are not thread-safe.
 */
#ifndef CCN_HIP_H
#define CCN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CCN_OK            0
#define CCN_EINVAL        1   /* bad argument / unsupported shape            */
#define CCN_EHIP          2   /* a HIP runtime call failed                   */
#define CCN_EWEIGHTS      3   /* missing / unexpected / mis-shaped parameter */
#define CCN_EWORKSPACE    4   /* workspace too small or misaligned           */
#define CCN_ESTATE        5   /* call order violated                         */

#define CCN_DTYPE_F32     0   /* fp32 storage, fp32 MFMA (v_mfma_f32_32x32x2_f32): parity mode        */
#define CCN_DTYPE_BF16    1   /* bf16 storage, bf16 MFMA (v_mfma_f32_16x16x32_bf16 / 32x32x16), fp32 accumulate */

#define CCN_MAX_MULT      8

typedef struct ccn_handle_s* ccn_handle_t;

/* Constructor arguments of CLIPCondUNet (models/unet.py:45) plus the arithmetic mode. */
typedef struct ccn_config {
    int32_t z_dim;                 /* 512                                         */
    int32_t base;                  /* 128                                         */
    int32_t n_mult;                /* len(ch_mult)                                */
    int32_t ch_mult[CCN_MAX_MULT]; /* (1,2,2); widths are a running product       */
    int32_t time_dim;              /* 256                                         */
    int32_t img_ch;                /* 3                                           */
    int32_t groups;                /* GroupNorm groups, 8 (models/blocks.py:31)   */
    int32_t dtype;                 /* CCN_DTYPE_*                                 */
} ccn_config_t;

/* ---- lifetime ------------------------------------------------------------------------------- */

/* CLIPCondUNet.__init__ + .to(device) (models/unet.py:45-79; cli/eval.py:50). Uses the current HIP device. */
int ccn_create(const ccn_config_t* cfg, ccn_handle_t* out);
int ccn_destroy(ccn_handle_t h);

/* Number of state-dict entries the handle expects, and the i-th key / shape: the key set of
 * CLIPCondUNet.state_dict() (192 entries at base=128, ch_mult=(1,2,2)). */
int ccn_num_params(ccn_handle_t h, int32_t* n);
int ccn_param_info(ccn_handle_t h, int32_t i, const char** name, int64_t shape[4], int32_t* ndim);

/* One entry of load_state_dict (cli/eval.py:51, cli/reconstruct_diffusion.py:48).  `data` is fp32,
 * contiguous, in the reference's own layout (Conv2d OIHW, ConvTranspose2d (Cin,Cout,4,4), Linear
 * (out,in)); host or device pointer.  The library copies; the caller keeps ownership. */
int ccn_load_param(ccn_handle_t h, const char* name, const float* data, const int64_t* shape, int32_t ndim);

/* How CCN_DTYPE_BF16 rounds the conv weights to bf16 when they are repacked (call before ccn_commit_params; no effect in fp32
 * mode).  A rounded weight is a static perturbation of the model that acts the same way in every one of the sampler's steps, and
 * that coherent accumulation is what moves the 50-step reconstructions: independent rounding loses 0.33 % contrast, 0.165 % PSNR
 * (max per record) against the fp32 reference path -- above north_star's 0.1 % gate (eval/metrics.py:22-29).
 *   CCN_ROUND_NEAREST          independent round-to-nearest-even, what `.to(torch.bfloat16)` of the checkpoint gives;
 *   CCN_ROUND_DIFFUSED         error diffusion along (cin, ky, kx) of every output channel: partial sums of an output channel's
 *                              weights stay within half an ulp of the fp32 sums;
 *   CCN_ROUND_DIFFUSED_PHASES  (default) the same, plus error diffusion ALONG THE DDIM STEPS: n such roundings of every weight whose
 *                              running sums track the fp32 weight, step i of ccn_sample uses version i % n (ccn_forward: version 0).
 *                              n = 8 (4 for models above 100 M conv weights): the mean weight over a period is accurate to 1/16 ulp;
 *                              costs n copies of the bf16 weights in HBM and nothing at run time. */
#define CCN_ROUND_NEAREST         0
#define CCN_ROUND_DIFFUSED        1
#define CCN_ROUND_DIFFUSED_PHASES 2
int ccn_set_weight_rounding(ccn_handle_t h, int32_t mode);

/* strict=True check (every key loaded exactly once with the right shape), then repack to the kernel
 * layouts ([tap][Cout][Cin], bf16 copies in bf16 mode) and upload.  Synchronises the device. */
int ccn_commit_params(ccn_handle_t h);

/* ---- the hot path --------------------------------------------------------------------------- */

/* Scratch needed by ccn_forward / ccn_sample for this shape; the caller allocates it (e.g. through
 * torch's caching allocator), 256-byte aligned, and keeps it alive and at the same address while a
 * captured graph for it is cached. `steps` = 1 for ccn_forward. */
int ccn_workspace_bytes(ccn_handle_t h, int32_t B, int32_t H, int32_t W, int32_t steps, size_t* bytes);

/* The caller is about to free (or reuse for something else) a workspace it handed to ccn_forward / ccn_sample: drains the
 * device, then drops every cached plan and captured graph that lives in it.  The plan cache is keyed by (B, H, W, steps,
 * workspace address) and a plan keeps state in its workspace between calls (the uploaded timestep table, the zeroed split-K
 * hand-off flags and arrival counters), so a LATER allocation that lands on the same address must not find the old plan.
 * Without this call a freed workspace must never be reused at the same address with the same shape.  (No reference
 * counterpart: torch owns all memory there; this is the ownership rule of SURVEY.md section 8b, "Python owns the workspace".) */
int ccn_release_workspace(ccn_handle_t h, void* workspace_dev);

/* CLIPCondUNet.forward(x_t, z_clip, t) (models/unet.py:81-106).
 * x_dev (B,img_ch,H,W) fp32 NCHW; z_dev (B,z_dim) fp32; t_dev (B,) int64; eps_dev (B,img_ch,H,W) fp32 NCHW. */
int ccn_forward(ccn_handle_t h, const float* x_dev, const float* z_dev, const int64_t* t_dev,
                float* eps_dev, int32_t B, int32_t H, int32_t W,
                void* workspace_dev, size_t workspace_bytes, void* stream);

/* DDIMSampler.sample(model, z_clip, shape, steps, x_T=...) for eta = 0 (diffusion/ddim.py:21-45).
 * ts_host[steps]: the timestep table (linspace(T-1,0,steps).long());
 * coef_host[steps][4]: fp32 (sqrt(1-ab_t), sqrt(ab_t), sqrt(ab_s), sqrt(ab_s - sigma^2)) per step, so that
 *     x0 = clamp((x - c0*eps)/c1, -1, 1);  x = c2*x0 + c3*eps      -- every op rounded to fp32
 * x_T_dev -> x_out_dev (may alias), (B,img_ch,H,W) fp32 NCHW, unclamped like the reference.
 * use_graph != 0: the whole steps-long loop is captured once into a hipGraph (cached per
 * shape/steps/table/workspace address) and replayed with one launch. */
int ccn_sample(ccn_handle_t h, const float* z_dev, const float* x_T_dev, float* x_out_dev,
               int32_t B, int32_t H, int32_t W, int32_t steps,
               const int32_t* ts_host, const float* coef_host,
               void* workspace_dev, size_t workspace_bytes, void* stream, int32_t use_graph);

/* The same loop for eta > 0 (diffusion/ddim.py:41-45): sigma_host[steps] fp32 per-step sigma (0 on steps without noise, e.g. the
 * last one), coef_host as above with c3 = sqrt(ab_s - sigma^2); noise_dev (steps,B,img_ch,H,W) fp32 holds the N(0,1) draws of every
 * step, made by the caller (the reference draws torch.randn_like per step: a torch caller fills slice i with its i-th draw so that
 * the fused loop consumes the generator exactly like the step-by-step loop); the update adds sigma*noise, rounded like the torch ops.
 * The captured graph is cached per (table, sigma, noise address). */
int ccn_sample_eta(ccn_handle_t h, const float* z_dev, const float* x_T_dev, float* x_out_dev,
                   int32_t B, int32_t H, int32_t W, int32_t steps,
                   const int32_t* ts_host, const float* coef_host, const float* sigma_host, const float* noise_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream, int32_t use_graph);

/* One DDIM update outside the fused loop (eta > 0, or a caller-driven loop; diffusion/ddim.py:34-45):
 * x = c2*clamp((x - c0*eps)/c1) + c3*eps [+ sigma*noise]; noise_dev may be NULL. In place on x_dev. */
int ccn_ddim_step(float* x_dev, const float* eps_dev, const float* noise_dev,
                  float c0, float c1, float c2, float c3, float sigma, int64_t n, void* stream);

/* NoiseScheduler.q_sample (diffusion/scheduler.py:46-49): out[b] = a[b]*x0[b] + s[b]*noise[b];
 * a_dev/s_dev are the gathered per-sample coefficients, (B,) fp32; per_sample = C*H*W. */
int ccn_q_sample(float* out_dev, const float* x0_dev, const float* noise_dev,
                 const float* a_dev, const float* s_dev, int32_t B, int64_t per_sample, void* stream);

/* NoiseScheduler.predict_x0_from_eps (diffusion/scheduler.py:51-55): out[b] = (x_t[b] - s[b]*eps[b]) / a[b]. */
int ccn_predict_x0(float* out_dev, const float* x_t_dev, const float* eps_dev,
                   const float* a_dev, const float* s_dev, int32_t B, int64_t per_sample, void* stream);

/* ---- operator-level entry points (what the reference's unit tests exercise) ------------------ */

/* FiLM.forward (models/blocks.py:22-25) with explicit parameters: y = x*(1 + Ws h + bs) + (Wh h + bh).
 * x/y (B,C,H,W) fp32 NCHW; h (B,D); Ws/Wh (C,D); bs/bh (C,). All device pointers.
 * scratch_dev: at least 2*B*C floats. */
int ccn_film_forward(const float* x_dev, const float* h_dev, const float* ws_dev, const float* bs_dev,
                     const float* wh_dev, const float* bh_dev, float* y_dev,
                     int32_t B, int32_t C, int32_t H, int32_t W, int32_t D,
                     float* scratch_dev, void* stream);

/* ResBlock.forward (models/blocks.py:40-44) of the block whose state-dict prefix is `prefix`
 * ("down.0", "mid1", "up.4", ...), using the handle's committed weights and arithmetic mode.
 * x/y (B,C,H,W) fp32 NCHW; h (B,time_dim) fp32.  Workspace: ccn_workspace_bytes(h,B,H,W,1) is enough
 * for any block at resolution HxW. */
int ccn_resblock_forward(ccn_handle_t h, const char* prefix, const float* x_dev, const float* cond_dev,
                         float* y_dev, int32_t B, int32_t H, int32_t W,
                         void* workspace_dev, size_t workspace_bytes, void* stream);

/* timestep_embedding(t, dim) (models/unet.py:22-39): (n,) int64 -> (n,dim) fp32, [cos | sin] order. */
int ccn_timestep_embedding(const int64_t* t_dev, float* out_dev, int32_t n, int32_t dim, void* stream);

/* ---- introspection (tests, profiling) -------------------------------------------------------- */

/* After ccn_forward / ccn_sample: copy an intermediate activation out as fp32 NCHW.  Names are the
 * reference module paths: "in_conv", "down.0" ... "mid2", "up.8" (after the skip add), and
 * "<block>.film" for the tensor entering norm2.  Synchronises `stream`. */
int ccn_read_activation(ccn_handle_t h, const char* name, float* out_dev, size_t out_elems, void* stream);

/* Device-side failures are sticky: a kernel that detects a broken hand-off (a split-K partial tile that never arrived) sets a
 * bit in the handle's error word and carries on; the NEXT call that takes the handle -- or this one, e.g. after the caller has
 * synchronised its stream -- returns CCN_EHIP once and clears it.  Results enqueued since the last successful check are then
 * invalid.  (The reference has no counterpart: torch raises asynchronous device errors the same way, at the next call.) */
int ccn_poll_errors(ccn_handle_t h);

/* Per-kernel-family device time of the next ccn_sample / ccn_forward calls, measured with HIP events
 * recorded on the launch stream around every kernel (the loop then runs launch by launch instead of
 * as one graph).  ccn_profile_read synchronises, then fills arrays of capacity `cap`: family name,
 * summed milliseconds, launch count, and the summed ALGORITHMIC flops (2*MAC) and HBM bytes of those
 * launches; *n receives the number of families. */
int ccn_profile_enable(ccn_handle_t h, int32_t on);
int ccn_profile_read(ccn_handle_t h, const char** names, float* ms, int32_t* calls,
                     double* flops, double* bytes, int32_t cap, int32_t* n);

/* Algorithmic work of one UNet forward for (B,H,W): conv/linear FLOPs (2*MAC) and minimal HBM bytes
 * under this handle's storage dtype. */
int ccn_algorithmic_work(ccn_handle_t h, int32_t B, int32_t H, int32_t W, double* flops, double* bytes);

/* ---- training step (train/diffusion_train.py:119-124,137-140) --------------------------------- *
 * The reference trains with `eps_hat = net(x_t, z, t); loss = F.mse_loss(eps_hat, noise); loss.backward(); opt.step()`.
 * A trainer handle reads the parameters from ONE flat fp32 device buffer owned by the caller (the entries of
 * CLIPCondUNet.state_dict(), in registration order, each at the offset ccn_train_param_info reports -- a torch caller makes
 * every nn.Parameter a view into it) and accumulates gradients into a second flat buffer of the same layout.
 * Arithmetic mode as for inference: CCN_DTYPE_F32 (parity) or CCN_DTYPE_BF16 (activations and conv operands in bf16,
 * fp32 accumulation, fp32 GroupNorm statistics, fp32 gradients and optimiser state: what torch.autocast(bfloat16) does
 * at train/diffusion_train.py:121). */
typedef struct ccn_trainer_s* ccn_trainer_t;

/* net = CLIPCondUNet(...).to(device); net.train() (train/diffusion_train.py:103,109) */
int ccn_train_create(const ccn_config_t* cfg, ccn_trainer_t* out);
int ccn_train_destroy(ccn_trainer_t tr);

/* Number of parameters, total floats of the flat buffer; i-th key / shape / offset (in floats) into the flat buffer. */
int ccn_train_num_params(ccn_trainer_t tr, int32_t* n, int64_t* total_floats);
int ccn_train_param_info(ccn_trainer_t tr, int32_t i, const char** name, int64_t shape[4], int32_t* ndim, int64_t* offset);

/* Scratch for one forward+backward at this shape (every activation of the forward is kept for the backward). */
int ccn_train_workspace_bytes(ccn_trainer_t tr, int32_t B, int32_t H, int32_t W, size_t* bytes);

/* eps_hat = net(x_t, z, t) (train/diffusion_train.py:123; models/unet.py:81-106), activations kept in the workspace.
 * x_t_dev (B,img_ch,H,W) fp32 NCHW; z_dev (B,z_dim); t_dev (B,) int64; eps_dev (B,img_ch,H,W) fp32 NCHW. */
int ccn_train_forward(ccn_trainer_t tr, const float* params_dev, const float* x_t_dev, const float* z_dev,
                      const int64_t* t_dev, float* eps_dev, int32_t B, int32_t H, int32_t W,
                      void* workspace_dev, size_t workspace_bytes, void* stream);

/* The backward pass of that forward (loss.backward(), train/diffusion_train.py:137) for a given d loss / d eps_hat
 * (B,img_ch,H,W) fp32 NCHW: grads_dev[offset_i ...] += d loss / d param_i for every parameter (the caller zeroes the
 * buffer when it wants plain gradients, as opt.zero_grad does at :140).  Must follow ccn_train_forward with the same
 * shape, workspace and x_t_dev / z_dev contents; the parameters must not have changed in between. */
int ccn_train_backward(ccn_trainer_t tr, const float* params_dev, float* grads_dev, const float* x_t_dev,
                       const float* z_dev, const float* d_eps_dev, int32_t B, int32_t H, int32_t W,
                       void* workspace_dev, size_t workspace_bytes, void* stream);

/* The same backward for a data-parallel caller that overlaps the gradient all-reduce with it (DistributedDataParallel's buckets):
 * cb(user, lo, hi) is called from the calling thread as soon as every gradient in the flat range [lo, hi) (floats) is complete
 * in stream order on `stream` -- the caller enqueues its collective on that range right there.  Ranges are disjoint, handed out
 * from the end of the buffer towards its start (the backward visits the layers in reverse registration order), each at least
 * bucket_floats long except the last, and together cover the whole buffer.  The FiLM linears are then differentiated block by
 * block instead of in one grouped launch at the end. */
typedef void (*ccn_grad_ready_cb)(void* user, int64_t lo_float, int64_t hi_float);
int ccn_train_backward_bucketed(ccn_trainer_t tr, const float* params_dev, float* grads_dev, const float* x_t_dev,
                                const float* z_dev, const float* d_eps_dev, int32_t B, int32_t H, int32_t W,
                                void* workspace_dev, size_t workspace_bytes, void* stream, int64_t bucket_floats,
                                ccn_grad_ready_cb cb, void* user);

/* on != 0: ccn_train_forward / ccn_train_backward capture their launch sequence into a hipGraph the first time they see a set of
 * pointer arguments (parameters, gradients, inputs, outputs, workspace) and replay it afterwards -- for callers that keep their
 * buffers at fixed addresses (a training loop with static input / output tensors).  Off by default. */
int ccn_train_set_graph(ccn_trainer_t tr, int32_t on);

/* Per-family timing of the training step (HIP events on the stream the launches go to): enable, run steps, read.
 * names/ms/calls/flops/bytes: arrays of `cap` (>= 16) entries; flops = algorithmic conv FLOPs of the family's launches,
 * bytes = algorithmic HBM bytes of the bandwidth-bound families (0 for the others).
 * Reading synchronises the device and resets the counters. */
int ccn_train_profile_enable(ccn_trainer_t tr, int32_t on);
int ccn_train_profile_read(ccn_trainer_t tr, const char** names, float* ms, int32_t* calls, double* flops,
                           double* bytes, int32_t cap, int32_t* n);

/* F.mse_loss(eps_hat, noise) (train/diffusion_train.py:124) and its gradient: *loss_dev = mean((eps - target)^2),
 * d_eps_dev = 2 (eps - target) / n (may be NULL).  scratch_dev: at least 1024 floats. */
int ccn_mse_loss_grad(const float* eps_dev, const float* target_dev, int64_t n, float* loss_dev, float* d_eps_dev,
                      float* scratch_dev, void* stream);

/* One torch.optim.AdamW step over a flat buffer (train/diffusion_train.py:105,138): p *= 1 - lr*wd;
 * m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr/(1-b1^step) * m / (sqrt(v)/sqrt(1-b2^step) + eps). step >= 1. */
int ccn_adamw_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);
/* The same step followed by opt.zero_grad() (train/diffusion_train.py:138-139) in the same pass: every gradient is read once and
 * its slot left at zero (one launch and one 4-byte write per parameter instead of a second pass over the buffer). */
int ccn_adamw_step_zero_grad(float* params_dev, float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n,
                             float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step, void* stream);

const char* ccn_last_error(void);
const char* ccn_version(void);

#ifdef __cplusplus
}
#endif
#endif /* CCN_HIP_H */
