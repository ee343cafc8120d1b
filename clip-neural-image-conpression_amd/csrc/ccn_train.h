// Internal declarations of the training-step kernels (ccn_train_kernels.hip) used by the host side (ccn_train.hip).
// The step they implement is the body of the reference's training loop, train/diffusion_train.py:119-124,137-140:
// eps_hat = net(x_t, z, t) with every activation kept, then the backward pass of that forward for a given d(eps_hat).
#pragma once
#include "ccn_internal.h"

namespace ccn {

// ---- weight repacking on the device (the fp32 master weights change every optimiser step) ------------------------
// Destination: the [tap][N_pad][K_pad] operand layout of the implicit-GEMM kernels (ccn_kernels.hip), element type T.
enum PackMode {
    PK_CONV3 = 0,     // Conv2d (O, I, 3, 3)           -> forward operand, 9 taps,  N = O, K = I
    PK_CONVT = 1,     // ConvTranspose2d (I, O, 4, 4)  -> forward operand, 16 taps, N = O, K = I
    PK_DG3S1 = 2,     // Conv2d 3x3 s1                 -> data-gradient operand: taps flipped, N = I, K = O
    PK_DG3S2 = 3,     // Conv2d 3x3 s2                 -> data-gradient operand for the ConvTranspose kernel: 4x4 taps, row/col 3 zero
    PK_DGT = 4,       // ConvTranspose2d 4x4 s2        -> data-gradient operand for the stride-2 kernel run with 16 taps
    PK_STEM = 5,      // Conv2d (O, img_ch, 3, 3)      -> im2col operand, 1 "tap", K = img_ch * 9
    PK_HEAD_DG = 6,   // Conv2d (img_ch, C, 3, 3)      -> data gradient through the im2col kernel: N = C, K = img_ch * 9, taps flipped
    PK_FRAG3 = 7,     // Conv2d 3x3 s1, bf16           -> MFMA fragment order of the persistent kernel (ccn_conv_pr.hip), forward operand
    PK_FRAG3_DG = 8,  // Conv2d 3x3 s1, bf16           -> the same order for the data-gradient operand (taps flipped, N = I, K = O)
    PK_FRAG_S2 = 9,   // Conv2d 3x3 s2, bf16           -> plane-pass fragment order of the persistent kernel's stride-2 form (prs2_frag_index)
    PK_FRAG_CT = 10,  // ConvTranspose2d 4x4 s2, bf16  -> parity / tap fragment order of its ConvTranspose form (prct_frag_index)
    PK_FRAG_STEM = 13,   // Conv2d (O, img_ch, 3, 3) + bias, bf16 -> register-fragment order of the stem kernel (ccn_stem.hip; bias as k = img_ch * 9)
    PK_FRAG_STEM_HEAD_DG = 14,  // Conv2d (img_ch, C, 3, 3), bf16 -> the head's data gradient as a stem conv on d eps: N = C, k = co * 9 + tap, taps flipped
    PK_FRAG_P4_DG = 12,  // ConvTranspose2d 4x4 s2, bf16 -> its data-gradient operand (a 4x4 s2 conv) as plane passes of 2x2 taps (prp4_frag_index; N = I, K = O)
    PK_FRAG_CT_DG = 11,  // Conv2d 3x3 s2, bf16        -> its data-gradient operand in that order (4x4 taps, row / column 3 zero; N = I, K = O)
};

// all repacks of one step in ONE launch: descriptor table built once at create time (offsets into the flat parameter buffer)
struct PackDesc { long long src_off; void* dst; int mode, O, I, taps, Np, Kp; long long aux_off; };   // aux_off: the bias (PK_FRAG_STEM)
hipError_t launch_pack_group(int dtype, const float* params, const PackDesc* descs_dev, int n, hipStream_t s);

// the FiLM linears of every ResBlock (to_scale / to_shift: models/blocks.py:19-20) share their input h: one launch each for the
// forward, the weight/bias gradients and the input gradient.  out_off: column of the (B, F) FiLM table.
struct LinDesc { long long w_off, b_off; int N, out_off; };
hipError_t launch_film_group_fwd(const float* params, const LinDesc* d, int n, int maxN, const float* h, float* film, int B, int K, int F, hipStream_t s);
hipError_t launch_film_group_dw(float* grads, const LinDesc* d, int n, int maxN, const float* dfilm, const float* h, int B, int K, int F, hipStream_t s);
hipError_t launch_film_group_dx(const float* params, const LinDesc* d, int n, const float* dfilm, float* dh, int B, int K, int F, hipStream_t s);

// ---- GroupNorm ------------------------------------------------------------------------------------------------------
// partial sums -> scale/shift table (as gn_finalize) plus (mean, rstd) per (sample, group)
hipError_t launch_gn_stats(const float2* part, int B, int G, int n_sp, int n_nt, int bn, int cpg, int C, double count,
                           const float* gamma, const float* beta, float eps, float2* ab, float2* stats, hipStream_t s);

struct GnBwdGeom { int nslb, pstep, ppb, nblk, zblocks; };
GnBwdGeom gn_bwd_geom(int dtype, int B, int HW, int C);
// pass 1: per (sample, pixel block, channel) sums of da and da * xhat, da = dA * silu'(a x + c) (or dA)
hipError_t launch_gn_bwd_reduce(int dtype, const void* x, const void* dA, const float2* ab, const float2* stats, float2* part,
                                int B, int HW, int C, int cpg, int G, int silu, hipStream_t s);
// per (sample, group): m1 = mean(gamma da), m2 = mean(gamma da xhat); dgamma += sum da xhat, dbeta += sum da
hipError_t launch_gn_bwd_finalize(const float2* part, int nblk, int B, int C, int cpg, int G, double count, const float* gamma,
                                  float2* gstat, float* dgamma, float* dbeta, hipStream_t s);
// pass 2: dx = a da - rstd (m1 + xhat m2) [+ addend]; with FiLM (blocks.py:22-25): out = dx (1 + s) and per-block sums of dx, dx * x
hipError_t launch_gn_bwd_apply(int dtype, const void* x, const void* dA, const float2* ab, const float2* stats, const float2* gstat,
                               const void* addend, void* out, const float* film, int film_bstride, float2* fpart,
                               int B, int HW, int C, int cpg, int G, int silu, hipStream_t s);
hipError_t launch_film_bwd_finalize(const float2* fpart, int nblk, const float* film, int film_bstride, float* dfilm, float* dbias, int B,
                                    int C, hipStream_t s);
// without FiLM the apply pass leaves per-(sample, block, channel) sums of its OUTPUT in fpart: the bias gradient of the conv
// whose output gradient that is
hipError_t launch_colsum_from_pairs(const float2* part, int rows, int C, float* db, hipStream_t s);
// per-channel sum over (B, HW) of an NHWC tensor, added to `db` (bias gradients); scratch: B * nblk * C floats
hipError_t launch_colsum(int dtype, const void* dy, float* scratch, float* db, int B, int HW, int C, hipStream_t s);
// per-channel sum of an NCHW fp32 tensor, added to db
hipError_t launch_nchw_chansum(const float* x, float* db, int B, int C, int64_t hw, hipStream_t s);

// ---- weight gradients -------------------------------------------------------------------------------------------------
struct WgArgs {
    const void* x;            // A source NHWC T [B][Hin][Win][Cin] (the forward conv's raw input)
    const float2* gn_ab;      // forward prologue GroupNorm (+SiLU when `silu`) to redo on the fly, or null
    int silu;
    const void* dy;           // NHWC T [B][Hout][Wout][Cout]
    float* part;              // [nsplit][taps_w][Cout][Cin] fp32 partial sums
    int B, Hin, Win, Cin, Hout, Wout, Cout, MH, MW, OS, npar, ntaps, taps_w, n_ty, n_tx, nsplit;
    int dbg;                // timing ablations (CCN_WG_DBG): 1 producers idle after the first tile, 2 consumers skip the MFMA loop, 4 no partial stores
    int tapinfo[16];        // as ConvArgs::tapinfo
    __host__ __device__ int tapinfo_dy(int i) const { return (tapinfo[i] & 3) - 1; }
    __host__ __device__ int tapinfo_dx(int i) const { return ((tapinfo[i] >> 2) & 3) - 1; }
    __host__ __device__ int tapinfo_w(int i) const { return tapinfo[i] >> 4; }
};
// concurrent: the launch shares the GPU with the main stream's kernels (side stream), so it asks for fewer workgroups
int wgrad_nsplit(int dtype, int kind, int B, int MH, int MW, int Cin, int Cout, bool concurrent);
hipError_t launch_wgrad(int dtype, int kind, const WgArgs& a, hipStream_t s);
hipError_t wgrad_prepare();
// grads[(o*I + i)*taps + t] (Conv2d) or grads[(i*O + o)*taps + t] (ConvTranspose2d) += sum over splits
// Ov / Iv: the rows / columns of the partial tiles that exist in the parameter (padding channels are dropped)
hipError_t launch_wgrad_reduce(const float* part, int nsplit, int taps, int O, int I, int Ov, int Iv, int transposed, float* grad, hipStream_t s);
// layout changes feeding the generic kernel for the stem / head weights: im2col of the NCHW fp32 image (k = ci*9 + tap, 32 columns)
// and NCHW fp32 -> NHWC T with the channels padded to Cp
hipError_t launch_im2col27(int dtype, const float* x, void* dst, int B, int C, int H, int W, hipStream_t s);
hipError_t launch_nchw_to_nhwc_pad(int dtype, const float* x, void* dst, int B, int C, int Cp, int64_t hw, hipStream_t s);
// ---- conditioning (small fp32 linears) ----------------------------------------------------------------------------------
hipError_t launch_tlinear_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, float* u, int R, int K, int N,
                              int silu, hipStream_t s);
hipError_t launch_tlinear_dw(const float* dy, int lddy, const float* x, int ldx, float* dW, float* db, int R, int K, int N, hipStream_t s);
hipError_t launch_tlinear_dx(const float* dy, int lddy, const float* W, float* dx, int lddx, int R, int K, int N, int accumulate, hipStream_t s);
hipError_t launch_silu_bwd(float* du, const float* dy, const float* u, int64_t n, hipStream_t s);
hipError_t launch_add2(float* y, const float* a, const float* b, int64_t n, hipStream_t s);

// ---- loss and optimiser ---------------------------------------------------------------------------------------------------
// loss = mean((eps - target)^2) (F.mse_loss, train/diffusion_train.py:124); d_eps = 2 (eps - target) / n; scratch >= 1024 floats
hipError_t launch_mse_loss_grad(const float* eps, const float* target, int64_t n, float* loss, float* d_eps, float* scratch, hipStream_t s);
// torch.optim.AdamW step (train/diffusion_train.py:105,138): decoupled weight decay, bias-corrected moments
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, int64_t n, float lr, float b1, float b2, float eps, float wd,
                        int step, hipStream_t s, bool zero_grad = false);

}  // namespace ccn
