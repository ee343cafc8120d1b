// Internal declarations shared by the kernel file and the host API file of libccn_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <stdlib.h>

// ---- diagnostics ---------------------------------------------------------------------------------
// The product library (plain `make`) reads NO environment variable and contains no ablation code: every A/B switch
// (kernel choice, split-K policy, pre-pass placement ...) is fixed at its measured default, the CCN_DBG / CCN_WG_DBG
// ablation bits (which produce WRONG results on purpose) and the CCN_STAMPS in-kernel timers compile to nothing.
// `make diag` builds libccn_hip_diag.so with -DCCN_DIAG, where the switches documented in DESIGN.md section 4 are live;
// tools/ select it through CCN_HIP_LIB.
#ifdef CCN_DIAG
#define CCN_DBG_BIT(a, bit) ((((a).dbg) & (bit)) != 0)
#define CCN_STAMPS_PTR(a) ((a).stamps)
namespace ccn { inline const char* diag_env(const char* name) { return getenv(name); } }
#else
#define CCN_DBG_BIT(a, bit) false
#define CCN_STAMPS_PTR(a) ((unsigned long long*)nullptr)
namespace ccn { inline const char* diag_env(const char*) { return nullptr; } }
#endif

namespace ccn {

// ---- weight fragment order of the persistent kernel's 3x3 stride-1 form (ccn_conv_pr.hip, v_mfma_f32_16x16x32_bf16) --------
// Per 64-channel Cin chunk and 32-channel Cout column: 36 fragments of 1 KiB (64 lanes x 8 bf16) in the order the consumer wave
// uses them: f = ((dx * 2 + k32) * 3 + dy) * 2 + c, with k32 the 32-channel K slice and c the 16-channel half of the column.  Lane
// l of a fragment holds output channel 16c + (l & 15) and the eight input channels of the chunk's 16-byte slice
// pr3_slice(l >> 4, k32); the consumers read the same slice of the staged input for that lane group.
// Element (output channel n, input channel k, tap t = dy * 3 + dx) of a layer with Np padded output channels lands at:
__host__ __device__ inline int pr3_slice(int g, int k32) { return ((g & 1) << 2) | (k32 << 1) | (g >> 1); }
__host__ __device__ inline size_t pr3_frag_index(int n, int k, int t, int Np)
{
    const int chunk = k >> 6, s = (k & 63) >> 3, e = k & 7;
    const int g = ((s >> 2) & 1) | ((s & 1) << 1), k32 = (s >> 1) & 1;           // inverse of pr3_slice
    const int nn = n >> 5, c = (n >> 4) & 1, lane = g * 16 + (n & 15);
    const int dy = t / 3, dx = t - dy * 3;
    const int f = ((dx * 2 + k32) * 3 + dy) * 2 + c;
    return ((((size_t)chunk * (size_t)(Np >> 5) + nn) * 36 + f) * 64 + lane) * 8 + e;
}

// ---- weight fragment order of the persistent kernel's 4x2-fragment forms (v_mfma_f32_32x32x16_bf16) ---------------------------
// A 1-KiB fragment covers 32 output channels x one 16-byte K slice per lane half: lane l holds output channel (l & 31) of its
// 32-channel column and input channels (2 kk + (l >> 5)) * 8 .. + 7 of the 64-channel chunk; a step's four K slices kk follow each other.
// `group` counts the (..., tap) prefix of the form's order:
__host__ __device__ inline size_t pr4_frag_elem(size_t group, int n, int k)
{
    const int kw = k & 63;
    return ((group * 4 + (size_t)(kw >> 4)) * 64 + (size_t)((((kw >> 3) & 1) << 5) | (n & 31))) * 8 + (size_t)(kw & 7);
}
// ConvTranspose 4x4 s2 p1: [output parity][chunk][Np / 32][tap 0..3]; tap t of parity par reads kernel element prct_tap(par, t)
// (even outputs <- k in {1, 3}, odd <- {0, 2}; the same order as fill_taps)
__host__ __device__ inline int prct_tap(int par, int t)
{
    const int ky = (par >> 1) ? ((t >> 1) ? 2 : 0) : ((t >> 1) ? 3 : 1), kx = (par & 1) ? ((t & 1) ? 2 : 0) : ((t & 1) ? 3 : 1);
    return ky * 4 + kx;
}
__host__ __device__ inline size_t prct_frag_index(int par, int t, int n, int k, int Np, int nch)
{
    return pr4_frag_elem((((size_t)par * nch + (k >> 6)) * (size_t)(Np >> 5) + (n >> 5)) * 4 + t, n, k);
}
// 3x3 stride 2 as plane passes: [chunk][pass 0..4][Np / 32][slot 0..1]; slot -> tap (dy * 3 + dx) of the 3x3 kernel, -1 = the empty
// tenth slot (pass table in ccn_conv_pr.hip)
__host__ __device__ inline int prs2_tap(int pass, int slot)
{
    const int dy = pass == 0 ? 0 : (pass == 1 ? 2 : (pass == 2 ? (slot ? 2 : 0) : 1));
    const int dx = pass == 2 ? 1 : (pass == 4 ? 1 : (slot ? 2 : 0));
    return (pass == 4 && slot) ? -1 : dy * 3 + dx;
}
__host__ __device__ inline size_t prs2_frag_index(int pass, int slot, int n, int k, int Np)
{
    return pr4_frag_elem((((size_t)(k >> 6) * 5 + pass) * (size_t)(Np >> 5) + (n >> 5)) * 2 + slot, n, k);
}
// 4x4 stride 2 pad 1 as plane passes (the ConvTranspose's data gradient): [chunk][pass 0..3][Np / 32][tap 0..3]; pass p stages input
// plane (py, px) = (p < 2, !(p & 1)), tap t = (i, j) of it is kernel element (ky, kx) at plane offset (i - py, j - px)
__host__ __device__ inline int prp4_tap(int pass, int t)
{
    const int py = pass < 2 ? 1 : 0, px = (pass & 1) ? 0 : 1, i = t >> 1, j = t & 1;
    const int ky = py ? (i ? 2 : 0) : (i ? 3 : 1), kx = px ? (j ? 2 : 0) : (j ? 3 : 1);
    return ky * 4 + kx;
}
__host__ __device__ inline size_t prp4_frag_index(int pass, int t, int n, int k, int Np)
{
    return pr4_frag_elem((((size_t)(k >> 6) * 4 + pass) * (size_t)(Np >> 5) + (n >> 5)) * 4 + t, n, k);
}

// ---- implicit-GEMM convolution ------------------------------------------------------------------
// One kernel family covers every contraction on the path (models/unet.py:55,63,75,79 and
// models/blocks.py:34,36).  M-space = the pixel grid a block tiles (TH=4 x TW=32 pixels per block):
//   3x3 s1  : M = output = input grid,         in = m + d,      out = m
//   3x3 s2  : M = output grid,                 in = 2m + d,     out = m
//   convT4  : M = input grid, 4 output parities, in = m + d,    out = 2m + parity   (2x2 taps each)
//   stem    : im2col of the NCHW fp32 image (K = img_ch*9, one "tap")
enum ConvKind { KIND_C3S1 = 0, KIND_C3S2 = 1, KIND_CT4 = 2, KIND_STEM = 3, KIND_HEAD = 4 };

struct ConvArgs {
    const void* in;         // NHWC T [B][Hin][Win][Cin]   (stem: NCHW fp32 [B][Cin][Hin][Win])
    const void* w;          // packed [tap][Cout_pad][Cin_pad] T
    const void* wfrag;      // layers with Cout_pad % 128 == 0 (bf16 mode): the same weights in MFMA fragment order for ccn_conv_pr.hip
                            // (3x3 s1: pr3_frag_index above; ConvTranspose / stride 2: see pack_convT / pack_conv3), else null
    const float* bias;      // [Cout]
    void* out;              // NHWC T [B][Hout][Wout][Cout] (head: unused)
    const float2* gn_ab;    // prologue GroupNorm: per sample C float2 slots, pair-interleaved {scale(2p), scale(2p+1), shift(2p), shift(2p+1)}, or null
    // persistent kernel: the input GroupNorm's finalize done in-kernel from the producer's partial sums (gn_ab then null):
    const float2* gs_part;  // [B][8][gs_nsp * gs_nnt] partial sums of the input tensor, or null
    const float* gs_gamma; const float* gs_beta;   // [Cin]
    double gs_inv_count;    // 1 / elements per (sample, group)
    int gs_nsp, gs_nnt, gs_bn, gs_cpg;              // layout of the partial sums (as gn_finalize takes it), channels per group
    const float* film;      // epilogue FiLM for this step: [B][film_bstride], s at [n], shift at [Cout+n]; or null
    const void* res;        // epilogue residual / skip, NHWC T like out; or null
    float2* part;           // epilogue GroupNorm partial sums [B][G][nslot] (sum, sum of squares); or null
    // fused GroupNorm finalize: the last workgroup of a sample to arrive reduces that sample's partial sums and writes the
    // consuming norm's per-channel (scale, shift); null fin_counter = separate gn_finalize launch
    unsigned* fin_counter;  // [B] arrival counters, zero between launches
    const float* fin_gamma; const float* fin_beta;
    float2* fin_ab;         // [B][Cout]
    double fin_count;       // elements per (sample, group)
    int fin_blocks;         // workgroups per sample
    int bn;                 // N-tile width of this launch
    float* x_state;         // head: DDIM state, NCHW fp32, updated in place when do_ddim
    float* eps_out;         // head: eps NCHW fp32, or null
    float c0, c1, c2, c3;   // head: DDIM coefficients
    const float* noise;     // head: eta > 0 -- this step's N(0,1) tensor, NCHW fp32 like x_state (added as sigma * noise), or null
    float sigma;
    int do_ddim;
    int film_bstride;
    int B, Hin, Win, Cin, Cin_pad, Hout, Wout, Cout, Cout_pad;
    int MH, MW, OS, npar;
    unsigned long long* stamps;   // in-kernel s_memtime stamps [block][8] (diagnostic builds of the launch only), or null
    int dbg;                // ablation switches for profiling (CCN_DBG env): 1 no A staging in loop, 2 no B staging, 4 no MFMA, 8 no epilogue
    int th;                 // tile rows of 32 pixels per workgroup (4; 8 for the large warp-specialised tiles)
    int use_pr;             // decided at plan time: this launch runs the persistent kernel (its GroupNorm slot layout differs)
    int use_stem2;          // decided at plan time: dedicated bf16 stem kernel (ccn_stem.hip)
    unsigned fd_nt, fd_ntp, fd_tx, fd_sp, fd_cpg, fd_gscpg, fd_gsbn, fd_gsnsp;   // persistent kernel (set by launch_conv_pr): magic numbers ceil(2^32 / d) of its
                            // launch-constant divisors n_nt, n_nt*npar, n_tx, n_tx*n_ty, cpg, gs_cpg (0 for d == 1)
    int prod_first;         // persistent kernel (set by launch_conv_pr): producer waves at a higher priority than the consumers
    int ksplit;             // persistent kernel: 2 = split the Cin chunks over two workgroups per tile (small layers), else 1
    void* kpart;            // split-K: bf16 partial tensor, laid out like `out`
    unsigned* kflag;        // split-K: [tiles][4] hand-off flags, zero between launches
    unsigned* err;          // device-visible error word of the handle (pinned host memory): bit 0 = a split-K hand-off timed out; or null
    int n_ty, n_tx, n_nt, nchunk, ntaps;
    int silu;               // SiLU after the prologue GroupNorm
    int cpg, G, nslot;      // output GroupNorm geometry
    // per tap: bits 0-1 = dy+1, bits 2-3 = dx+1, bits 4.. = weight tap index; convT: entry [parity*4 + tap]
    int tapinfo[16];
    __host__ __device__ int tapinfo_dy(int i) const { return (tapinfo[i] & 3) - 1; }
    __host__ __device__ int tapinfo_dx(int i) const { return ((tapinfo[i] >> 2) & 3) - 1; }
    __host__ __device__ int tapinfo_w(int i) const { return tapinfo[i] >> 4; }
    static int make_tap(int dy, int dx, int wt) { return (dy + 1) | ((dx + 1) << 2) | (wt << 4); }
};

int conv_bn_for(int cout, int kind);                       // N-tile width used for a layer
size_t conv_lds_bytes(int dtype, int kind, int bn);
hipError_t conv_prepare();                                 // raise dynamic-LDS limits once
hipError_t launch_conv(int dtype, int kind, int bn, const ConvArgs& a, hipStream_t s);
// rows of 32 pixels per workgroup tile for this layer (decided once, at plan time)
int conv_tile_rows(int kind, int bn, int B, int MH, int MW, int npar, int n_nt);
// warp-specialised kernel (ccn_conv_ws.hip): 3x3 s1 and ConvTranspose parities, BN 64/128
bool conv_ws_enabled();
bool conv_ws_supported(int kind, int bn);
hipError_t conv_ws_prepare();
hipError_t launch_conv_ws(int dtype, int bn, const ConvArgs& a, hipStream_t s);
// free-running variant (ccn_conv_fr.hip): private per-wave weight rings by LDS-DMA, no barrier inside a Cin chunk
hipError_t conv_fr_prepare();
hipError_t launch_conv_fr(int dtype, int bn, const ConvArgs& a, hipStream_t s);

// persistent variant (ccn_conv_pr.hip): weights streamed from L2 into registers, double-buffered input chunks, per-wave
// epilogue; GroupNorm partial slots are per consumer wave: n_sp*2 spatial x n_nt*2 channel columns of 64
bool conv_pr_supported(int kind, int bn, int th);
bool conv_pr_selected(int dtype, int kind, int bn, int th);   // supported and chosen by the variant switch
hipError_t conv_pr_prepare();
hipError_t launch_conv_pr(int dtype, const ConvArgs& a, hipStream_t s);

// dedicated stem kernel (ccn_stem.hip): bf16 mode, K = img_ch*9 <= 31, Cout 32/64/128; weights + bias in fragment order in wfrag
bool stem2_supported(int dtype, int cin, int cout, int G);
int stem2_blocks(int H, int W, int* upw_out);
hipError_t launch_stem2(const ConvArgs& a, hipStream_t s);

// dedicated head (ccn_head.hip): bf16 mode, out_norm folded into per-sample weights, the nine taps in the MFMA N dimension
bool head2_supported(int dtype, int cin, int cout, int G);
size_t head2_scratch_bytes(int B, int C);
hipError_t launch_head2(const ConvArgs& a, const float2* ab, const float* w_f32, void* scratch, int step, hipStream_t s);

// GroupNorm-apply + SiLU as its own pass (NHWC T -> NHWC T).  Used in front of convs whose input is re-staged by
// several N tiles (Cout >= 256): the transform then runs once per element instead of once per (N tile x halo).
bool conv_wants_preact(int kind, int bn, int n_nt);
hipError_t launch_gn_act(int dtype, const void* x, const float2* ab, void* y, int B, int HW, int C, hipStream_t s);
// the same with the GroupNorm finalize of the input folded in (partial sums -> scale/shift inside every workgroup)
hipError_t launch_gn_act_fused(int dtype, const void* x, void* y, int B, int HW, int C, const float2* part, int G, int n_sp, int n_nt,
                               int bn, int cpg, double count, const float* gamma, const float* beta, float eps, hipStream_t s,
                               float2* ab_out = nullptr, float2* stats_out = nullptr);

// ---- GroupNorm finalize: partial sums -> per-(b,channel) scale/shift --------------------------------
hipError_t launch_gn_finalize(const float2* part, int B, int G, int n_sp, int n_nt, int bn, int cpg, int C,
                              double count, const float* gamma, const float* beta, float eps,
                              float2* ab, hipStream_t s);

// ---- conditioning (tiny GEMVs, fp32) --------------------------------------------------------------
hipError_t launch_temb_i64(const int64_t* t, float* out, int n, int dim, hipStream_t s);
hipError_t launch_temb_i32(const int32_t* t, float* out, int n, int dim, hipStream_t s);
// y[r][n] = act(sum_k x[r][k] W[n][k] + b[n]);  x row r is xa[ra(r)] (+ xb[rb(r)] when xb != null) where
// ra(r) = r / a_div, rb(r) = r % b_mod (a_div = 1, b_mod = R gives plain rows).
hipError_t launch_linear(const float* xa, const float* xb, int a_div, int b_mod, const float* W, const float* b,
                         float* y, int R, int K, int N, int act_silu, hipStream_t s);

// ---- elementwise ---------------------------------------------------------------------------------
hipError_t launch_ddim_step(float* x, const float* eps, const float* noise, float c0, float c1, float c2, float c3,
                            float sigma, int64_t n, hipStream_t s);
hipError_t launch_q_sample(float* out, const float* x0, const float* noise, const float* a, const float* sg,
                           int B, int64_t per, hipStream_t s);
hipError_t launch_predict_x0(float* out, const float* xt, const float* eps, const float* a, const float* sg,
                             int B, int64_t per, hipStream_t s);
hipError_t launch_film_nchw(const float* x, const float* sc, const float* sh, float* y, int B, int C, int64_t hw,
                            hipStream_t s);
hipError_t launch_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, hipStream_t s);
hipError_t launch_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, hipStream_t s);
// GroupNorm partial sums of an NHWC tensor that no conv epilogue produced (operator-level entry points)
hipError_t launch_gn_partials(int dtype, const void* x, float2* part, int B, int HW, int C, int cpg, int G,
                              int nslot, hipStream_t s);

}  // namespace ccn
