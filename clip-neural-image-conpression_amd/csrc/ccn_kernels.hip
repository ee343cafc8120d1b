// Hand-written gfx950 (CDNA4) kernels of the DDIM reconstruction path.
//
//   conv_igemm_kernel   every convolution of CLIPCondUNet as an LDS-tiled implicit GEMM on MFMA:
//                       3x3 s1 (models/blocks.py:34,36), 3x3 s2 (models/unet.py:63), ConvTranspose 4x4 s2 as four
//                       2x2-tap parity sub-convolutions (models/unet.py:75), stem (models/unet.py:55, im2col K=27)
//                       and head (models/unet.py:79) with the DDIM update (diffusion/ddim.py:34-45) in its epilogue.
//                       Prologue fusion: GroupNorm-apply + SiLU while staging the input halo tile (blocks.py:41,43).
//                       Epilogue fusion: bias, FiLM (blocks.py:22-25), residual / skip add (blocks.py:44, unet.py:104),
//                       and the NEXT GroupNorm's partial sums, so a normalised tensor is never re-read for statistics.
//   gn_finalize_kernel  partial sums -> per-(sample, channel) scale/shift (fp64 combine).
//   linear/temb         conditioning vector and FiLM tables for every step, computed once before the loop.
//   ddim_step/q_sample  stand-alone elementwise forms.
//
// Layout: activations NHWC (channel-contiguous), element type T = float (parity mode, v_mfma_f32_32x32x2_f32,
// bit-identical to an fmaf chain) or bf16 (throughput mode, v_mfma_f32_32x32x16_bf16, fp32 accumulate).
// Both types use the same byte geometry: one LDS row = 128 B = 8 chunks of 16 B = 32 floats or 64 bf16; lane
// half h reads chunk 2*kk+h of its A row (a pixel) and of its B row (an output channel), so the K order seen by
// the two operands is identical by construction.  LDS rows are XOR-swizzled by ((row>>1)&7) at 16-B granularity:
// 32 consecutive rows read at one chunk index hit 16 distinct 16-B slots per ds_read_b128 lane group.
#include "ccn_device.h"
#include <cstdlib>

namespace ccn {

__host__ __device__ constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int AK, int IS> struct HaloGeom {
    static constexpr int HOFF = AK == A_NHWC ? 1 : 0;
    // IS*(TH-1)+3 with TH = 4; one more row at stride 2 so that a 4x4 kernel (taps -1..2: the data gradient of the
    // ConvTranspose, ccn_train.hip) fits the same staging
    static constexpr int ROWS = AK == A_NHWC ? (IS == 1 ? 6 : 10) : 4;
    static constexpr int PITCH = AK == A_NHWC ? (IS == 1 ? 34 : 66) : 32;  // >= IS*(TW-1)+3 with TW = 32
    static constexpr int BYTES = ROWS * PITCH * 128;
};
template <int AK, int IS, int BN> struct LdsGeom {
    static constexpr int A_BYTES = HaloGeom<AK, IS>::BYTES;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int CP = BN + 4;                       // fp32 epilogue tile pitch (floats)
    static constexpr int CS_BYTES = 128 * CP * 4;
    static constexpr int RED_BYTES = 4 * BN * 2 * 4 + BN * 2 * 4;
    static constexpr int TOTAL = cmax(A_BYTES + 2 * B_BYTES, CS_BYTES + RED_BYTES);
};

template <typename T, int AK, int IS, int EPI, int WM, int WN, int MF, int NF>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs a)
{
    static_assert(WM * WN == 4 && WM * MF == 4, "4 waves, 4 M-fragments of 32 pixels (TH=4 rows of TW=32)");
    constexpr int BN = WN * NF * 32;
    constexpr int EPC = Vec16<T>::EPC;
    constexpr int CKE = 8 * EPC;
    using HG = HaloGeom<AK, IS>;
    using LG = LdsGeom<AK, IS, BN>;
    constexpr int HPITCH = HG::PITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;
    unsigned char* const Bs = smem + LG::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    int bid = blockIdx.x;
    const int nt = bid % a.n_nt; bid /= a.n_nt;
    const int par = bid % a.npar; bid /= a.npar;
    const int tx = bid % a.n_tx; bid /= a.n_tx;
    const int ty = bid % a.n_ty;
    const int b = bid / a.n_ty;
    const int my0 = ty * 4, mx0 = tx * 32, n0 = nt * BN;
    const int py = par >> 1, px_ = par & 1;
    const int par_off = par * 4;

    const unsigned char* const wbase = (const unsigned char*)a.w;
    const unsigned char* const inb = (const unsigned char*)a.in;

    // ---- weight (B operand) staging: global -> registers -> LDS, double buffered -----------------
    constexpr int BU = BN * 8 / 256;
    u32x4 breg[BU];
#define CCN_LOAD_B(IT)                                                                                              \
    {                                                                                                               \
        const int chunk_ = (IT) / a.ntaps, tap_ = (IT) - chunk_ * a.ntaps;                                          \
        const int wt_ = a.tapinfo_w(par_off + tap_);                                                                \
        _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                                            \
            const int idx = tid + 256 * u, n = idx >> 3, ck = idx & 7;                                              \
            const size_t off = ((size_t)(wt_ * a.Cout_pad + n0 + n) * a.Cin_pad + (size_t)chunk_ * CKE) * sizeof(T) + ck * 16; \
            breg[u] = *(const u32x4*)(wbase + off);                                                                 \
        }                                                                                                           \
    }
#define CCN_STORE_B(BUF)                                                                                            \
    {                                                                                                               \
        _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                                            \
            const int idx = tid + 256 * u, n = idx >> 3, ck = idx & 7;                                              \
            *(u32x4*)(Bs + (BUF) * LG::B_BYTES + n * 128 + (((ck ^ (n >> 1)) & 7) << 4)) = breg[u];                \
        }                                                                                                           \
    }

    // ---- input (A operand) staging ---------------------------------------------------------------
    auto stage_A = [&](int chunk) __attribute__((always_inline)) {
        const int ck = tid & 7;
        if constexpr (AK == A_NHWC) {
            constexpr int AU = HG::ROWS * HPITCH * 8;
            constexpr int AIT = (AU + 255) / 256;
            constexpr int GRP = 7;
            const int cbase = chunk * CKE + ck * EPC;
            const bool cvalid = cbase < a.Cin;
            const bool gn = a.gn_ab != nullptr;
            GnCoef<T> gk;
            gk.load(a.gn_ab + (size_t)b * a.Cin + (cvalid ? cbase : 0), gn && cvalid);
            const int iy0 = IS * my0 - 1, ix0 = IS * mx0 - 1;
#pragma unroll
            for (int g0 = 0; g0 < AIT; g0 += GRP) {
                u32x4 raw[GRP]; unsigned okm = 0;
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int i = g0 + u;
                    raw[u] = u32x4{0u, 0u, 0u, 0u};
                    if (i < AIT) {
                        const int px = (tid >> 3) + 32 * i;
                        const int hy = px / HPITCH, hx = px - hy * HPITCH;
                        const int iy = iy0 + hy, ix = ix0 + hx;
                        const bool okv = px < HG::ROWS * HPITCH && cvalid && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
                        if (okv) {
                            okm |= 1u << u;
                            raw[u] = *(const u32x4*)(inb + (((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + cbase) * sizeof(T));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < GRP; ++u) {
                    const int i = g0 + u;
                    if (i < AIT) {
                        const int px = (tid >> 3) + 32 * i;
                        if (px < HG::ROWS * HPITCH) {
                            u32x4 o = raw[u];
                            if (((okm >> u) & 1u) && gn)       // zero padding applies AFTER GroupNorm+SiLU; head: no SiLU (unet.py:105)
                                o = gk.template apply<EPI == EPI_NHWC>(raw[u]);
                            *(u32x4*)(As + px * 128 + (((ck ^ (px >> 1)) & 7) << 4)) = o;
                        }
                    }
                }
            }
        } else {   // A_IM2COL: stem, K = Cin*9 taken from the NCHW fp32 image; row m = pixel of the 4x32 tile
            const float* xin = (const float*)a.in;
            const int kreal = a.Cin * 9;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = (tid >> 3) + 32 * i;
                const int y = my0 + (m >> 5), x = mx0 + (m & 31);
                float v[EPC];
#pragma unroll
                for (int e = 0; e < EPC; ++e) {
                    const int k = chunk * CKE + ck * EPC + e;
                    float val = 0.0f;
                    if (k < kreal) {
                        const int ci = k / 9, tp = k - ci * 9, ky = tp / 3, kx = tp - ky * 3;
                        const int iy = y + ky - 1, ix = x + kx - 1;
                        if (iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win)
                            val = xin[((size_t)(b * a.Cin + ci) * a.Hin + iy) * a.Win + ix];
                    }
                    v[e] = val;
                }
                *(u32x4*)(As + m * 128 + (((ck ^ (m >> 1)) & 7) << 4)) = Vec16<T>::pack(v);
            }
        }
    };

    // ---- main loop: (Cin chunk) x (tap), one barrier per iteration -------------------------------
    f32x16 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    int pbase[MF], nrow[NF];
#pragma unroll
    for (int i = 0; i < MF; ++i) pbase[i] = (IS * (wm * MF + i) + HG::HOFF) * HPITCH + IS * r + HG::HOFF;
#pragma unroll
    for (int j = 0; j < NF; ++j) nrow[j] = (wn * NF + j) * 32 + r;

    const int n_it = a.nchunk * a.ntaps;
    CCN_LOAD_B(0)
    for (int it = 0; it < n_it; ++it) {
        const int chunk = it / a.ntaps, tap = it - chunk * a.ntaps;
        if (tap == 0) {
            if (it > 0) __syncthreads();
            stage_A(chunk);
        }
        CCN_STORE_B(it & 1)
        __syncthreads();
        if (it + 1 < n_it) CCN_LOAD_B(it + 1)
        const int dy = a.tapinfo_dy(par_off + tap), dx = a.tapinfo_dx(par_off + tap);
        const unsigned char* const Bb = Bs + (it & 1) * LG::B_BYTES;
        int pa[MF];
#pragma unroll
        for (int i = 0; i < MF; ++i) pa[i] = pbase[i] + dy * HPITCH + dx;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            u32x4 av[MF], bv[NF];
#pragma unroll
            for (int i = 0; i < MF; ++i)
                av[i] = *(const u32x4*)(As + pa[i] * 128 + ((((2 * kk + h) ^ (pa[i] >> 1)) & 7) << 4));
#pragma unroll
            for (int j = 0; j < NF; ++j)
                bv[j] = *(const u32x4*)(Bb + nrow[j] * 128 + ((((2 * kk + h) ^ (nrow[j] >> 1)) & 7) << 4));
#pragma unroll
            for (int i = 0; i < MF; ++i)
#pragma unroll
                for (int j = 0; j < NF; ++j) mfma16<T>(acc[i][j], av[i], bv[j]);
        }
    }

#undef CCN_LOAD_B
#undef CCN_STORE_B
    // ---- epilogue: accumulators -> LDS fp32 tile -> coalesced 16-B rows ----------------------------
    __syncthreads();
    float* const Cs = (float*)smem;
    constexpr int CP = LG::CP;
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = (wm * MF + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                Cs[m * CP + (wn * NF + j) * 32 + r] = acc[i][j][q];
            }
    __syncthreads();

    if constexpr (EPI == EPI_HEAD) {
        if (tid < 128) {
            const int m = tid, my = my0 + (m >> 5), mx = mx0 + (m & 31);
            if (my < a.MH && mx < a.MW) {
                for (int c = 0; c < a.Cout; ++c) {
                    const float e = Cs[m * CP + c] + a.bias[c];
                    const size_t idx = ((size_t)(b * a.Cout + c) * a.Hout + my) * a.Wout + mx;
                    if (a.eps_out) a.eps_out[idx] = e;
                    if (a.do_ddim) {
                        // diffusion/ddim.py:38-43 -- every op rounds to fp32 separately, like the torch ops
                        const float x = a.x_state[idx];
                        float x0 = __fdiv_rn(__fsub_rn(x, __fmul_rn(a.c0, e)), a.c1);
                        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
                        float xn = __fadd_rn(__fmul_rn(a.c2, x0), __fmul_rn(a.c3, e));
                        if (a.noise) xn = __fadd_rn(xn, __fmul_rn(a.sigma, a.noise[idx]));      // eta > 0 (diffusion/ddim.py:44-45)
                        a.x_state[idx] = xn;
                    }
                }
            }
        }
    } else {
        constexpr int NOCT = BN / 8, PSL = 256 / NOCT, NIT = 128 / PSL;
        const int o = tid % NOCT, ps = tid / NOCT;
        const int nb = n0 + o * 8;
        const bool nvalid = nb < a.Cout;
        float bias8[8], f1[8], f2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { bias8[e] = 0.f; f1[e] = 1.f; f2[e] = 0.f; }
        if (nvalid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = a.bias[nb + e];
            if (a.film) {
                const float* fp = a.film + (size_t)b * a.film_bstride;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f1[e] = 1.0f + fp[nb + e]; f2[e] = fp[a.Cout + nb + e]; }
            }
        }
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
        unsigned char* const outb = (unsigned char*)a.out;
        const unsigned char* const resb = (const unsigned char*)a.res;
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp) {
            const int m = itp * PSL + ps;
            const int my = my0 + (m >> 5), mx = mx0 + (m & 31);
            if (nvalid && my < a.MH && mx < a.MW) {
                const int oy = my * a.OS + py, ox = mx * a.OS + px_;
                const size_t eoff = ((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + nb;
                float v[8];
                const f32x4 c0 = *(const f32x4*)(Cs + m * CP + o * 8), c1 = *(const f32x4*)(Cs + m * CP + o * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = c0[e]; v[4 + e] = c1[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e] + bias8[e], f1[e], f2[e]);
                if (resb) {
                    float rv[8];
                    if constexpr (EPC == 8) {
                        Vec16<T>::unpack(*(const u32x4*)(resb + eoff * sizeof(T)), rv);
                    } else {
                        Vec16<T>::unpack(*(const u32x4*)(resb + eoff * sizeof(T)), rv);
                        Vec16<T>::unpack(*(const u32x4*)(resb + eoff * sizeof(T) + 16), rv + 4);
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                if constexpr (EPC == 8) {
                    *(u32x4*)(outb + eoff * sizeof(T)) = Vec16<T>::pack(v);
                } else {
                    *(u32x4*)(outb + eoff * sizeof(T)) = Vec16<T>::pack(v);
                    *(u32x4*)(outb + eoff * sizeof(T) + 16) = Vec16<T>::pack(v + 4);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
            }
        }
        if (a.part) {
            // fixed-order reduction: lanes sharing an octet (xor strides >= NOCT), then waves, then channels of a group
#pragma unroll
            for (int s = NOCT; s < 64; s <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], s); s2[e] += __shfl_xor(s2[e], s); }
            float* const red = (float*)(smem + LG::CS_BYTES);       // [4][BN][2]
            float* const chs = red + 4 * BN * 2;                    // [BN][2]
            if (lane < NOCT) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    red[(wave * BN + lane * 8 + e) * 2 + 0] = s1[e];
                    red[(wave * BN + lane * 8 + e) * 2 + 1] = s2[e];
                }
            }
            __syncthreads();
            if (tid < BN) {
                float t1 = 0.f, t2 = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
                chs[tid * 2] = t1; chs[tid * 2 + 1] = t2;
            }
            __syncthreads();
            if (n0 < a.Cout) {
                const int nend = min(n0 + BN, a.Cout);
                const int g = n0 / a.cpg + tid;
                if (g <= (nend - 1) / a.cpg) {
                    const int clo = max(g * a.cpg, n0), chi = min((g + 1) * a.cpg, nend);
                    float t1 = 0.f, t2 = 0.f;
                    for (int c = clo; c < chi; ++c) { t1 += chs[(c - n0) * 2]; t2 += chs[(c - n0) * 2 + 1]; }
                    const int slot = (((ty * a.n_tx + tx) * a.npar + par) * a.n_nt) + nt;
                    part_store(a.part + (size_t)(b * a.G + g) * a.nslot + slot, t1, t2);
                }
            }
            if (a.fin_counter) gn_fused_finalize<256>(a, b, (unsigned*)red, tid);
        }
    }
}

// ---- dispatch ----------------------------------------------------------------------------------------
int conv_bn_for(int cout, int kind)
{
    if (kind == KIND_HEAD) return 32;
    if (cout >= 128) return 128;
    if (cout >= 64) return 64;
    return 32;
}

template <int AK, int IS> static size_t lds_for(int bn)
{
    switch (bn) {
        case 128: return LdsGeom<AK, IS, 128>::TOTAL;
        case 64: return LdsGeom<AK, IS, 64>::TOTAL;
        default: return LdsGeom<AK, IS, 32>::TOTAL;
    }
}
size_t conv_lds_bytes(int dtype, int kind, int bn)
{
    (void)dtype;
    switch (kind) {
        case KIND_C3S2: return lds_for<A_NHWC, 2>(bn);
        case KIND_STEM: return lds_for<A_IM2COL, 1>(bn);
        default: return lds_for<A_NHWC, 1>(bn);
    }
}

typedef void (*conv_fn_t)(const ConvArgs);

template <typename T, int AK, int IS> static conv_fn_t pick_bn(int bn)
{
    switch (bn) {
        case 128: return conv_igemm_kernel<T, AK, IS, EPI_NHWC, 2, 2, 2, 2>;
        case 64: return conv_igemm_kernel<T, AK, IS, EPI_NHWC, 2, 2, 2, 1>;
        default: return conv_igemm_kernel<T, AK, IS, EPI_NHWC, 4, 1, 1, 1>;
    }
}
template <typename T> static conv_fn_t pick_kind(int kind, int bn)
{
    switch (kind) {
        case KIND_C3S2: return pick_bn<T, A_NHWC, 2>(bn);
        case KIND_STEM: return pick_bn<T, A_IM2COL, 1>(bn);
        case KIND_HEAD: return conv_igemm_kernel<T, A_NHWC, 1, EPI_HEAD, 4, 1, 1, 1>;
        default: return pick_bn<T, A_NHWC, 1>(bn);
    }
}
static conv_fn_t pick(int dtype, int kind, int bn)
{
    return dtype == 0 ? pick_kind<float>(kind, bn) : pick_kind<__bf16>(kind, bn);
}

hipError_t conv_prepare()
{
    static const int kinds[] = {KIND_C3S1, KIND_C3S2, KIND_STEM, KIND_HEAD};
    static const int bns[] = {32, 64, 128};
    for (int dt = 0; dt < 2; ++dt)
        for (int kind : kinds)
            for (int bn : bns) {
                if (kind == KIND_HEAD && bn != 32) continue;
                hipError_t e = hipFuncSetAttribute((const void*)pick(dt, kind, bn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)conv_lds_bytes(dt, kind, bn));
                if (e != hipSuccess) return e;
            }
    hipError_t e;
    if ((e = conv_ws_prepare()) != hipSuccess) return e;
    if ((e = conv_fr_prepare()) != hipSuccess) return e;
    return conv_pr_prepare();
}

bool conv_ws_enabled()
{
    static const int on = diag_env("CCN_CONV_V1") ? 0 : 1;      // CCN_CONV_V1=1: A/B switch back to the 4-wave kernel
    return on != 0;
}

int conv_tile_rows(int kind, int bn, int B, int MH, int MW, int npar, int n_nt)
{
    if (!conv_ws_enabled() || !conv_ws_supported(kind, bn)) return 4;
    // 8-row tiles halve the weight traffic through L2 and LDS (the 4-row tile is LDS-bandwidth bound: 1 KB of
    // fragment reads per MFMA plus 32 B/clk of weight ds_writes) and cut the halo overhead; keep every CU busy
    const long blocks8 = (long)B * ((MH + 7) / 8) * ((MW + 31) / 32) * npar * n_nt;
    static const long min8 = diag_env("CCN_MIN8") ? atol(diag_env("CCN_MIN8")) : 256;
    return blocks8 >= min8 ? 8 : 4;
}

// 0 (or 1) ws everywhere, 2 free-running everywhere, 3 free-running on 8-row tiles, 4 (default) persistent
// register-weight kernel where it applies, else as 3.  CCN_CONV_DMA overrides; tests switch it between handles.
static int g_conv_variant = -1;
static int conv_variant()
{
    if (g_conv_variant < 0) g_conv_variant = diag_env("CCN_CONV_DMA") ? atoi(diag_env("CCN_CONV_DMA")) : 4;
    return g_conv_variant;
}
extern "C" int ccn_internal_set_conv_variant(int v) { const int old = conv_variant(); g_conv_variant = v; return old; }

bool conv_pr_selected(int dtype, int kind, int bn, int th)
{
    return dtype == 1 && conv_ws_enabled() && conv_variant() == 4 && conv_pr_supported(kind, bn, th);
}

hipError_t launch_conv(int dtype, int kind, int bn, const ConvArgs& a, hipStream_t s)
{
    if (a.use_stem2) return launch_stem2(a, s);
    if (a.use_pr && kind == KIND_C3S2) return launch_conv_pr(dtype, a, s);
    if (conv_ws_enabled() && conv_ws_supported(kind, bn)) {
        static const int dbg = diag_env("CCN_DBG") ? atoi(diag_env("CCN_DBG")) : 0;
        ConvArgs d = a; d.dbg = dbg;
        if (a.use_pr) return launch_conv_pr(dtype, d, s);        // decided at plan time (variant 4, bf16, 8-row tiles)
        // free-running kernel on 8-row tiles only: on 4-row tiles a consumer wave would issue 8 DMA pieces per 16 MFMAs and
        // the private weight copies double the L2 traffic of the already weight-heavy 128-pixel tile
        const int variant = conv_variant();
        if (variant == 2 || (variant >= 3 && a.th == 8)) return launch_conv_fr(dtype, bn, d, s);
        return launch_conv_ws(dtype, bn, d, s);
    }
    if (a.th != 4) return hipErrorInvalidValue;
    const conv_fn_t fn = pick(dtype, kind, bn);
    const unsigned grid = (unsigned)(a.B * a.n_ty * a.n_tx * a.npar * a.n_nt);
    hipLaunchKernelGGL(fn, dim3(grid), dim3(256), conv_lds_bytes(dtype, kind, bn), s, a);
    return hipGetLastError();
}

// ---- GroupNorm-apply + SiLU pre-pass ----------------------------------------------------------------------
bool conv_wants_preact(int kind, int bn, int n_nt)
{
    static const int mode = diag_env("CCN_PREACT") ? atoi(diag_env("CCN_PREACT")) : 1;    // 0 never, 1 when n_nt >= 2, 2 always
    if (!conv_ws_enabled() || !conv_ws_supported(kind, bn) || kind != KIND_C3S1) return false;
    return mode == 2 || (mode == 1 && n_nt >= 2);   // (launch_gn_act needs C/EPC <= 256; wider layers keep the fused prologue)
}

// thread -> fixed 16-byte channel slice (coefficients loaded once) x 4 pixels, all four loads in flight together
template <typename T>
__global__ __launch_bounds__(256) void gn_act_kernel(const T* __restrict__ x, const float2* __restrict__ ab, T* __restrict__ y,
                                                      int HW, int C)
{
    constexpr int EPC = Vec16<T>::EPC;
    const int nsl = C / EPC;                         // 16-byte slices per pixel (<= 256)
    const int pstep = 256 / nsl;                     // pixels covered by one pass of the block
    const int b = blockIdx.y, tid = threadIdx.x;
    const int sl = tid % nsl, pp = tid / nsl;
    if (pp >= pstep) return;
    const int p0 = blockIdx.x * 4 * pstep + pp;
    GnCoef<T> gk;
    gk.load(ab + (size_t)b * C + sl * EPC, true);
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k * pstep;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (p < HW) v[k] = *(const u32x4*)(x + ((size_t)b * HW + p) * C + sl * EPC);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k * pstep;
        if (p < HW) *(u32x4*)(y + ((size_t)b * HW + p) * C + sl * EPC) = gk.template apply<true>(v[k]);
    }
}
// The same pass with the GroupNorm finalize folded in: every workgroup first reduces the producer's partial sums of its
// sample (G groups x a few hundred slots, L2 resident; a fixed summation order, so deterministic) and
// forms its threads' scale/shift in registers -- one launch and one ~5 us dependency step less per pre-activated conv.
// `ab_out` / `stats_out` (training forward, else null): the first workgroup of every sample also writes the scale / shift table and
// (mean, rstd) per group, which the backward pass and the weight-gradient kernel read -- the separate gn_stats launch disappears.
template <typename T>
__global__ __launch_bounds__(256) void gn_act_fused_kernel(const T* __restrict__ x, T* __restrict__ y, int HW, int C,
                                                            const float2* __restrict__ part, int G, int nslot, int n_nt, int bn, int cpg,
                                                            double count, const float* __restrict__ gamma, const float* __restrict__ beta, float eps,
                                                            float2* __restrict__ ab_out, float2* __restrict__ stats_out)
{
    constexpr int EPC = Vec16<T>::EPC;
    __shared__ double st[64][2];
    const int nsl = C / EPC, pstep = 256 / nsl;
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sl = tid % nsl, pp = tid / nsl;
    const int p0 = blockIdx.x * 4 * pstep + pp;
    const bool act = pp < pstep;
    u32x4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                       // the data loads fly while the statistics are reduced
        const int p = p0 + k * pstep;
        v[k] = u32x4{0u, 0u, 0u, 0u};
        if (act && p < HW) v[k] = *(const u32x4*)(x + ((size_t)b * HW + p) * C + sl * EPC);
    }
    for (int g = wave; g < G; g += 4) {
        double mean, rstd;
        gn_group_stats(part, b, g, G, nslot, n_nt, bn, cpg, count, eps, lane, mean, rstd);
        if (lane == 0) { st[g][0] = mean; st[g][1] = rstd; }
    }
    __syncthreads();
    if (ab_out && blockIdx.x == 0) {
        for (int c = tid; c < C; c += 256) {
            const int g = c / cpg;
            const double sc = (double)gamma[c] * st[g][1];
            float* const row = (float*)(ab_out + (size_t)b * C) + 4 * (c >> 1) + (c & 1);      // pair-interleaved (GnCoef::load)
            row[0] = (float)sc; row[2] = (float)((double)beta[c] - st[g][0] * sc);
        }
        if (tid < G) stats_out[b * G + tid] = make_float2((float)st[tid][0], (float)st[tid][1]);
    }
    if (!act) return;
    GnCoef<T> gk;
#pragma unroll
    for (int e = 0; e < EPC; ++e) {
        const int c = sl * EPC + e, g = c / cpg;
        const double sc = (double)gamma[c] * st[g][1];
        gk.a[e] = (float)sc; gk.c[e] = (float)((double)beta[c] - st[g][0] * sc);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k * pstep;
        if (p < HW) *(u32x4*)(y + ((size_t)b * HW + p) * C + sl * EPC) = gk.template apply<true>(v[k]);
    }
}
hipError_t launch_gn_act_fused(int dtype, const void* x, void* y, int B, int HW, int C, const float2* part, int G, int n_sp, int n_nt,
                               int bn, int cpg, double count, const float* gamma, const float* beta, float eps, hipStream_t s,
                               float2* ab_out, float2* stats_out)
{
    const int epc = dtype == 0 ? 4 : 8, nsl = C / epc;
    if (nsl > 256 || nsl <= 0 || G > 64) return hipErrorInvalidValue;
    const int pstep = 256 / nsl;
    const dim3 grid((HW + 4 * pstep - 1) / (4 * pstep), B);
    if (dtype == 0) hipLaunchKernelGGL(gn_act_fused_kernel<float>, grid, dim3(256), 0, s, (const float*)x, (float*)y, HW, C, part, G, n_sp * n_nt, n_nt, bn, cpg, count, gamma, beta, eps, ab_out, stats_out);
    else hipLaunchKernelGGL(gn_act_fused_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, (__bf16*)y, HW, C, part, G, n_sp * n_nt, n_nt, bn, cpg, count, gamma, beta, eps, ab_out, stats_out);
    return hipGetLastError();
}

hipError_t launch_gn_act(int dtype, const void* x, const float2* ab, void* y, int B, int HW, int C, hipStream_t s)
{
    const int epc = dtype == 0 ? 4 : 8, nsl = C / epc;
    if (nsl > 256 || nsl <= 0) return hipErrorInvalidValue;
    const int pstep = 256 / nsl;
    const dim3 grid((HW + 4 * pstep - 1) / (4 * pstep), B);
    if (dtype == 0) hipLaunchKernelGGL(gn_act_kernel<float>, grid, dim3(256), 0, s, (const float*)x, ab, (float*)y, HW, C);
    else hipLaunchKernelGGL(gn_act_kernel<__bf16>, grid, dim3(256), 0, s, (const __bf16*)x, ab, (__bf16*)y, HW, C);
    return hipGetLastError();
}

// ---- GroupNorm finalize ----------------------------------------------------------------------------------
// One workgroup of 256 threads per (sample, group): the launch is pure latency (a dependency step between two convs), so
// the slot loop is spread over four waves (plain loads: the producer kernel has completed) and combined in a fixed order.
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float2* __restrict__ part, int G, int n_sp, int n_nt, int bn,
                                                           int cpg, int C, double count, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, float2* __restrict__ ab)
{
    __shared__ double red[4][2];
    const int bg = blockIdx.x, b = bg / G, g = bg % G, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jlo = (g * cpg) / bn, jhi = ((g + 1) * cpg - 1) / bn, nj = jhi - jlo + 1, ne = n_sp * nj;
    const float2* base = part + (size_t)(b * G + g) * n_sp * n_nt;
    double s1 = 0.0, s2 = 0.0;
    for (int e = tid; e < ne; e += 256) {
        const int sp = e / nj, j = jlo + (e - sp * nj);
        const float2 v = base[(size_t)sp * n_nt + j];
        s1 += (double)v.x; s2 += (double)v.y;
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) { s1 += __shfl_xor(s1, s); s2 += __shfl_xor(s2, s); }
    if (lane == 0) { red[wave][0] = s1; red[wave][1] = s2; }
    __syncthreads();
    s1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    s2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double mean = s1 / count;
    double var = s2 / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    for (int c = g * cpg + tid; c < (g + 1) * cpg; c += 256) {
        const double sc = (double)gamma[c] * rstd;
        // pair-interleaved: channels (2p, 2p+1) -> {scale, scale, shift, shift} (see GnCoef::load)
        float* const row = (float*)(ab + (size_t)b * C) + 4 * (c >> 1) + (c & 1);
        row[0] = (float)sc; row[2] = (float)((double)beta[c] - mean * sc);
    }
}

hipError_t launch_gn_finalize(const float2* part, int B, int G, int n_sp, int n_nt, int bn, int cpg, int C, double count,
                              const float* gamma, const float* beta, float eps, float2* ab, hipStream_t s)
{
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B * G), dim3(256), 0, s, part, G, n_sp, n_nt, bn, cpg, C, count, gamma, beta, eps, ab);
    return hipGetLastError();
}

// GroupNorm partial sums of an NHWC tensor (operator-level entry points only; the hot path gets them from the
// producing conv's epilogue).  Block (slot, b): a contiguous pixel range; thread -> (group, sub-range), fixed order.
template <typename T>
__global__ __launch_bounds__(256) void gn_partials_kernel(const T* __restrict__ x, float2* __restrict__ part, int HW, int C,
                                                           int cpg, int G, int nslot)
{
    __shared__ float sh[256 * 2];
    const int slot = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int per = (HW + nslot - 1) / nslot;
    const int p0 = slot * per, p1 = min(HW, p0 + per);
    const int nsub = 256 / G;
    const int g = tid % G, sub = tid / G;
    float s1 = 0.f, s2 = 0.f;
    if (sub < nsub)
        for (int p = p0 + sub; p < p1; p += nsub) {
            const T* row = x + ((size_t)b * HW + p) * C + g * cpg;
            for (int j = 0; j < cpg; ++j) { const float v = (float)row[j]; s1 += v; s2 = fmaf(v, v, s2); }
        }
    sh[tid * 2] = s1; sh[tid * 2 + 1] = s2;
    __syncthreads();
    if (tid < G) {
        float t1 = 0.f, t2 = 0.f;
        for (int q = 0; q < nsub; ++q) { t1 += sh[(q * G + tid) * 2]; t2 += sh[(q * G + tid) * 2 + 1]; }
        part[(size_t)(b * G + tid) * nslot + slot] = make_float2(t1, t2);
    }
}
hipError_t launch_gn_partials(int dtype, const void* x, float2* part, int B, int HW, int C, int cpg, int G, int nslot, hipStream_t s)
{
    if (G > 256) return hipErrorInvalidValue;
    if (dtype == 0) hipLaunchKernelGGL(gn_partials_kernel<float>, dim3(nslot, B), dim3(256), 0, s, (const float*)x, part, HW, C, cpg, G, nslot);
    else hipLaunchKernelGGL(gn_partials_kernel<__bf16>, dim3(nslot, B), dim3(256), 0, s, (const __bf16*)x, part, HW, C, cpg, G, nslot);
    return hipGetLastError();
}

// ---- conditioning ------------------------------------------------------------------------------------------
// timestep_embedding (models/unet.py:33-36): freqs = exp(-ln(1e4) * i / half) in fp32, [cos(t f) | sin(t f)].
template <typename TI>
__global__ void temb_kernel(const TI* __restrict__ t, float* __restrict__ out, int n, int dim)
{
    const int half = dim / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * dim) return;
    const int row = idx / dim, i = idx - row * dim;
    float v = 0.0f;
    if (i < 2 * half) {
        const int fi = i < half ? i : i - half;
        const float f = expf(__fdiv_rn(__fmul_rn(-9.210340371976184f, (float)fi), (float)half));
        const float arg = __fmul_rn((float)t[row], f);
        v = i < half ? cosf(arg) : sinf(arg);
    }
    out[idx] = v;   // odd dim: last column zero (F.pad, models/unet.py:37-38)
}
hipError_t launch_temb_i64(const int64_t* t, float* out, int n, int dim, hipStream_t s)
{
    hipLaunchKernelGGL(temb_kernel<int64_t>, dim3((n * dim + 255) / 256), dim3(256), 0, s, t, out, n, dim);
    return hipGetLastError();
}
hipError_t launch_temb_i32(const int32_t* t, float* out, int n, int dim, hipStream_t s)
{
    hipLaunchKernelGGL(temb_kernel<int32_t>, dim3((n * dim + 255) / 256), dim3(256), 0, s, t, out, n, dim);
    return hipGetLastError();
}

// One wave per output column n, 16 rows per block; lanes stride K, xor-shuffle reduce (fixed order).
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ xa, const float* __restrict__ xb, int a_div, int b_mod,
                                                      const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ y,
                                                      int R, int K, int N, int act_silu)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x * 4 + wave;
    if (n >= N) return;
    const int r0 = blockIdx.y * 16, r1 = min(R, r0 + 16);
    const float* wrow = W + (size_t)n * K;
    for (int rr = r0; rr < r1; ++rr) {
        const float* pa = xa + (size_t)(rr / a_div) * K;
        const float* pb = xb ? xb + (size_t)(rr % b_mod) * K : nullptr;
        float acc = 0.f;
        for (int k = lane; k < K; k += 64) {
            float xv = pa[k];
            if (pb) xv += pb[k];
            acc = fmaf(xv, wrow[k], acc);
        }
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) acc += __shfl_xor(acc, s);
        if (lane == 0) {
            float v = acc + (bias ? bias[n] : 0.f);
            if (act_silu) v = v / (1.0f + expf(-v));
            y[(size_t)rr * N + n] = v;
        }
    }
}
hipError_t launch_linear(const float* xa, const float* xb, int a_div, int b_mod, const float* W, const float* b, float* y,
                         int R, int K, int N, int act_silu, hipStream_t s)
{
    hipLaunchKernelGGL(linear_kernel, dim3((N + 3) / 4, (R + 15) / 16), dim3(256), 0, s, xa, xb, a_div, b_mod, W, b, y, R, K, N, act_silu);
    return hipGetLastError();
}

// ---- elementwise ----------------------------------------------------------------------------------------------
__global__ void ddim_step_kernel(float* __restrict__ x, const float* __restrict__ eps, const float* __restrict__ noise,
                                 float c0, float c1, float c2, float c3, float sigma, int64_t n)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const float e = eps[i];
        float x0 = __fdiv_rn(__fsub_rn(x[i], __fmul_rn(c0, e)), c1);
        x0 = fminf(fmaxf(x0, -1.0f), 1.0f);
        float xn = __fadd_rn(__fmul_rn(c2, x0), __fmul_rn(c3, e));
        if (noise) xn = __fadd_rn(xn, __fmul_rn(sigma, noise[i]));
        x[i] = xn;
    }
}
hipError_t launch_ddim_step(float* x, const float* eps, const float* noise, float c0, float c1, float c2, float c3, float sigma,
                            int64_t n, hipStream_t s)
{
    const int grid = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(ddim_step_kernel, dim3(grid > 0 ? grid : 1), dim3(256), 0, s, x, eps, noise, c0, c1, c2, c3, sigma, n);
    return hipGetLastError();
}

__global__ void q_sample_kernel(float* __restrict__ out, const float* __restrict__ x0, const float* __restrict__ noise,
                                const float* __restrict__ a, const float* __restrict__ sg, int64_t per)
{
    const int b = blockIdx.y;
    const float ca = a[b], cs = sg[b];
    const size_t base = (size_t)b * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x)
        out[base + i] = __fadd_rn(__fmul_rn(ca, x0[base + i]), __fmul_rn(cs, noise[base + i]));
}
hipError_t launch_q_sample(float* out, const float* x0, const float* noise, const float* a, const float* sg, int B, int64_t per,
                           hipStream_t s)
{
    const int gx = (int)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024);
    hipLaunchKernelGGL(q_sample_kernel, dim3(gx > 0 ? gx : 1, B), dim3(256), 0, s, out, x0, noise, a, sg, per);
    return hipGetLastError();
}

__global__ void predict_x0_kernel(float* __restrict__ out, const float* __restrict__ xt, const float* __restrict__ eps,
                                  const float* __restrict__ a, const float* __restrict__ sg, int64_t per)
{
    const int b = blockIdx.y;
    const float ca = a[b], cs = sg[b];
    const size_t base = (size_t)b * per;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (int64_t)gridDim.x * blockDim.x)
        out[base + i] = __fdiv_rn(__fsub_rn(xt[base + i], __fmul_rn(cs, eps[base + i])), ca);
}
hipError_t launch_predict_x0(float* out, const float* xt, const float* eps, const float* a, const float* sg, int B, int64_t per,
                             hipStream_t s)
{
    const int gx = (int)((per + 255) / 256 < 1024 ? (per + 255) / 256 : 1024);
    hipLaunchKernelGGL(predict_x0_kernel, dim3(gx > 0 ? gx : 1, B), dim3(256), 0, s, out, xt, eps, a, sg, per);
    return hipGetLastError();
}

// y = x*(1+s) + b on NCHW fp32 (stand-alone FiLM, models/blocks.py:25; the UNet path fuses this into conv1's epilogue)
__global__ void film_nchw_kernel(const float* __restrict__ x, const float* __restrict__ sc, const float* __restrict__ sh,
                                 float* __restrict__ y, int64_t hw)
{
    const int bc = blockIdx.y;
    const float s1 = __fadd_rn(1.0f, sc[bc]), b1 = sh[bc];
    const size_t base = (size_t)bc * hw;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x)
        y[base + i] = __fadd_rn(__fmul_rn(x[base + i], s1), b1);
}
hipError_t launch_film_nchw(const float* x, const float* sc, const float* sh, float* y, int B, int C, int64_t hw, hipStream_t s)
{
    const int gx = (int)((hw + 255) / 256 < 256 ? (hw + 255) / 256 : 256);
    hipLaunchKernelGGL(film_nchw_kernel, dim3(gx > 0 ? gx : 1, B * C), dim3(256), 0, s, x, sc, sh, y, hw);
    return hipGetLastError();
}

template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int C, int HW, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        const int64_t bp = i / C, p = bp % HW, b = bp / HW;
        dst[i] = (T)src[((size_t)b * C + c) * HW + p];
    }
}
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ src, float* __restrict__ dst, int C, int HW, int64_t total)
{
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = i % HW, bc = i / HW;
        const int c = (int)(bc % C);
        const int64_t b = bc / C;
        dst[i] = (float)src[((size_t)b * HW + p) * C + c];
    }
}
hipError_t launch_nchw_to_nhwc(int dtype, const float* src, void* dst, int B, int C, int H, int W, hipStream_t s)
{
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == 0) hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3(grid), dim3(256), 0, s, src, (float*)dst, C, H * W, total);
    else hipLaunchKernelGGL(nchw_to_nhwc_kernel<__bf16>, dim3(grid), dim3(256), 0, s, src, (__bf16*)dst, C, H * W, total);
    return hipGetLastError();
}
hipError_t launch_nhwc_to_nchw(int dtype, const void* src, float* dst, int B, int C, int H, int W, hipStream_t s)
{
    const int64_t total = (int64_t)B * C * H * W;
    const int grid = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (dtype == 0) hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)src, dst, C, H * W, total);
    else hipLaunchKernelGGL(nhwc_to_nchw_kernel<__bf16>, dim3(grid), dim3(256), 0, s, (const __bf16*)src, dst, C, H * W, total);
    return hipGetLastError();
}

}  // namespace ccn
