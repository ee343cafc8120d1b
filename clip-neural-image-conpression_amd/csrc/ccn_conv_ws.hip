// Warp-specialised implicit-GEMM convolution for the stride-1 3x3 convs and the ConvTranspose parities (the
// 3x3-s1 ResBlock convs are 87 % of the path's FLOPs).
//
// One workgroup = 8 waves on one CU:
//   waves 0-3  consumers: nothing but ds_read_b128 + MFMA over the staged tiles (2x2 waves, MF x NF fragments each)
//   waves 4-7  producers: run one pipeline step ahead -- global loads of the next input-halo chunk and weight
//              stage, the GroupNorm-apply + SiLU transform (VALU, transcendental-heavy), ds_write into the
//              *other* LDS buffer
// so the VALU prologue fusion and the staging traffic overlap the matrix pipe instead of alternating with it
// (MFMA and VALU are separate pipes; a wave issues in order, so the overlap has to come from different waves).
// Both operands are double buffered; ONE workgroup barrier per (Cin-chunk, tap) iteration orders everything:
//   iteration it:  consumers read A[c&1], B[it&1];  producers write B[(it+1)&1] and a slice of A[(c+1)&1]
// The epilogue (accumulators -> LDS fp32 tile -> bias / FiLM / residual -> coalesced 16-B NHWC rows + the next
// GroupNorm's partial sums) is shared by all 8 waves, 128 pixels per pass, with the residual rows prefetched
// into registers before the pass barrier.
//
// Tile: TH x 32 output-space pixels (TH = 4*MF/2: 4 or 8 rows of 32) x BN = 64*NF output channels.
#include "ccn_device.h"

namespace ccn {

namespace {

template <int TH> struct WsGeom {
    static constexpr int HROWS = TH + 2, HPITCH = 34;
    static constexpr int A_BYTES = HROWS * HPITCH * 128;
    static constexpr int AU = HROWS * HPITCH * 8;            // 16-byte units per chunk
};
template <int TH, int BN> struct WsLds {
    static constexpr int A_BYTES = WsGeom<TH>::A_BYTES;
    static constexpr int B_BYTES = BN * 128;
    static constexpr int LOOP = 2 * A_BYTES + 2 * B_BYTES;
    static constexpr int CP = BN + 4;
    static constexpr int CS_BYTES = 128 * CP * 4;
    static constexpr int RED_BYTES = 8 * BN * 2 * 4 + BN * 2 * 4;
    static constexpr int TOTAL = LOOP > CS_BYTES + RED_BYTES ? LOOP : CS_BYTES + RED_BYTES;
};

}  // namespace

template <typename T, int MF, int NF>
__global__ __launch_bounds__(512) void conv_ws_kernel(const ConvArgs a)
{
    constexpr int WM = 2, WN = 2;
    constexpr int TH = WM * MF;                 // tile rows of 32 pixels
    constexpr int BN = WN * NF * 32;
    constexpr int EPC = Vec16<T>::EPC;
    constexpr int CKE = 8 * EPC;
    using G = WsGeom<TH>;
    using L = WsLds<TH, BN>;
    constexpr int HPITCH = G::HPITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;                         // 2 buffers
    unsigned char* const Bs = smem + 2 * L::A_BYTES;        // 2 buffers

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool producer = wave >= 4;
    const int r = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    const int nt = bid % a.n_nt; bid /= a.n_nt;
    const int par = bid % a.npar; bid /= a.npar;
    const int tx = bid % a.n_tx; bid /= a.n_tx;
    const int ty = bid % a.n_ty;
    const int b = bid / a.n_ty;
    const int my0 = ty * TH, mx0 = tx * 32, n0 = nt * BN;
    const int py = par >> 1, px_ = par & 1;
    const int par_off = par * 4;
    const int n_it = a.nchunk * a.ntaps;

    const unsigned char* const wbase = (const unsigned char*)a.w;
    const unsigned char* const inb = (const unsigned char*)a.in;
    const bool gn = a.gn_ab != nullptr;
    const int iy0 = my0 - 1, ix0 = mx0 - 1;

    // A unit u of a chunk: halo pixel px = u >> 3, 16-byte channel slice ck = u & 7
    auto a_src = [&](int chunk, int px, int ck, bool& ok) -> const unsigned char* {
        const int hy = px / HPITCH, hx = px - hy * HPITCH;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const int cbase = chunk * CKE + ck * EPC;
        ok = px < G::HROWS * HPITCH && cbase < a.Cin && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
        return inb + (((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + cbase) * sizeof(T);
    };
    auto b_src = [&](int it, int n, int ck) -> const unsigned char* {
        const int chunk = it / a.ntaps, tap = it - chunk * a.ntaps;
        const int wt = a.tapinfo_w(par_off + tap);
        return wbase + ((size_t)(wt * a.Cout_pad + n0 + n) * a.Cin_pad + (size_t)chunk * CKE) * sizeof(T) + ck * 16;
    };

    // ------------------------------------------------------------------ prologue: chunk 0 and weight stage 0 by all 512 threads
    {
        const int ck = tid & 7;
        GnCoef<T> gk;
        const bool cv = ck * EPC < a.Cin;
        gk.load(a.gn_ab + (size_t)b * a.Cin + (cv ? ck * EPC : 0), gn && cv);
        constexpr int PIT = (G::AU + 511) / 512;
        u32x4 raw[PIT];
        unsigned okm = 0;
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            bool ok;
            const unsigned char* src = a_src(0, (tid >> 3) + 64 * i, ck, ok);
            raw[i] = u32x4{0u, 0u, 0u, 0u};
            if (ok) { raw[i] = *(const u32x4*)src; okm |= 1u << i; }
        }
        constexpr int BU0 = BN * 8 / 512;
        u32x4 b0[BU0 > 0 ? BU0 : 1];
#pragma unroll
        for (int u = 0; u < BU0; ++u) { const int idx = tid + 512 * u; b0[u] = *(const u32x4*)b_src(0, idx >> 3, idx & 7); }
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int px = (tid >> 3) + 64 * i;
            if (px < G::HROWS * HPITCH) {
                u32x4 o = raw[i];
                if (((okm >> i) & 1u) && gn) o = gk.template apply<true>(raw[i]);
                *(u32x4*)(As + px * 128 + (((ck ^ (px >> 1)) & 7) << 4)) = o;
            }
        }
#pragma unroll
        for (int u = 0; u < BU0; ++u) {
            const int idx = tid + 512 * u, n = idx >> 3, ckb = idx & 7;
            *(u32x4*)(Bs + n * 128 + (((ckb ^ (n >> 1)) & 7) << 4)) = b0[u];
        }
    }

    f32x16 acc[MF][NF];
#pragma unroll
    for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    if (producer) {
        // ------------------------------------------------------------------ producers
        // Pipeline depths are set by memory latency, not by the iteration length (~1k cycles):
        //   weights (L2-resident): two register sets, loaded two iterations ahead of their ds_write;
        //   input halo (HBM): ALL of the next chunk's 16-byte units are requested at tap 0 of the current chunk and
        //   retired (GroupNorm+SiLU, ds_write) a few per iteration over the remaining taps.
        const int ptid = tid - 256, ck = ptid & 7;
        constexpr int BU = BN * 8 / 256;                          // weight units per producer thread per stage
        constexpr int AIT = (G::AU + 255) / 256;                  // A units per producer thread per chunk
        u32x4 bset0[BU], bset1[BU];
        u32x4 areg[AIT];
        unsigned aok = 0, avalid = 0;
        unsigned aoff[AIT];                                       // byte offset of unit i inside the input tensor (chunk 0)
        unsigned boff[BU];
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            const int px = (ptid >> 3) + 32 * i;
            const int hy = px / HPITCH, hx = px - hy * HPITCH;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = px < G::HROWS * HPITCH && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            if (ok) avalid |= 1u << i;
            aoff[i] = ok ? (unsigned)((((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + ck * EPC) * sizeof(T)) : 0u;
        }
#pragma unroll
        for (int u = 0; u < BU; ++u) {
            const int idx = ptid + 256 * u;
            boff[u] = (unsigned)(((size_t)(n0 + (idx >> 3)) * a.Cin_pad * sizeof(T)) + (idx & 7) * 16);
        }
        const int upi = (AIT + a.ntaps - 2) / (a.ntaps - 1);      // units retired per iteration at taps 1..ntaps-1
        GnCoef<T> gk;

#define WS_LOAD_B(SET, IT)                                                                                          \
        if ((IT) < n_it) {                                                                                          \
            const int chunk_ = (IT) / a.ntaps, tap_ = (IT) - chunk_ * a.ntaps;                                      \
            const unsigned char* wb_ = wbase + ((size_t)a.tapinfo_w(par_off + tap_) * a.Cout_pad * a.Cin_pad + (size_t)chunk_ * CKE) * sizeof(T); \
            _Pragma("unroll") for (int u = 0; u < BU; ++u) SET[u] = *(const u32x4*)(wb_ + boff[u]);                \
        }
#define WS_STORE_B(SET, IT)                                                                                         \
        if ((IT) < n_it) {                                                                                          \
            _Pragma("unroll") for (int u = 0; u < BU; ++u) {                                                        \
                const int idx = ptid + 256 * u, n = idx >> 3, ckb = idx & 7;                                        \
                *(u32x4*)(Bs + ((IT) & 1) * L::B_BYTES + n * 128 + (((ckb ^ (n >> 1)) & 7) << 4)) = SET[u];        \
            }                                                                                                       \
        }
        // A work of iteration IT (chunk c, tap t): t == 0 requests chunk c+1, t >= 1 retires units [(t-1)*upi, t*upi)
#define WS_A_WORK(IT)                                                                                               \
        {                                                                                                           \
            const int c_ = (IT) / a.ntaps, t_ = (IT) - c_ * a.ntaps;                                                \
            if (c_ + 1 < a.nchunk) {                                                                                \
                const int cb_ = (c_ + 1) * CKE + ck * EPC;                                                          \
                const bool cv_ = cb_ < a.Cin;                                                                       \
                if (t_ == 0) {                                                                                      \
                    gk.load(a.gn_ab + (size_t)b * a.Cin + (cv_ ? cb_ : 0), gn && cv_);                              \
                    aok = cv_ ? avalid : 0u;                                                                        \
                    const unsigned char* ab_ = inb + (size_t)(c_ + 1) * CKE * sizeof(T);                            \
                    _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                               \
                        areg[i] = u32x4{0u, 0u, 0u, 0u};                                                            \
                        if ((aok >> i) & 1u) areg[i] = *(const u32x4*)(ab_ + aoff[i]);                              \
                    }                                                                                               \
                } else {                                                                                            \
                    unsigned char* const Ad_ = As + ((c_ + 1) & 1) * L::A_BYTES;                                    \
                    const int lo_ = (t_ - 1) * upi, hi_ = lo_ + upi;                                                \
                    _Pragma("unroll") for (int i = 0; i < AIT; ++i) {                                               \
                        if (i >= lo_ && i < hi_) {                                                                  \
                            const int px = (ptid >> 3) + 32 * i;                                                    \
                            if (px < G::HROWS * HPITCH) {                                                           \
                                u32x4 o_ = areg[i];                                                                 \
                                if (((aok >> i) & 1u) && gn) o_ = gk.template apply<true>(areg[i]);                \
                                *(u32x4*)(Ad_ + px * 128 + (((ck ^ (px >> 1)) & 7) << 4)) = o_;                     \
                            }                                                                                       \
                        }                                                                                           \
                    }                                                                                               \
                }                                                                                                   \
            }                                                                                                       \
        }

        WS_LOAD_B(bset1, 1)
        WS_LOAD_B(bset0, 2)
        __syncthreads();                                           // prologue tiles visible
        for (int it = 0; it < n_it; it += 2) {
            if (!(a.dbg & 2)) { WS_STORE_B(bset1, it + 1)          // stage it+1, requested two iterations ago
            WS_LOAD_B(bset1, it + 3) }
            if (!(a.dbg & 1)) WS_A_WORK(it)
            __syncthreads();
            if (it + 1 < n_it) {
                if (!(a.dbg & 2)) { WS_STORE_B(bset0, it + 2)
                WS_LOAD_B(bset0, it + 4) }
                if (!(a.dbg & 1)) WS_A_WORK(it + 1)
                __syncthreads();
            }
        }
#undef WS_LOAD_B
#undef WS_STORE_B
#undef WS_A_WORK
    } else {
        // ------------------------------------------------------------------ consumers
        const int wm = wave / WN, wn = wave % WN;
        int pbase[MF], nrow[NF];
#pragma unroll
        for (int i = 0; i < MF; ++i) pbase[i] = ((wm * MF + i) + 1) * HPITCH + r + 1;
#pragma unroll
        for (int j = 0; j < NF; ++j) nrow[j] = (wn * NF + j) * 32 + r;
        __syncthreads();                                           // prologue tiles visible
        for (int it = 0; it < n_it; ++it) {
            const int chunk = it / a.ntaps, tap = it - chunk * a.ntaps;
            const int dy = a.tapinfo_dy(par_off + tap), dx = a.tapinfo_dx(par_off + tap);
            const unsigned char* const Ab = As + (chunk & 1) * L::A_BYTES;
            const unsigned char* const Bb = Bs + (it & 1) * L::B_BYTES;
            int pa[MF];
#pragma unroll
            for (int i = 0; i < MF; ++i) pa[i] = pbase[i] + dy * HPITCH + dx;
            if (!(a.dbg & 4))
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                u32x4 av[MF], bv[NF];
#pragma unroll
                for (int i = 0; i < MF; ++i)
                    av[i] = *(const u32x4*)(Ab + pa[i] * 128 + ((((2 * kk + h) ^ (pa[i] >> 1)) & 7) << 4));
#pragma unroll
                for (int j = 0; j < NF; ++j)
                    bv[j] = *(const u32x4*)(Bb + nrow[j] * 128 + ((((2 * kk + h) ^ (nrow[j] >> 1)) & 7) << 4));
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j) mfma16<T>(acc[i][j], av[i], bv[j]);
            }
            __syncthreads();
        }
    }

    // ------------------------------------------------------------------ epilogue, 128 pixels (4 tile rows) per pass, all 8 waves
    float* const Cs = (float*)smem;
    constexpr int CP = L::CP;
    constexpr int NOCT = BN / 8, PSL = 512 / NOCT, NIT = 128 / PSL;
    const int o = tid % NOCT, ps = tid / NOCT;
    const int nb = n0 + o * 8;
    const bool nvalid = nb < a.Cout;
    float f1[8], f2[8];                         // v = acc * f1 + f2  with f2 = bias * f1 + shift
#pragma unroll
    for (int e = 0; e < 8; ++e) { f1[e] = 1.f; f2[e] = 0.f; }
    if (nvalid) {
#pragma unroll
        for (int e = 0; e < 8; ++e) f2[e] = a.bias[nb + e];
        if (a.film) {
            const float* fp = a.film + (size_t)b * a.film_bstride;
#pragma unroll
            for (int e = 0; e < 8; ++e) { f1[e] = 1.0f + fp[nb + e]; f2[e] = fmaf(f2[e], f1[e], fp[a.Cout + nb + e]); }
        }
    }
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    unsigned char* const outb = (unsigned char*)a.out;
    const unsigned char* const resb = (const unsigned char*)a.res;
    constexpr int NPASS = TH / 4;
    if (!(a.dbg & 8))
#pragma unroll
    for (int pass = 0; pass < NPASS; ++pass) {
        if (pass > 0) __syncthreads();                              // previous pass finished reading Cs
        if (!producer) {
            const int wm = wave / WN, wn = wave % WN;
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int row = wm * MF + i;                        // tile row of this fragment
                if (row / 4 == pass) {
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int q = 0; q < 16; ++q) {
                            const int m = (row & 3) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                            Cs[m * CP + (wn * NF + j) * 32 + r] = acc[i][j][q];
                        }
                }
            }
        }
        // residual / skip rows of this pass: issue every load before the barrier
        u32x4 rres[NIT][EPC == 8 ? 1 : 2];
        size_t eoff[NIT];
        unsigned vmask = 0;
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp) {
            const int m = itp * PSL + ps;
            const int my = my0 + pass * 4 + (m >> 5), mx = mx0 + (m & 31);
            const bool v = nvalid && my < a.MH && mx < a.MW;
            const int oy = my * a.OS + py, ox = mx * a.OS + px_;
            eoff[itp] = (((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + nb) * sizeof(T);
            if (v) vmask |= 1u << itp;
#pragma unroll
            for (int w = 0; w < (EPC == 8 ? 1 : 2); ++w) {
                rres[itp][w] = u32x4{0u, 0u, 0u, 0u};
                if (v && resb) rres[itp][w] = *(const u32x4*)(resb + eoff[itp] + 16 * w);
            }
        }
        __syncthreads();
#pragma unroll
        for (int itp = 0; itp < NIT; ++itp) {
            if ((vmask >> itp) & 1u) {
                const int m = itp * PSL + ps;
                float v[8];
                const f32x4 c0 = *(const f32x4*)(Cs + m * CP + o * 8), c1 = *(const f32x4*)(Cs + m * CP + o * 8 + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = c0[e]; v[4 + e] = c1[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], f1[e], f2[e]);
                if (resb) {
                    float rv[8];
                    Vec16<T>::unpack(rres[itp][0], rv);
                    if constexpr (EPC == 4) Vec16<T>::unpack(rres[itp][1], rv + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                *(u32x4*)(outb + eoff[itp]) = Vec16<T>::pack(v);
                if constexpr (EPC == 4) *(u32x4*)(outb + eoff[itp] + 16) = Vec16<T>::pack(v + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
            }
        }
    }
    if (a.part) {
        // fixed-order reduction: lanes sharing an octet, then the 8 waves, then the channels of each group
#pragma unroll
        for (int s = NOCT; s < 64; s <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], s); s2[e] += __shfl_xor(s2[e], s); }
        float* const red = (float*)(smem + L::CS_BYTES);        // [8][BN][2]
        float* const chs = red + 8 * BN * 2;                    // [BN][2]
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
            for (int w = 0; w < 8; ++w) { t1 += red[(w * BN + tid) * 2]; t2 += red[(w * BN + tid) * 2 + 1]; }
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
                a.part[(size_t)(b * a.G + g) * a.nslot + slot] = make_float2(t1, t2);
            }
        }
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------
typedef void (*ws_fn_t)(const ConvArgs);

static ws_fn_t pick_ws(int dtype, int th, int bn)
{
    if (dtype == 0) {
        if (th == 8) return bn == 128 ? conv_ws_kernel<float, 4, 2> : conv_ws_kernel<float, 4, 1>;
        return bn == 128 ? conv_ws_kernel<float, 2, 2> : conv_ws_kernel<float, 2, 1>;
    }
    if (th == 8) return bn == 128 ? conv_ws_kernel<__bf16, 4, 2> : conv_ws_kernel<__bf16, 4, 1>;
    return bn == 128 ? conv_ws_kernel<__bf16, 2, 2> : conv_ws_kernel<__bf16, 2, 1>;
}
static size_t ws_lds(int th, int bn)
{
    if (th == 8) return bn == 128 ? WsLds<8, 128>::TOTAL : WsLds<8, 64>::TOTAL;
    return bn == 128 ? WsLds<4, 128>::TOTAL : WsLds<4, 64>::TOTAL;
}

bool conv_ws_supported(int kind, int bn) { return (kind == KIND_C3S1 || kind == KIND_CT4) && (bn == 128 || bn == 64); }

hipError_t conv_ws_prepare()
{
    for (int dt = 0; dt < 2; ++dt)
        for (int th = 4; th <= 8; th += 4)
            for (int bn = 64; bn <= 128; bn += 64) {
                hipError_t e = hipFuncSetAttribute((const void*)pick_ws(dt, th, bn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_lds(th, bn));
                if (e != hipSuccess) return e;
            }
    return hipSuccess;
}

hipError_t launch_conv_ws(int dtype, int bn, const ConvArgs& a, hipStream_t s)
{
    const unsigned grid = (unsigned)(a.B * a.n_ty * a.n_tx * a.npar * a.n_nt);
    hipLaunchKernelGGL(pick_ws(dtype, a.th, bn), dim3(grid), dim3(512), ws_lds(a.th, bn), s, a);
    return hipGetLastError();
}

}  // namespace ccn
