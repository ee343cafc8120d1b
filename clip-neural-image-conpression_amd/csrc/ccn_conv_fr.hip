// "Free-running" warp-specialised implicit-GEMM convolution: no workgroup barrier inside a Cin chunk.
//
// Measurements on the barrier-per-stage kernels (ccn_conv_ws.hip and a shared LDS-DMA-ring variant of it) showed the consumer waves are the
// pole: they hardly wait at the barriers, but every stage restarts their LDS-read -> MFMA pipeline behind a barrier
// and the loop runs at ~50 % of the MFMA rate.  Here the consumers never synchronise with anybody inside a chunk:
//   * weights (B): every consumer wave owns a PRIVATE ring of three one-tap slots for the 32*NF output channels it
//     multiplies, fills it itself by LDS-DMA (buffer_load_dwordx4 ... lds) two taps ahead, and orders its own reads
//     with its own counted s_waitcnt vmcnt -- no cross-wave hand-off at all (the two waves that share channels each
//     fetch their copy: 2x weight traffic from L2, which an 8-row tile can afford);
//   * input (A): ONE LDS buffer per workgroup.  The four producer waves hold the next chunk in registers (requested a
//     whole chunk ahead), apply GroupNorm + SiLU in place while the consumers compute, and at the chunk boundary --
//     two barriers -- dump it into the buffer;
//   * so a tile costs 1 + 2*(nchunk-1) barriers in its main loop instead of nchunk*ntaps, the fragment prefetch runs
//     straight through tap boundaries, and the producers' ds_writes are confined to the boundaries.
#include "ccn_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace ccn {

namespace {

template <int TH> struct FrGeom {
    static constexpr int HROWS = TH + 2, HPITCH = 34;
    static constexpr int A_BYTES = HROWS * HPITCH * 128;
    static constexpr int AU = HROWS * HPITCH * 8;            // 16-byte units per chunk
};
template <int TH, int BN, int NF> struct FrLds {
    static constexpr int NBUF = 3;
    static constexpr int A_BYTES = FrGeom<TH>::A_BYTES;
    static constexpr int BW_BYTES = NF * 32 * 128;           // one tap of one consumer wave's channels
    static constexpr int B_BYTES = 4 * NBUF * BW_BYTES;
    static constexpr int LOOP = A_BYTES + B_BYTES;
    static constexpr int CP = BN + 4;
    static constexpr int CS1_BYTES = 128 * CP * 4;
    static constexpr int CS_BYTES = (TH / 4) * CS1_BYTES;
    static constexpr int RED_BYTES = 8 * BN * 2 * 4 + BN * 2 * 4;
    static constexpr int TOTAL = LOOP > CS_BYTES + RED_BYTES ? LOOP : CS_BYTES + RED_BYTES;
};

}  // namespace

template <typename T, int MF, int NF, int NTAPS>
__global__ __launch_bounds__(512) void conv_fr_kernel(const ConvArgs a)
{
    constexpr int WM = 2, WN = 2;
    constexpr int TH = WM * MF;
    constexpr int BN = WN * NF * 32;
    constexpr int EPC = Vec16<T>::EPC;
    constexpr int CKE = 8 * EPC;
    constexpr int NA = 4, A0 = 4;               // waves 4..7 stage the input operand
    using G = FrGeom<TH>;
    using L = FrLds<TH, BN, NF>;
    constexpr int NBUF = L::NBUF;
    constexpr int HPITCH = G::HPITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const As = smem;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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

    const unsigned char* const wbase = (const unsigned char*)a.w;
    const unsigned char* const inb = (const unsigned char*)a.in;
    const bool gn = a.gn_ab != nullptr;
    const int iy0 = my0 - 1, ix0 = mx0 - 1;
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a) && lane == 0 && (wave == 0 || wave == A0)) {
            unsigned long long* st = CCN_STAMPS_PTR(a) + ((size_t)blockIdx.x * 3 + (wave == 0 ? 0 : 2)) * 8;
            st[slot] = __builtin_amdgcn_s_memrealtime();
            if (slot == 1) st[5] = __builtin_amdgcn_s_memtime();
            if (slot == 2) st[6] = __builtin_amdgcn_s_memtime();
        }
    };
    unsigned long long bar_wait = 0;
    // raw barrier: drain only this wave's LDS operations; in-flight buffer loads (register prefetch, LDS-DMA) survive it
    auto raw_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto loop_barrier = [&]() __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a)) { const unsigned long long t0 = __builtin_amdgcn_s_memtime(); raw_barrier(); bar_wait += __builtin_amdgcn_s_memtime() - t0; }
        else raw_barrier();
    };
    auto stamp_wait = [&]() __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a) && lane == 0 && (wave == 0 || wave == A0))
            CCN_STAMPS_PTR(a)[((size_t)blockIdx.x * 3 + (wave == 0 ? 0 : 2)) * 8 + 4] = bar_wait;
    };
    stamp(0);
    const size_t wtap_bytes = (size_t)a.Cout_pad * a.Cin_pad * sizeof(T);
    constexpr unsigned OOB = 0x7FFFFFF0u;       // past num_records: the buffer load returns zeros without touching memory
    const unsigned in_bytes = (unsigned)((size_t)a.B * a.Hin * a.Win * a.Cin * sizeof(T));
    auto in_srd = [&](int chunk) __attribute__((always_inline)) {
        const unsigned off = (unsigned)((size_t)chunk * CKE * sizeof(T));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(inb + off), 0, in_bytes - off, 0x00020000);
    };
    auto w_srd = [&](int tap, int chunk) __attribute__((always_inline)) {
        return __builtin_amdgcn_make_buffer_rsrc((void*)(wbase + (size_t)a.tapinfo_w(par_off + tap) * wtap_bytes + (size_t)chunk * CKE * sizeof(T)),
                                                 0, (unsigned)wtap_bytes, 0x00020000);
    };

    // ------------------------------------------------------------------ prologue: chunk 0 by all 512 threads
    {
        const int ck = tid & 7;
        GnCoef<T> gk;
        const bool cv = ck * EPC < a.Cin;
        gk.load(a.gn_ab + (size_t)b * a.Cin + (cv ? ck * EPC : 0), gn && cv);
        constexpr int PIT = (G::AU + 511) / 512;
        u32x4 raw[PIT];
        unsigned okm = 0;
        const auto srd0 = in_srd(0);
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int px = (tid >> 3) + 64 * i;
            const int hy = px / HPITCH, hx = px - hy * HPITCH;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = px < G::HROWS * HPITCH && cv && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            const unsigned off = ok ? (unsigned)((((size_t)(b * a.Hin + iy) * a.Win + ix) * a.Cin + ck * EPC) * sizeof(T)) : OOB;
            raw[i] = __builtin_amdgcn_raw_buffer_load_b128(srd0, off, 0, 0);
            if (ok) okm |= 1u << i;
        }
#pragma unroll
        for (int i = 0; i < PIT; ++i) {
            const int px = (tid >> 3) + 64 * i;
            if (px < G::HROWS * HPITCH) {
                u32x4 o = raw[i];
                if (((okm >> i) & 1u) && gn) o = gk.template apply<true>(raw[i]);
                *(u32x4*)(As + px * 128 + (((ck ^ (px >> 1)) & 7) << 4)) = o;
            }
        }
    }

    // ------------------------------------------------------------------ epilogue pieces (used by every role after its loop)
    float* const Cs = (float*)smem;
    constexpr int CP = L::CP;
    constexpr int NOCT = BN / 8, PSL = 512 / NOCT, NIT = 128 / PSL;
    constexpr int NPASS = TH / 4;
    const int o = tid % NOCT, ps = tid / NOCT;
    const int nb = n0 + o * 8;
    const bool nvalid = nb < a.Cout;
    float f1[8], f2[8], s1[8], s2[8];
    unsigned char* const outb = (unsigned char*)a.out;
    const unsigned char* const resb = (const unsigned char*)a.res;
    auto epi_init = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) { f1[e] = 1.f; f2[e] = 0.f; s1[e] = 0.f; s2[e] = 0.f; }
        if (nvalid) {
#pragma unroll
            for (int e = 0; e < 8; ++e) f2[e] = a.bias[nb + e];
            if (a.film) {
                const float* fp = a.film + (size_t)b * a.film_bstride;
#pragma unroll
                for (int e = 0; e < 8; ++e) { f1[e] = 1.0f + fp[nb + e]; f2[e] = fmaf(f2[e], f1[e], fp[a.Cout + nb + e]); }
            }
        }
    };
    auto epi_all = [&]() __attribute__((always_inline)) {
        u32x4 rres[NPASS * NIT][EPC == 8 ? 1 : 2];
        size_t eoff[NPASS * NIT];
        unsigned vmask = 0;
#pragma unroll
        for (int q = 0; q < NPASS * NIT; ++q) {
            const int pass = q / NIT, itp = q - pass * NIT;
            const int m = itp * PSL + ps;
            const int my = my0 + pass * 4 + (m >> 5), mx = mx0 + (m & 31);
            const bool v = nvalid && my < a.MH && mx < a.MW;
            const int oy = my * a.OS + py, ox = mx * a.OS + px_;
            eoff[q] = (((size_t)(b * a.Hout + oy) * a.Wout + ox) * a.Cout + nb) * sizeof(T);
            if (v) vmask |= 1u << q;
#pragma unroll
            for (int w = 0; w < (EPC == 8 ? 1 : 2); ++w) {
                rres[q][w] = u32x4{0u, 0u, 0u, 0u};
                if (v && resb) rres[q][w] = *(const u32x4*)(resb + eoff[q] + 16 * w);
            }
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NPASS * NIT; ++q) {
            if ((vmask >> q) & 1u) {
                const int pass = q / NIT, itp = q - pass * NIT;
                const int m = itp * PSL + ps;
                const float* cs = Cs + pass * (L::CS1_BYTES / 4) + m * CP + o * 8;
                float v[8];
                const f32x4 c0 = *(const f32x4*)cs, c1 = *(const f32x4*)(cs + 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) { v[e] = c0[e]; v[4 + e] = c1[e]; }
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = fmaf(v[e], f1[e], f2[e]);
                if (resb) {
                    float rv[8];
                    Vec16<T>::unpack(rres[q][0], rv);
                    if constexpr (EPC == 4) Vec16<T>::unpack(rres[q][1], rv + 4);
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += rv[e];
                }
                *(u32x4*)(outb + eoff[q]) = Vec16<T>::pack(v);
                if constexpr (EPC == 4) *(u32x4*)(outb + eoff[q] + 16) = Vec16<T>::pack(v + 4);
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += v[e]; s2[e] = fmaf(v[e], v[e], s2[e]); }
            }
        }
    };
    const bool do_epi = !CCN_DBG_BIT(a, 8);

    if (wave >= A0) {
        // ------------------------------------------------------------------ A producers (4 waves)
        // registers hold chunk c+1 (requested during chunk c-1); GroupNorm + SiLU is applied in place while the consumers
        // work on chunk c; between the two boundary barriers the chunk is dumped into the single A buffer and the same
        // registers are re-requested for chunk c+2
        const int ptid = tid - A0 * 64, ck = ptid & 7;
        constexpr int AIT = (G::AU + NA * 64 - 1) / (NA * 64);
        u32x4 areg[AIT];
        GnCoef<T> gk;
        auto a_off = [&](int i) __attribute__((always_inline)) -> unsigned {
            const int px = (ptid >> 3) + NA * 8 * i;
            const int hy = px / HPITCH, hx = px - hy * HPITCH;
            const int iy = iy0 + hy, ix = ix0 + hx;
            const bool ok = px < G::HROWS * HPITCH && iy >= 0 && iy < a.Hin && ix >= 0 && ix < a.Win;
            return ok ? (unsigned)(((b * a.Hin + iy) * a.Win + ix) * a.Cin + ck * EPC) * (unsigned)sizeof(T) : OOB;
        };
        auto a_req_all = [&](int chunk) __attribute__((always_inline)) {
            const int cb = chunk * CKE + ck * EPC;
            const bool cv = chunk < a.nchunk && cb < a.Cin;
            gk.load(a.gn_ab + (size_t)b * a.Cin + (cv ? cb : 0), gn && cv);     // first: its wait must not wait for the HBM requests below
            const auto srd = in_srd(chunk < a.nchunk ? chunk : 0);
#pragma unroll
            for (int i = 0; i < AIT; ++i) areg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, cv ? a_off(i) : OOB, 0, 0);
        };
        a_req_all(1);
        raw_barrier();                                             // chunk 0 visible
        stamp(1);
        for (int chunk = 0; chunk + 1 < a.nchunk; ++chunk) {
            if (!CCN_DBG_BIT(a, 1)) {
                const bool cv = (chunk + 1) * CKE + ck * EPC < a.Cin;
                if (gn && cv) {
#pragma unroll
                    for (int i = 0; i < AIT; ++i)
                        if (a_off(i) != OOB) areg[i] = gk.template apply<true>(areg[i]);   // padding stays zero
                }
            }
            loop_barrier();                                        // consumers are done with chunk `chunk`
            if (!CCN_DBG_BIT(a, 1)) {
#pragma unroll
                for (int i = 0; i < AIT; ++i) {
                    const int px = (ptid >> 3) + NA * 8 * i;
                    if (px < G::HROWS * HPITCH) *(u32x4*)(As + px * 128 + (((ck ^ (px >> 1)) & 7) << 4)) = areg[i];
                }
            }
            loop_barrier();                                        // chunk `chunk + 1` visible
            if (!CCN_DBG_BIT(a, 1)) a_req_all(chunk + 2);
        }
        stamp(2); stamp_wait();
        if (do_epi) {
            __syncthreads();                                       // matches the consumers' barrier: LDS is about to become the fp32 tile
            epi_init(); epi_all();
        }
        stamp(3);
    } else {
        // ------------------------------------------------------------------ consumers (4 waves)
        __builtin_amdgcn_s_setprio(2);
        const int wm = wave / WN, wn = wave % WN;
        f32x16 acc[MF][NF];
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
        auto rbase = [&](int row) __attribute__((always_inline)) { return row * 128 + ((((row >> 1) & 6)) << 4) + (((h ^ (row >> 1)) & 1) << 4); };
        int prow[MF], bbase[NF], toff[NTAPS];
#pragma unroll
        for (int i = 0; i < MF; ++i) prow[i] = ((wm * MF + i) + 1) * HPITCH + r + 1;
#pragma unroll
        for (int j = 0; j < NF; ++j) bbase[j] = rbase(j * 32 + r);             // row index inside this wave's private tile
#pragma unroll
        for (int t = 0; t < NTAPS; ++t)
            toff[t] = NTAPS == 9 ? (t / 3 - 1) * HPITCH + (t % 3 - 1)
                                 : a.tapinfo_dy(par_off + t) * HPITCH + a.tapinfo_dx(par_off + t);
        // private weight ring: slot s of this wave at Bs + (wave*NBUF + s)*BW_BYTES; a tap is PP one-KiB pieces (8 rows each)
        constexpr int PP = NF * 32 / 8;
        const int b_wave = (int)L::A_BYTES + wave * NBUF * (int)L::BW_BYTES;
        auto b_dma = [&](int tg) __attribute__((always_inline)) {             // tg: tap index counted from the start of the tile
            int chunk = tg / NTAPS, tap = tg - chunk * NTAPS;
            if (chunk >= a.nchunk) { chunk = a.nchunk - 1; }                  // past the end: harmless re-read, keeps vmcnt uniform
            const auto srd = w_srd(tap, chunk);
            [[maybe_unused]] const int slot = tg % NBUF;             // (used by the device pass only)
#pragma unroll
            for (int k = 0; k < PP; ++k) {
                const int row = 8 * k + (lane >> 3);                           // row inside the private tile
                const unsigned voff = (unsigned)((size_t)(n0 + wn * NF * 32 + row) * a.Cin_pad * sizeof(T)) + ((((lane & 7) ^ (row >> 1)) & 7) << 4);
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(srd, (__attribute__((address_space(3))) void*)(smem + b_wave + slot * L::BW_BYTES + k * 1024),
                                                         16, voff, 0, 0, 0);
#else
                (void)voff; (void)srd;
#endif
            }
        };
        b_dma(0); b_dma(1);
        raw_barrier();                                             // chunk 0 visible; the two weight taps stay in flight
        stamp(1);
        int tg = 0;                                                // taps done since the start of the tile
        for (int chunk = 0; chunk < a.nchunk; ++chunk) {
#pragma unroll
            for (int i = 0; i < MF; ++i) asm volatile("" : "+v"(prow[i]));    // keep the address math inside the loop
            {
                constexpr int NSTEP = NTAPS * 4;                   // step = tap * 4 + kk
                u32x4 av[2][MF], bv[2][NF];
                int abase[MF];
                auto frag = [&](int j, int tg0, u32x4* av_, u32x4* bv_) __attribute__((always_inline)) {
                    const int tt = j >> 2, kk = j & 3;
                    if (kk == 0) {
#pragma unroll
                        for (int i = 0; i < MF; ++i) abase[i] = rbase(prow[i] + toff[tt]);
                    }
                    const int b_off = b_wave + ((tg0 + tt) % NBUF) * (int)L::BW_BYTES;
#pragma unroll
                    for (int i = 0; i < MF; ++i) av_[i] = *(const u32x4*)(smem + (abase[i] ^ (kk << 5)));
#pragma unroll
                    for (int jn = 0; jn < NF; ++jn) bv_[jn] = *(const u32x4*)(smem + b_off + (bbase[jn] ^ (kk << 5)));
                };
                // DMA(x) is issued during tap x-2, so at the start of tap tg the ring holds tg (needed now) and tg+1 (may fly)
                asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PP) : "memory");
                frag(0, tg, av[0], bv[0]);
#pragma unroll
                for (int j = 0; j < NSTEP; ++j) {
                    const bool tap_begin = (j & 3) == 0, tap_end = (j & 3) == 3;
                    __builtin_amdgcn_sched_barrier(0);
                    // the slot of the previous tap is free (its fragment reads fed MFMAs already issued): refill it two taps ahead
                    if (tap_begin) b_dma(tg + (j >> 2) + 2);
                    if (j + 1 < NSTEP) {
                        if (tap_end) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PP) : "memory");   // next tap landed (the one after may fly)
                        frag(j + 1, tg, av[(j + 1) & 1], bv[(j + 1) & 1]);
                    }
#pragma unroll
                    for (int i = 0; i < MF; ++i)
#pragma unroll
                        for (int jn = 0; jn < NF; ++jn) mfma16<T>(acc[i][jn], av[j & 1][i], bv[j & 1][jn]);
                    // An in-order wave that issues its MFMAs back to back leaves 24 of every 32 cycles of issue bandwidth
                    // unused and then runs the next step's address math / ds_reads while the matrix pipe drains (measured:
                    // a pure MFMA stream at 76 % of the pipe rate).  Interleave: one MFMA, then the DMA pieces of this tap
                    // (first step only), one LDS read and up to two VALU of the NEXT step's fragment fetch in its shadow.
                    constexpr int DPM = (PP + MF * NF - 1) / (MF * NF);     // DMA pieces per MFMA shadow
#pragma unroll
                    for (int m = 0; m < MF * NF; ++m) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // 1 MFMA
                        if (tap_begin) __builtin_amdgcn_sched_group_barrier(0x020, DPM, 0);   // VMEM reads (LDS-DMA pieces)
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // 1 DS read
                        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // 2 VALU
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            tg += NTAPS;
            if (chunk + 1 < a.nchunk) { loop_barrier(); loop_barrier(); }     // A buffer released / refilled
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // no DMA may land after the epilogue starts reusing LDS
        __builtin_amdgcn_s_setprio(0);
        stamp(2); stamp_wait();
        if (do_epi) {
            __syncthreads();                                       // every wave's DMA drained, every wave done reading A/B
            epi_init();
#pragma unroll
            for (int i = 0; i < MF; ++i) {
                const int row = wm * MF + i;
                float* const cst = Cs + (row / 4) * (L::CS1_BYTES / 4);
#pragma unroll
                for (int j = 0; j < NF; ++j)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int m = (row & 3) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                        cst[m * CP + (wn * NF + j) * 32 + r] = acc[i][j][q];
                    }
            }
            epi_all();
        }
        stamp(3);
    }
    if (!do_epi) return;
    if (a.part) {
#pragma unroll
        for (int s = NOCT; s < 64; s <<= 1)
#pragma unroll
            for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], s); s2[e] += __shfl_xor(s2[e], s); }
        float* const red = (float*)(smem + L::CS_BYTES);
        float* const chs = red + 8 * BN * 2;
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
                part_store(a.part + (size_t)(b * a.G + g) * a.nslot + slot, t1, t2);
            }
        }
        if (a.fin_counter) gn_fused_finalize<512>(a, b, (unsigned*)red, tid);
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------
typedef void (*fr_fn_t)(const ConvArgs);

template <typename T> static fr_fn_t pick_fr_t(int ntaps, int th, int bn)
{
    if (ntaps == 9) {
        if (th == 8) { if (bn == 128) return conv_fr_kernel<T, 4, 2, 9>; return conv_fr_kernel<T, 4, 1, 9>; }
        if (bn == 128) return conv_fr_kernel<T, 2, 2, 9>;
        return conv_fr_kernel<T, 2, 1, 9>;
    }
    if (th == 8) { if (bn == 128) return conv_fr_kernel<T, 4, 2, 4>; return conv_fr_kernel<T, 4, 1, 4>; }
    if (bn == 128) return conv_fr_kernel<T, 2, 2, 4>;
    return conv_fr_kernel<T, 2, 1, 4>;
}
static fr_fn_t pick_fr(int dtype, int ntaps, int th, int bn)
{
    return dtype == 0 ? pick_fr_t<float>(ntaps, th, bn) : pick_fr_t<__bf16>(ntaps, th, bn);
}
static size_t fr_lds(int th, int bn)
{
    if (th == 8) return bn == 128 ? FrLds<8, 128, 2>::TOTAL : FrLds<8, 64, 1>::TOTAL;
    return bn == 128 ? FrLds<4, 128, 2>::TOTAL : FrLds<4, 64, 1>::TOTAL;
}

hipError_t conv_fr_prepare()
{
    for (int dt = 0; dt < 2; ++dt)
        for (int ntaps = 4; ntaps <= 9; ntaps += 5)
            for (int th = 4; th <= 8; th += 4)
                for (int bn = 64; bn <= 128; bn += 64) {
                    hipError_t e = hipFuncSetAttribute((const void*)pick_fr(dt, ntaps, th, bn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                       (int)fr_lds(th, bn));
                    if (e != hipSuccess) return e;
                }
    return hipSuccess;
}

static unsigned long long* g_stamps = nullptr;
static unsigned g_stamp_grid = 0;
extern "C" int ccn_internal_dump_stamps_fr(const char* path)
{
    if (!g_stamps || !g_stamp_grid) return 1;
    std::vector<unsigned long long> h((size_t)g_stamp_grid * 24);
    if (hipMemcpy(h.data(), g_stamps, h.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    FILE* f = fopen(path, "w");
    if (!f) return 3;
    for (unsigned b = 0; b < g_stamp_grid; ++b)
        for (int k = 0; k < 24; ++k) fprintf(f, "%llu%c", h[(size_t)b * 24 + k], k == 23 ? '\n' : ' ');
    fclose(f);
    return 0;
}

hipError_t launch_conv_fr(int dtype, int bn, const ConvArgs& a, hipStream_t s)
{
    const unsigned grid = (unsigned)(a.B * a.n_ty * a.n_tx * a.npar * a.n_nt);
    static const char* env = diag_env("CCN_STAMPS");
    if (env) {
        unsigned want = (unsigned)atoi(env), want_taps = strchr(env, ':') ? (unsigned)atoi(strchr(env, ':') + 1) : 9u;
        if (grid == want && (unsigned)a.ntaps == want_taps && grid <= 8192) {
            if (!g_stamps) { if (hipMalloc((void**)&g_stamps, (size_t)8192 * 24 * 8) != hipSuccess) return hipErrorOutOfMemory; }
            g_stamp_grid = grid;
            ConvArgs d = a; d.stamps = g_stamps;
            hipLaunchKernelGGL(pick_fr(dtype, a.ntaps, a.th, bn), dim3(grid), dim3(512), fr_lds(a.th, bn), s, d);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(pick_fr(dtype, a.ntaps, a.th, bn), dim3(grid), dim3(512), fr_lds(a.th, bn), s, a);
    return hipGetLastError();
}

}  // namespace ccn
