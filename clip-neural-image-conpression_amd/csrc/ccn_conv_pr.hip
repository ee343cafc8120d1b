// Persistent implicit-GEMM 3x3 stride-1 convolution with the weight operand in registers (bf16 throughput mode).
//
// What the free-running kernel (ccn_conv_fr.hip) still pays, measured per 8x32-pixel tile of a C=128 layer: a ~6-8 us
// serial prologue (cold fetch of the first input chunk), a ~4.7 us serial epilogue, and a main loop that runs at ~55 %
// of the MFMA rate because the LDS port is saturated: 96 B/clk of fragment reads plus 32 B/clk of weight LDS-DMA
// writes against a 128 B/clk port.  This kernel removes all three:
//   * weights never touch LDS: the host packs them in MFMA fragment order (one 1 KiB wave load = one 32-channel x
//     16-byte fragment column) and every consumer wave streams its own fragments from L2 straight into a ring of D
//     steps of registers, D-1 steps (~1300 clk) ahead of use.  LDS traffic drops to the input fragments (64 B/clk);
//   * the input chunk buffers are double buffered in the space the weight rings used: a chunk boundary is ONE barrier
//     and the producers write the next chunk while the consumers compute;
//   * a workgroup is persistent: it walks tiles vb, vb+grid, ...  The producer waves run one chunk ahead ACROSS tile
//     boundaries and the weight ring keeps streaming, so a tile has no prologue;
//   * the consumers have no epilogue either: at the end of a tile they round the accumulators to bf16 into a staging
//     tile in LDS (32 ds_write_b64 per wave; the MFMA operands are swapped so a lane holds 4 consecutive channels of
//     one pixel) and go straight on to the next tile.  The PRODUCER waves -- idle most of a chunk -- finish the
//     previous tile during the next tile's first chunk: staging -> bias / FiLM / residual (rows prefetched into their
//     registers one chunk earlier) -> 16-byte NHWC stores, and the next GroupNorm's partial sums (the four
//     producer waves' per-channel sums are combined in LDS after the next barrier: ONE slot per tile).
// Round 2 (what the stamps of a one-tile-per-CU launch showed: 5 us until the first chunk is visible, 4 us of tail, and a
// 5 us finalize launch + kernel boundary in front of every conv):
//   * the input GroupNorm's finalize is done here: every producer wave reduces the producer kernel's partial sums of the
//     sample it is about to stage (<= 64 slots per group, one round of loads issued next to the first input loads);
//   * the consumers, idle until the first chunk is in LDS, stage the lower half of its halo rows themselves; the LAST tile's
//     bias / FiLM / residual operands are fetched before the final barrier of the chunk loop;
//   * TH = 4: the same kernel on 4-row tiles for layers with fewer than #CUs 8-row tiles.
// Built, measured in product builds and REMOVED (DESIGN.md / docs/EXPERIMENTS.md keep the numbers and the last commit that carries each):
// the last tile's epilogue split between the two roles (+-1 % by box), write-through output stores (-1..-2 %), blocked tile order with
// staged input kept in LDS between neighbouring tiles (-3..-4 %), a deeper weight ring, runtime split-K / GroupNorm-source switches.
// Round 3: the 3x3 consumers moved from v_mfma_f32_32x32x16_bf16 to v_mfma_f32_16x16x32_bf16 (same tile, same bytes, higher clock).
// Tile: TH (8 or 4) rows x 32 pixels x 128 output channels, 4 consumer waves (3x3: all rows x 32 channels each; other tap sets:
// 2x2 waves of 4x2 fragments of 32x32) + 4 producer waves.
// Rounding: the conv accumulator is rounded to bf16 once before the affine/residual and the sum once more on store
// (the other kernels round once); both are within the bf16 mode's error budget (tests/test_gpu_parity.py bounds).
#include "ccn_device.h"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

namespace ccn {

namespace {

// x / d for a divisor known at launch: d's magic number m = ceil(2^32 / d) comes in the launch arguments (0 for d == 1), the quotient
// is one v_mul_hi_u32 instead of the ~40-instruction float-reciprocal sequence of a runtime integer division -- a tile decode has five
// of them and the producers, the pole of the 128-channel layers, decode two tiles' worth per tile.  Exact while x * d < 2^32.
#define CCN_FDIV(x, m, d) ((m) ? (int)__umulhi((unsigned)(x), (m)) : (int)(x))

template <int TH_> struct PrLdsT {
    static constexpr int HROWS = TH_ + 2, HPITCH = 34;
    static constexpr int A_BYTES = HROWS * HPITCH * 128;       // one Cin chunk of the halo tile (43520 at 8 rows)
    static constexpr int SP = 272;                              // staging pitch in bytes: 128 bf16 channels + 16
    static constexpr int STG_BYTES = TH_ * 32 * SP;             // TH x 32 pixels
    static constexpr int CHS_BYTES = 4 * 128 * 2 * 4;           // per producer wave: per-channel (sum, sum of squares)
    static constexpr int TOTAL = 2 * A_BYTES + STG_BYTES + CHS_BYTES;
};
typedef PrLdsT<8> PrLds;
static_assert(PrLds::TOTAL <= 160 * 1024, "LDS budget");

}  // namespace

// MODE 1 (RES): the epilogue adds the residual tensor and ignores FiLM (ResBlock conv2, ConvTranspose + skip); MODE 0: bias /
// FiLM only (conv1, stride-2); MODE 2: MODE 0 for split-K launches (its second half adds the first half's partial tile, so it
// needs residual registers too).  Separate instantiations keep the producers' register footprint down.
//
// Split-K (a.ksplit == 2; layers with at most #CUs/2 tiles -- the 32-pixel level at C2): virtual tile v = 2*tile + kh computes
// the Cin chunks [kh*nchunk/2, (kh+1)*nchunk/2).  The kh = 0 half finishes its epilogue into the bf16 partial tensor `kpart`
// (bias / FiLM / residual already applied) and raises one flag per producer wave (release, agent scope); the kh = 1 half
// waits for that flag (bounded spin, acquire), adds the partial like a residual and writes the output and the GroupNorm
// statistics.  Partner workgroups are neighbours in the virtual tile order (same XCD) and never wait on each other in the
// other direction, so the launch cannot deadlock as long as its <= #CUs workgroups are co-resident.
//
// TH = 4 (3x3 stride 1 only): the same kernel on tiles of 4 rows for layers with fewer than #CUs tiles of 8 rows (the 32-pixel
// level at C2: 128 -> 256 tiles) -- every CU gets a tile without splitting K (no hand-off between workgroups), the producers
// stage 6 halo rows instead of 10, a consumer wave owns 4 rows x 32 channels (12 MFMAs per (dx, k-slice) group per 3 weight
// fragments: twice the weight stream per MFMA of the 8-row form, still from L2).
// GS: how the input's GroupNorm reaches the kernel -- 0 the finalized scale / shift table (a.gn_ab), 1 formed here from the producer
// kernel's partial sums (a.gs_part), 2 none (the input is already activated, or the form has no GroupNorm: ConvTranspose, stride 2).
// P4 (NTAPS == 4 only): the 4x4 stride-2 pad-1 convolution (the data gradient of the ConvTranspose, training) as FOUR plane passes per
// channel chunk with 2x2 taps each -- the ConvTranspose form's consumers on the stride-2 form's staging; see the pass table below.
template <int NTAPS, int D, int MODE, int TH = 8, int GS = 0, bool P4 = false>
__global__ __launch_bounds__(512) void conv_pr_kernel(const ConvArgs a, const int grid_tiles)
{
    static_assert(TH == 8 || (TH == 4 && NTAPS == 9), "4-row tiles: 3x3 stride-1 form only");
    static_assert(!P4 || (NTAPS == 4 && MODE != 2), "plane passes of 2x2 taps: the 4-tap consumers, no split-K");
    constexpr bool RES = MODE == 1;                            // residual registers: MODE 1 always, MODE 2 in the second K half
    constexpr bool RR = MODE != 0;
    typedef __bf16 T;
    constexpr int MF = 4, NF = 2;
    constexpr int BN = 128;
    constexpr int HROWS = TH + 2;
    // output stride / parities of the form (launch_conv_pr checks a.OS / a.npar against them): compile-time, so the tile decodes and
    // the epilogue's address arithmetic fold for the 3x3 forms
    constexpr int OSC = (NTAPS == 4 && !P4) ? 2 : 1, NPARC = (NTAPS == 4 && !P4) ? 4 : 1;
    constexpr int EPC = 8, CKE = 64;
    constexpr int NSTEP = NTAPS * 4;                           // step = tap * 4 + kk (one 16-byte K slice per lane half)
    // NTAPS == 2: the stride-2 3x3 conv.  Its 9 taps fall on the four parity planes P[py][px](i, j) = in(2i+py, 2j+px) of the
    // input (out(y,x) needs rows 2y-1, 2y, 2y+1 = plane 1 at i = y-1, plane 0 at y, plane 1 at y), so it is run as a stride-1
    // problem whose "Cin chunks" are (channel chunk, plane pass) pairs with two taps each: the producers stage a 10 x 34
    // halo tile of ONE plane per pass (input stride 2), the consumers see effective chunks of 8 steps.  Five passes per
    // channel chunk (plane (1,1) twice, the tenth tap slot has zero weights):
    //   pass 0: plane (1,1), taps (dy,dx) (0,0) (0,2) at offsets (-1,-1) (-1,0)      pass 1: plane (1,1), (2,0) (2,2) at (0,-1) (0,0)
    //   pass 2: plane (1,0), (0,1) (2,1) at (-1,0) (0,0)      pass 3: plane (0,1), (1,0) (1,2) at (0,-1) (0,0)      pass 4: plane (0,0), (1,1) at (0,0)
    // P4: out(y,x) = sum over ky, kx = 0..3 of w[ky][kx] in(2y + ky - 1, 2x + kx - 1).  Row 2y + ky - 1 lies on plane py = 1 for
    // ky = 0 (i = y - 1) and ky = 2 (i = y), on plane py = 0 for ky = 1 (i = y) and ky = 3 (i = y + 1); columns alike: every plane
    // carries 2x2 taps at offsets (i - py, j - px), i, j = 0..1.  Pass p of a channel chunk stages plane (py, px) = (p < 2, !(p & 1)).
    constexpr bool S2 = NTAPS == 2;
    constexpr int NPASS = S2 ? 5 : (P4 ? 4 : 1);               // plane passes per channel chunk
    constexpr int IS = (S2 || P4) ? 2 : 1;                     // input stride of the staged plane
    static_assert(NSTEP % D == 0, "the register ring must wrap at the chunk boundary");
    using L = PrLdsT<TH>;
    constexpr int HPITCH = L::HPITCH;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* const stg = smem + 2 * L::A_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;

    // XCD-aware block -> tile-stream mapping: consecutive hardware block ids go round-robin over the 8 XCDs, so give
    // XCD x the contiguous virtual ids [x*grid/8, (x+1)*grid/8): neighbouring tiles (shared halos, the two N tiles of one
    // pixel tile) then meet in the same L2.
    const int grid = (int)gridDim.x;
    const int vb = (grid & 7) == 0 ? ((int)blockIdx.x & 7) * (grid >> 3) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
    const int ntiles = grid_tiles;                             // virtual tiles (x2 under split-K)
    // Tile order of a workgroup: strided (vb, vb + grid, ...): at any moment the grid works on one contiguous band of tiles
    const int my_tiles = (ntiles - vb + grid - 1) / grid;          // >= 1 (host: grid <= ntiles)
    auto vt = [&](int ti) __attribute__((always_inline)) { return vb + ti * grid; };
    // split-K launches run the MODE 2 instantiation only (launch_conv_pr): everywhere else ks folds to 1 at compile time and the
    // hand-off branches of the epilogue disappear
    const int ks = (MODE == 2 && a.ksplit == 2) ? 2 : 1;
    // XCD this workgroup runs on: HW_REG_XCC_ID (id 20), bits [3:0] -- only the split-K hand-off uses it
    const unsigned xcc_id = MODE == 2 ? (unsigned)__builtin_amdgcn_s_getreg((3 << 11) | 20) : 0u;
    const int nck = a.nchunk / ks;                             // chunks per virtual tile
    const int ktotal = my_tiles * nck;
    auto vt_tile = [&](int v) __attribute__((always_inline)) { return ks == 2 ? (v >> 1) : v; };
    auto vt_kh = [&](int v) __attribute__((always_inline)) { return ks == 2 ? (v & 1) : 0; };

    const unsigned char* const inb = (const unsigned char*)a.in;
    // GroupNorm of the input: scale/shift per (sample, channel) either from the table a separate finalize launch wrote (gn_ab),
    // or -- gs_part != null -- formed HERE from the producing kernel's partial sums (no finalize launch between two convs)
    constexpr bool gstat = GS == 1;                                           // (launch_conv_pr: GS == 1 exactly when a.gs_part is set)
    static_assert(GS != 1 || (NTAPS == 9 && MODE != 2), "in-kernel statistics: 3x3 stride-1 layers without split-K");
    // (the ConvTranspose / stride-2 forms never have a GroupNorm on their input -- launch_conv_pr refuses it -- so the transform and
    // its coefficient traffic compile out of those instantiations)
    // GS: 0 = scale / shift table (a.gn_ab), 1 = formed here from partial sums, 2 = no GroupNorm on the input (pre-activated input)
    const bool gn = NTAPS == 9 && GS != 2 && !CCN_DBG_BIT(a, 32);
    constexpr unsigned OOB = 0x7FFFFFF0u;
    auto raw_barrier = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto stamp = [&](int slot) __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a) && lane == 0 && (wave == 0 || wave == 4))
            CCN_STAMPS_PTR(a)[((size_t)blockIdx.x * 2 + (wave == 0 ? 0 : 1)) * 8 + slot] = __builtin_amdgcn_s_memrealtime();
    };
    // diagnostics (CCN_STAMPS): shader-clock cycles spent waiting at barriers / in phases, per role
    [[maybe_unused]] unsigned long long t_bar = 0, t_a = 0, t_b = 0, t_w = 0, t_r = 0;
    const unsigned long long t_begin = CCN_STAMPS_PTR(a) ? __builtin_amdgcn_s_memtime() : 0;
    auto timed_barrier = [&]() __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a)) { const unsigned long long t0 = __builtin_amdgcn_s_memtime(); raw_barrier(); t_bar += __builtin_amdgcn_s_memtime() - t0; }
        else raw_barrier();
    };
    auto stamp_cycles = [&]() __attribute__((always_inline)) {
        if (CCN_STAMPS_PTR(a) && lane == 0 && (wave == 0 || wave == 4)) {
            unsigned long long* st = CCN_STAMPS_PTR(a) + ((size_t)blockIdx.x * 2 + (wave == 0 ? 0 : 1)) * 8;
            st[3] = t_bar; st[4] = __builtin_amdgcn_s_memtime() - t_begin; st[5] = t_a; st[6] = t_b; st[7] = t_r;
        }
    };
    stamp(0);

    // ---- epilogue machinery (run by the producer waves).  thread -> fixed channel octet o16 of the tile's 128, pixels pr + 16*it
    // (it < 2 TH) of its 32 TH: row it>>1, column pr + 16*(it&1); residual rows, bias and FiLM are fetched during the tile's last chunk
    const int etid = tid & 255, ew = wave & 3;
    const int o16 = etid & 15, pr = etid >> 4;
    unsigned char* const outb = (unsigned char*)a.out;
    const unsigned char* const resb = (const unsigned char*)a.res;
    const unsigned out_bytes = (unsigned)((size_t)a.B * a.Hout * a.Wout * a.Cout * sizeof(T));
    // per-wave channel sums of an epilogue, behind the staging tile
    float* const chs = (float*)(stg + L::STG_BYTES) + ew * 256;
    constexpr int NQ = TH / 2;               // epilogue batches of 4 items (two tile rows each)
    u32x4 rr[RR ? 2 : 1][4];                 // residual rows, batches of 4 items (two tile rows), two batches in flight
    f32x4 fb[2], fs[RES ? 1 : 2], ft[RES ? 1 : 2];   // raw bias / FiLM scale / FiLM shift of this thread's octet
    int e_b = 0, e_ty = 0, e_tx = 0, e_nt = 0, e_par = 0, e_kh = 0, e_tile = 0;
    bool pd_on = false;                      // a tile's per-wave GroupNorm sums wait in LDS for combine()
    int pd_b = 0, pd_slot = 0, pd_n0 = 0;
    unsigned char* const kpartb = (unsigned char*)a.kpart;
    unsigned e_base = 0;
    int e_rows = 0;                          // wave-uniform: valid rows of the tile
    unsigned e_m0 = OOB, e_m1 = OOB;         // 0 when this thread's first / second column (and its octet) is inside the tensor
    // byte offset of item it (row it>>1, column pr + 16*(it&1)); masked-out items land past num_records.  Callers pass a
    // laundered copy of e_base: otherwise the 16 offsets are computed once per tile and kept live across dump() and the barrier
    auto item_off = [&](unsigned eb, int it) __attribute__((always_inline)) -> unsigned {
        const unsigned rmask = (it >> 1) < e_rows ? 0u : OOB;                 // wave-uniform
        return (eb + (unsigned)(it >> 1) * (unsigned)(OSC * a.Wout * a.Cout * (int)sizeof(T)) + (unsigned)((it & 1) * 16) * (unsigned)(OSC * a.Cout * (int)sizeof(T)))
               | ((it & 1) ? e_m1 : e_m0) | rmask;
    };
    // during the tile's last chunk: decode the tile
    auto epi_setup = [&](int v) __attribute__((always_inline)) {
        const int tile = vt_tile(v);
        e_kh = vt_kh(v); e_tile = tile;
        const int t2 = CCN_FDIV(tile, a.fd_nt, a.n_nt);
        e_nt = tile - t2 * a.n_nt;
        e_par = t2 % NPARC;
        const int sp = t2 / NPARC;
        {
            const int row = CCN_FDIV(sp, a.fd_tx, a.n_tx);                    // (b, ty) row of tiles
            e_tx = sp - row * a.n_tx;
            e_b = CCN_FDIV(sp, a.fd_sp, a.n_tx * a.n_ty);
            e_ty = row - e_b * a.n_ty;
        }
        const int nb = e_nt * BN + o16 * 8;
        const bool nvalid = nb < a.Cout;
        // output pixel of M-space pixel (my, mx): (my*OS + py, mx*OS + px) -- OS = 2 and 4 parities for the ConvTranspose
        e_base = (unsigned)(((e_b * a.Hout + e_ty * TH * OSC + (e_par >> 1)) * a.Wout + (e_tx * 32 + pr) * OSC + (e_par & 1)) * a.Cout + nb) * (unsigned)sizeof(T);
        e_rows = a.MH - e_ty * TH;
        e_m0 = (nvalid && e_tx * 32 + pr < a.MW) ? 0u : OOB;
        e_m1 = (nvalid && e_tx * 32 + pr + 16 < a.MW) ? 0u : OOB;
    };
    // residual source of the current epilogue: the residual tensor, or (second K half) the first half's partial tile
    auto res_on = [&]() __attribute__((always_inline)) -> bool { return RES || (MODE == 2 && ks == 2 && e_kh == 1); };
    auto res_batch = [&](int q, u32x4* dst) __attribute__((always_inline)) {
        if constexpr (RR) {
            if (res_on()) {
                const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)((ks == 2 && e_kh == 1) ? kpartb : (resb ? resb : outb)), 0, out_bytes, 0x00020000);
                unsigned eb = e_base; asm volatile("" : "+v"(eb));
                if (ks == 2 && e_kh == 1) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, item_off(eb, q * 4 + i), 0, 1);   // sc0: bypass L1
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!CCN_DBG_BIT(a, 16384)) dst[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, item_off(eb, q * 4 + i), 0, 2);   // nt: read once (+1.2 %)
                        else dst[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, item_off(eb, q * 4 + i), 0, 0);
                    }
                }
            }
        }
    };
    // start of the epilogue iteration: bias / FiLM (raw: folding them here would wait for the loads) and the first two
    // residual batches; all of it flies during request()
    auto epi_request = [&](int q0, int q1) __attribute__((always_inline)) {
        if (ks == 2 && e_kh == 1) {
            // second K half: the partner's partial tile must be complete.  Thread t reads exactly what thread t of the partner
            // wrote, so one flag per producer wave is enough.  Bounded spin: a protocol bug must not hang the GPU.
            unsigned* const fl = a.kflag + (size_t)e_tile * 4 + ew;
            unsigned seen = 0u;
            if (lane == 0) {
                int spins = 0;
                while ((seen = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(2);
                // the partner never arrived (it cannot happen while both halves are co-resident, which launch_conv_pr
                // guarantees): the tile below is then wrong -- say so in the handle's error word, which the next API call
                // (or ccn_poll_errors) turns into CCN_EHIP, instead of falling through silently
                if (spins >= (1 << 22) && a.err) __hip_atomic_fetch_or(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(fl, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
            }
            // No agent-scope acquire fence on the expected path (it would invalidate the XCD's whole L2): the partner workgroup runs
            // on the same XCD -- partners are virtual tiles 2t, 2t+1 and launch_conv_pr makes the split-K grid a multiple of 16, so
            // that every XCD's contiguous range [x*grid/8, (x+1)*grid/8) starts at an even id and holds whole pairs -- its
            // stores are write-through to that L2, and this CU cannot hold stale lines of the partial tensor (L1 is
            // invalidated at kernel start and the lines are read for the first time now, with the L1-bypass bit set).
            // That placement is an ASSUMPTION about the dispatcher (round-robin over the XCDs), so it is checked: the flag carries
            // the partner's XCC_ID + 1; on a mismatch (a CU mask, another partition mode, a future driver) this wave takes the
            // agent-scope acquire after all and reports bit 2 in the handle's error word (tests assert it is never set on the box).
            seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
            if (seen != 0u && seen != xcc_id + 1u) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                if (lane == 0 && a.err) __hip_atomic_fetch_or(a.err, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("" ::: "memory");
        }
        const int nb = e_nt * BN + o16 * 8;
        const int nbs = nb < a.Cout ? nb : 0;
        fb[0] = *(const f32x4*)(a.bias + nbs); fb[1] = *(const f32x4*)(a.bias + nbs + 4);
        if constexpr (!RES) {
            if (a.film) {
                const float* fp = a.film + (size_t)e_b * a.film_bstride;
                fs[0] = *(const f32x4*)(fp + nbs); fs[1] = *(const f32x4*)(fp + nbs + 4);
                ft[0] = *(const f32x4*)(fp + a.Cout + nbs); ft[1] = *(const f32x4*)(fp + a.Cout + nbs + 4);
            }
        }
        res_batch(q0, rr[q0 & 1]);
        if constexpr (RR) { if (q0 + 1 < q1) res_batch(q0 + 1, rr[(q0 + 1) & 1]); }
    };
    // the staging tile of the tile described by e_* is complete (the consumers wrote it before the last barrier)
    // items of batches [q0, q1) (call sites pass constants: the loop below unrolls and folds)
    auto epilogue = [&](int q0, int q1) __attribute__((always_inline)) {
        // scalar fp32 math only (packed-fp32 ops starve next to the consumers' MFMA stream, see GnCoef), and the running
        // sums pinned per item: left alone the compiler sums ACROSS the 16 unrolled items at the end and keeps all 128
        // output values alive until then
        const bool first = ks == 2 && e_kh == 0, second = ks == 2 && e_kh == 1;
        // the first K half carries bias / FiLM shift / residual; the second only scales by the FiLM factor and adds the partial
        float f1[RES ? 1 : 8], f2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            f2[e] = second ? 0.f : fb[e >> 2][e & 3];
            if constexpr (!RES) {
                f1[e] = 1.f;
                if (a.film) { f1[e] = 1.0f + fs[e >> 2][e & 3]; f2[e] = second ? 0.f : fmaf(f2[e], f1[e], ft[e >> 2][e & 3]); }
            }
        }
        const auto osrd_c = __builtin_amdgcn_make_buffer_rsrc((void*)(first ? kpartb : outb), 0, out_bytes, 0x00020000);
        const bool radd = res_on();
        float s1[8], s2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < q0 || q >= q1) continue;
            unsigned eb = e_base; asm volatile("" : "+v"(eb));
            int sbase = pr * L::SP + o16 * 16; asm volatile("" : "+v"(sbase));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int it = q * 4 + i;
                const unsigned off = item_off(eb, it);
                const u32x4 sv = *(const u32x4*)(stg + sbase + ((it >> 1) * 32 + (it & 1) * 16) * L::SP);
                const bool live = off < OOB;                          // masked items contribute nothing to the statistics
                float x[8];
                Vec16<T>::unpack(sv, x);
                if constexpr (RES) {
                    float rv[8];
                    Vec16<T>::unpack(rr[q & 1][i], rv);
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = (x[e] + f2[e]) + rv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x[e] = fmaf(x[e], f1[e], f2[e]);
                    if constexpr (MODE == 2) {
                        if (radd) {
                            float rv[8];
                            Vec16<T>::unpack(rr[q & 1][i], rv);
#pragma unroll
                            for (int e = 0; e < 8; ++e) x[e] += rv[e];
                        }
                    }
                }
                // Plain (write-back) output stores: write-through (sc1) stores are 1-2 % slower in product builds (the next conv then
                // reads its input from beyond L2)
                __builtin_amdgcn_raw_buffer_store_b128(Vec16<T>::pack(x), osrd_c, off, 0, 0);
                if (live) {                                           // exec-masked: costs scalar ops, not 8 VALU multiplies
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s1[e] += x[e]; s2[e] = fmaf(x[e], x[e], s2[e]); }
                }
                asm volatile("" : "+v"(s1[0]), "+v"(s1[1]), "+v"(s1[2]), "+v"(s1[3]), "+v"(s1[4]), "+v"(s1[5]), "+v"(s1[6]), "+v"(s1[7]));
                asm volatile("" : "+v"(s2[0]), "+v"(s2[1]), "+v"(s2[2]), "+v"(s2[3]), "+v"(s2[4]), "+v"(s2[5]), "+v"(s2[6]), "+v"(s2[7]));
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (RR) { if (q + 2 < q1) res_batch(q + 2, rr[q & 1]); }   // refill the buffer just consumed, one batch ahead
        }
        if (first) {
            // publish the partial tile: every store of this wave visible at agent scope, then its flag
            // (stores are write-through to the XCD's L2; waiting for their acknowledgement is the release -- an agent-scope
            // release fence would write back the whole L2)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_store(a.kflag + (size_t)e_tile * 4 + ew, xcc_id + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (a.part && !first) {
#pragma unroll
            for (int s = 16; s < 64; s <<= 1)
#pragma unroll
                for (int e = 0; e < 8; ++e) { s1[e] += __shfl_xor(s1[e], s); s2[e] += __shfl_xor(s2[e], s); }
            if (lane < 16) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { chs[(lane * 8 + e) * 2] = s1[e]; chs[(lane * 8 + e) * 2 + 1] = s2[e]; }
            }
            // the four waves' per-channel sums are combined into ONE slot per tile after the next workgroup barrier (combine())
            pd_on = true; pd_b = e_b; pd_n0 = e_nt * BN;
            pd_slot = ((e_ty * a.n_tx + e_tx) * NPARC + e_par) * a.n_nt + e_nt;
        }
    };
    // After the barrier that follows an epilogue: group sums over the four producer waves' per-channel sums, one slot per
    // (tile, group) -- a quarter of the slots a per-wave publication needs, which is what lets the consuming conv reduce
    // them itself in one round of loads (stats()).  Wave ew takes every fourth group of the tile's channel range;
    // fixed summation order (lane tree), so the statistics stay run-to-run deterministic.
    auto combine = [&]() __attribute__((always_inline)) {
        if (!pd_on) return;
        pd_on = false;
        const float* const all = (const float*)(stg + L::STG_BYTES);
        const int n0 = pd_n0;
        if (n0 >= a.Cout) return;
        const int nend = min(n0 + BN, a.Cout);
        const int g1 = CCN_FDIV(nend - 1, a.fd_cpg, a.cpg);
        for (int g = CCN_FDIV(n0, a.fd_cpg, a.cpg) + ew; g <= g1; g += 4) {
            const int clo = max(g * a.cpg, n0), chi = min((g + 1) * a.cpg, nend);
            float t1 = 0.f, t2 = 0.f;
            const float* const wa = all + (lane >> 4) * 256;
            for (int c = clo + (lane & 15); c < chi; c += 16) {
                t1 += wa[(c - n0) * 2]; t2 += wa[(c - n0) * 2 + 1];
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) { t1 += __shfl_xor(t1, m); t2 += __shfl_xor(t2, m); }
            if (lane == 0) part_store(a.part + (size_t)(pd_b * a.G + g) * a.nslot + pd_slot, t1, t2);
        }
    };

    // ---- input staging machinery (role-neutral like the epilogue's: the producers run it for every chunk; the consumers,
    // idle until the first chunk is in LDS, stage the lower half of the halo rows of that first chunk themselves)
    const int ptid = tid & 255;
    const int ck = ptid & 7, pcol = ptid >> 3;                // 16-byte channel slice, halo column 0..31
    // the two halo columns 32, 33 (HROWS rows x 8 slices = 160 units at 8 rows) go to threads 0..HROWS*16-1 as one more item
    const int xrow = pcol >> 1, xcol = 32 + (pcol & 1);
    const bool xthr = ptid < HROWS * 16;
    constexpr int AIT = HROWS + 1;
    u32x4 areg[AIT];
    GnCoef<T> gk;
    unsigned rowm = 0;                       // wave-uniform: bit i = halo row i inside the image (for the chunk in areg)
    bool colv = false, xv = false;           // this thread's column / extra item inside the image
    int rq_ti = 0, rq_c = 0;
    // A request is split in two: prep() at the START of an iteration decodes the tile and fetches the GroupNorm
    // coefficients of the chunk into a second register set; issue() after dump() sends the 11 input loads and adopts the
    // coefficients.  (Fetched inside issue(), the coefficient loads made the wave wait a full L2 latency right there.)
    GnCoef<T> gkn;
    int q_b = 0, q_iy0 = 0, q_ix0 = 0, q_c = 0;
    bool q_tv = false;
    // in-kernel finalize (gstat): every producer wave reduces the partial sums of the sample it is about to stage by itself
    // (no cross-wave step): lane l sums slots (l & 7), (l & 7) + 8, ... of group l >> 3 -- 8 groups x 8 lanes -- in fp64 and
    // three xor-shuffles leave (mean, 1/sqrt(var + eps)) of group l >> 3 in every lane; a thread's 8 channels lie in one
    // group (cpg % 8 == 0, checked on the host) whose statistics it fetches with a lane permute; gamma / beta of those
    // channels are loaded raw in prep() and folded into (scale, shift) in issue(), one dump later, so nothing waits for them.
    float st_mean = 0.f, st_rstd = 1.f, qn_mean = 0.f, qn_rstd = 1.f;
    int st_b = -1;
    f32x4 qn_g[2], qn_bt[2];
    unsigned svmask = 0u;                    // valid slots of sv[]
    float2 sv[8];                            // one round of partial-sum loads (stats_issue -> stats_reduce)
    auto stats_issue = [&](int b) __attribute__((always_inline)) {
        // branch-free: the eight loads leave back to back (a per-slot `if` made the compiler wait for each load in its own block,
        // eight serial round trips on the cold-start path); slots past the group's count read slot 0 and are masked in stats_reduce
        const int g = lane >> 3, sub = lane & 7;
        const int cpg = a.gs_cpg, pnt = a.gs_nnt, nsp = a.gs_nsp;
        const int jlo = CCN_FDIV(g * cpg, a.fd_gsbn, a.gs_bn), jhi = CCN_FDIV((g + 1) * cpg - 1, a.fd_gsbn, a.gs_bn), nj = jhi - jlo + 1;
        const float2* const base = a.gs_part + (size_t)(b * 8 + g) * ((size_t)nsp * pnt) + jlo;   // nsp * nj <= 64 (host: in_kernel_stats)
        int off[8];
        svmask = 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int e = sub + 8 * u;                                 // slot e = jj * nsp + sp
            const int jj = CCN_FDIV(e, a.fd_gsnsp, nsp), sp = e - jj * nsp;
            const bool ok = jj < nj;
            off[u] = ok ? sp * pnt + jj : 0;
            svmask |= ok ? (1u << u) : 0u;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) sv[u] = base[off[u]];
    };
    auto stats_reduce = [&]() __attribute__((always_inline)) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const bool ok = (svmask >> u) & 1u; s1 += ok ? (double)sv[u].x : 0.0; s2 += ok ? (double)sv[u].y : 0.0; }
#pragma unroll
        for (int m = 1; m < 8; m <<= 1) { s1 += __shfl_xor(s1, m); s2 += __shfl_xor(s2, m); }
        const double mean = s1 * a.gs_inv_count;
        double var = s2 * a.gs_inv_count - mean * mean;
        if (var < 0.0) var = 0.0;
        st_mean = (float)mean;
        st_rstd = __builtin_amdgcn_rsqf((float)(var + 1e-5));
    };
    // prep() = decode (tile / chunk of the next request) + coefficients; split so that the FIRST request of the kernel can put
    // its input loads in front of the statistics' round trip (request_first below)
    int q_cbs = 0; bool q_cv = false, q_cv_i = false;
    auto decode = [&]() __attribute__((always_inline)) -> bool {
        const int v = vt(rq_ti);
        if (rq_c == 0) {                                       // new tile: decode it once, not once per chunk (the divisions are ~100 SALU ops)
            q_tv = rq_ti < my_tiles;
            const int tile = vt_tile(q_tv ? v : vt(0));
            const int sp = CCN_FDIV(tile, a.fd_ntp, a.n_nt * NPARC);   // tile = ((spatial tile) * npar + parity) * n_nt + N tile
            const int row = CCN_FDIV(sp, a.fd_tx, a.n_tx);
            const int tx = sp - row * a.n_tx;
            q_b = CCN_FDIV(sp, a.fd_sp, a.n_tx * a.n_ty);
            const int ty = row - q_b * a.n_ty;
            q_iy0 = ty * TH - 1; q_ix0 = tx * 32 - 1;
        }
        q_c = vt_kh(v) * nck + rq_c;
        const int cb = (q_c / NPASS) * CKE + ck * EPC;
        q_cv = q_tv && cb < a.Cin;
        q_cbs = q_cv ? cb : 0;
        const bool fresh = gstat && rq_c == 0 && q_tv && q_b != st_b;  // wave-uniform: statistics of another sample needed
        if (++rq_c == nck) { rq_c = 0; ++rq_ti; }
        return fresh;
    };
    auto coef_loads = [&]() __attribute__((always_inline)) {
        if (gstat) {
            qn_g[0] = *(const f32x4*)(a.gs_gamma + q_cbs); qn_g[1] = *(const f32x4*)(a.gs_gamma + q_cbs + 4);
            qn_bt[0] = *(const f32x4*)(a.gs_beta + q_cbs); qn_bt[1] = *(const f32x4*)(a.gs_beta + q_cbs + 4);
        } else gkn.load(a.gn_ab + (size_t)q_b * a.Cin + q_cbs, gn && q_cv);
    };
    auto coef_sel = [&]() __attribute__((always_inline)) {
        if (gstat) {
            const int src = CCN_FDIV(q_cbs, a.fd_gscpg, a.gs_cpg) * 8; // a lane that holds the statistics of the channels' group
            qn_mean = __shfl(st_mean, src); qn_rstd = __shfl(st_rstd, src);
        }
    };
    auto prep = [&]() __attribute__((always_inline)) {
        const bool fresh = decode();
        coef_loads();
        if (fresh) { stats_issue(q_b); stats_reduce(); st_b = q_b; }
        coef_sel();
    };
    auto adopt = [&]() __attribute__((always_inline)) {
        if (gstat) gk.from_raw(qn_g, qn_bt, qn_mean, qn_rstd, gn && q_cv_i);
        else gk = gkn;
    };
    // rows [r0, r1) of the halo tile (+ the columns-32/33 item when with_x); call sites pass constants
    auto issue_loads = [&](int r0 = 0, int r1 = TH + 2, bool with_x = true) __attribute__((always_inline)) {
        // (readfirstlane: these are wave-uniform by construction; saying so keeps the descriptor in SGPRs)
        const int b = __builtin_amdgcn_readfirstlane(q_b), iy0 = __builtin_amdgcn_readfirstlane(q_iy0), ix0 = __builtin_amdgcn_readfirstlane(q_ix0);
        const int cc = __builtin_amdgcn_readfirstlane(q_c / NPASS);           // channel chunk
        const int pass = q_c - cc * NPASS;
        const int py = S2 ? (pass <= 2 ? 1 : 0) : (P4 ? (pass < 2 ? 1 : 0) : 0);                          // plane of this pass
        const int px = S2 ? ((pass <= 1 || pass == 3) ? 1 : 0) : (P4 ? ((pass & 1) ? 0 : 1) : 0);
        const int cb = cc * CKE + ck * EPC;
        const bool cv = q_tv && cb < a.Cin;
        q_cv_i = cv;
        // The descriptor is WAVE-UNIFORM (a lane-dependent base makes the compiler wrap every load in a readfirstlane waterfall
        // loop) and covers exactly sample b from this chunk's channels on: halo rows above the image give negative = huge
        // unsigned offsets, rows below it run past num_records, so the hardware range check zero-fills both with no
        // per-row compare; only the column validity (and the channel tail) is per lane, ORed in as an out-of-range constant.
        const unsigned img_bytes = (unsigned)(a.Hin * a.Win * a.Cin) * (unsigned)sizeof(T);
        const unsigned coff = (unsigned)(((q_tv && cc * CKE < a.Cin) ? cc : 0) * CKE) * (unsigned)sizeof(T);
        const auto srd = __builtin_amdgcn_make_buffer_rsrc((void*)(inb + (size_t)b * img_bytes + coff), 0, img_bytes - coff, 0x00020000);
        const int rs = IS * a.Win * a.Cin * (int)sizeof(T);                   // row stride of the staged plane in bytes
        const int ixr = IS * (ix0 + pcol) + px;                               // input column of this thread's halo column
        colv = cv && ixr >= 0 && ixr < a.Win;
        const unsigned cmask = colv ? 0u : OOB;
        const int base = ((IS * iy0 + py) * a.Win + ixr) * a.Cin * (int)sizeof(T) + ck * 16;
        // halo rows inside the image: [r_lo, r_hi) (wave-uniform), as a bit mask for the padding-stays-zero select in dump()
        const int first = IS * iy0 + py;                                      // input row of halo row 0
        const int r_lo = first < 0 ? (-first + IS - 1) / IS : 0;
        int r_hi = (a.Hin - first + IS - 1) / IS; r_hi = r_hi > HROWS ? HROWS : (r_hi < 0 ? 0 : r_hi);
        rowm = q_tv ? (((1u << r_hi) - 1u) & ~((1u << r_lo) - 1u)) : 0u;
#pragma unroll
        for (int i = 0; i < HROWS; ++i) {
            if (i < r0 || i >= r1) continue;
            areg[i] = __builtin_amdgcn_raw_buffer_load_b128(srd, (unsigned)(base + i * rs) | cmask, 0, 0);
        }
        {
            const int iyr = IS * (iy0 + xrow) + py, ixx = IS * (ix0 + xcol) + px;
            xv = xthr && cv && iyr >= 0 && iyr < a.Hin && ixx >= 0 && ixx < a.Win;
            const int off = (iyr * a.Win + ixx) * a.Cin * (int)sizeof(T) + ck * 16;
            if (with_x) areg[HROWS] = __builtin_amdgcn_raw_buffer_load_b128(srd, (unsigned)off | (xv ? 0u : OOB), 0, 0);
        }
    };
    auto issue = [&]() __attribute__((always_inline)) { issue_loads(); adopt(); };
    auto request = [&]() __attribute__((always_inline)) { prep(); issue(); };
    // the kernel's first request: input loads go out BEFORE the wave waits for the partial sums (one memory round trip for
    // everything the first dump needs instead of two)
    auto request_first = [&](int r0, int r1, bool with_x) __attribute__((always_inline)) {
        const bool fresh = decode();
        if (fresh) stats_issue(q_b);
        coef_loads();
        issue_loads(r0, r1, with_x);
        if (fresh) { stats_reduce(); st_b = q_b; }
        coef_sel();
        adopt();
    };
    auto dump = [&](int buf, int r0 = 0, int r1 = TH + 2, bool with_x = true) __attribute__((always_inline)) {
        unsigned char* const As = smem + buf * L::A_BYTES;
        int pc = pcol; asm volatile("" : "+v"(pc));               // LDS addresses recomputed here, not kept live across the loop
        // LDS rows of 128 B, 16-byte slices XOR-swizzled by the halo COLUMN ((hx >> 1) & 7): a tap's dy then moves a
        // fragment address by a constant, which lets the consumers address all three dy taps with immediate offsets
#pragma unroll
        for (int i = 0; i < AIT; ++i) {
            if (i < HROWS ? (i < r0 || i >= r1) : !with_x) continue;
            const int px = i < HROWS ? i * HPITCH + pc : (pc >> 1) * HPITCH + 32 + (pc & 1);
            // 3x3 form (16x16x32 MFMA): slices swizzled by (hx >> 1) & 3; the 4-tap / 2-tap forms (32x32x16): by (hx >> 1) & 7
            const int sw = i < HROWS ? (NTAPS == 9 ? (pc >> 1) & 3 : (pc >> 1)) : 0;   // extra item: columns 32, 33 -> swizzle 0
            u32x4 v = areg[i];
            const bool ok = i < HROWS ? (((rowm >> i) & 1u) && colv) : xv;
            if (gn) {
                const u32x4 tr = gk.template apply<true>(v);
                v = u32x4{0u, 0u, 0u, 0u};                                  // padding stays zero
                if (ok) v = tr;                                             // (exec-masked move: scalar ops instead of 4 selects)
            }
            if (i < HROWS || xthr) *(u32x4*)(As + px * 128 + (((ck ^ sw) & 7) << 4)) = v;
            if ((i & 1) == 1) __builtin_amdgcn_sched_barrier(0);        // bound the scheduler's appetite for registers
        }
    };

    // first chunk of the launch: halo rows [0, HSPLIT) + the extra columns by the producers, [HSPLIT, HROWS) by the consumers
    constexpr int HSPLIT = HROWS / 2;

    if (wave >= 4) {
        // ------------------------------------------------------------------ producers (4 waves): input chunks + tile epilogues
        // The VALU is the scarce resource here (GroupNorm + SiLU costs ~45 VALU per 16 bytes, a quarter of them
        // transcendental, next to a consumer wave that owns the SIMD's issue priority), so the per-item address math is
        // reduced to one add: item i of a thread is halo row i at a fixed column, offsets are base + i * row stride, row
        // validity is wave-uniform, and everything is branch-free (out-of-range offsets make loads return zero).
        // the producers outrank the consumers (priority 2) where THEY are the pole (launch_conv_pr decides per layer)
        if (a.prod_first) __builtin_amdgcn_s_setprio(3);
        if (CCN_DBG_BIT(a, 128)) __builtin_amdgcn_s_setprio(1);      // diagnostics: producer priority experiments
        if (CCN_DBG_BIT(a, 256)) __builtin_amdgcn_s_setprio(3);
        request_first(0, HSPLIT, true);
        dump(0, 0, HSPLIT, true);
        request();
        raw_barrier();                                             // chunk 0 visible
        stamp(1);
        int c = 0, ti = 0;
        for (int k = 0; k < ktotal; ++k) {
            unsigned long long t0 = CCN_STAMPS_PTR(a) ? __builtin_amdgcn_s_memtime() : 0;
            if (CCN_STAMPS_PTR(a)) {                                        // diagnostic: separate the wait for the chunk registers from the work
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned long long t1 = __builtin_amdgcn_s_memtime(); t_w += t1 - t0; t0 = t1;
            }
            if (k + 1 < ktotal) { prep(); dump((k + 1) & 1); }
            if (CCN_STAMPS_PTR(a)) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); t_a += t1 - t0; t0 = t1; }
            const bool epi = c == 0 && ti > 0;                     // previous tile: its staging was complete at the last barrier
            if (epi) epi_request(0, NQ);
            if (k + 1 < ktotal) issue();
            if (c == nck - 1) epi_setup(vt(ti));                   // this tile finishes in this iteration
            if (CCN_STAMPS_PTR(a)) { const unsigned long long t1 = __builtin_amdgcn_s_memtime(); t_r += t1 - t0; t0 = t1; }
            if (epi) epilogue(0, NQ);
            if (CCN_STAMPS_PTR(a)) t_b += __builtin_amdgcn_s_memtime() - t0;
            // last iteration (always the LAST chunk of a tile, nck >= 2, so no epilogue ran above and its registers are free):
            // fetch the last tile's bias / FiLM / first residual rows now, behind the consumers' last chunk, instead of after the
            // final barrier where nothing hides their latency.  (The second half of a split-K pair first has to see its
            // partner's flag; that wait stays behind the barrier so that it cannot hold this workgroup's consumers up.)
            const bool early = k + 1 == ktotal && !(ks == 2 && e_kh == 1);
            if (early) epi_request(0, NQ);
            timed_barrier();                                       // chunk k+1 visible, chunk k released, staging complete
            combine();                                             // (the sums of an epilogue that ran in this iteration)
            if (++c == nck) { c = 0; ++ti; }
        }
        if (ks == 2 && e_kh == 1) epi_request(0, NQ);
        epilogue(0, NQ);                                           // last tile
        if (pd_on) {
            // the consumers have left (or are leaving) the kernel: the barrier waits for the surviving waves only, the four producers
            raw_barrier();
            combine();
        }
        stamp(2); stamp_cycles();
        return;
    }

    // ---------------------------------------------------------------------- consumers (4 waves)
    if (!CCN_DBG_BIT(a, 16)) __builtin_amdgcn_s_setprio(2);
    if constexpr (NTAPS == 9) {
        // 3x3, column-per-wave form: wave w owns ALL TH tile rows of the 32 output channels nt*128 + 32w, on
        // v_mfma_f32_16x16x32_bf16 (round 3; rounds 1-2 ran v_mfma_f32_32x32x16_bf16 here).  The chip holds a higher clock on this
        // shape: the same loop on random operands, same LDS and weight bytes, 1.90 vs 1.65 GHz = +12.5 % by wall at 3 % more cycles
        // (tools/ubench/consumer_loop.hip shape_ab, profiles/r03_power_ubench.txt; MI355X_MICROARCH.md DVFS give-back item 7).
        //   * wave tile = pixel halves p x channel halves c: quarter q = 2p + c of the row's 16-register accumulator tuple (one
        //     tuple per output row: 32 separate 4-register tuples make the allocator rotate them over the chunk loop's back edge);
        //     operands swapped (weights = A): a lane's quarter is pixel 16p + (lane & 15), channels 16c + 4(lane >> 4) + 0..3;
        //   * one group = (dx, 32-channel K slice k32): 6 weight fragments (dy x c) + 2 HROWS row fragments (halo row x p), each row
        //     fragment read ONCE from LDS and used for up to six MFMAs (three output rows x both channel halves) when it arrives;
        //   * the MFMA's k-block (lane >> 4) of slice k32 is the chunk's 16-byte channel slice s = ((g & 1) << 2 | k32 << 1 | g >> 1)
        //     (pr3_slice, ccn_internal.h; the host / device packers put the weights in the same order) and dump() swizzles the
        //     slices of halo column hx by (hx >> 1) & 3: with that pairing every ds_read_b128 of the loop is bank-conflict free
        //     (its 16-lane groups mix k-blocks {0,1} or {2,3}, whose slices differ in bit 2, which the swizzle never touches);
        //   * weights through a register ring of two groups: the six fragments of group g+1 leave in six six-MFMA steps of group g
        //     (one 1-KiB load per step, ~1500 cycles ahead of use), wrapping into the next chunk / tile.
        constexpr int NROW = HROWS, NSG = 2 * NROW;                // steps of a group: (halo row, pixel half)
        constexpr int NG = 6;                                      // groups of a chunk: (dx, k32)
        constexpr int WIN = 6, PF = 4;                             // row-fragment window / prefetch distance (steps of <= 96 cycles)
        static_assert((NG * NSG) % WIN == 0, "static window indexing");
        f32x16 acc[TH];
        const int n32 = a.Cout_pad / 32;
        constexpr unsigned COLB = 36 * 1024;
        const unsigned wtotal = (unsigned)((size_t)a.nchunk * n32 * COLB);
        const auto wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.wfrag, 0, wtotal, 0x00020000);
        const unsigned lane16 = (unsigned)lane * 16u;
        auto wbase_of = [&](int tile, int chunk) __attribute__((always_inline)) -> unsigned {
            const int nt = tile - CCN_FDIV(tile, a.fd_nt, a.n_nt) * a.n_nt;
            return (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)(chunk * n32 + nt * 4 + wave) * COLB));
        };
        const int px16 = lane & 15, g4 = lane >> 4;
        int b16x[3];                                               // halo row 0, column px16 + dx (dx = 0..2 <-> -1..+1), pixel half 0, k32 = 0
#pragma unroll
        for (int d = 0; d < 3; ++d) { const int hx = px16 + d; b16x[d] = hx * 128 + ((((g4 & 1) << 2) | ((g4 >> 1) ^ ((hx >> 1) & 3))) << 4); }
        u32x4 bq[2][3][2];
        {
            const unsigned wb = wbase_of(vt_tile(vt(0)), vt_kh(vt(0)) * nck);
#pragma unroll
            for (int f = 0; f < 6; ++f) bq[0][f >> 1][f & 1] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, lane16, wb + f * 1024, 0);
        }
        // staging address of this lane: pixel i*32 + 16p + px16, channels 32*wave + 16c + 4*g4 .. +3
        const int stg_lane = px16 * L::SP + (wave * 32 + 4 * g4) * 2;
        if constexpr (HSPLIT < HROWS) {
            request_first(HSPLIT, HROWS, false);                   // the consumers' share of the first chunk (they are idle until it is staged)
            if (CCN_STAMPS_PTR(a)) {                               // diagnostics: the cold start in three parts [100 MHz ticks since this wave's start]
                const unsigned long long r0 = CCN_STAMPS_PTR(a)[((size_t)blockIdx.x * 2) * 8];
                t_a = __builtin_amdgcn_s_memrealtime() - r0;       // statistics reduced, coefficients ready, input loads issued
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                t_b = __builtin_amdgcn_s_memrealtime() - r0;       // input (and first weights) arrived
            }
            dump(0, HSPLIT, HROWS, false);
            if (CCN_STAMPS_PTR(a)) t_r = __builtin_amdgcn_s_memrealtime() - CCN_STAMPS_PTR(a)[((size_t)blockIdx.x * 2) * 8];   // staged
        }
        raw_barrier();                                             // chunk 0 visible
        stamp(1);
        int k = 0;
        // Cout not a multiple of 128 (C4's 192-wide level): the last N tile is part padding.  Every tile of a workgroup has the same N
        // tile when the grid is a multiple of n_nt (tiles are visited with stride `grid`), so a consumer wave whose 32 channels are ALL
        // padding has nothing to compute in this launch: it only keeps the barrier protocol (the epilogue masks those channels anyway).
        const bool idle_w = ks == 1 && grid % a.n_nt == 0 &&
                            (vt_tile(vt(0)) - CCN_FDIV(vt_tile(vt(0)), a.fd_nt, a.n_nt) * a.n_nt) * BN + wave * 32 >= a.Cout;
        if (idle_w) { for (int kk = 0; kk < ktotal; ++kk) timed_barrier(); }
        else
        for (int ti = 0; ti < my_tiles; ++ti) {
            const int v = vt(ti), v_next = ti + 1 < my_tiles ? vt(ti + 1) : v;
            const int tile = vt_tile(v), c0 = vt_kh(v) * nck;
#pragma unroll
            for (int i = 0; i < TH; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][q] = 0.0f;
            for (int chunk = 0; chunk < nck; ++chunk, ++k) {
                const unsigned wb_cur = wbase_of(tile, c0 + chunk);
                const unsigned wb_nxt = chunk + 1 < nck ? wbase_of(tile, c0 + chunk + 1) : wbase_of(vt_tile(v_next), vt_kh(v_next) * nck);
                int bufoff = (k & 1) * L::A_BYTES; asm volatile("" : "+v"(bufoff));
                u32x4 rw[WIN];
                int ag = 0;
                // row fragment `s` of the chunk's stream: group s / NSG = (dx, k32), halo row (s % NSG) / 2, pixel half s & 1; the
                // row / half offset is the ds_read's immediate
                auto rload = [&](int s_) __attribute__((always_inline)) {
                    const int g = s_ / NSG, hp = s_ % NSG;
                    if (hp == 0) ag = (bufoff + b16x[g >> 1]) ^ ((g & 1) << 5);
                    rw[s_ % WIN] = *(const u32x4*)(smem + ag + ((hp >> 1) * HPITCH + (hp & 1) * 16) * 128);
                };
                if (!CCN_DBG_BIT(a, 64)) {
#pragma unroll
                for (int s_ = 0; s_ < PF; ++s_) rload(s_);
#pragma unroll
                for (int s_ = 0; s_ < NG * NSG; ++s_) {
                    const int g = s_ / NSG, hh = (s_ % NSG) >> 1, p = s_ & 1;
                    __builtin_amdgcn_sched_barrier(0);
                    // fragment f of group g+1 (wraps into the next chunk / tile) leaves in the f-th six-MFMA step of this group
                    const bool wl = TH == 8 ? (p == 0 && hh >= 2 && hh <= 7) : (hh >= 1 && hh <= 3);
                    if (wl) {
                        const int f = TH == 8 ? hh - 2 : (hh - 1) * 2 + p;
                        const unsigned off = g + 1 < NG ? wb_cur + (unsigned)((g + 1) * 6 + f) * 1024u : wb_nxt + (unsigned)f * 1024u;
                        bq[(g + 1) & 1][f >> 1][f & 1] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, lane16, off, 0);
                    }
                    if (s_ + PF < NG * NSG) rload(s_ + PF);
                    int nm = 0;
                    // The six (dy, c) weight fragments are visited back and forth from step to step: a step ends on the fragment the next
                    // one starts with, so ONE MFMA operand changes per MFMA instead of 7 per 6 -- less operand toggling, less energy per
                    // MFMA on a power-limited chip: +0.4 ... +0.75 % images/s (three interleaved A/B rounds).  Every accumulator quarter
                    // still receives its contributions in the same order (one MFMA per quarter per step): results are bit-identical.
#pragma unroll
                    for (int q6 = 0; q6 < 6; ++q6) {
                        const int o = (s_ & 1) ? 5 - q6 : q6;
                        const int dy = o >> 1, c = o & 1;
                        const int i = hh - dy;
                        if (i >= 0 && i < TH) { mfma16q(acc[i], p * 2 + c, bq[g & 1][dy][c], rw[s_ % WIN]); ++nm; }
                    }
#pragma unroll
                    for (int m = 0; m < 6; ++m) {
                        if (m < nm) __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (m == 0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        if (m == 2 && wl) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                        if (m & 1) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                }
                if (chunk + 1 == nck) {
                    // hand the tile to the producers: bf16 staging, 8 bytes (4 channels) per store
#pragma unroll
                    for (int i = 0; i < TH; ++i)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x16& c = acc[i];
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            const u32x2 pk = {pack_bf2(c[q * 4], c[q * 4 + 1]), pack_bf2(c[q * 4 + 2], c[q * 4 + 3])};
                            *(u32x2*)(stg + stg_lane + (i * 32 + (q >> 1) * 16) * L::SP + ((q & 1) * 16) * 2) = pk;
                        }
                }
                timed_barrier();                                   // chunk k+1 visible, chunk k released, staging complete
            }
        }
        __builtin_amdgcn_s_setprio(0);
        stamp(2); stamp_cycles();
    } else {
    // ConvTranspose parities (4 taps) and stride-2 plane passes (2 taps): 2 x 2 waves of 4 x 2 fragments of v_mfma_f32_32x32x16_bf16
    static_assert(TH == 8, "the 4x2 fragment form works on 8-row tiles");
    const int wm = wave >> 1, wn = wave & 1;
    f32x16 acc[MF][NF];
    // byte offset of this lane's 16-byte slice (k-slice 0) of halo pixel (row_lin = hy*HPITCH + hx): slice index h ^ (hx >> 1)
    // in the low bit and (hx >> 1) & 6 above it; k-slice kk flips bits 5-6 of the address (XOR with kk << 5)
    auto rbase = [&](int row_lin, int hx) __attribute__((always_inline)) { return row_lin * 128 + (((hx >> 1) & 6) << 4) + (((h ^ (hx >> 1)) & 1) << 4); };
    int prow[MF];
#pragma unroll
    for (int i = 0; i < MF; ++i) prow[i] = ((wm * MF + i) + 1) * HPITCH + r + 1;
    // weight fragments: [chunk][Cout_pad/32][tap][kk][lane] x 16 B (host-packed); this wave owns columns nt*4 + wn*2 + {0,1}
    const int n32 = a.Cout_pad / 32;
    constexpr unsigned COLB = NSTEP * 1024;                        // bytes of one 32-channel column of one chunk
    const unsigned wtotal = (unsigned)((size_t)NPARC * a.nchunk * n32 * COLB);
    const auto wsrd = __builtin_amdgcn_make_buffer_rsrc((void*)a.wfrag, 0, wtotal, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    auto wbase_of = [&](int tile, int chunk) __attribute__((always_inline)) -> unsigned {
        const int t2 = CCN_FDIV(tile, a.fd_nt, a.n_nt);
        const int nt = tile - t2 * a.n_nt, par = t2 % NPARC;               // ConvTranspose: [parity][chunk][column][tap][kk][lane]
        return (unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned)((par * a.nchunk + chunk) * n32 + nt * 4 + wn * NF) * COLB));
    };
    u32x4 bq[D][NF];
    {
        const unsigned wb = wbase_of(vt_tile(vt(0)), vt_kh(vt(0)) * nck);
#pragma unroll
        for (int s = 0; s < D - 1; ++s)
#pragma unroll
            for (int jn = 0; jn < NF; ++jn) bq[s][jn] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, lane16, wb + s * 1024 + jn * COLB, 0);
    }
    // staging address of this lane: pixel (wm*4 + i)*32 + r, channels wn*64 + j*32 + g*8 + 4*h .. +3
    const int stg_lane = (wm * 4 * 32 + r) * L::SP + (wn * 64 + 4 * h) * 2;

    if constexpr (HSPLIT < HROWS) {
        request_first(HSPLIT, HROWS, false);
        dump(0, HSPLIT, HROWS, false);
    }
    raw_barrier();                                                 // chunk 0 visible
    stamp(1);
    int k = 0;
    for (int ti = 0; ti < my_tiles; ++ti) {
        const int v = vt(ti), v_next = ti + 1 < my_tiles ? vt(ti + 1) : v;
        const int tile = vt_tile(v), c0 = vt_kh(v) * nck;        // first chunk of this virtual tile
        int toffs[NTAPS], tdxs[NTAPS];                            // ConvTranspose: the parity's 2x2 taps (wave-uniform)
        if constexpr (NTAPS == 4 && !P4) {
            const int par = CCN_FDIV(tile, a.fd_nt, a.n_nt) % NPARC;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) { tdxs[t] = a.tapinfo_dx(par * 4 + t); toffs[t] = a.tapinfo_dy(par * 4 + t) * HPITCH + tdxs[t]; }
        }
#pragma unroll
        for (int i = 0; i < MF; ++i)
#pragma unroll
            for (int j = 0; j < NF; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;
        for (int chunk = 0; chunk < nck; ++chunk, ++k) {
            const unsigned wb_cur = wbase_of(tile, c0 + chunk);
            const unsigned wb_nxt = chunk + 1 < nck ? wbase_of(tile, c0 + chunk + 1) : wbase_of(vt_tile(v_next), vt_kh(v_next) * nck);
            const int bufoff = (k & 1) * L::A_BYTES;
            if constexpr (S2) {                                    // this pass's two taps inside the staged plane (table in the header)
                const int pass = (c0 + chunk) % 5;
                tdxs[0] = (pass == 0 || pass == 1 || pass == 3) ? -1 : 0; tdxs[1] = 0;
                toffs[0] = ((pass == 0 || pass == 2) ? -HPITCH : 0) + tdxs[0];
                toffs[1] = pass == 0 ? -HPITCH : 0;
            }
            if constexpr (P4) {                                    // this pass's 2x2 taps inside the staged plane
                const int pass = (c0 + chunk) & 3;
                const int py = pass < 2 ? 1 : 0, px = (pass & 1) ? 0 : 1;
#pragma unroll
                for (int t = 0; t < NTAPS; ++t) { tdxs[t] = (t & 1) - px; toffs[t] = ((t >> 1) - py) * HPITCH + tdxs[t]; }
            }
#pragma unroll
            for (int i = 0; i < MF; ++i) asm volatile("" : "+v"(prow[i]));    // keep the address math inside the loop
            u32x4 av[2][MF];
            int abase[MF];
            // steps in (tap, kk) order, one address set per tap
            auto frag = [&](int j, u32x4* av_) __attribute__((always_inline)) {
                const int tt = j >> 2, kk = j & 3;
                if (kk == 0) {
#pragma unroll
                    for (int i = 0; i < MF; ++i) abase[i] = bufoff + rbase(prow[i] + toffs[tt], r + 1 + tdxs[tt]);
                }
#pragma unroll
                for (int i = 0; i < MF; ++i) av_[i] = *(const u32x4*)(smem + (abase[i] ^ (kk << 5)));
            };
            if (!CCN_DBG_BIT(a, 64)) {                                   // CCN_DBG=64: consumers idle (timing experiments only)
            frag(0, av[0]);
#pragma unroll
            for (int j = 0; j < NSTEP; ++j) {
                __builtin_amdgcn_sched_barrier(0);
                {
                    // refill the ring slot step j-1 just released with the fragments of step j+D-1 (wraps into the next chunk / tile)
                    const int p = j + D - 1;
                    const unsigned off = p < NSTEP ? wb_cur + (unsigned)p * 1024u : wb_nxt + (unsigned)(p - NSTEP) * 1024u;
#pragma unroll
                    for (int jn = 0; jn < NF; ++jn)
                        bq[p % D][jn] = __builtin_amdgcn_raw_buffer_load_b128(wsrd, lane16, off + jn * COLB, 0);
                }
                if (j + 1 < NSTEP) frag(j + 1, av[(j + 1) & 1]);
                // operands swapped (weights as the MFMA A operand): a lane's accumulator registers run over output
                // channels ((q&3) + 8*(q>>2) + 4*h) of pixel r, i.e. 4 consecutive channels per register quad
                // (the second weight fragment walks the row fragments backwards: it starts on the one the first ended on -- one operand
                // change less per step, bit-identical results; see the 3x3 form)
#pragma unroll
                for (int jn = 0; jn < NF; ++jn)
#pragma unroll
                    for (int ii = 0; ii < MF; ++ii) { const int i = (jn & 1) ? MF - 1 - ii : ii; mfma16<T>(acc[i][jn], bq[j % D][jn], av[j & 1][i]); }
                // one MFMA, then (in its shadow) one weight load, one LDS fragment read and a little address math of the next step
#pragma unroll
                for (int m = 0; m < MF * NF; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (m < NF) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if (chunk + 1 == nck) {
                // hand the tile to the producers: bf16 staging, 8 bytes (4 channels) per store
#pragma unroll
                for (int i = 0; i < MF; ++i)
#pragma unroll
                    for (int j = 0; j < NF; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x16& c = acc[i][j];
                            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
                            const u32x2 pk = {pack_bf2(c[g * 4], c[g * 4 + 1]), pack_bf2(c[g * 4 + 2], c[g * 4 + 3])};
                            *(u32x2*)(stg + stg_lane + i * 32 * L::SP + (j * 32 + g * 8) * 2) = pk;
                        }
            }
            timed_barrier();                                       // chunk k+1 visible, chunk k released, staging complete
        }
    }
    __builtin_amdgcn_s_setprio(0);
    stamp(2); stamp_cycles();
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------
constexpr int PR_D = 6;                                          // weight ring depth (steps) of the 4-tap / 2-tap forms
typedef void (*pr_fn_t)(const ConvArgs, int);
static pr_fn_t pick_pr(int ntaps, int mode, int th = 8, int gs = 0)
{
    if (ntaps == 16) return mode == 1 ? (pr_fn_t)conv_pr_kernel<4, 8, 1, 8, 0, true> : (pr_fn_t)conv_pr_kernel<4, 8, 0, 8, 0, true>;   // 4x4 s2 as plane passes
    if (gs == 1 && ntaps == 9 && mode != 2) {
        if (th == 4) return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1, 4, 1> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0, 4, 1>;
        return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1, 8, 1> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0, 8, 1>;
    }
    if (gs == 2 && ntaps == 9) {
        if (th == 4) return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1, 4, 2> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 2, 4, 2> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0, 4, 2>);
        return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1, 8, 2> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 2, 8, 2> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0, 8, 2>);
    }
    if (ntaps == 9 && th == 4) return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1, 4> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 2, 4> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0, 4>);
    if (ntaps == 9) return mode == 1 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 1> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<9, PR_D, 2> : (pr_fn_t)conv_pr_kernel<9, PR_D, 0>);
    if (ntaps == 2) return mode == 1 ? (pr_fn_t)conv_pr_kernel<2, 8, 1> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<2, 8, 2> : (pr_fn_t)conv_pr_kernel<2, 8, 0>);
    return mode == 1 ? (pr_fn_t)conv_pr_kernel<4, 8, 1> : (mode == 2 ? (pr_fn_t)conv_pr_kernel<4, 8, 2> : (pr_fn_t)conv_pr_kernel<4, 8, 0>);
}

bool conv_pr_supported(int kind, int bn, int th)
{
    return bn == 128 && (((kind == KIND_C3S1 || kind == KIND_CT4 || kind == KIND_C3S2) && th == 8) || (kind == KIND_C3S1 && th == 4));
}

static int g_cus = 0;
hipError_t conv_pr_prepare()
{
    hipError_t e = hipSuccess;
    for (int ntaps : {2, 4, 9, 16})
        for (int mode = 0; mode < 3; ++mode) {
            if (ntaps == 16 && mode == 2) continue;
            e = hipFuncSetAttribute((const void*)pick_pr(ntaps, mode), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PrLds::TOTAL);
            if (e != hipSuccess) return e;
        }
    for (int mode = 0; mode < 3; ++mode) {
        e = hipFuncSetAttribute((const void*)pick_pr(9, mode, 4), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PrLdsT<4>::TOTAL);
        if (e != hipSuccess) return e;
    }
    for (int gs = 1; gs <= 2; ++gs)
        for (int mode = 0; mode < 3; ++mode) {
            if (gs == 1 && mode == 2) continue;
            e = hipFuncSetAttribute((const void*)pick_pr(9, mode, 8, gs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PrLds::TOTAL);
            if (e != hipSuccess) return e;
            e = hipFuncSetAttribute((const void*)pick_pr(9, mode, 4, gs), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PrLdsT<4>::TOTAL);
            if (e != hipSuccess) return e;
        }
    int dev = 0;
    e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return e;
    g_cus = prop.multiProcessorCount;
    return hipSuccess;
}

static unsigned long long* g_stamps = nullptr;
static unsigned g_stamp_grid = 0;
extern "C" int ccn_internal_dump_stamps_pr(const char* path)
{
    if (!g_stamps || !g_stamp_grid) return 1;
    std::vector<unsigned long long> hbuf((size_t)g_stamp_grid * 16);
    if (hipMemcpy(hbuf.data(), g_stamps, hbuf.size() * 8, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    FILE* f = fopen(path, "w");
    if (!f) return 3;
    for (unsigned b = 0; b < g_stamp_grid; ++b)
        for (int k = 0; k < 16; ++k) fprintf(f, "%llu%c", hbuf[(size_t)b * 16 + k], k == 15 ? '\n' : ' ');
    fclose(f);
    return 0;
}

hipError_t launch_conv_pr(int dtype, const ConvArgs& a, hipStream_t s)
{
    // a.ntaps as the persistent kernel sees the layer: 9 (3x3 s1), 4 (ConvTranspose parity), 2 (3x3 s2 as plane passes, nchunk = 5 x channel chunks)
    const bool c3 = a.ntaps == 9 && a.npar == 1 && a.OS == 1, ct = a.ntaps == 4 && a.npar == 4 && a.OS == 2;
    const bool s2 = a.ntaps == 2 && a.npar == 1 && a.OS == 1 && !a.gn_ab && (a.nchunk % 5) == 0;
    // a.ntaps == 16: the 4x4 stride-2 conv as four plane passes of 2x2 taps (nchunk = 4 x channel chunks; training's ConvTranspose data gradient)
    const bool p4 = a.ntaps == 16 && a.npar == 1 && a.OS == 1 && (a.nchunk % 4) == 0 && a.ksplit != 2;
    if ((ct || s2 || p4) && (a.gn_ab || a.gs_part)) return hipErrorInvalidValue;     // (no input GroupNorm in those forms)
    if (dtype != 1 || !a.wfrag || !(c3 || ct || s2 || p4) || !(a.th == 8 || (a.th == 4 && c3)) || (a.Cout_pad & 127) || a.nchunk < 2 || a.fin_counter || (a.res && a.film)) return hipErrorInvalidValue;
    const int ks = a.ksplit == 2 ? 2 : 1;
    if (ks == 2 && (!a.kpart || !a.kflag || (a.nchunk & 1) || (s2 && (a.nchunk / 2) % 5) || a.res)) return hipErrorInvalidValue;   // (split-K: MODE 2, no residual)
    const int ntiles = a.B * a.n_ty * a.n_tx * a.npar * a.n_nt * ks;
    static const int cap = diag_env("CCN_PR_GRID") ? atoi(diag_env("CCN_PR_GRID")) : 0;       // diagnostics build only
    int grid = cap > 0 ? cap : (g_cus > 0 ? g_cus : 256);
    if (grid > ntiles) grid = ntiles;
    // split-K partners (virtual tiles 2t, 2t+1) must run at the same time AND on the same XCD (the hand-off has no agent-scope
    // fence): a grid that is a multiple of 16 gives every XCD an even-sized, even-aligned contiguous range of virtual ids in
    // every round of the persistent loop.  Fewer than 16 virtual tiles: the layer is too small for this kernel anyway.
    if (ks == 2) { grid &= ~15; if (grid < 16) return hipErrorInvalidValue; }
    ConvArgs d = a;
    {
        auto magic = [](long dv) -> unsigned { return dv <= 1 ? 0u : (unsigned)((0x100000000ull + (unsigned long long)dv - 1) / (unsigned long long)dv); };
        const long dmax = std::max<long>(std::max<long>((long)a.n_nt * a.npar, (long)a.n_tx * a.n_ty), std::max<long>(a.cpg, a.gs_cpg));
        if ((unsigned long long)std::max<long>(ntiles, 8192) * (unsigned long long)dmax >= 0x100000000ull) return hipErrorInvalidValue;   // CCN_FDIV exactness
        d.fd_nt = magic(a.n_nt); d.fd_ntp = magic((long)a.n_nt * a.npar); d.fd_tx = magic(a.n_tx); d.fd_sp = magic((long)a.n_tx * a.n_ty);
        d.fd_cpg = magic(a.cpg); d.fd_gscpg = magic(a.gs_cpg);
        d.fd_gsbn = magic(a.gs_bn); d.fd_gsnsp = magic(a.gs_nsp);
    }
    // Producer waves outrank the consumers on the 2-chunk 3x3 layers (the 128-channel levels, where the producers are the pole and
    // the consumers wait a quarter of their time at barriers): +0.8 % on two boxes in product builds; on every layer: -0.5..+1.7 %
    // by box; on the deep-K layers only: 0.
    d.prod_first = (a.nchunk <= 2 && c3) ? 1 : 0;
    static const char* env = diag_env("CCN_STAMPS");
    if (env && (unsigned)atoi(env) == (unsigned)ntiles && (!strchr(env, ':') || atoi(strchr(env, ':') + 1) == a.ntaps)) {   // CCN_STAMPS=<tiles>[:<ntaps>]
        if (!g_stamps) { if (hipMalloc((void**)&g_stamps, (size_t)1024 * 24 * 8) != hipSuccess) return hipErrorOutOfMemory; }
        g_stamp_grid = (unsigned)grid;
        d.stamps = g_stamps;
    } else d.stamps = nullptr;
    if (a.gs_part && !(c3 && ks == 1)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(pick_pr(a.ntaps, a.res ? 1 : (ks == 2 ? 2 : 0), a.th, a.gs_part ? 1 : (a.gn_ab ? 0 : 2)), dim3((unsigned)grid), dim3(512),
                       a.th == 4 ? PrLdsT<4>::TOTAL : PrLds::TOTAL, s, d, ntiles);
    return hipGetLastError();
}

}  // namespace ccn
