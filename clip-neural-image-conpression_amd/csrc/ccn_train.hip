// Host side of the training step: ccn_train_* of include/ccn_hip.h.
//
// Reference: the loop body of train/diffusion_train.py:119-124,137-140 -- eps_hat = net(x_t, z, t); loss = mse(eps_hat, noise);
// loss.backward(); opt.step().  ccn_train_forward is CLIPCondUNet.forward (models/unet.py:81-106) with every tensor the
// backward needs kept in the caller's workspace; ccn_train_backward takes d loss / d eps_hat and ACCUMULATES the gradient of
// every parameter into a flat fp32 buffer laid out like the flat parameter buffer (the order of CLIPCondUNet.state_dict()).
//
// Parameters are read from the caller's flat fp32 buffer on every call (they change with every optimiser step) and repacked on
// the device into the operand layouts of the convolution kernels.  Activations and activation gradients are NHWC in the
// handle's arithmetic type (fp32 parity mode / bf16); GroupNorm statistics, FiLM, the conditioning MLP, all parameter
// gradients and the optimiser state are fp32.
//
// Layout of the workspace: [shared scratch regions][tensors in allocation order].  Every call walks the fixed layer list with a
// bump allocator, so the forward walk of ccn_train_backward (launches disabled) finds the tensors the forward call wrote.
#include "../../include/ccn_hip.h"
#include "ccn_train.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

using namespace ccn;

extern "C" void ccn_internal_set_error(const char* msg);   // ccn_api.hip: thread-local message behind ccn_last_error()

namespace {

int tfail(int code, const std::string& msg) { ccn_internal_set_error(msg.c_str()); return code; }
#define THIP(expr)                                                                                       \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) { err = std::string(#expr) + ": " + hipGetErrorString(e_); return false; } \
    } while (0)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

struct PInfo { std::string name; std::vector<int64_t> shape; size_t off = 0; size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; } };

struct TConvW {
    int kind = KIND_C3S1, Cin = 0, Cout = 0;
    int pw = -1, pb = -1;                       // parameter indices (weight, bias)
    int BN = 0, Cin_pad = 0, Cout_pad = 0;      // forward operand
    int dkind = KIND_C3S1, dBN = 0, dCin_pad = 0, dCout_pad = 0, dtaps = 9;   // data-gradient operand (roles of Cin / Cout swapped)
    void* wf = nullptr; void* wd = nullptr;
    void* wf_frag = nullptr; void* wd_frag = nullptr;   // 3x3 s1, bf16: fragment-ordered operands for the persistent kernel, else null
    PackDesc pd_f{}, pd_d{}, pf_f{}, pf_d{};             // repack descriptors: plain / fragment, forward / data gradient
};
struct TNorm { int C = 0, pg = -1, pb = -1; };
struct TLin { int pw = -1, pb = -1, N = 0, K = 0; };
struct TRes { int C = 0; TNorm n1, n2; TConvW c1, c2; TLin fs, fh; int film_off = 0; };
enum LType { L_STEM, L_RES, L_DOWN, L_UP, L_HEAD };
struct Layer { LType type; int idx; };

struct TT {                                     // an NHWC activation and the partial sums of the GroupNorm that reads it
    void* p = nullptr; int C = 0, H = 0, W = 0;
    float2* part = nullptr; int n_sp = 0, n_nt = 0, bn = 0;
};

enum { TF_PACK = 0, TF_COND, TF_CONV_FWD, TF_GN_STATS, TF_PREACT, TF_CONV_DGRAD, TF_WGRAD, TF_WGRAD_REDUCE, TF_BIAS, TF_GN_BWD, TF_SMALL, TF_COUNT };
const char* kTrainFamilies[TF_COUNT] = {"weight_repack", "conditioning_fwd_bwd", "conv_forward", "gn_stats", "gn_silu_prepass", "conv_data_grad", "conv_weight_grad",
                                        "weight_grad_reduce", "bias_grad", "groupnorm_silu_film_bwd", "stem_head_weight_grad"};
struct Mark { int fam; hipEvent_t ev; double flops, bytes; };

struct ShapeInfo { size_t scr_wg = 0, scr_gn = 0, scr_film = 0, scr_col = 0, tensors = 0, total = 0; PackDesc* packs = nullptr; int n_packs = 0, n_packs_fwd = 0; };

}  // namespace

struct ccn_trainer_s {
    ccn_config_t cfg{};
    int elem = 4, G = 8;
    std::vector<PInfo> params;
    size_t total = 0;
    TConvW stem, head;
    TNorm out_norm;
    TLin tp0, tp2, zp;
    std::vector<TRes> res;
    std::vector<TConvW> downs, ups;
    std::vector<Layer> layers;
    int F = 0;
    float* zero_bias = nullptr;
    // captured hipGraphs of one forward / one backward, keyed by every pointer argument (ccn_train_set_graph)
    struct TGraph { std::vector<const void*> key; hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr; };
    bool use_graph = false;
    hipStream_t cap_stream = nullptr;
    std::vector<TGraph> graphs;
    hipStream_t side = nullptr;           // weight gradients run here, beside the data-gradient chain (nothing downstream reads them)
    hipEvent_t pack_dg_done = nullptr;    // the data-gradient operands' repack, enqueued on `side` by the last forward
    bool pack_dg_pending = false;
    hipEvent_t fwd_fork = nullptr;
    std::vector<hipEvent_t> sync_pool; size_t sync_used = 0;
    LinDesc* lin_descs = nullptr; int n_lin = 0, max_lin_n = 0;
    std::vector<void*> allocs;
    std::map<std::string, ShapeInfo> shapes;
    // profiling (ccn_train_profile_*): an event before every group of launches; the time up to the next event is the group's
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool; size_t ev_used = 0;
    std::vector<Mark> marks;
    // state of the last forward (checked by backward)
    int fB = 0, fH = 0, fW = 0; void* fws = nullptr;
};

namespace {

int add_param(ccn_trainer_s* tr, const std::string& name, std::vector<int64_t> shape)
{
    PInfo p; p.name = name; p.shape = std::move(shape); p.off = tr->total;
    tr->total += align_up(p.numel(), 4);                    // 16-byte aligned views
    tr->params.push_back(p);
    return (int)tr->params.size() - 1;
}

// models/unet.py:45-79 registration order (the key order of state_dict())
void build_arch(ccn_trainer_s* tr)
{
    const ccn_config_t& c = tr->cfg;
    const int td = c.time_dim;
    auto lin = [&](const std::string& n, int N, int K) { TLin l; l.N = N; l.K = K; l.pw = add_param(tr, n + ".weight", {N, K}); l.pb = add_param(tr, n + ".bias", {N}); return l; };
    auto conv = [&](const std::string& n, int kind, int Cin, int Cout) {
        TConvW w; w.kind = kind; w.Cin = Cin; w.Cout = Cout;
        if (kind == KIND_CT4) w.pw = add_param(tr, n + ".weight", {Cin, Cout, 4, 4});
        else w.pw = add_param(tr, n + ".weight", {Cout, Cin, 3, 3});
        w.pb = add_param(tr, n + ".bias", {Cout});
        return w;
    };
    auto norm = [&](const std::string& n, int C) { TNorm g; g.C = C; g.pg = add_param(tr, n + ".weight", {C}); g.pb = add_param(tr, n + ".bias", {C}); return g; };
    tr->tp0 = lin("time_proj.0", td * 4, td);
    tr->tp2 = lin("time_proj.2", td, td * 4);
    tr->zp = lin("z_proj.0", td, c.z_dim);
    tr->stem = conv("in_conv", KIND_STEM, c.img_ch, c.base);
    tr->layers.push_back({L_STEM, 0});
    int ch = c.base, film_off = 0;
    auto add_res = [&](const std::string& name, int cc) {
        TRes r; r.C = cc;
        r.n1 = norm(name + ".norm1", cc); r.c1 = conv(name + ".conv1", KIND_C3S1, cc, cc);
        r.n2 = norm(name + ".norm2", cc); r.c2 = conv(name + ".conv2", KIND_C3S1, cc, cc);
        r.fs = lin(name + ".film.to_scale", cc, td); r.fh = lin(name + ".film.to_shift", cc, td);
        r.film_off = film_off; film_off += 2 * cc;
        tr->res.push_back(r);
        tr->layers.push_back({L_RES, (int)tr->res.size() - 1});
    };
    for (int i = 0; i < c.n_mult; ++i) {
        const int m = c.ch_mult[i];
        add_res("down." + std::to_string(3 * i), ch);
        add_res("down." + std::to_string(3 * i + 1), ch);
        tr->downs.push_back(conv("down." + std::to_string(3 * i + 2), KIND_C3S2, ch, ch * m));
        tr->layers.push_back({L_DOWN, i});
        ch *= m;
    }
    add_res("mid1", ch);
    add_res("mid2", ch);
    for (int i = 0; i < c.n_mult; ++i) {
        const int m = c.ch_mult[c.n_mult - 1 - i];
        add_res("up." + std::to_string(3 * i), ch);
        add_res("up." + std::to_string(3 * i + 1), ch);
        tr->ups.push_back(conv("up." + std::to_string(3 * i + 2), KIND_CT4, ch, ch / m));
        tr->layers.push_back({L_UP, i});
        ch /= m;
    }
    tr->out_norm = norm("out_norm", ch);
    tr->head = conv("out", KIND_HEAD, ch, c.img_ch);
    tr->layers.push_back({L_HEAD, 0});
    tr->F = film_off;
}

bool alloc_dev(ccn_trainer_s* tr, size_t bytes, void** out, std::string& err)
{
    void* p = nullptr;
    THIP(hipMalloc(&p, bytes ? bytes : 4));
    tr->allocs.push_back(p);
    *out = p;
    return true;
}

// operand geometry and device buffers of one convolution (forward and data-gradient operands)
bool setup_conv(ccn_trainer_s* tr, TConvW& w, std::string& err)
{
    const int cke = tr->elem == 2 ? 64 : 32;
    const int fkind = w.kind;
    w.BN = conv_bn_for(w.Cout, fkind);
    w.Cout_pad = (int)align_up(w.Cout, w.BN);
    w.Cin_pad = fkind == KIND_STEM ? cke : (int)align_up(w.Cin, cke);
    const int ftaps = fkind == KIND_CT4 ? 16 : (fkind == KIND_STEM ? 1 : 9);
    if (!alloc_dev(tr, (size_t)ftaps * w.Cout_pad * w.Cin_pad * tr->elem, &w.wf, err)) return false;
    const long long src = (long long)tr->params[w.pw].off;
    switch (fkind) {
        case KIND_CT4: w.pd_f = {src, w.wf, PK_CONVT, w.Cout, w.Cin, 16, w.Cout_pad, w.Cin_pad}; break;
        case KIND_STEM: w.pd_f = {src, w.wf, PK_STEM, w.Cout, w.Cin, 1, w.Cout_pad, w.Cin_pad}; break;
        default: w.pd_f = {src, w.wf, PK_CONV3, w.Cout, w.Cin, 9, w.Cout_pad, w.Cin_pad}; break;
    }
    if (fkind == KIND_STEM) {
        // bf16: the register-weight stem kernel (ccn_stem.hip) instead of the generic one on an im2col tile
        if (stem2_supported(tr->cfg.dtype, w.Cin, w.Cout, tr->G)) {
            if (!alloc_dev(tr, (size_t)w.Cout * 32 * 2, &w.wf_frag, err)) return false;
            w.pf_f = {src, w.wf_frag, PK_FRAG_STEM, w.Cout, w.Cin, 1, w.Cout, 32, (long long)tr->params[w.pb].off};
        }
        return true;                                         // the image needs no gradient
    }
    // data gradient: a convolution from Cout back to Cin
    switch (fkind) {
        case KIND_C3S1: w.dkind = KIND_C3S1; w.dtaps = 9; break;
        case KIND_C3S2: w.dkind = KIND_CT4; w.dtaps = 16; break;
        case KIND_CT4: w.dkind = KIND_C3S2; w.dtaps = 16; break;
        case KIND_HEAD: w.dkind = KIND_STEM; w.dtaps = 1; break;
    }
    w.dBN = conv_bn_for(w.Cin, w.dkind);
    w.dCout_pad = (int)align_up(w.Cin, w.dBN);
    w.dCin_pad = w.dkind == KIND_STEM ? cke : (int)align_up(w.Cout, cke);
    if (!alloc_dev(tr, (size_t)w.dtaps * w.dCout_pad * w.dCin_pad * tr->elem, &w.wd, err)) return false;
    switch (fkind) {
        case KIND_C3S1: w.pd_d = {src, w.wd, PK_DG3S1, w.Cout, w.Cin, 9, w.dCout_pad, w.dCin_pad}; break;
        case KIND_C3S2: w.pd_d = {src, w.wd, PK_DG3S2, w.Cout, w.Cin, 16, w.dCout_pad, w.dCin_pad}; break;
        case KIND_CT4: w.pd_d = {src, w.wd, PK_DGT, w.Cout, w.Cin, 16, w.dCout_pad, w.dCin_pad}; break;
        case KIND_HEAD: w.pd_d = {src, w.wd, PK_HEAD_DG, w.Cout, w.Cin, 1, w.dCout_pad, w.dCin_pad}; break;
    }
    if (fkind == KIND_HEAD && stem2_supported(tr->cfg.dtype, w.Cout, w.Cin, tr->G)) {
        // the head's data gradient is a stem conv (img_ch -> C, taps flipped) on d eps, which arrives as NCHW fp32 like the image
        if (!alloc_dev(tr, (size_t)w.Cin * 32 * 2, &w.wd_frag, err)) return false;
        w.pf_d = {src, w.wd_frag, PK_FRAG_STEM_HEAD_DG, w.Cout, w.Cin, 1, w.Cin, 32, 0};
    }
    if (fkind == KIND_C3S1 && tr->elem == 2) {
        if (w.BN == 128 && w.Cin_pad >= 128) {
            if (!alloc_dev(tr, (size_t)9 * w.Cout_pad * w.Cin_pad * 2, &w.wf_frag, err)) return false;
            w.pf_f = {src, w.wf_frag, PK_FRAG3, w.Cout, w.Cin, 9, w.Cout_pad, w.Cin_pad};
        }
        if (w.dBN == 128 && w.dCin_pad >= 128) {
            if (!alloc_dev(tr, (size_t)9 * w.dCout_pad * w.dCin_pad * 2, &w.wd_frag, err)) return false;
            w.pf_d = {src, w.wd_frag, PK_FRAG3_DG, w.Cout, w.Cin, 9, w.dCout_pad, w.dCin_pad};
        }
    }
    // stride-2 conv / ConvTranspose on the persistent kernel's plane-pass / parity forms (bf16, 128-wide N tiles, >= 2 Cin chunks):
    // the forward operands, and the stride-2 conv's data gradient (a ConvTranspose with the 3x3 kernel zero-padded to 4x4)
    if (tr->elem == 2 && (fkind == KIND_C3S2 || fkind == KIND_CT4) && w.BN == 128 && w.Cin_pad >= 128) {
        const int slots = fkind == KIND_C3S2 ? 10 : 16;
        if (!alloc_dev(tr, (size_t)slots * w.Cout_pad * w.Cin_pad * 2, &w.wf_frag, err)) return false;
        w.pf_f = {src, w.wf_frag, fkind == KIND_C3S2 ? PK_FRAG_S2 : PK_FRAG_CT, w.Cout, w.Cin, slots, w.Cout_pad, w.Cin_pad};
    }
    if (tr->elem == 2 && (fkind == KIND_C3S2 || fkind == KIND_CT4) && w.dBN == 128 && w.dCin_pad >= 128) {
        // (the ConvTranspose's data gradient: a 4x4 stride-2 conv, run as four plane passes of 2x2 taps)
        if (!alloc_dev(tr, (size_t)16 * w.dCout_pad * w.dCin_pad * 2, &w.wd_frag, err)) return false;
        w.pf_d = {src, w.wd_frag, fkind == KIND_C3S2 ? PK_FRAG_CT_DG : PK_FRAG_P4_DG, w.Cout, w.Cin, 16, w.dCout_pad, w.dCin_pad};
    }
    return true;
}

void fill_taps(int* tapinfo, int kind, bool four_by_four)
{
    std::memset(tapinfo, 0, 16 * sizeof(int));
    if (kind == KIND_STEM) { tapinfo[0] = ConvArgs::make_tap(0, 0, 0); return; }
    if (kind == KIND_CT4) {
        // out = 2*in - 1 + k (ConvTranspose2d k=4, s=2, p=1): even out <- k in {1 (d=0), 3 (d=-1)}; odd out <- k in {0 (d=+1), 2 (d=0)}
        static const int kk[2][2] = {{1, 3}, {0, 2}}, dd[2][2] = {{0, -1}, {1, 0}};
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        tapinfo[(py * 2 + px) * 4 + i * 2 + j] = ConvArgs::make_tap(dd[py][i], dd[px][j], kk[py][i] * 4 + kk[px][j]);
        return;
    }
    if (four_by_four) {                                      // 4x4 s2 p1 convolution: in = 2 m + (k - 1), k = 0..3
        for (int ky = 0; ky < 4; ++ky)
            for (int kx = 0; kx < 4; ++kx) tapinfo[ky * 4 + kx] = ConvArgs::make_tap(ky - 1, kx - 1, ky * 4 + kx);
        return;
    }
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) tapinfo[ky * 3 + kx] = ConvArgs::make_tap(ky - 1, kx - 1, ky * 3 + kx);
}

struct Geom { int Hout, Wout, MH, MW, OS, npar, ntaps; };
Geom geom_of(int kind, int Hin, int Win, bool four_by_four)
{
    Geom g{};
    switch (kind) {
        case KIND_C3S2: g.Hout = (Hin - 1) / 2 + 1; g.Wout = (Win - 1) / 2 + 1; g.MH = g.Hout; g.MW = g.Wout; g.OS = 1; g.npar = 1; g.ntaps = four_by_four ? 16 : 9; break;
        case KIND_CT4: g.Hout = Hin * 2; g.Wout = Win * 2; g.MH = Hin; g.MW = Win; g.OS = 2; g.npar = 4; g.ntaps = 4; break;
        case KIND_STEM: g.Hout = Hin; g.Wout = Win; g.MH = Hin; g.MW = Win; g.OS = 1; g.npar = 1; g.ntaps = 1; break;
        default: g.Hout = Hin; g.Wout = Win; g.MH = Hin; g.MW = Win; g.OS = 1; g.npar = 1; g.ntaps = 9; break;
    }
    return g;
}

// One walk over the layer list.  `launch` off: only the allocation sequence (and the scratch maxima) are reproduced.
struct Walk {
    ccn_trainer_s* tr;
    int B, H, W;
    char* base; size_t off = 0;
    bool launch;
    hipStream_t st;
    const float* P; float* Gd;            // flat parameters / gradients
    ShapeInfo need;                       // scratch maxima seen by this walk
    float *scr_wg = nullptr, *scr_col = nullptr; float2 *scr_gn = nullptr, *scr_film = nullptr;
    std::string err;
    // conditioning buffers
    float *temb = nullptr, *u0 = nullptr, *t1 = nullptr, *tp = nullptr, *uz = nullptr, *zpv = nullptr, *hv = nullptr, *film = nullptr;
    float *dfilm = nullptr, *dh = nullptr, *dt1 = nullptr, *du0 = nullptr, *duz = nullptr;
    // saved by the forward walk
    struct ResSave { TT x, y, o, xa, ya; bool pre = false; float2 *ab1, *st1, *ab2, *st2; };
    std::vector<ResSave> rs;
    TT stem_out; std::vector<TT> down_in, down_out, up_in, up_out;
    TT head_in; float2 *ab_o = nullptr, *st_o = nullptr;
    std::vector<PackDesc> pack_list;      // repacks this shape needs, collected by the measuring walk (one per conv launch)
    int g_sum_rows = 0;                   // > 0: scr_film holds per-block channel sums of the current gradient tensor (rows = B * blocks)

    Walk(ccn_trainer_s* t, int B_, int H_, int W_, void* ws, bool l, hipStream_t s, const float* p, float* g)
        : tr(t), B(B_), H(H_), W(W_), base((char*)ws), launch(l), st(s), P(p), Gd(g) {}

    // profiling: the launches that follow belong to family `fam` (flops: their algorithmic work); fam < 0 closes the sequence
    void mark(int fam, double flops = 0.0, double bytes = 0.0)
    {
        if (!launch || !tr->profiling) return;
        if (tr->ev_used == tr->ev_pool.size()) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; tr->ev_pool.push_back(e); }
        hipEvent_t e = tr->ev_pool[tr->ev_used++];
        if (hipEventRecord(e, st) != hipSuccess) return;
        tr->marks.push_back({fam, e, flops, bytes});
    }
    // fork: the side stream continues from this point of the main stream; join: the main stream waits for everything on the side
    hipStream_t wg_stream = nullptr;
    // bucketed backward: cb(user, lo, hi) as soon as the gradients of flat range [lo, hi) are complete in stream order on `st`
    ccn_grad_ready_cb bucket_cb = nullptr; void* bucket_user = nullptr; long long bucket_floats = 0, hi_pending = 0;
    const PackDesc* shape_packs = nullptr; int n_shape_packs = 0, n_shape_packs_fwd = 0;
    hipEvent_t sync_event_fwd()
    {
        // (the forward call's own fork event: the pool is reset at the start of every backward call, after this event has done its work)
        if (!tr->fwd_fork && hipEventCreateWithFlags(&tr->fwd_fork, hipEventDisableTiming) != hipSuccess) tr->fwd_fork = nullptr;
        return tr->fwd_fork;
    }
    bool side_active() const
    {
        static const bool off_ = diag_env("CCN_TRAIN_NO_SIDE_STREAM") != nullptr;
        return !off_ && !tr->profiling && tr->side != nullptr;
    }
    bool fork_side()
    {
        wg_stream = st;
        if (!side_active()) return true;
        if (tr->sync_used == tr->sync_pool.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return true; tr->sync_pool.push_back(e); }
        hipEvent_t e = tr->sync_pool[tr->sync_used++];
        if (hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(tr->side, e, 0) != hipSuccess) { err = "stream fork failed"; return false; }
        wg_stream = tr->side;
        return true;
    }
    hipEvent_t sync_event()
    {
        if (tr->sync_used == tr->sync_pool.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return nullptr; tr->sync_pool.push_back(e); }
        return tr->sync_pool[tr->sync_used++];
    }
    bool join_side()
    {
        if (!tr->side || tr->sync_used == 0) return true;
        if (tr->sync_used == tr->sync_pool.size()) { hipEvent_t e; if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { err = "event create failed"; return false; } tr->sync_pool.push_back(e); }
        hipEvent_t e = tr->sync_pool[tr->sync_used++];
        if (hipEventRecord(e, tr->side) != hipSuccess || hipStreamWaitEvent(st, e, 0) != hipSuccess) { err = "stream join failed"; return false; }
        return true;
    }
    void* take(size_t bytes) { off = align_up(off, 256); void* p = base ? base + off : nullptr; off += bytes; return p; }
    TT new_tensor(int C, int h, int w) { TT t; t.C = C; t.H = h; t.W = w; t.p = take((size_t)B * h * w * C * tr->elem); return t; }
    int groups_for(int C) const { return C < tr->G ? C : tr->G; }
    const float* par(int i) const { return P + tr->params[i].off; }
    float* grad(int i) const { return Gd + tr->params[i].off; }
    bool ok(hipError_t e, const char* what) { if (e != hipSuccess) { err = std::string(what) + ": " + hipGetErrorString(e); return false; } return true; }

    void place_scratch(const ShapeInfo& si)
    {
        scr_wg = (float*)take(si.scr_wg); scr_gn = (float2*)take(si.scr_gn); scr_film = (float2*)take(si.scr_film);
        scr_col = (float*)take(si.scr_col);
    }

    // ---- forward pieces --------------------------------------------------------------------------------------------------
    // generic launch of the forward conv kernels.  `kind`: kernel family; N = output channels of this launch, K = input channels
    // input GroupNorm formed inside the persistent kernel from the producer's partial sums (its GS = 1; the inference plan's in_kernel_stats)
    struct GsIn { const float2* part; int n_sp, n_nt, bn, cpg; const float* gamma; const float* beta; double inv_count; };
    bool run_conv(int fam, int kind, bool four, const PackDesc& plain, const PackDesc& frag, int BN, int K, int Kpad, int N, int Npad, const float* bias,
                  const void* in, int Hin, int Win, TT* out, void* out_p, const float2* gn_ab, const float* film, const void* res, bool want_part, float* eps_out,
                  const GsIn* gs = nullptr, bool* gs_used = nullptr)
    {
        const Geom g = geom_of(kind, Hin, Win, four);
        const int cke = tr->elem == 2 ? 64 : 32;
        const int n_nt = Npad / BN;
        int th = conv_tile_rows(kind, BN, B, g.MH, g.MW, g.npar, n_nt);
        // the persistent register-weight kernel (ccn_conv_pr.hip) where the inference plan would use it: bf16, 3x3 s1 on 8- or 4-row tiles;
        // stride-2 conv (plane passes) and ConvTranspose (parities) on 8-row tiles as soon as half the CUs get one
        static const bool no_pr = diag_env("CCN_TRAIN_NO_PR") != nullptr;     // A/B switches
        static const bool no_pr2 = diag_env("CCN_TRAIN_NO_PR_S2CT") != nullptr;
        static const long pr2_min = diag_env("CCN_TRAIN_PR2_MIN") ? atol(diag_env("CCN_TRAIN_PR2_MIN")) : 64;
        bool pr = false;
        if (!no_pr && frag.dst && Kpad / cke >= 2 && !(res && film) && (double)B * g.Hout * g.Wout * N * tr->elem < 2.0e9) {
            if (kind == KIND_C3S1) pr = conv_pr_selected(tr->cfg.dtype, kind, BN, th);
            else if (!no_pr2 && !gn_ab && (kind == KIND_C3S2 || (kind == KIND_CT4 && !four)) && conv_pr_selected(tr->cfg.dtype, kind, BN, 8) &&
                     (long)B * ceil_div(g.MH, 8) * ceil_div(g.MW, 32) * g.npar * n_nt >= pr2_min) { pr = true; th = 8; }
        }
        static const bool no_stem2 = diag_env("CCN_TRAIN_NO_STEM2") != nullptr;
        const bool stem2 = !no_stem2 && kind == KIND_STEM && frag.dst && stem2_supported(tr->cfg.dtype, K, N, tr->G) && !gn_ab && !film && !res;
        if (!base) pack_list.push_back((pr || stem2) ? frag : plain);      // measuring walk: this shape's repack list
        ConvArgs a{};
        a.in = in; a.w = plain.dst; a.wfrag = pr ? frag.dst : nullptr; a.use_pr = pr ? 1 : 0; a.bias = bias; a.out = out_p;
        a.gn_ab = gn_ab; a.film = film; a.res = res;
        if (gs_used) *gs_used = false;
        if (gs && pr && kind == KIND_C3S1) {                    // (no split-K in the training step)
            a.gn_ab = nullptr;
            a.gs_part = gs->part; a.gs_gamma = gs->gamma; a.gs_beta = gs->beta; a.gs_inv_count = gs->inv_count;
            a.gs_nsp = gs->n_sp; a.gs_nnt = gs->n_nt; a.gs_bn = gs->bn; a.gs_cpg = gs->cpg;
            if (gs_used) *gs_used = true;
        }
        a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = K; a.Cin_pad = Kpad;
        a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = N; a.Cout_pad = Npad;
        a.MH = g.MH; a.MW = g.MW; a.OS = g.OS; a.npar = g.npar; a.ntaps = g.ntaps;
        a.th = th; a.n_ty = ceil_div(g.MH, th); a.n_tx = ceil_div(g.MW, 32); a.n_nt = n_nt;
        a.nchunk = Kpad / cke;
        if (pr && kind == KIND_C3S2) {                                     // plane passes (ccn_conv_pr.hip)
            if (four) a.nchunk *= 4;                                       // 4x4: four passes of 2x2 taps (a.ntaps stays 16)
            else { a.nchunk *= 5; a.ntaps = 2; }                           // 3x3: five passes of two tap slots
        }
        a.silu = 1; a.ksplit = 1;
        a.G = groups_for(N); a.cpg = N / a.G;
        a.nslot = a.n_ty * a.n_tx * g.npar * n_nt;                        // (the persistent kernel too: one partial per tile)
        if (stem2) { a.use_stem2 = 1; a.wfrag = frag.dst; a.nslot = 4 * stem2_blocks(Hin, Win, nullptr); }   // one slot per wave
        a.film_bstride = tr->F;
        a.bn = BN;
        a.fin_blocks = a.n_ty * a.n_tx * g.npar * n_nt;
        a.eps_out = eps_out;
        // (measured and dropped: capping the data-gradient convs' persistent grid to the CUs the side stream's weight-gradient kernel
        // leaves free -- 192 / 160 / 128 workgroups: 652-658 vs 663 images/s uncapped; docs/EXPERIMENTS.md R3.12)
        fill_taps(a.tapinfo, kind, four);
        if (want_part && out) {
            out->part = (float2*)take((size_t)B * a.G * a.nslot * sizeof(float2));
            out->n_sp = a.n_ty * a.n_tx * g.npar; out->n_nt = n_nt; out->bn = BN;
            if (stem2) { out->n_sp = a.nslot; out->n_nt = 1; out->bn = 1 << 30; }
            a.part = out->part;
        }
        if (!launch) return true;
        mark(fam, 2.0 * B * g.Hout * g.Wout * (double)N * (kind == KIND_CT4 ? 4 : (kind == KIND_STEM ? 1 : g.ntaps)) * (kind == KIND_STEM ? 9.0 * K : (double)K));
        return ok(launch_conv(tr->cfg.dtype, kind, BN, a, st), "conv");
    }
    bool conv_fwd(const TConvW& w, const TT& in, TT& out, const float2* gn_ab, const float* film, const TT* res, bool want_part, const void* in_override = nullptr,
                  float* eps_out = nullptr, const GsIn* gs = nullptr, bool* gs_used = nullptr)
    {
        return run_conv(TF_CONV_FWD, w.kind, false, w.pd_f, w.pf_f, w.BN, w.Cin, w.Cin_pad, w.Cout, w.Cout_pad, launch ? par(w.pb) : nullptr, in_override ? in_override : in.p, in.H, in.W,
                        &out, out.p, gn_ab, film, res ? res->p : nullptr, want_part, eps_out, gs, gs_used);
    }
    // ---- forward ResBlock conv with its GroupNorm + SiLU, side-stream form (bf16, plain stream launches) --------------------------------
    // The conv transforms its raw input in its producer waves (the inference kernels' prologue: statistics formed in-kernel from the
    // partial sums where they are few, else from the table of a 5-us statistics launch), so the pass that WRITES the activated tensor
    // -- which only the weight-gradient kernel reads, in the backward pass -- leaves the forward's critical path: it runs on the side
    // stream beside the convs (HBM-bound next to MFMA-bound).  The backward pass waits for pack_dg_done, recorded after the last of them.
    bool fwd_side_used = false;
    bool side_pre_ok() const
    {
        static const bool off_ = diag_env("CCN_TRAIN_NO_SIDE_PRE") != nullptr;       // A/B switch
        return launch && !off_ && side_active() && !tr->use_graph && tr->pack_dg_done && tr->elem == 2;
    }
    bool in_kernel_stats_ok(const TT& t, int C) const
    {
        const int G = groups_for(C), cpg = C / G;
        if (tr->elem != 2 || t.n_sp <= 0 || G != 8 || (cpg % 8) != 0) return false;
        int nj = 1;
        for (int g = 0; g < G; ++g) { const int n = ((g + 1) * cpg - 1) / t.bn - (g * cpg) / t.bn + 1; if (n > nj) nj = n; }
        return (long)t.n_sp * nj <= 64;
    }
    // (allocation order of the pre-pass form, which the never-launching measuring / replay walks go through: activated tensor -- by the
    // caller --, tables, conv output, the output's partial sums)
    bool norm_conv_fwd_side(const TT& x, const TNorm& n, const TConvW& w, float2*& ab, float2*& stats, const TT& xa, TT& y, const float* film_r, const TT* res)
    {
        const int G = groups_for(x.C), cpg = x.C / G;
        ab = (float2*)take((size_t)B * x.C * sizeof(float2));
        stats = (float2*)take((size_t)B * G * sizeof(float2));
        y = new_tensor(w.Cout, x.H, x.W);
        const double count = (double)cpg * x.H * x.W;
        auto fork = [&]() -> bool {
            hipEvent_t e = sync_event();
            if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(tr->side, e, 0) != hipSuccess) { err = "stream fork failed"; return false; }
            fwd_side_used = true;
            return true;
        };
        const int dt = tr->cfg.dtype;
        // (the second condition is run_conv's own size limit for the persistent kernel: past it the conv takes the table route below)
        if (launch && in_kernel_stats_ok(x, x.C) && (double)B * x.H * x.W * w.Cout * tr->elem < 2.0e9) {
            // statistics in the conv itself; the side stream's pass writes the activated tensor AND the tables the backward pass reads
            const GsIn gs{x.part, x.n_sp, x.n_nt, x.bn, cpg, par(n.pg), par(n.pb), 1.0 / count};
            bool used = false;
            if (!fork()) return false;
            if (!ok(launch_gn_act_fused(dt, x.p, xa.p, B, x.H * x.W, x.C, x.part, G, x.n_sp, x.n_nt, x.bn, cpg, count, par(n.pg), par(n.pb), 1e-5f, tr->side, ab, stats),
                    "gn_act_fused")) return false;
            if (!conv_fwd(w, x, y, nullptr, film_r, res, true, nullptr, nullptr, &gs, &used)) return false;
            if (used) return true;
            // (the conv did not land on the persistent kernel: it ran WITHOUT its GroupNorm -- cannot happen for the shapes side_pre_ok()
            // admits, checked here rather than assumed)
            err = "internal: in-kernel statistics requested for a conv outside the persistent kernel";
            return false;
        }
        if (launch) {
            if (!ok(launch_gn_stats(x.part, B, G, x.n_sp, x.n_nt, x.bn, cpg, x.C, count, par(n.pg), par(n.pb), 1e-5f, ab, stats, st), "gn_stats")) return false;
            if (!fork()) return false;
            if (!ok(launch_gn_act(dt, x.p, ab, xa.p, B, x.H * x.W, x.C, tr->side), "gn_act")) return false;
        }
        return conv_fwd(w, x, y, ab, film_r, res, true);
    }
    // dX = conv'(dY): N = the forward conv's Cin
    bool conv_dgrad(const TConvW& w, const void* dy, int Hdy, int Wdy, void* dx, const void* res)
    {
        g_sum_rows = 0;                                         // the gradient tensor that follows comes out of a conv: no channel sums
        return run_conv(TF_CONV_DGRAD, w.dkind, w.kind == KIND_CT4, w.pd_d, w.pf_d, w.dBN, w.kind == KIND_HEAD ? tr->cfg.img_ch : w.Cout, w.dCin_pad, w.Cin, w.dCout_pad, tr->zero_bias, dy, Hdy, Wdy,
                        nullptr, dx, nullptr, nullptr, res, false, nullptr);
    }
    bool preact(const TT& t, const float2* ab, const TT& out)
    {
        if (!launch) return true;
        mark(TF_PREACT, 0.0, 2.0 * B * t.H * t.W * t.C * tr->elem);
        return ok(launch_gn_act(tr->cfg.dtype, t.p, ab, out.p, B, t.H * t.W, t.C, st), "gn_act");
    }
    // GroupNorm statistics AND the pre-activated tensor SiLU(GN(t)) in one launch (round 3: 29 gn_stats launches of a step gone):
    // every workgroup of the pass reduces the producer's partial sums itself, the first one of each sample also writes the scale /
    // shift table and (mean, rstd) for the backward pass
    bool gn_fwd_preact(const TT& t, const TNorm& n, float2*& ab, float2*& stats, const TT& out)
    {
        const int G = groups_for(t.C), cpg = t.C / G;
        ab = (float2*)take((size_t)B * t.C * sizeof(float2));
        stats = (float2*)take((size_t)B * G * sizeof(float2));
        if (!launch) return true;
        // every workgroup redoes the reduction of G x slots partial sums: worth it while that is a few KB (below the 256-pixel level);
        // at 256 px (256+ slots per group, 1024 workgroups per sample) the separate 5-us statistics launch stays
        if ((long)t.n_sp * t.n_nt > 128) {
            mark(TF_GN_STATS);
            if (!ok(launch_gn_stats(t.part, B, G, t.n_sp, t.n_nt, t.bn, cpg, t.C, (double)cpg * t.H * t.W, par(n.pg), par(n.pb), 1e-5f, ab, stats, st), "gn_stats")) return false;
            return preact(t, ab, out);
        }
        mark(TF_PREACT, 0.0, 2.0 * B * t.H * t.W * t.C * tr->elem);
        return ok(launch_gn_act_fused(tr->cfg.dtype, t.p, out.p, B, t.H * t.W, t.C, t.part, G, t.n_sp, t.n_nt, t.bn, cpg, (double)cpg * t.H * t.W,
                                      par(n.pg), par(n.pb), 1e-5f, st, ab, stats), "gn_act_fused");
    }
    bool gn_fwd(const TT& t, const TNorm& n, float2*& ab, float2*& stats)
    {
        const int G = groups_for(t.C), cpg = t.C / G;
        ab = (float2*)take((size_t)B * t.C * sizeof(float2));
        stats = (float2*)take((size_t)B * G * sizeof(float2));
        if (!launch) return true;
        mark(TF_GN_STATS);
        return ok(launch_gn_stats(t.part, B, G, t.n_sp, t.n_nt, t.bn, cpg, t.C, (double)cpg * t.H * t.W, par(n.pg), par(n.pb), 1e-5f, ab, stats, st), "gn_stats");
    }

    bool conditioning(const float* z, const int64_t* t)
    {
        const ccn_config_t& c = tr->cfg;
        const int td = c.time_dim;
        temb = (float*)take((size_t)B * td * 4); u0 = (float*)take((size_t)B * td * 16); t1 = (float*)take((size_t)B * td * 16);
        tp = (float*)take((size_t)B * td * 4); uz = (float*)take((size_t)B * td * 4); zpv = (float*)take((size_t)B * td * 4);
        hv = (float*)take((size_t)B * td * 4); film = (float*)take((size_t)B * tr->F * 4);
        if (!launch) return true;
        mark(TF_COND);
        if (!ok(launch_temb_i64(t, temb, B, td, st), "temb")) return false;
        if (!ok(launch_tlinear_fwd(temb, td, par(tr->tp0.pw), par(tr->tp0.pb), t1, 4 * td, u0, B, td, 4 * td, 1, st), "time_proj.0")) return false;
        if (!ok(launch_tlinear_fwd(t1, 4 * td, par(tr->tp2.pw), par(tr->tp2.pb), tp, td, nullptr, B, 4 * td, td, 0, st), "time_proj.2")) return false;
        if (!ok(launch_tlinear_fwd(z, c.z_dim, par(tr->zp.pw), par(tr->zp.pb), zpv, td, uz, B, c.z_dim, td, 1, st), "z_proj")) return false;
        if (!ok(launch_add2(hv, tp, zpv, (int64_t)B * td, st), "h")) return false;
        return ok(launch_film_group_fwd(P, tr->lin_descs, tr->n_lin, tr->max_lin_n, hv, film, B, td, tr->F, st), "film");
    }

    bool forward(const float* x_t, const float* z, const int64_t* t, float* eps)
    {
        if (launch && eps) tr->sync_used = 0;                    // (fork events of this call; a wait keeps the record it saw when it was enqueued)
        if (launch) {
            mark(TF_PACK);
            // The forward convs' operands now, on this stream; the data-gradient convs' operands (the other half of the 0.18 ms repack)
            // on the side stream beside the forward pass -- the backward pass waits for them (pack_dg_done).  One launch under a graph
            // capture (the side branch would have to rejoin inside the forward's graph) and while profiling.
            static const bool no_split = diag_env("CCN_TRAIN_NO_PACK_SPLIT") != nullptr;     // A/B switch
            const bool split = !no_split && side_active() && tr->pack_dg_done && !tr->use_graph && n_shape_packs_fwd > 0 && n_shape_packs_fwd < n_shape_packs;
            if (!ok(launch_pack_group(tr->cfg.dtype, P, shape_packs, split ? n_shape_packs_fwd : n_shape_packs, st), "pack")) return false;
            tr->pack_dg_pending = false;
            if (split) {
                hipEvent_t e = sync_event_fwd();
                if (!e || hipEventRecord(e, st) != hipSuccess || hipStreamWaitEvent(tr->side, e, 0) != hipSuccess) { err = "stream fork failed"; return false; }
                if (!ok(launch_pack_group(tr->cfg.dtype, P, shape_packs + n_shape_packs_fwd, n_shape_packs - n_shape_packs_fwd, tr->side), "pack")) return false;
                if (hipEventRecord(tr->pack_dg_done, tr->side) != hipSuccess) { err = "event record failed"; return false; }
                tr->pack_dg_pending = true;
            }
        }
        if (!conditioning(z, t)) return false;
        TT x; std::vector<TT> skips;
        TT img; img.C = tr->cfg.img_ch; img.H = H; img.W = W;
        for (const Layer& L : tr->layers) {
            switch (L.type) {
                case L_STEM: {
                    x = new_tensor(tr->stem.Cout, H, W);
                    if (!conv_fwd(tr->stem, img, x, nullptr, nullptr, nullptr, true, x_t)) return false;
                    stem_out = x;
                    break;
                }
                case L_RES: {
                    const TRes& r = tr->res[L.idx];
                    // The activated tensors SiLU(GN(.)) are written once by a pre-pass and kept: the forward conv AND the weight-gradient
                    // kernel then stage plain copies.  Redoing the transform while staging costs both of them VALU issue slots next to
                    // their MFMAs (per pixel tile the weight-gradient kernel spent more cycles in exp/rcp than in MFMAs); the pre-pass is
                    // one HBM-bound read + write per norm.  Layers the pre-pass kernel cannot take keep the fused prologue.
                    static const bool no_pre = diag_env("CCN_TRAIN_NO_PREACT") != nullptr;       // A/B switch
                    ResSave s; s.x = x;
                    s.pre = !no_pre && r.C / (tr->elem == 2 ? 8 : 4) <= 256;
                    // (side form: same allocation sequence as the pre-pass form -- activated tensor, tables, conv output -- so the
                    // measuring and replay walks, which never launch, need not know which one ran)
                    const bool sidef = s.pre && side_pre_ok() && conv_pr_selected(tr->cfg.dtype, KIND_C3S1, r.c1.BN, 8) && r.c1.pf_f.dst && r.c1.Cin_pad / 64 >= 2;
                    if (sidef) {
                        s.xa = new_tensor(r.C, x.H, x.W);
                        if (!norm_conv_fwd_side(x, r.n1, r.c1, s.ab1, s.st1, s.xa, s.y, film ? film + r.film_off : nullptr, nullptr)) return false;
                        s.ya = new_tensor(r.C, x.H, x.W);
                        if (!norm_conv_fwd_side(s.y, r.n2, r.c2, s.ab2, s.st2, s.ya, s.o, nullptr, &x)) return false;
                    } else {
                    if (s.pre) { s.xa = new_tensor(r.C, x.H, x.W); if (!gn_fwd_preact(x, r.n1, s.ab1, s.st1, s.xa)) return false; }
                    else if (!gn_fwd(x, r.n1, s.ab1, s.st1)) return false;
                    s.y = new_tensor(r.C, x.H, x.W);
                    if (!conv_fwd(r.c1, s.pre ? s.xa : x, s.y, s.pre ? nullptr : s.ab1, film ? film + r.film_off : nullptr, nullptr, true)) return false;
                    if (s.pre) { s.ya = new_tensor(r.C, x.H, x.W); if (!gn_fwd_preact(s.y, r.n2, s.ab2, s.st2, s.ya)) return false; }
                    else if (!gn_fwd(s.y, r.n2, s.ab2, s.st2)) return false;
                    s.o = new_tensor(r.C, x.H, x.W);
                    if (!conv_fwd(r.c2, s.pre ? s.ya : s.y, s.o, s.pre ? nullptr : s.ab2, nullptr, &x, true)) return false;
                    }
                    rs.push_back(s);
                    x = s.o;
                    break;
                }
                case L_DOWN: {
                    const TConvW& w = tr->downs[L.idx];
                    skips.push_back(x);
                    TT o = new_tensor(w.Cout, x.H / 2, x.W / 2);
                    if (!conv_fwd(w, x, o, nullptr, nullptr, nullptr, true)) return false;
                    down_in.push_back(x); down_out.push_back(o);
                    x = o;
                    break;
                }
                case L_UP: {
                    const TConvW& w = tr->ups[L.idx];
                    TT o = new_tensor(w.Cout, x.H * 2, x.W * 2);
                    TT sk = skips.back(); skips.pop_back();
                    if (!conv_fwd(w, x, o, nullptr, nullptr, &sk, true)) return false;
                    up_in.push_back(x); up_out.push_back(o);
                    x = o;
                    break;
                }
                case L_HEAD: {
                    head_in = x;
                    if (!gn_fwd(x, tr->out_norm, ab_o, st_o)) return false;
                    // bf16: the inference plan's head kernels (out_norm folded into per-sample weights, nine taps in the N dimension) instead
                    // of the generic kernel: 20 instead of 45 us.  (Same allocation in every walk: the scratch is taken whenever the kernel
                    // would be chosen, which depends on the configuration only.)
                    static const bool no_head2 = diag_env("CCN_TRAIN_NO_HEAD2") != nullptr;      // A/B switch
                    if (!no_head2 && head2_supported(tr->cfg.dtype, tr->head.Cin, tr->head.Cout, tr->G) && (double)B * H * W * x.C * 2 < 2.0e9) {
                        void* scratch = take(head2_scratch_bytes(B, x.C));
                        if (!launch) break;
                        ConvArgs a{};
                        a.in = x.p; a.bias = par(tr->head.pb); a.B = B; a.Hin = H; a.Win = W; a.Cin = tr->head.Cin; a.Cout = tr->head.Cout;
                        a.Hout = H; a.Wout = W; a.eps_out = eps;
                        mark(TF_CONV_FWD, 2.0 * B * H * W * (double)tr->head.Cout * 9.0 * tr->head.Cin);
                        if (!ok(launch_head2(a, ab_o, par(tr->head.pw), scratch, 0, st), "head")) return false;
                        break;
                    }
                    TT none;
                    if (!run_conv(TF_CONV_FWD, KIND_HEAD, false, tr->head.pd_f, tr->head.pf_f, tr->head.BN, tr->head.Cin, tr->head.Cin_pad, tr->head.Cout, tr->head.Cout_pad,
                                  launch ? par(tr->head.pb) : nullptr, x.p, x.H, x.W, &none, nullptr, ab_o, nullptr, nullptr, false, eps)) return false;
                    break;
                }
            }
        }
        if (launch && (fwd_side_used || tr->pack_dg_pending)) {
            // Everything this forward put on the side stream -- the data-gradient operands' repack (reads the parameters), the activated-
            // tensor passes (write into the workspace) -- is joined HERE, on the caller's stream: the side work ended long before the
            // last convs of the main chain do, so the wait is free, and whatever the caller does next with the parameters or the
            // workspace (an optimizer step without a backward pass, freeing the workspace) is ordered behind it.
            if (hipEventRecord(tr->pack_dg_done, tr->side) != hipSuccess || hipStreamWaitEvent(st, tr->pack_dg_done, 0) != hipSuccess) { err = "stream join failed"; return false; }
            tr->pack_dg_pending = false;
        }
        return true;
    }

    // ---- backward pieces ---------------------------------------------------------------------------------------------------
    void want(size_t& slot, size_t bytes) { if (bytes > slot) slot = bytes; }

    // dW (and db) of a conv: A = act(GN(x)) redone on the fly, dY given
    bool conv_wgrad(const TConvW& w, const TT& xin, const float2* gn_ab, const void* dy, int Hdy, int Wdy, bool bias_done = false)
    {
        if (!wgrad_generic(w.kind, xin.p, xin.H, xin.W, w.Cin, w.Cin, gn_ab, 1, dy, w.Cout, w.Cout, grad_or_null(w.pw))) return false;
        const GnBwdGeom gg = gn_bwd_geom(tr->cfg.dtype, B, Hdy * Wdy, w.Cout);
        want(need.scr_col, (size_t)B * gg.nblk * w.Cout * 4);
        if (!launch || bias_done) return true;
        mark(TF_BIAS);
        if (g_sum_rows > 0) return ok(launch_colsum_from_pairs(scr_film, g_sum_rows, w.Cout, grad(w.pb), st), "bias_grad");
        return ok(launch_colsum(tr->cfg.dtype, dy, scr_col, grad(w.pb), B, Hdy * Wdy, w.Cout, st), "bias_grad");
    }
    float* grad_or_null(int i) const { return Gd ? grad(i) : nullptr; }
    // kind: the forward conv's family (KIND_STEM = 1x1 on an im2col'ed input); Cin / Cout as stored (multiples of 8), Civ / Cov the
    // channels that exist in the parameter
    bool wgrad_generic(int kind, const void* x, int Hin, int Win, int Cin, int Civ, const float2* gn_ab, int silu, const void* dy, int Cout, int Cov, float* gdst)
    {
        const Geom g = geom_of(kind, Hin, Win, false);
        WgArgs a{};
        a.x = x; a.gn_ab = gn_ab; a.silu = silu; a.dy = dy; a.part = scr_wg;
        a.B = B; a.Hin = Hin; a.Win = Win; a.Cin = Cin; a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = Cout;
        a.MH = g.MH; a.MW = g.MW; a.OS = g.OS; a.npar = g.npar; a.ntaps = g.ntaps; a.taps_w = kind == KIND_CT4 ? 16 : (kind == KIND_STEM ? 1 : 9);
        a.n_ty = ceil_div(g.MH, 4); a.n_tx = ceil_div(g.MW, 32);
        const int ns_alone = wgrad_nsplit(tr->cfg.dtype, kind, B, g.MH, g.MW, Cin, Cout, false), ns_conc = wgrad_nsplit(tr->cfg.dtype, kind, B, g.MH, g.MW, Cin, Cout, true);
        a.nsplit = side_active() ? ns_conc : ns_alone;
        fill_taps(a.tapinfo, kind, false);
        want(need.scr_wg, (size_t)(ns_alone > ns_conc ? ns_alone : ns_conc) * a.taps_w * Cout * Cin * 4);
        if (!launch) return true;
        if (!fork_side()) return false;
        mark(TF_WGRAD, 2.0 * B * g.Hout * g.Wout * (double)Cov * (kind == KIND_CT4 ? 4 : (kind == KIND_STEM ? 1 : 9)) * Civ);
        if (!ok(launch_wgrad(tr->cfg.dtype, kind, a, wg_stream), "wgrad")) return false;
        mark(TF_WGRAD_REDUCE);
        return ok(launch_wgrad_reduce(scr_wg, a.nsplit, a.taps_w, Cout, Cin, Cov, Civ, kind == KIND_CT4 ? 1 : 0, gdst, wg_stream), "wgrad_reduce");
    }
    // GroupNorm(+SiLU) backward of the norm reading tensor `x`: dA -> out (may alias dA)
    bool gn_bwd(const TT& x, const TNorm& n, const float2* ab, const float2* stats, const void* dA, void* out, bool silu, const void* addend,
                int film_off, float* film_bias = nullptr)
    {
        const int G = groups_for(x.C), cpg = x.C / G, HW = x.H * x.W;
        const GnBwdGeom gg = gn_bwd_geom(tr->cfg.dtype, B, HW, x.C);
        want(need.scr_gn, (size_t)B * gg.nblk * x.C * sizeof(float2));
        want(need.scr_film, (size_t)B * gg.nblk * x.C * sizeof(float2));
        const float* film_r = film_off >= 0 && film ? film + film_off : nullptr;
        float* dfilm_r = film_off >= 0 && dfilm ? dfilm + film_off : nullptr;
        g_sum_rows = film_off >= 0 ? 0 : B * gg.nblk;
        float2* gstat = (float2*)take((size_t)B * G * sizeof(float2));
        if (!launch) return true;
        const int dt = tr->cfg.dtype;
        // algorithmic HBM bytes: pass 1 reads x and dA, pass 2 reads them again (+ the residual gradient) and writes dx
        mark(TF_GN_BWD, 0.0, (double)B * HW * x.C * tr->elem * (addend ? 6.0 : 5.0));
        // (round 3: the three passes as ONE launch with one 1024-thread workgroup per (sample, group) for the levels below 256 px was
        // built, parity-green, and 0.95 ms SLOWER per step (2.47 vs 1.53 ms for the family): a group's channels are 32-128 contiguous
        // bytes per pixel, so 32 workgroups pull uncoalesced 32-byte segments at a few tens of GB/s each; docs/EXPERIMENTS.md R3.5)
        if (!ok(launch_gn_bwd_reduce(dt, x.p, dA, ab, stats, scr_gn, B, HW, x.C, cpg, G, silu ? 1 : 0, st), "gn_bwd_reduce")) return false;
        if (!ok(launch_gn_bwd_finalize(scr_gn, gg.nblk, B, x.C, cpg, G, (double)cpg * HW, par(n.pg), gstat, grad(n.pg), grad(n.pb), st), "gn_bwd_finalize")) return false;
        if (!ok(launch_gn_bwd_apply(dt, x.p, dA, ab, stats, gstat, addend, out, film_r, tr->F, scr_film, B, HW, x.C, cpg, G, silu ? 1 : 0, st), "gn_bwd_apply"))
            return false;
        if (film_r) return ok(launch_film_bwd_finalize(scr_film, gg.nblk, film_r, tr->F, dfilm_r, film_bias, B, x.C, st), "film_bwd");
        return true;
    }
    bool flush_bucket(long long lo, bool force)
    {
        if (!bucket_cb || !launch) return true;
        if (!force && hi_pending - lo < bucket_floats) return true;
        if (lo >= hi_pending) return true;
        if (!join_side()) return false;
        bucket_cb(bucket_user, lo, hi_pending);
        hi_pending = lo;
        return true;
    }
    bool lin_bwd(const TLin& l, const float* dy, int lddy, const float* x, int ldx, float* dx, int lddx, int accumulate)
    {
        if (!launch) return true;
        if (!ok(launch_tlinear_dw(dy, lddy, x, ldx, grad(l.pw), grad(l.pb), B, l.K, l.N, st), "linear_dw")) return false;
        if (dx) return ok(launch_tlinear_dx(dy, lddy, par(l.pw), dx, lddx, B, l.K, l.N, accumulate, st), "linear_dx");
        return true;
    }

    bool backward(const float* x_t, const float* z, const float* d_eps)
    {
        const ccn_config_t& c = tr->cfg;
        const int td = c.time_dim, dt = c.dtype;
        dfilm = (float*)take((size_t)B * tr->F * 4); dh = (float*)take((size_t)B * td * 4);
        dt1 = (float*)take((size_t)B * td * 16); du0 = (float*)take((size_t)B * td * 16); duz = (float*)take((size_t)B * td * 4);
        hi_pending = (long long)tr->total;
        if (launch && tr->pack_dg_pending) {                   // the data-gradient operands repacked beside the forward pass
            if (hipStreamWaitEvent(st, tr->pack_dg_done, 0) != hipSuccess) { err = "stream wait failed"; return false; }
            tr->pack_dg_pending = false;
        }
        if (launch && bucket_cb && hipMemsetAsync(dh, 0, (size_t)B * td * 4, st) != hipSuccess) { err = "hipMemsetAsync failed"; return false; }
        TT g;                                                  // gradient w.r.t. the current tensor
        std::vector<void*> dskip;                              // gradients waiting at the skip connections (pushed by the up path)
        int ri = (int)rs.size() - 1, di = (int)down_in.size() - 1, ui = (int)up_in.size() - 1;
        for (int li = (int)tr->layers.size() - 1; li >= 0; --li) {
            const Layer& L = tr->layers[li];
            switch (L.type) {
                case L_HEAD: {
                    const TConvW& w = tr->head;
                    const TT& u = head_in;
                    // out.weight: the generic kernel on d eps as an NHWC tensor (channels padded to 8), A = out_norm(u) without SiLU
                    void* de = take((size_t)B * H * W * 8 * tr->elem);
                    if (launch) {
                        mark(TF_SMALL);
                        if (!ok(launch_nchw_to_nhwc_pad(dt, d_eps, de, B, c.img_ch, 8, (int64_t)H * W, st), "head_deps_layout")) return false;
                        if (!ok(launch_nchw_chansum(d_eps, grad(w.pb), B, c.img_ch, (int64_t)H * W, st), "head_bias")) return false;
                    }
                    if (!wgrad_generic(KIND_C3S1, u.p, H, W, u.C, u.C, ab_o, 0, de, 8, c.img_ch, grad_or_null(w.pw))) return false;
                    g = new_tensor(u.C, H, W);
                    if (!conv_dgrad(w, d_eps, H, W, g.p, nullptr)) return false;
                    if (!gn_bwd(u, tr->out_norm, ab_o, st_o, g.p, g.p, false, nullptr, -1)) return false;
                    break;
                }
                case L_UP: {
                    const TConvW& w = tr->ups[L.idx];
                    const TT& xin = up_in[ui]; --ui;
                    dskip.push_back(g.p);                       // x = convT(xin) + skip
                    if (!conv_wgrad(w, xin, nullptr, g.p, g.H, g.W)) return false;
                    TT d = new_tensor(xin.C, xin.H, xin.W);
                    if (!conv_dgrad(w, g.p, g.H, g.W, d.p, nullptr)) return false;
                    g = d;
                    break;
                }
                case L_DOWN: {
                    const TConvW& w = tr->downs[L.idx];
                    const TT& xin = down_in[di]; --di;
                    if (!conv_wgrad(w, xin, nullptr, g.p, g.H, g.W)) return false;
                    TT d = new_tensor(xin.C, xin.H, xin.W);
                    const void* sk = dskip.back(); dskip.pop_back();
                    if (!conv_dgrad(w, g.p, g.H, g.W, d.p, sk)) return false;
                    g = d;
                    break;
                }
                case L_RES: {
                    const TRes& r = tr->res[L.idx];
                    const ResSave& s = rs[ri]; --ri;
                    // out = x + conv2(A2), A2 = silu(gn2(F)), F = film(conv1(A1)), A1 = silu(gn1(x))
                    if (!conv_wgrad(r.c2, s.pre ? s.ya : s.y, s.pre ? nullptr : s.ab2, g.p, g.H, g.W)) return false;
                    TT g1 = new_tensor(r.C, g.H, g.W);
                    if (!conv_dgrad(r.c2, g.p, g.H, g.W, g1.p, nullptr)) return false;
                    if (!gn_bwd(s.y, r.n2, s.ab2, s.st2, g1.p, g1.p, true, nullptr, r.film_off, launch ? grad(r.c1.pb) : nullptr)) return false;
                    if (launch && bucket_cb) {
                        // bucketed mode: this block's FiLM linears now (side stream) instead of one grouped launch at the end, so that the
                        // block's whole parameter range is complete when its bucket is handed out
                        if (!fork_side()) return false;
                        const LinDesc* d2 = tr->lin_descs + 2 * L.idx;
                        if (!ok(launch_film_group_dw(Gd, d2, 2, r.C, dfilm, hv, B, td, tr->F, wg_stream), "film_dw")) return false;
                        if (!ok(launch_film_group_dx(P, d2, 2, dfilm, dh, B, td, tr->F, wg_stream), "film_dx")) return false;
                    }
                    if (!conv_wgrad(r.c1, s.pre ? s.xa : s.x, s.pre ? nullptr : s.ab1, g1.p, g.H, g.W, true)) return false;
                    TT g2 = new_tensor(r.C, g.H, g.W);
                    if (!conv_dgrad(r.c1, g1.p, g.H, g.W, g2.p, nullptr)) return false;
                    if (!gn_bwd(s.x, r.n1, s.ab1, s.st1, g2.p, g2.p, true, g.p, -1)) return false;
                    g = g2;
                    break;
                }
                case L_STEM: {
                    const TConvW& w = tr->stem;
                    // in_conv.weight: 1x1 weight gradient against the im2col of the image (27 of 32 columns)
                    const GnBwdGeom gg = gn_bwd_geom(dt, B, H * W, w.Cout);
                    want(need.scr_col, (size_t)B * gg.nblk * w.Cout * 4);
                    void* col = take((size_t)B * H * W * 32 * tr->elem);
                    if (launch) {
                        mark(TF_SMALL);
                        if (!ok(launch_im2col27(dt, x_t, col, B, c.img_ch, H, W, st), "stem_im2col")) return false;
                        mark(TF_BIAS);
                        if (g_sum_rows > 0) { if (!ok(launch_colsum_from_pairs(scr_film, g_sum_rows, w.Cout, grad(w.pb), st), "stem_bias")) return false; }
                        else if (!ok(launch_colsum(dt, g.p, scr_col, grad(w.pb), B, H * W, w.Cout, st), "stem_bias")) return false;
                    }
                    if (!wgrad_generic(KIND_STEM, col, H, W, 32, c.img_ch * 9, nullptr, 0, g.p, w.Cout, w.Cout, grad_or_null(w.pw))) return false;
                    break;
                }
            }
            if (bucket_cb) {
                int first = 0;                                  // first parameter of this layer: everything from its offset up is complete
                switch (L.type) {
                    case L_HEAD: first = tr->out_norm.pg; break;
                    case L_UP: first = tr->ups[L.idx].pw; break;
                    case L_DOWN: first = tr->downs[L.idx].pw; break;
                    case L_RES: first = tr->res[L.idx].n1.pg; break;
                    case L_STEM: first = tr->stem.pw; break;
                }
                if (L.type != L_STEM && !flush_bucket((long long)tr->params[first].off, false)) return false;
            }
        }
        // conditioning: film_r = h W_r^T + b_r for every block; h = time_proj(temb(t)) + z_proj(z)
        if (!launch) return true;
        mark(TF_COND);
        if (bucket_cb) {
            if (!join_side()) return false;                      // dh was accumulated block by block on the side stream
        } else {
            if (hipMemsetAsync(dh, 0, (size_t)B * td * 4, st) != hipSuccess) { err = "hipMemsetAsync failed"; return false; }
            if (!ok(launch_film_group_dw(Gd, tr->lin_descs, tr->n_lin, tr->max_lin_n, dfilm, hv, B, td, tr->F, st), "film_dw")) return false;
            if (!ok(launch_film_group_dx(P, tr->lin_descs, tr->n_lin, dfilm, dh, B, td, tr->F, st), "film_dx")) return false;
        }
        if (!lin_bwd(tr->tp2, dh, td, t1, 4 * td, dt1, 4 * td, 0)) return false;
        if (!ok(launch_silu_bwd(du0, dt1, u0, (int64_t)B * 4 * td, st), "silu_bwd")) return false;
        if (!lin_bwd(tr->tp0, du0, 4 * td, temb, td, nullptr, 0, 0)) return false;
        if (!ok(launch_silu_bwd(duz, dh, uz, (int64_t)B * td, st), "silu_bwd")) return false;
        if (!lin_bwd(tr->zp, duz, td, z, c.z_dim, nullptr, 0, 0)) return false;
        return flush_bucket(0, true);
    }
};

// the kernel choice (tests switch it between handles) decides which repacks a shape needs, so it is part of the key
std::string shape_key(int B, int H, int W)
{
    return std::to_string(B) + "x" + std::to_string(H) + "x" + std::to_string(W) + (conv_pr_selected(1, KIND_C3S1, 128, 8) ? "p" : "n");
}

int shape_info(ccn_trainer_s* tr, int B, int H, int W, ShapeInfo* out)
{
    if (B <= 0 || H <= 0 || W <= 0) return tfail(CCN_EINVAL, "B, H, W must be positive");
    const int div = 1 << tr->cfg.n_mult;
    if (H % div || W % div) return tfail(CCN_EINVAL, "H and W must be divisible by 2^len(ch_mult)");
    const std::string key = shape_key(B, H, W);
    auto it = tr->shapes.find(key);
    if (it != tr->shapes.end()) { *out = it->second; return CCN_OK; }
    Walk w(tr, B, H, W, nullptr, false, nullptr, nullptr, nullptr);
    if (!w.forward(nullptr, nullptr, nullptr, nullptr)) return tfail(CCN_EINVAL, w.err);
    const int n_packs_fwd = (int)w.pack_list.size();            // the forward convs' operands come first in the list
    if (!w.backward(nullptr, nullptr, nullptr)) return tfail(CCN_EINVAL, w.err);
    ShapeInfo si = w.need;
    si.n_packs_fwd = n_packs_fwd;
    si.tensors = align_up(w.off, 256);
    si.total = si.tensors + align_up(si.scr_wg, 256) + align_up(si.scr_gn, 256) + align_up(si.scr_film, 256) + align_up(si.scr_col, 256) +
               5 * 256;
    {
        std::string err;
        void* dev = nullptr;
        if (!alloc_dev(tr, w.pack_list.size() * sizeof(PackDesc), &dev, err)) return tfail(CCN_EHIP, err);
        if (hipMemcpy(dev, w.pack_list.data(), w.pack_list.size() * sizeof(PackDesc), hipMemcpyHostToDevice) != hipSuccess) return tfail(CCN_EHIP, "descriptor upload failed");
        si.packs = (PackDesc*)dev; si.n_packs = (int)w.pack_list.size();
    }
    tr->shapes[key] = si;
    *out = si;
    return CCN_OK;
}

// replay the cached graph for `key`, or capture `body` (which enqueues on the capture stream) and replay it
template <typename F>
int run_graphed(ccn_trainer_s* tr, const std::vector<const void*>& key, hipStream_t user, F&& body)
{
    for (auto& g : tr->graphs)
        if (g.key == key) {
            if (hipGraphLaunch(g.exec, user) != hipSuccess) return tfail(CCN_EHIP, "hipGraphLaunch failed");
            return CCN_OK;
        }
    if (!tr->cap_stream && hipStreamCreateWithFlags(&tr->cap_stream, hipStreamNonBlocking) != hipSuccess) return tfail(CCN_EHIP, "capture stream");
    if (hipStreamBeginCapture(tr->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) return tfail(CCN_EHIP, "hipStreamBeginCapture failed");
    std::string err;
    const bool good = body(tr->cap_stream, err);
    hipGraph_t graph = nullptr;
    const hipError_t ce = hipStreamEndCapture(tr->cap_stream, &graph);
    if (!good) { if (graph) (void)hipGraphDestroy(graph); return tfail(CCN_EHIP, err); }
    if (ce != hipSuccess) return tfail(CCN_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
    ccn_trainer_s::TGraph g; g.key = key; g.graph = graph;
    if (hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0) != hipSuccess) { (void)hipGraphDestroy(graph); return tfail(CCN_EHIP, "hipGraphInstantiate failed"); }
    if (tr->graphs.size() >= 8) {
        (void)hipGraphExecDestroy(tr->graphs.front().exec); (void)hipGraphDestroy(tr->graphs.front().graph);
        tr->graphs.erase(tr->graphs.begin());
    }
    tr->graphs.push_back(g);
    if (hipGraphLaunch(g.exec, user) != hipSuccess) return tfail(CCN_EHIP, "hipGraphLaunch failed");
    return CCN_OK;
}

}  // namespace

extern "C" {

int ccn_train_set_graph(ccn_trainer_t tr, int32_t on)
{
    if (!tr) return tfail(CCN_EINVAL, "null handle");
    tr->use_graph = on != 0;
    return CCN_OK;
}

int ccn_train_create(const ccn_config_t* cfg, ccn_trainer_t* out)
{
    if (!cfg || !out) return tfail(CCN_EINVAL, "null argument");
    if (cfg->n_mult <= 0 || cfg->n_mult > CCN_MAX_MULT || cfg->base <= 0 || cfg->time_dim <= 0 || cfg->z_dim <= 0 || cfg->img_ch <= 0 || cfg->img_ch > 3)
        return tfail(CCN_EINVAL, "bad config");
    if (cfg->dtype != CCN_DTYPE_F32 && cfg->dtype != CCN_DTYPE_BF16) return tfail(CCN_EINVAL, "bad dtype");
    if (cfg->base % 8) return tfail(CCN_EINVAL, "training path needs base % 8 == 0");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return tfail(CCN_EHIP, "no HIP device: libccn_hip.so has no CPU fallback");
    ccn_trainer_s* tr = new ccn_trainer_s();
    tr->cfg = *cfg;
    tr->elem = cfg->dtype == CCN_DTYPE_BF16 ? 2 : 4;
    tr->G = cfg->groups > 0 ? cfg->groups : 8;
    build_arch(tr);
    std::string err;
    bool good = true;
    good = good && setup_conv(tr, tr->stem, err) && setup_conv(tr, tr->head, err);
    for (TRes& r : tr->res) good = good && setup_conv(tr, r.c1, err) && setup_conv(tr, r.c2, err);
    for (TConvW& w : tr->downs) good = good && setup_conv(tr, w, err);
    for (TConvW& w : tr->ups) good = good && setup_conv(tr, w, err);
    int maxc = cfg->base;
    for (const TRes& r : tr->res) if (r.C > maxc) maxc = r.C;
    void* zb = nullptr;
    good = good && alloc_dev(tr, (size_t)(maxc + 256) * 4, &zb, err);
    if (good && hipMemset(zb, 0, (size_t)(maxc + 256) * 4) != hipSuccess) { good = false; err = "hipMemset failed"; }
    if (good && (conv_prepare() != hipSuccess || wgrad_prepare() != hipSuccess)) { good = false; err = "kernel attribute setup failed"; }
    if (good && hipStreamCreateWithFlags(&tr->side, hipStreamNonBlocking) != hipSuccess) { tr->side = nullptr; }
    if (good && tr->side && hipEventCreateWithFlags(&tr->pack_dg_done, hipEventDisableTiming) != hipSuccess) { tr->pack_dg_done = nullptr; }
    if (good) {
        // descriptor tables of the grouped launches (offsets into the caller's flat buffers are fixed by the architecture)
        std::vector<LinDesc> ld;
        for (const TRes& r : tr->res) {
            ld.push_back({(long long)tr->params[r.fs.pw].off, (long long)tr->params[r.fs.pb].off, r.C, r.film_off});
            ld.push_back({(long long)tr->params[r.fh.pw].off, (long long)tr->params[r.fh.pb].off, r.C, r.film_off + r.C});
            if (r.C > tr->max_lin_n) tr->max_lin_n = r.C;
        }
        void* ldd = nullptr;
        good = alloc_dev(tr, ld.size() * sizeof(LinDesc), &ldd, err);
        if (good && hipMemcpy(ldd, ld.data(), ld.size() * sizeof(LinDesc), hipMemcpyHostToDevice) != hipSuccess) { good = false; err = "descriptor upload failed"; }
        tr->lin_descs = (LinDesc*)ldd; tr->n_lin = (int)ld.size();
    }
    if (!good) { ccn_train_destroy(tr); return tfail(CCN_EHIP, err); }
    tr->zero_bias = (float*)zb;
    *out = tr;
    return CCN_OK;
}

int ccn_train_destroy(ccn_trainer_t tr)
{
    if (!tr) return CCN_OK;
    for (void* p : tr->allocs) (void)hipFree(p);
    for (hipEvent_t e : tr->ev_pool) (void)hipEventDestroy(e);
    for (auto& g : tr->graphs) { if (g.exec) (void)hipGraphExecDestroy(g.exec); if (g.graph) (void)hipGraphDestroy(g.graph); }
    if (tr->cap_stream) (void)hipStreamDestroy(tr->cap_stream);
    for (hipEvent_t e : tr->sync_pool) (void)hipEventDestroy(e);
    if (tr->side) (void)hipStreamDestroy(tr->side);
    if (tr->pack_dg_done) (void)hipEventDestroy(tr->pack_dg_done);
    if (tr->fwd_fork) (void)hipEventDestroy(tr->fwd_fork);
    delete tr;
    return CCN_OK;
}

int ccn_train_num_params(ccn_trainer_t tr, int32_t* n, int64_t* total_floats)
{
    if (!tr || !n) return tfail(CCN_EINVAL, "null argument");
    *n = (int32_t)tr->params.size();
    if (total_floats) *total_floats = (int64_t)tr->total;
    return CCN_OK;
}

int ccn_train_param_info(ccn_trainer_t tr, int32_t i, const char** name, int64_t shape[4], int32_t* ndim, int64_t* offset)
{
    if (!tr || i < 0 || i >= (int32_t)tr->params.size()) return tfail(CCN_EINVAL, "parameter index out of range");
    const PInfo& p = tr->params[i];
    if (name) *name = p.name.c_str();
    if (ndim) *ndim = (int32_t)p.shape.size();
    if (shape) for (size_t k = 0; k < p.shape.size() && k < 4; ++k) shape[k] = p.shape[k];
    if (offset) *offset = (int64_t)p.off;
    return CCN_OK;
}

int ccn_train_workspace_bytes(ccn_trainer_t tr, int32_t B, int32_t H, int32_t W, size_t* bytes)
{
    if (!tr || !bytes) return tfail(CCN_EINVAL, "null argument");
    ShapeInfo si;
    const int rc = shape_info(tr, B, H, W, &si);
    if (rc) return rc;
    *bytes = si.total;
    return CCN_OK;
}

int ccn_train_forward(ccn_trainer_t tr, const float* params_dev, const float* x_t_dev, const float* z_dev, const int64_t* t_dev, float* eps_dev,
                      int32_t B, int32_t H, int32_t W, void* workspace_dev, size_t workspace_bytes, void* stream)
{
    if (!tr || !params_dev || !x_t_dev || !z_dev || !t_dev || !eps_dev) return tfail(CCN_EINVAL, "null argument");
    ShapeInfo si;
    int rc = shape_info(tr, B, H, W, &si);
    if (rc) return rc;
    if (!workspace_dev || ((uintptr_t)workspace_dev & 255)) return tfail(CCN_EWORKSPACE, "workspace must be non-null and 256-byte aligned");
    if (workspace_bytes < si.total) return tfail(CCN_EWORKSPACE, "workspace too small: need " + std::to_string(si.total));
    auto body = [&](hipStream_t st, std::string& err) {
        Walk w(tr, B, H, W, workspace_dev, true, st, params_dev, nullptr);
        w.place_scratch(si);
        w.shape_packs = si.packs; w.n_shape_packs = si.n_packs; w.n_shape_packs_fwd = si.n_packs_fwd;
        if (!w.forward(x_t_dev, z_dev, t_dev, eps_dev)) { err = w.err; return false; }
        w.mark(-1);
        return true;
    };
    if (tr->use_graph && !tr->profiling) {
        const std::vector<const void*> key = {(const void*)1, params_dev, x_t_dev, z_dev, t_dev, eps_dev, workspace_dev, (const void*)(uintptr_t)B,
                                              (const void*)(uintptr_t)H, (const void*)(uintptr_t)W};
        rc = run_graphed(tr, key, (hipStream_t)stream, body);
        if (rc) return rc;
    } else {
        std::string err;
        if (!body((hipStream_t)stream, err)) return tfail(CCN_EHIP, err);
    }
    tr->fB = B; tr->fH = H; tr->fW = W; tr->fws = workspace_dev;
    return CCN_OK;
}

int ccn_train_backward(ccn_trainer_t tr, const float* params_dev, float* grads_dev, const float* x_t_dev, const float* z_dev, const float* d_eps_dev,
                       int32_t B, int32_t H, int32_t W, void* workspace_dev, size_t workspace_bytes, void* stream)
{
    if (!tr || !params_dev || !grads_dev || !x_t_dev || !z_dev || !d_eps_dev) return tfail(CCN_EINVAL, "null argument");
    if (tr->fB != B || tr->fH != H || tr->fW != W || tr->fws != workspace_dev)
        return tfail(CCN_ESTATE, "ccn_train_backward must follow ccn_train_forward with the same shape and workspace");
    ShapeInfo si;
    int rc = shape_info(tr, B, H, W, &si);
    if (rc) return rc;
    if (workspace_bytes < si.total) return tfail(CCN_EWORKSPACE, "workspace too small");
    auto body = [&](hipStream_t st, std::string& err) {
        Walk w(tr, B, H, W, workspace_dev, false, st, params_dev, grads_dev);
        w.place_scratch(si);
        if (!w.forward(x_t_dev, z_dev, nullptr, nullptr)) { err = w.err; return false; }
        w.launch = true;
        tr->sync_used = 0;
        if (!w.backward(x_t_dev, z_dev, d_eps_dev) || !w.join_side()) { err = w.err; return false; }
        w.mark(-1);
        return true;
    };
    if (tr->use_graph && !tr->profiling) {
        const std::vector<const void*> key = {(const void*)2, params_dev, grads_dev, x_t_dev, z_dev, d_eps_dev, workspace_dev, (const void*)(uintptr_t)B,
                                              (const void*)(uintptr_t)H, (const void*)(uintptr_t)W};
        return run_graphed(tr, key, (hipStream_t)stream, body);
    }
    std::string err;
    if (!body((hipStream_t)stream, err)) return tfail(CCN_EHIP, err);
    return CCN_OK;
}

int ccn_train_backward_bucketed(ccn_trainer_t tr, const float* params_dev, float* grads_dev, const float* x_t_dev, const float* z_dev, const float* d_eps_dev,
                                int32_t B, int32_t H, int32_t W, void* workspace_dev, size_t workspace_bytes, void* stream, int64_t bucket_floats,
                                ccn_grad_ready_cb cb, void* user)
{
    if (!tr || !params_dev || !grads_dev || !x_t_dev || !z_dev || !d_eps_dev || !cb) return tfail(CCN_EINVAL, "null argument");
    if (tr->fB != B || tr->fH != H || tr->fW != W || tr->fws != workspace_dev)
        return tfail(CCN_ESTATE, "ccn_train_backward_bucketed must follow ccn_train_forward with the same shape and workspace");
    ShapeInfo si;
    int rc = shape_info(tr, B, H, W, &si);
    if (rc) return rc;
    if (workspace_bytes < si.total) return tfail(CCN_EWORKSPACE, "workspace too small");
    Walk w(tr, B, H, W, workspace_dev, false, (hipStream_t)stream, params_dev, grads_dev);
    w.place_scratch(si);
    if (!w.forward(x_t_dev, z_dev, nullptr, nullptr)) return tfail(CCN_EHIP, w.err);
    w.launch = true;
    w.bucket_cb = cb; w.bucket_user = user; w.bucket_floats = bucket_floats > 0 ? bucket_floats : 1;
    tr->sync_used = 0;
    if (!w.backward(x_t_dev, z_dev, d_eps_dev) || !w.join_side()) return tfail(CCN_EHIP, w.err);
    return CCN_OK;
}

int ccn_train_profile_enable(ccn_trainer_t tr, int32_t on)
{
    if (!tr) return tfail(CCN_EINVAL, "null handle");
    tr->profiling = on != 0;
    tr->marks.clear(); tr->ev_used = 0;
    return CCN_OK;
}

int ccn_train_profile_read(ccn_trainer_t tr, const char** names, float* ms, int32_t* calls, double* flops, double* bytes, int32_t cap, int32_t* n)
{
    if (!tr || !names || !ms || !calls || !flops || !bytes || !n) return tfail(CCN_EINVAL, "null argument");
    if (cap < TF_COUNT) return tfail(CCN_EINVAL, "cap too small");
    if (hipDeviceSynchronize() != hipSuccess) return tfail(CCN_EHIP, "hipDeviceSynchronize failed");
    for (int f = 0; f < TF_COUNT; ++f) { names[f] = kTrainFamilies[f]; ms[f] = 0.f; calls[f] = 0; flops[f] = 0.0; bytes[f] = 0.0; }
    for (size_t i = 0; i + 1 < tr->marks.size(); ++i) {
        const Mark& m = tr->marks[i];
        if (m.fam < 0) continue;
        float t = 0.f;
        if (hipEventElapsedTime(&t, m.ev, tr->marks[i + 1].ev) != hipSuccess) return tfail(CCN_EHIP, "hipEventElapsedTime failed");
        ms[m.fam] += t; calls[m.fam] += 1; flops[m.fam] += m.flops; bytes[m.fam] += m.bytes;
    }
    *n = TF_COUNT;
    tr->marks.clear(); tr->ev_used = 0;
    return CCN_OK;
}

int ccn_mse_loss_grad(const float* eps_dev, const float* target_dev, int64_t n, float* loss_dev, float* d_eps_dev, float* scratch_dev, void* stream)
{
    if (!eps_dev || !target_dev || !loss_dev || !scratch_dev || n <= 0) return tfail(CCN_EINVAL, "bad argument");
    if (launch_mse_loss_grad(eps_dev, target_dev, n, loss_dev, d_eps_dev, scratch_dev, (hipStream_t)stream) != hipSuccess) return tfail(CCN_EHIP, "mse launch failed");
    return CCN_OK;
}

int ccn_adamw_step(float* params_dev, const float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, void* stream)
{
    if (!params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev || n <= 0 || step <= 0) return tfail(CCN_EINVAL, "bad argument");
    if (launch_adamw(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream) != hipSuccess)
        return tfail(CCN_EHIP, "adamw launch failed");
    return CCN_OK;
}

int ccn_adamw_step_zero_grad(float* params_dev, float* grads_dev, float* exp_avg_dev, float* exp_avg_sq_dev, int64_t n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int32_t step, void* stream)
{
    if (!params_dev || !grads_dev || !exp_avg_dev || !exp_avg_sq_dev || n <= 0 || step <= 0) return tfail(CCN_EINVAL, "bad argument");
    if (launch_adamw(params_dev, grads_dev, exp_avg_dev, exp_avg_sq_dev, n, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream, true) != hipSuccess)
        return tfail(CCN_EHIP, "adamw launch failed");
    return CCN_OK;
}

}  // extern "C"
