// Host side of libccn_hip.so: the C ABI of include/ccn_hip.h.
//
// A handle owns the repacked weights of one CLIPCondUNet.  ccn_forward / ccn_sample build (once per
// shape and workspace address) a *plan*: the fixed list of kernel launches of one UNet evaluation with
// every activation, GroupNorm partial-sum and scale/shift buffer placed in the caller's workspace.
// ccn_sample replays the plan `steps` times -- conditioning (timestep MLP, z projection, all FiLM
// linears) is hoisted in front of the loop for all steps, because t is batch-uniform and known ahead
// (diffusion/ddim.py:25,32) -- either launch by launch or as one captured hipGraph.
#include "../../include/ccn_hip.h"
#include "ccn_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

using namespace ccn;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            return fail(CCN_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

}  // namespace
// ccn_train.hip reports its failures through the same thread-local message
extern "C" void ccn_internal_set_error(const char* msg) { g_err = msg ? msg : ""; }
namespace {

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

uint16_t f2bf_host(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

struct ParamInfo {
    std::string name;
    std::vector<int64_t> shape;
    size_t numel() const { size_t n = 1; for (auto d : shape) n *= (size_t)d; return n; }
};

struct ConvW {
    int kind = KIND_C3S1;
    int Cin = 0, Cout = 0, Cin_pad = 0, Cout_pad = 0, BN = 0, ntaps_w = 0;
    void* w = nullptr;
    void* wfrag = nullptr;      // MFMA-fragment-ordered copy for the persistent 3x3 kernel (Cout_pad % 128 == 0), else null
    float* bias = nullptr;
    // bf16 mode, CCN_ROUND_DIFFUSED_PHASES: further roundings of the same weights, used by DDIM step i as version i % nphase
    // (version 0 = w / wfrag above); see diffuse_round_phases()
    int nphase = 1;
    void* w_ph[7] = {};
    void* wfrag_ph[7] = {};
};
struct NormW { int C = 0; float* gamma = nullptr; float* beta = nullptr; };
struct ResW {
    std::string prefix;
    int C = 0;
    NormW n1, n2;
    ConvW c1, c2;
    int film_off = 0;   // row offset into the concatenated FiLM linear: [off, off+C) scale, [off+C, off+2C) shift
};
enum LType { L_STEM, L_RES, L_DOWN, L_UP, L_HEAD };
struct Layer { LType type; std::string name; int idx; };

struct TensorRef {
    void* p = nullptr;
    int C = 0, H = 0, W = 0;
    float2* part = nullptr;      // GroupNorm partial sums written by the producer
    int n_sp = 0, n_nt = 0, bn = 0;
    bool pr = false;             // partial sums laid out per consumer wave (persistent kernel)
    std::shared_ptr<ConvArgs> prod;   // launch arguments of the producing conv (patched when its GroupNorm finalize is fused in)
};

struct StepCtx {
    int step = 0;
    const float* x_in = nullptr;   // NCHW fp32 image entering the stem
    float* x_state = nullptr;      // DDIM state updated by the head (== x_in inside the loop)
    float* eps_out = nullptr;
    int do_ddim = 0;
    float c[4] = {0, 0, 0, 0};
    const float* noise = nullptr;  // eta > 0: this step's noise tensor (sigma > 0), else null
    float sigma = 0.f;
};

struct Launch {
    int family;
    double flops, bytes;
    std::function<hipError_t(hipStream_t, const StepCtx&)> fn;
};

const char* kFamilies[] = {"conv3x3_s1_igemm", "conv3x3_s2_igemm", "convT4x4_s2_igemm", "stem_conv_igemm",
                           "head_conv_ddim", "gn_finalize", "conditioning", "gn_silu_prepass"};
enum { F_C3S1 = 0, F_C3S2, F_CT4, F_STEM, F_HEAD, F_GNF, F_COND, F_LAYOUT, F_COUNT };

struct Bump {
    char* base; size_t off = 0; bool measure;
    Bump(void* b, bool m) : base((char*)b), measure(m) {}
    void* take(size_t bytes) { off = align_up(off, 256); void* p = measure ? nullptr : base + off; off += bytes; return p; }
};

constexpr int kMaxNorms = 128;

struct GraphEntry {
    int steps = 0;
    std::vector<int32_t> ts;
    std::vector<float> coef;
    std::vector<float> sigma;            // eta > 0 (empty: eta = 0)
    const float* noise = nullptr;        // ... and the address of the caller's noise tensor the graph was captured with
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

constexpr size_t kMaxGraphsPerPlan = 4;
struct Plan {
    int B = 0, H = 0, W = 0, steps = 0;
    void* ws = nullptr;
    size_t bytes = 0;
    // conditioning buffers
    int32_t* ts_dev = nullptr;
    float *temb = nullptr, *t1 = nullptr, *tp = nullptr, *zp = nullptr, *film = nullptr;
    float *zbuf = nullptr, *xstate = nullptr;
    std::vector<std::pair<void*, size_t>> zero_once;   // regions zeroed when the plan is created (split-K hand-off flags)
    unsigned* counters = nullptr;        // arrival counters of the fused GroupNorm finalizes, [kMaxNorms][B], zero at rest
    int n_counters = 0;
    std::vector<int32_t> ts_keep;        // the timestep table currently in ts_dev (uploaded synchronously, only when it changes)
    std::vector<Launch> ops;             // one UNet evaluation (+ DDIM update in the head)
    std::map<std::string, TensorRef> named;
    std::vector<GraphEntry> graphs;
    ~Plan() {
        for (auto& g : graphs) {
            if (g.exec) (void)hipGraphExecDestroy(g.exec);
            if (g.graph) (void)hipGraphDestroy(g.graph);
        }
    }
};

}  // namespace

struct ccn_handle_s {
    ccn_config_t cfg{};
    int elem = 4;                       // bytes per activation / weight element
    std::vector<ParamInfo> params;
    std::map<std::string, std::vector<float>> host;   // loaded fp32 copies until commit
    bool committed = false;
    std::vector<void*> dev_allocs;
    // model
    ConvW stem, head;
    NormW out_norm;
    std::vector<ResW> res;
    std::vector<ConvW> downs, ups;
    std::vector<Layer> layers;
    float *tp0_w = nullptr, *tp0_b = nullptr, *tp2_w = nullptr, *tp2_b = nullptr, *zp_w = nullptr, *zp_b = nullptr;
    float *film_w = nullptr, *film_b = nullptr;
    float* head_w_f32 = nullptr;       // (img_ch, C, 3, 3) fp32 copy of out.weight for the dedicated head kernel
    int F = 0;                          // rows of the concatenated FiLM linear
    int G = 8;
    int weight_rounding = CCN_ROUND_DIFFUSED_PHASES;   // how bf16 mode rounds the conv weights in ccn_commit_params (ccn_set_weight_rounding)
    std::map<std::string, std::vector<float>> host_ph[7];   // versions 1.. of the rounded conv weights until commit
    int nphase = 1;                                     // versions per conv weight of the committed model (diffuse_round_phases)
    std::vector<std::unique_ptr<Plan>> plans;
    hipStream_t cap_stream = nullptr;
    unsigned* err_host = nullptr;       // pinned, device-mapped error word the kernels OR into (ConvArgs::err)
    unsigned* err_dev = nullptr;
    // profiling
    bool profiling = false;
    std::vector<hipEvent_t> ev_pool;
    struct Rec { int family; double flops, bytes; int e0, e1; };
    std::vector<Rec> recs;
    size_t ev_used = 0;
    std::vector<std::string> fam_names;
};

namespace {

// ---- architecture -> parameter list (models/unet.py:45-79 registration order) -------------------------
void add_res_params(std::vector<ParamInfo>& v, const std::string& p, int c, int d)
{
    v.push_back({p + ".norm1.weight", {c}}); v.push_back({p + ".norm1.bias", {c}});
    v.push_back({p + ".conv1.weight", {c, c, 3, 3}}); v.push_back({p + ".conv1.bias", {c}});
    v.push_back({p + ".norm2.weight", {c}}); v.push_back({p + ".norm2.bias", {c}});
    v.push_back({p + ".conv2.weight", {c, c, 3, 3}}); v.push_back({p + ".conv2.bias", {c}});
    v.push_back({p + ".film.to_scale.weight", {c, d}}); v.push_back({p + ".film.to_scale.bias", {c}});
    v.push_back({p + ".film.to_shift.weight", {c, d}}); v.push_back({p + ".film.to_shift.bias", {c}});
}

void build_arch(ccn_handle_s* h)
{
    const ccn_config_t& c = h->cfg;
    auto& v = h->params;
    const int td = c.time_dim;
    v.push_back({"time_proj.0.weight", {td * 4, td}}); v.push_back({"time_proj.0.bias", {td * 4}});
    v.push_back({"time_proj.2.weight", {td, td * 4}}); v.push_back({"time_proj.2.bias", {td}});
    v.push_back({"z_proj.0.weight", {td, c.z_dim}}); v.push_back({"z_proj.0.bias", {td}});
    v.push_back({"in_conv.weight", {c.base, c.img_ch, 3, 3}}); v.push_back({"in_conv.bias", {c.base}});
    h->layers.push_back({L_STEM, "in_conv", 0});
    int ch = c.base, film_off = 0;
    auto add_res = [&](const std::string& name, int cc) {
        add_res_params(v, name, cc, td);
        ResW r; r.prefix = name; r.C = cc; r.film_off = film_off; film_off += 2 * cc;
        h->res.push_back(r);
        h->layers.push_back({L_RES, name, (int)h->res.size() - 1});
    };
    for (int i = 0; i < c.n_mult; ++i) {
        const int m = c.ch_mult[i];
        add_res("down." + std::to_string(3 * i), ch);
        add_res("down." + std::to_string(3 * i + 1), ch);
        const std::string dn = "down." + std::to_string(3 * i + 2);
        v.push_back({dn + ".weight", {ch * m, ch, 3, 3}}); v.push_back({dn + ".bias", {ch * m}});
        ConvW d; d.kind = KIND_C3S2; d.Cin = ch; d.Cout = ch * m; h->downs.push_back(d);
        h->layers.push_back({L_DOWN, dn, i});
        ch *= m;
    }
    add_res("mid1", ch);
    add_res("mid2", ch);
    for (int i = 0; i < c.n_mult; ++i) {
        const int m = c.ch_mult[c.n_mult - 1 - i];
        add_res("up." + std::to_string(3 * i), ch);
        add_res("up." + std::to_string(3 * i + 1), ch);
        const std::string un = "up." + std::to_string(3 * i + 2);
        v.push_back({un + ".weight", {ch, ch / m, 4, 4}}); v.push_back({un + ".bias", {ch / m}});
        ConvW u; u.kind = KIND_CT4; u.Cin = ch; u.Cout = ch / m; h->ups.push_back(u);
        h->layers.push_back({L_UP, un, i});
        ch /= m;
    }
    v.push_back({"out_norm.weight", {ch}}); v.push_back({"out_norm.bias", {ch}});
    v.push_back({"out.weight", {c.img_ch, ch, 3, 3}}); v.push_back({"out.bias", {c.img_ch}});
    h->layers.push_back({L_HEAD, "out", 0});
    h->F = film_off;
}

// ---- weight repacking ------------------------------------------------------------------------------------
int upload(ccn_handle_s* h, const void* src, size_t bytes, void** dst)
{
    void* p = nullptr;
    HIPCHK(hipMalloc(&p, bytes ? bytes : 4));
    h->dev_allocs.push_back(p);
    if (bytes) HIPCHK(hipMemcpy(p, src, bytes, hipMemcpyHostToDevice));
    *dst = p;
    return CCN_OK;
}
int upload_f32(ccn_handle_s* h, const std::string& name, float** dst)
{
    const auto& v = h->host.at(name);
    return upload(h, v.data(), v.size() * 4, (void**)dst);
}

float bf16_value(float f)
{
    const uint32_t u = (uint32_t)f2bf_host(f) << 16;
    float r;
    std::memcpy(&r, &u, 4);
    return r;
}

// bf16 mode, CCN_ROUND_DIFFUSED: the conv weights are rounded to bf16 with ERROR DIFFUSION along the contraction of every output
// channel instead of independently (round-to-nearest-even): walking (cin, ky, kx) with kx fastest, each element is rounded after
// the accumulated rounding error of its predecessors has been added to it, so that every partial sum of an output channel's
// weights -- over the taps of one input channel, over input channels -- stays within half an ulp of the fp32 sum.  Independent
// rounding perturbs the model statically: the response of each conv to the smooth / mean part of its input shifts by a random
// walk over K = 9 Cin roundings, the same way in all 50 steps, and the reconstructions lose 0.33 % contrast (+0.11 % PSNR against
// the fp32 mode, above north_star's 0.1 % gate; tools/weight_rounding_probe.py, profiles/r03_weight_rounding_probe.txt).  With
// diffusion the same bf16 kernels land within 0.04 % mean / 0.075 % max.  Costs nothing at run time: the values are ordinary bf16.
// The weights are replaced IN PLACE by bf16-representable fp32 values, so every packer below rounds them exactly.
// ConvTranspose2d (Cin, Cout, 4, 4): an output pixel of parity (py, px) sees only 4 of the 16 taps (ky in {1,3} or {0,2}), so the
// diffusion runs per (output channel, parity) over (cin, those 4 taps).
void diffuse_round_conv(std::vector<float>& w, int O, int I, int taps)        // Conv2d (O, I, k, k)
{
    for (int o = 0; o < O; ++o) {
        double err = 0.0;
        float* row = w.data() + (size_t)o * I * taps;
        for (size_t k = 0; k < (size_t)I * taps; ++k) {
            const double t = (double)row[k] + err;
            const float r = bf16_value((float)t);
            err = t - (double)r;
            row[k] = r;
        }
    }
}
void diffuse_round_convT(std::vector<float>& w, int I, int O)                 // ConvTranspose2d (I, O, 4, 4), stride 2, padding 1
{
    static const int kk2[2][2] = {{1, 3}, {0, 2}};                            // kernel indices feeding even / odd outputs (fill_taps)
    for (int o = 0; o < O; ++o)
        for (int par = 0; par < 4; ++par) {
            double err = 0.0;
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 4; ++t) {
                    const int wt = kk2[par >> 1][t >> 1] * 4 + kk2[par & 1][t & 1];
                    float& v = w[((size_t)i * O + o) * 16 + wt];
                    const double tt = (double)v + err;
                    const float r = bf16_value((float)tt);
                    err = tt - (double)r;
                    v = r;
                }
        }
}

// CCN_ROUND_DIFFUSED_PHASES: error diffusion ALSO ALONG THE DDIM STEPS.  A rounded weight is a static perturbation of the model:
// the same error acts in each of the 50 evaluations of one trajectory, and because x changes slowly from step to step its effect
// adds up coherently -- that, not the size of one forward's error, is what moved the PSNR (zero-mean noise of twice the size drawn
// afresh every step moves it 30x less; tools/bf16_bias_probe.py).  So the sampler uses n roundings W_0 .. W_{n-1} of every conv
// weight in turn (step i takes version i % n), built with the rounding error fed forward from version to version:
//     W_k = round( (k + 1) W - (W_0 + ... + W_{k-1}) )      (each `round` = the spatial diffusion above)
// Every partial sum W_0 + ... + W_k stays within half a bf16 ulp of (k + 1) W: the MEAN weight over a period is accurate to
// ulp / (2n) instead of ulp / 2, and what remains alternates in sign from step to step with period <= n steps, which the sampler
// averages out.  Costs n copies of the bf16 weights in HBM (65 MB each at C2) and nothing at run time: each step reads one version.
// n = 8 (bench workload, mean / max PSNR delta per record: 1 version 5.2e-4 / 8.6e-4, 4 versions 4.5e-4 / 9.4e-4, 8 versions
// 3.9e-4 / 8.5e-4); 4 for models above 100 M conv weights (C4: 816 M -- commit time and 1.6 GB per version).
int phases_for(const ccn_handle_s* h)
{
    size_t total = 0;
    for (auto& p : h->params) if (p.shape.size() == 4) total += p.numel();
    return total > (size_t)100000000 ? 4 : 8;
}
void diffuse_round_phases(ccn_handle_s* h, const ParamInfo& p)
{
    auto& w0 = h->host.at(p.name);
    const std::vector<float> orig = w0;
    const bool convT = p.shape[2] == 4;
    auto round_one = [&](std::vector<float>& w) {
        if (convT) diffuse_round_convT(w, (int)p.shape[0], (int)p.shape[1]);
        else diffuse_round_conv(w, (int)p.shape[0], (int)p.shape[1], (int)(p.shape[2] * p.shape[3]));
    };
    round_one(w0);
    std::vector<double> sum(orig.size());
    for (size_t i = 0; i < orig.size(); ++i) sum[i] = (double)w0[i];
    for (int k = 1; k < h->nphase; ++k) {
        std::vector<float> wk(orig.size());
        for (size_t i = 0; i < orig.size(); ++i) wk[i] = (float)((double)(k + 1) * (double)orig[i] - sum[i]);
        round_one(wk);
        for (size_t i = 0; i < orig.size(); ++i) sum[i] += (double)wk[i];
        h->host_ph[k - 1][p.name] = std::move(wk);
    }
}

}  // namespace
// Host-only arithmetic of the bf16 weight rounding, exported for tests/test_host_logic.py (not part of include/ccn_hip.h; no GPU
// involved): `w` is a Conv2d (O, I, taps) or -- convT != 0 -- a ConvTranspose2d (I, O, 4, 4) weight; out[p] receives version p of
// `phases` versions (phases == 1: the plain error-diffused rounding), each bf16-representable fp32.
extern "C" int ccn_internal_round_weights(const float* w, int O, int I, int taps, int convT, int phases, float* out)
{
    if (!w || !out || O <= 0 || I <= 0 || phases < 1 || phases > 8 || (convT && taps != 16)) return 1;
    const size_t n = (size_t)O * I * taps;
    std::vector<float> orig(w, w + n);
    std::vector<double> sum(n, 0.0);
    for (int k = 0; k < phases; ++k) {
        std::vector<float> wk(n);
        for (size_t i = 0; i < n; ++i) wk[i] = (float)((double)(k + 1) * (double)orig[i] - sum[i]);
        if (convT) diffuse_round_convT(wk, I, O); else diffuse_round_conv(wk, O, I, taps);
        for (size_t i = 0; i < n; ++i) sum[i] += (double)wk[i];
        std::memcpy(out + (size_t)k * n, wk.data(), n * 4);
    }
    return 0;
}
namespace {

// element (tap, o, i) of the packed [taps][Cout_pad][Cin_pad] tensor
template <typename F>
int pack_and_upload(ccn_handle_s* h, ConvW& cw, int taps, F&& at)
{
    const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
    cw.BN = conv_bn_for(cw.Cout, cw.kind);
    cw.Cout_pad = (int)align_up(cw.Cout, cw.BN);
    cw.ntaps_w = taps;
    const size_t n = (size_t)taps * cw.Cout_pad * cw.Cin_pad;
    (void)cke;
    if (h->cfg.dtype == CCN_DTYPE_BF16) {
        std::vector<uint16_t> buf(n, 0);
        for (int t = 0; t < taps; ++t)
            for (int o = 0; o < cw.Cout; ++o)
                for (int i = 0; i < cw.Cin_pad; ++i) buf[((size_t)t * cw.Cout_pad + o) * cw.Cin_pad + i] = f2bf_host(at(t, o, i));
        return upload(h, buf.data(), n * 2, &cw.w);
    }
    std::vector<float> buf(n, 0.f);
    for (int t = 0; t < taps; ++t)
        for (int o = 0; o < cw.Cout; ++o)
            for (int i = 0; i < cw.Cin_pad; ++i) buf[((size_t)t * cw.Cout_pad + o) * cw.Cin_pad + i] = at(t, o, i);
    return upload(h, buf.data(), n * 4, &cw.w);
}

int pack_conv3(ccn_handle_s* h, ConvW& cw, const std::string& name)     // Conv2d weight (O, I, 3, 3)
{
    const float* w = h->host.at(name + ".weight").data();
    const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
    cw.Cin_pad = (int)align_up(cw.Cin, cke);
    const int I = cw.Cin;
    int rc = pack_and_upload(h, cw, 9, [&](int t, int o, int i) { return i < I ? w[((size_t)o * I + i) * 9 + t] : 0.f; });
    if (rc) return rc;
    if (cw.kind == KIND_C3S2 && cw.BN == 128 && h->cfg.dtype == CCN_DTYPE_BF16) {
        // plane-pass order for the persistent kernel (ccn_conv_pr.hip, NTAPS == 2): prs2_frag_index / prs2_tap (ccn_internal.h), the tenth
        // tap slot stays zero; the training step's device-side packer (PK_FRAG_S2) goes through the same functions
        const int nch = cw.Cin_pad / cke, n32 = cw.Cout_pad / 32, O = cw.Cout;
        std::vector<uint16_t> fr((size_t)nch * 5 * n32 * 2 * 4 * 64 * 8, 0);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int ps = 0; ps < 10; ++ps) {
                    const int tp = prs2_tap(ps >> 1, ps & 1);
                    if (tp >= 0) fr[prs2_frag_index(ps >> 1, ps & 1, o, i, cw.Cout_pad)] = f2bf_host(w[((size_t)o * I + i) * 9 + tp]);
                }
        if ((rc = upload(h, fr.data(), fr.size() * 2, &cw.wfrag))) return rc;
    }
    if (cw.kind == KIND_C3S1 && cw.BN == 128 && h->cfg.dtype == CCN_DTYPE_BF16) {
        // fragment order of the persistent kernel's 3x3 form (pr3_frag_index, ccn_internal.h): one wave load = 1 KiB contiguous
        const int nch = cw.Cin_pad / cke, n32 = cw.Cout_pad / 32, O = cw.Cout;
        std::vector<uint16_t> fr((size_t)nch * n32 * 36 * 64 * 8, 0);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int t = 0; t < 9; ++t) fr[pr3_frag_index(o, i, t, cw.Cout_pad)] = f2bf_host(w[((size_t)o * I + i) * 9 + t]);
        if ((rc = upload(h, fr.data(), fr.size() * 2, &cw.wfrag))) return rc;
    }
    return upload_f32(h, name + ".bias", &cw.bias);
}
int pack_convT(ccn_handle_s* h, ConvW& cw, const std::string& name)     // ConvTranspose2d weight (I, O, 4, 4)
{
    const float* w = h->host.at(name + ".weight").data();
    const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
    cw.Cin_pad = (int)align_up(cw.Cin, cke);
    const int I = cw.Cin, O = cw.Cout;
    int rc = pack_and_upload(h, cw, 16, [&](int t, int o, int i) { return i < I ? w[((size_t)i * O + o) * 16 + t] : 0.f; });
    if (rc) return rc;
    if (cw.BN == 128 && h->cfg.dtype == CCN_DTYPE_BF16) {
        // fragment order for ccn_conv_pr.hip: [parity][chunk][Cout_pad/32][tap 0..3][kk][lane][8] = prct_frag_index, tap -> kernel
        // element prct_tap (ccn_internal.h; the same order as fill_taps: even outputs <- k in {1, 3}, odd <- {0, 2}); the training
        // step's device-side packer (PK_FRAG_CT) goes through the same functions
        const int nch = cw.Cin_pad / cke, n32 = cw.Cout_pad / 32;
        std::vector<uint16_t> fr((size_t)4 * nch * n32 * 4 * 4 * 64 * 8, 0);
        for (int o = 0; o < O; ++o)
            for (int i = 0; i < I; ++i)
                for (int pt = 0; pt < 16; ++pt)
                    fr[prct_frag_index(pt >> 2, pt & 3, o, i, cw.Cout_pad, nch)] = f2bf_host(w[((size_t)i * O + o) * 16 + prct_tap(pt >> 2, pt & 3)]);
        if ((rc = upload(h, fr.data(), fr.size() * 2, &cw.wfrag))) return rc;
    }
    return upload_f32(h, name + ".bias", &cw.bias);
}
int pack_stem(ccn_handle_s* h, ConvW& cw, const std::string& name)      // Conv2d weight (O, img_ch, 3, 3) as K = I*9
{
    const float* w = h->host.at(name + ".weight").data();
    const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
    cw.Cin_pad = cke;
    const int K = cw.Cin * 9;
    int rc = pack_and_upload(h, cw, 1, [&](int, int o, int k) { return k < K ? w[(size_t)o * K + k] : 0.f; });
    if (rc) return rc;
    if (stem2_supported(h->cfg.dtype, cw.Cin, cw.Cout, h->G)) {
        // fragment order for ccn_stem.hip: [Cout/32][k-step 2][lane = h*32 + r][8 bf16]; lane (r, h) holds output channel
        // j*32 + r, k = 16 s + 8 h + e; k = K is the bias (multiplied by a constant-one im2col element)
        const float* bias = h->host.at(name + ".bias").data();
        const int nt = cw.Cout / 32;
        std::vector<uint16_t> fr((size_t)nt * 2 * 64 * 8, 0);
        for (int j = 0; j < nt; ++j)
            for (int s = 0; s < 2; ++s)
                for (int ln = 0; ln < 64; ++ln)
                    for (int e = 0; e < 8; ++e) {
                        const int o = j * 32 + (ln & 31), k = 16 * s + 8 * (ln >> 5) + e;
                        const float v = k < K ? w[(size_t)o * K + k] : (k == K ? bias[o] : 0.f);
                        fr[(((size_t)j * 2 + s) * 64 + ln) * 8 + e] = f2bf_host(v);
                    }
        if ((rc = upload(h, fr.data(), fr.size() * 2, &cw.wfrag))) return rc;
    }
    return upload_f32(h, name + ".bias", &cw.bias);
}

// ---- plan -------------------------------------------------------------------------------------------------
struct ConvGeom { int Hout, Wout, MH, MW, OS, npar, ntaps, n_ty, n_tx, n_nt, th; };

ConvGeom conv_geom(const ConvW& cw, int B, int Hin, int Win)
{
    ConvGeom g{};
    switch (cw.kind) {
        case KIND_C3S2: g.Hout = (Hin - 1) / 2 + 1; g.Wout = (Win - 1) / 2 + 1; g.MH = g.Hout; g.MW = g.Wout; g.OS = 1; g.npar = 1; g.ntaps = 9; break;
        case KIND_CT4: g.Hout = Hin * 2; g.Wout = Win * 2; g.MH = Hin; g.MW = Win; g.OS = 2; g.npar = 4; g.ntaps = 4; break;
        case KIND_STEM: g.Hout = Hin; g.Wout = Win; g.MH = Hin; g.MW = Win; g.OS = 1; g.npar = 1; g.ntaps = 1; break;
        default: g.Hout = Hin; g.Wout = Win; g.MH = Hin; g.MW = Win; g.OS = 1; g.npar = 1; g.ntaps = 9; break;
    }
    g.n_nt = cw.Cout_pad / cw.BN;
    g.th = conv_tile_rows(cw.kind, cw.BN, B, g.MH, g.MW, g.npar, g.n_nt);
    g.n_ty = ceil_div(g.MH, g.th); g.n_tx = ceil_div(g.MW, 32);
    return g;
}

void fill_taps(ConvArgs& a, int kind)
{
    std::memset(a.tapinfo, 0, sizeof(a.tapinfo));
    if (kind == KIND_STEM) { a.tapinfo[0] = ConvArgs::make_tap(0, 0, 0); return; }
    if (kind == KIND_CT4) {
        // out = 2*in - 1 + k  (ConvTranspose2d k=4, s=2, p=1): even out <- k in {1 (d=0), 3 (d=-1)}; odd out <- k in {0 (d=+1), 2 (d=0)}
        static const int kk[2][2] = {{1, 3}, {0, 2}}, dd[2][2] = {{0, -1}, {1, 0}};
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px)
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j)
                        a.tapinfo[(py * 2 + px) * 4 + i * 2 + j] = ConvArgs::make_tap(dd[py][i], dd[px][j], kk[py][i] * 4 + kk[px][j]);
        return;
    }
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) a.tapinfo[ky * 3 + kx] = ConvArgs::make_tap(ky - 1, kx - 1, ky * 3 + kx);
}

struct PlanBuilder {
    ccn_handle_s* h;
    Plan* plan;
    Bump bump;
    int B;
    int film_stride;     // floats between consecutive samples' FiLM rows
    PlanBuilder(ccn_handle_s* h_, Plan* p, void* ws, bool measure) : h(h_), plan(p), bump(ws, measure), B(p->B), film_stride(h_->F) {}

    TensorRef new_tensor(int C, int H, int W)
    {
        TensorRef t; t.C = C; t.H = H; t.W = W;
        t.p = bump.take((size_t)B * H * W * C * h->elem);
        return t;
    }
    float2* new_ab(int C) { return (float2*)bump.take((size_t)B * C * sizeof(float2)); }
    int groups_for(int C) const { return C < h->G ? C : h->G; }

    // 2 when the layer should run split-K on the persistent kernel (see conv()), else 1
    int split_k_for(const ConvW& cw, const ConvGeom& g, bool has_gn, bool s2pr) const
    {
        static const bool off = diag_env("CCN_NO_SPLITK") != nullptr;
        if (off || h->cfg.dtype != CCN_DTYPE_BF16 || !cw.wfrag || cw.BN != 128) return 1;
        // Measured at C2: a single-tile-per-workgroup launch of the persistent kernel carries ~25 us of fixed cost (cold first
        // chunk, serial last epilogue, hand-off), so halving the K loop of the 32-pixel 3x3 s1 layers (56 us) does not beat the
        // 4-row kernel (53 us); it does help the stride-2 conv into that level (63 -> 56 us), whose alternative is 128 tiles.
        // CCN_SPLITK_S1=1 enables it for the stride-1 layers too.
        static const bool s1_too = diag_env("CCN_SPLITK_S1") != nullptr;
        if (!((cw.kind == KIND_C3S1 && s1_too) || (cw.kind == KIND_C3S2 && s2pr))) return 1;
        if (!conv_pr_selected(h->cfg.dtype, cw.kind, cw.BN, 8)) return 1;
        const int cke = 64;
        const long tiles8 = (long)B * ceil_div(g.MH, 8) * g.n_tx * g.npar * g.n_nt;
        int nchunk = cw.Cin_pad / cke;
        if (cw.kind == KIND_C3S2) { if ((nchunk & 1)) return 1; nchunk *= 5; }
        (void)has_gn;
        return (tiles8 >= 32 && tiles8 <= 128 && nchunk >= 4 && (nchunk & 1) == 0) ? 2 : 1;
    }

    // conv launch; `want_part`: also emit the partial sums of the output's GroupNorm
    // `in_norm` (instead of gn_ab): the input goes through this GroupNorm (+ SiLU) first; its finalize either becomes part of
    // the conv launch itself (persistent kernel, see in_kernel_stats) or a gn_finalize launch in front of it
    void conv(const ConvW& cw, int family, const TensorRef& in, TensorRef& out, const float2* gn_ab, int film_off,
              const TensorRef* res, bool want_part, bool is_stem = false, bool is_head = false, const NormW* in_norm = nullptr)
    {
        ConvGeom g = conv_geom(cw, B, in.H, in.W);
        // stride-2 convs: the 4-wave kernel works on 4-row tiles; the persistent kernel (bf16) takes them on 8-row tiles as soon
        // as half the CUs get a tile (it is ~2.5x more efficient per tile)
        const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
        const bool s2pr = cw.kind == KIND_C3S2 && cw.wfrag && conv_pr_selected(h->cfg.dtype, cw.kind, cw.BN, 8) && !gn_ab &&
                          cw.Cin_pad / cke >= 2 && (double)B * g.Hout * g.Wout * cw.Cout * h->elem < 2.0e9 &&
                          (long)B * ceil_div(g.MH, 8) * g.n_tx * g.n_nt >= 128;
        if (s2pr) { g.th = 8; g.n_ty = ceil_div(g.MH, 8); }
        // small layers (at most #CUs/2 tiles of 8 rows: the 32-pixel level at C2): 8-row tiles on the persistent kernel with the
        // Cin chunks split over two workgroups per tile instead of 4-row tiles on the LDS-bound kernel
        const int ksplit = res ? 1 : split_k_for(cw, g, gn_ab != nullptr, s2pr);      // (the split-K instantiation has no residual path of its own)
        if (ksplit == 2 && g.th != 8) { g.th = 8; g.n_ty = ceil_div(g.MH, 8); }
        std::shared_ptr<ConvArgs> ap(new ConvArgs());
        ConvArgs& a = *ap;
        a.in = in.p; a.w = cw.w; a.wfrag = cw.wfrag; a.bias = cw.bias; a.out = out.p;
        a.film = nullptr; a.res = res ? res->p : nullptr;
        a.B = B; a.Hin = in.H; a.Win = in.W; a.Cin = cw.Cin; a.Cin_pad = cw.Cin_pad;
        a.Hout = g.Hout; a.Wout = g.Wout; a.Cout = cw.Cout; a.Cout_pad = cw.Cout_pad;
        a.MH = g.MH; a.MW = g.MW; a.OS = g.OS; a.npar = g.npar; a.ntaps = g.ntaps;
        a.n_ty = g.n_ty; a.n_tx = g.n_tx; a.n_nt = g.n_nt; a.th = g.th;
        a.nchunk = cw.Cin_pad / cke;
        a.silu = 1;
        a.G = groups_for(cw.Cout); a.cpg = cw.Cout / a.G;
        a.nslot = g.n_ty * g.n_tx * g.npar * g.n_nt;
        if (s2pr) { a.nchunk = 5 * (cw.Cin_pad / cke); a.ntaps = 2; }   // plane passes (ccn_conv_pr.hip)
        const bool pr = cw.wfrag && conv_pr_selected(h->cfg.dtype, cw.kind, cw.BN, g.th) && cw.Cin_pad / cke >= 2 && (cw.kind != KIND_C3S2 || s2pr) &&
                        (double)B * g.Hout * g.Wout * cw.Cout * h->elem < 2.0e9;
        if (!pr) a.wfrag = nullptr;
        a.use_pr = pr ? 1 : 0;
        if (in_norm) {
            if (pr && ksplit != 2 && cw.kind == KIND_C3S1 && in_kernel_stats(in, cw.Cin)) {
                const int cpg_in = cw.Cin / groups_for(cw.Cin);
                a.gs_part = in.part; a.gs_gamma = in_norm->gamma; a.gs_beta = in_norm->beta;
                a.gs_inv_count = 1.0 / ((double)cpg_in * in.H * in.W);
                a.gs_nsp = in.n_sp; a.gs_nnt = in.n_nt; a.gs_bn = in.bn; a.gs_cpg = cpg_in;
            } else gn_ab = gn(in, *in_norm);                     // (pushes the finalize launch in front of this conv)
        }
        a.gn_ab = gn_ab;
        a.ksplit = 1;
        if (pr && ksplit == 2) {
            a.ksplit = 2;
            a.kpart = bump.take((size_t)B * g.Hout * g.Wout * cw.Cout * h->elem);
            const size_t fbytes = (size_t)B * g.n_ty * g.n_tx * g.npar * g.n_nt * 4 * sizeof(unsigned);
            a.kflag = (unsigned*)bump.take(fbytes);
            if (a.kflag) plan->zero_once.push_back({a.kflag, fbytes});
        }
        // (the persistent kernel publishes ONE partial per tile: its four producer waves' sums are combined in LDS first)
        const bool stem2 = is_stem && cw.wfrag && stem2_supported(h->cfg.dtype, cw.Cin, cw.Cout, h->G);
        a.use_stem2 = stem2 ? 1 : 0;
        if (stem2) { a.wfrag = cw.wfrag; a.nslot = 4 * stem2_blocks(in.H, in.W, nullptr); }   // one slot per wave
        a.film_bstride = film_stride;
        a.err = h->err_dev;
        fill_taps(a, cw.kind);
        if (want_part) {
            out.part = (float2*)bump.take((size_t)B * a.G * a.nslot * sizeof(float2));
            out.n_sp = g.n_ty * g.n_tx * g.npar; out.n_nt = g.n_nt; out.bn = cw.BN;
            if (pr) out.pr = true;
            if (stem2) { out.n_sp = a.nslot; out.n_nt = 1; out.bn = 1 << 30; out.pr = true; }
        }
        a.part = out.part;
        a.bn = cw.BN;
        a.fin_blocks = g.n_ty * g.n_tx * g.npar * g.n_nt;
        if (want_part) out.prod = ap;
        const double macs = (double)B * g.Hout * g.Wout * cw.Cout * (double)(cw.kind == KIND_CT4 ? 4 : (cw.kind == KIND_STEM ? 9 : 9)) * cw.Cin;
        double bytes = ((double)B * in.H * in.W * cw.Cin + (double)B * g.Hout * g.Wout * cw.Cout + (res ? (double)B * g.Hout * g.Wout * cw.Cout : 0.0)
                        + (double)(cw.kind == KIND_CT4 ? 16 : 9) * cw.Cin * cw.Cout) * h->elem;
        if (is_stem) bytes = (double)B * in.H * in.W * cw.Cin * 4 + ((double)B * g.Hout * g.Wout * cw.Cout + 9.0 * cw.Cin * cw.Cout) * h->elem;
        if (is_head) bytes = ((double)B * in.H * in.W * cw.Cin + 9.0 * cw.Cin * cw.Cout) * h->elem + 3.0 * (double)B * g.Hout * g.Wout * cw.Cout * 4;
        const int dtype = h->cfg.dtype, kind = cw.kind, bn = cw.BN;
        float* film_tab = plan->film;
        const int F = film_stride;
        const int Bc = B;
        const ConvW cwv = cw;                                 // (device pointers of every weight version)
        const bool frag_used = a.wfrag != nullptr;
        Launch L{family, 2.0 * macs, bytes, nullptr};
        L.fn = [=](hipStream_t s, const StepCtx& c) -> hipError_t {
            ConvArgs k = *ap;                                 // read at launch time: gn() may have fused its finalize in
            if (cwv.nphase > 1 && (c.step % cwv.nphase)) {    // DDIM step i runs on weight version i % nphase (diffuse_round_phases)
                const int v = c.step % cwv.nphase - 1;
                k.w = cwv.w_ph[v];
                if (frag_used) k.wfrag = cwv.wfrag_ph[v];
            }
            if (film_off >= 0) k.film = film_tab + (size_t)c.step * Bc * F + film_off;
            if (is_stem) k.in = c.x_in;
            if (is_head) {
                k.x_state = c.x_state; k.eps_out = c.eps_out; k.do_ddim = c.do_ddim;
                k.c0 = c.c[0]; k.c1 = c.c[1]; k.c2 = c.c[2]; k.c3 = c.c[3];
                k.noise = c.noise; k.sigma = c.sigma;
            }
            return launch_conv(dtype, kind, bn, k, s);
        };
        plan->ops.push_back(std::move(L));
    }

    // The persistent kernel can form the scale/shift of its input's GroupNorm itself from the producer's partial sums (every
    // producer wave reduces 8 groups x 8 lanes): 8 groups, a thread's 8-channel slice inside one group, and few enough slots
    // per group that the reduction (slots / 8 loads per lane, redone whenever a workgroup moves to another sample) stays
    // cheaper than the ~6.5 us of a finalize launch + kernel boundary -- true below the 256-pixel level at C2.
    bool in_kernel_stats(const TensorRef& t, int C) const
    {
        static const bool off = diag_env("CCN_NO_INSTAT") != nullptr;       // diagnostics build only
        if (off || h->cfg.dtype != CCN_DTYPE_BF16 || t.n_sp <= 0 || t.C != C) return false;
        const int G = groups_for(C), cpg = C / G;
        if (G != 8 || (cpg % 8) != 0) return false;
        int nj = 1;
        for (int g = 0; g < G; ++g) { const int n = ((g + 1) * cpg - 1) / t.bn - (g * cpg) / t.bn + 1; if (n > nj) nj = n; }
        return (long)t.n_sp * nj <= 64;                            // 8 slots per lane: one round of loads
    }

    // GroupNorm finalize of tensor `t` for the norm (gamma, beta): returns the scale/shift table
    const float2* gn(const TensorRef& t, const NormW& n)
    {
        float2* ab = new_ab(t.C);
        const int G = groups_for(t.C), cpg = t.C / G;
        const double count = (double)cpg * t.H * t.W;
        // measured slower than the 5 us finalize launch it removes (every workgroup drains its stores and pays an atomic
        // round trip before exiting): 50.4 vs 52.1 img/s at C2, so opt-in only
        static const bool fuse = diag_env("CCN_FUSED_FINALIZE") != nullptr;
        if (fuse && t.prod && !t.pr && plan->counters && plan->n_counters < kMaxNorms) {
            // the producing conv's last workgroup per sample does the finalize (no launch, no kernel boundary)
            ConvArgs& pa = *t.prod;
            pa.fin_counter = plan->counters + (size_t)plan->n_counters * B;
            plan->n_counters++;
            pa.fin_gamma = n.gamma; pa.fin_beta = n.beta; pa.fin_ab = ab; pa.fin_count = count;
            return ab;
        }
        const float2* part = t.part; const int n_sp = t.n_sp, n_nt = t.n_nt, bn = t.bn, C = t.C, Bc = B;
        const float* gamma = n.gamma; const float* beta = n.beta;
        Launch L{F_GNF, 0.0, (double)B * G * n_sp * n_nt * 8.0 + (double)B * C * 8.0, nullptr};
        L.fn = [=](hipStream_t s, const StepCtx&) -> hipError_t {
            return launch_gn_finalize(part, Bc, G, n_sp, n_nt, bn, cpg, C, count, gamma, beta, 1e-5f, ab, s);
        };
        plan->ops.push_back(std::move(L));
        return ab;
    }

    // GroupNorm-apply + SiLU as a separate pass when the consuming conv would redo it per N tile (C >= 256)
    TensorRef preact(const TensorRef& t, const float2* ab)
    {
        TensorRef o = new_tensor(t.C, t.H, t.W);
        const int dtype = h->cfg.dtype, Bc = B, HW = t.H * t.W, C = t.C;
        const void* src = t.p; void* dst = o.p;
        Launch L{F_LAYOUT, 0.0, 2.0 * B * HW * (double)C * h->elem, nullptr};
        L.fn = [=](hipStream_t s, const StepCtx&) -> hipError_t { return launch_gn_act(dtype, src, ab, dst, Bc, HW, C, s); };
        plan->ops.push_back(std::move(L));
        return o;
    }

    // the same pass with the finalize of `t`'s GroupNorm folded in (no gn() launch, no scale/shift table)
    TensorRef preact_fused(const TensorRef& t, const NormW& n)
    {
        TensorRef o = new_tensor(t.C, t.H, t.W);
        const int dtype = h->cfg.dtype, Bc = B, HW = t.H * t.W, C = t.C;
        const int G = groups_for(t.C), cpg = t.C / G;
        const double count = (double)cpg * t.H * t.W;
        const float2* part = t.part; const int n_sp = t.n_sp, n_nt = t.n_nt, bn = t.bn;
        const float* gamma = n.gamma; const float* beta = n.beta;
        const void* src = t.p; void* dst = o.p;
        Launch L{F_LAYOUT, 0.0, 2.0 * B * HW * (double)C * h->elem, nullptr};
        L.fn = [=](hipStream_t s, const StepCtx&) -> hipError_t {
            return launch_gn_act_fused(dtype, src, dst, Bc, HW, C, part, G, n_sp, n_nt, bn, cpg, count, gamma, beta, 1e-5f, s);
        };
        plan->ops.push_back(std::move(L));
        return o;
    }

    // the 3x3 s1 conv `cw` on an H x W input will run on the persistent kernel (same conditions as conv())
    bool will_use_pr(const ConvW& cw, int H, int W) const
    {
        ConvGeom g = conv_geom(cw, B, H, W);
        const int cke = h->cfg.dtype == CCN_DTYPE_BF16 ? 64 : 32;
        if (split_k_for(cw, g, true, false) == 2) g.th = 8;
        return cw.wfrag && conv_pr_selected(h->cfg.dtype, cw.kind, cw.BN, g.th) && cw.Cin_pad / cke >= 2 &&
               (double)B * g.Hout * g.Wout * cw.Cout * h->elem < 2.0e9;
    }

    // x + conv2(SiLU(GN2(FiLM(conv1(SiLU(GN1(x))))))) -- models/blocks.py:40-44
    TensorRef resblock(const ResW& r, const TensorRef& x, bool out_feeds_gn, int film_off = -2)
    {
        // pre-pass only where the conv kernel would redo the transform per N tile AND has no idle VALU for it: the persistent
        // kernel's producers absorb it up to 4 N tiles (measured at the 512-channel level of C2, 4-row tiles: 69.5 vs 68.8
        // images/s against pre-pass + finalize-free conv; CCN_PREACT_PR=1 restores the pre-pass in front of it for A/B runs)
        static const bool preact_pr = diag_env("CCN_PREACT_PR") != nullptr;
        const bool pre = conv_wants_preact(r.c1.kind, r.c1.BN, r.c1.Cout_pad / r.c1.BN) && r.C / (h->elem == 2 ? 8 : 4) <= 256 &&
                         (preact_pr || r.c1.Cout_pad / r.c1.BN >= 6 || !will_use_pr(r.c1, x.H, x.W));
        static const bool fuse_act = !diag_env("CCN_NO_FUSED_GNACT");       // finalize folded into the pre-pass (A/B switch)
        const bool f1 = pre && fuse_act && x.n_sp > 0, f2 = pre && fuse_act;     // (n_sp, not the pointer: null while measuring)
        TensorRef y = new_tensor(r.C, x.H, x.W);
        if (f1) { TensorRef xa = preact_fused(x, r.n1); conv(r.c1, F_C3S1, xa, y, nullptr, film_off == -2 ? r.film_off : film_off, nullptr, true); }
        else if (pre) {
            const float2* ab1 = gn(x, r.n1);
            TensorRef xa = preact(x, ab1); conv(r.c1, F_C3S1, xa, y, nullptr, film_off == -2 ? r.film_off : film_off, nullptr, true);
        } else conv(r.c1, F_C3S1, x, y, nullptr, film_off == -2 ? r.film_off : film_off, nullptr, true, false, false, &r.n1);
        plan->named[r.prefix + ".film"] = y;
        TensorRef o = new_tensor(r.C, x.H, x.W);
        if (f2) { TensorRef ya = preact_fused(y, r.n2); conv(r.c2, F_C3S1, ya, o, nullptr, -1, &x, out_feeds_gn); }
        else if (pre) {
            const float2* ab2 = gn(y, r.n2);
            TensorRef ya = preact(y, ab2); conv(r.c2, F_C3S1, ya, o, nullptr, -1, &x, out_feeds_gn);
        } else conv(r.c2, F_C3S1, y, o, nullptr, -1, &x, out_feeds_gn, false, false, &r.n2);
        plan->named[r.prefix] = o;
        return o;
    }
};

int build_plan(ccn_handle_s* h, Plan* plan, void* ws, bool measure)
{
    const ccn_config_t& c = h->cfg;
    PlanBuilder pb(h, plan, ws, measure);
    const int B = plan->B, H = plan->H, W = plan->W, S = plan->steps;
    const int TR = B > S ? B : S, FR = S * B > B ? S * B : B;
    plan->ts_dev = (int32_t*)pb.bump.take((size_t)S * 4);
    plan->temb = (float*)pb.bump.take((size_t)TR * c.time_dim * 4);
    plan->t1 = (float*)pb.bump.take((size_t)TR * c.time_dim * 4 * 4);
    plan->tp = (float*)pb.bump.take((size_t)TR * c.time_dim * 4);
    plan->zp = (float*)pb.bump.take((size_t)B * c.time_dim * 4);
    plan->film = (float*)pb.bump.take((size_t)FR * h->F * 4);
    plan->zbuf = (float*)pb.bump.take((size_t)B * c.z_dim * 4);
    plan->xstate = (float*)pb.bump.take((size_t)B * c.img_ch * H * W * 4);
    plan->counters = (unsigned*)pb.bump.take((size_t)kMaxNorms * B * 4);
    plan->n_counters = 0;

    plan->ops.clear(); plan->named.clear();
    TensorRef img; img.C = c.img_ch; img.H = H; img.W = W;     // NCHW fp32, pointer supplied per call
    TensorRef x;
    std::vector<TensorRef> skips;
    const size_t nl = h->layers.size();
    for (size_t li = 0; li < nl; ++li) {
        const Layer& L = h->layers[li];
        const bool next_is_gn = li + 1 < nl && (h->layers[li + 1].type == L_RES || h->layers[li + 1].type == L_HEAD);
        switch (L.type) {
            case L_STEM: {
                x = pb.new_tensor(h->stem.Cout, H, W);
                pb.conv(h->stem, F_STEM, img, x, nullptr, -1, nullptr, next_is_gn, true, false);
                plan->named[L.name] = x;
                break;
            }
            case L_RES: {
                x = pb.resblock(h->res[L.idx], x, next_is_gn);
                break;
            }
            case L_DOWN: {
                skips.push_back(x);
                const ConvW& cw = h->downs[L.idx];
                if ((x.H % 2) || (x.W % 2)) return fail(CCN_EINVAL, "H and W must be divisible by 2^len(ch_mult)");
                TensorRef o = pb.new_tensor(cw.Cout, x.H / 2, x.W / 2);
                pb.conv(cw, F_C3S2, x, o, nullptr, -1, nullptr, next_is_gn);
                plan->named[L.name] = o;
                x = o;
                break;
            }
            case L_UP: {
                const ConvW& cw = h->ups[L.idx];
                TensorRef o = pb.new_tensor(cw.Cout, x.H * 2, x.W * 2);
                TensorRef sk = skips.back(); skips.pop_back();
                if (sk.C != cw.Cout || sk.H != o.H || sk.W != o.W) return fail(CCN_EINVAL, "skip shape mismatch");
                pb.conv(cw, F_CT4, x, o, nullptr, -1, &sk, next_is_gn);
                plan->named[L.name] = o;
                x = o;
                break;
            }
            case L_HEAD: {
                static const bool no_head2 = diag_env("CCN_NO_HEAD2") != nullptr;
                if (!no_head2 && head2_supported(c.dtype, h->head.Cin, h->head.Cout, h->G)) {
                    // dedicated kernel pair: out_norm's finalize folded into per-sample head weights, taps in the N dimension
                    std::shared_ptr<ConvArgs> ap(new ConvArgs());
                    ConvArgs& a = *ap;
                    a.in = x.p; a.bias = h->head.bias; a.B = B; a.Hin = H; a.Win = W; a.Cin = h->head.Cin; a.Cout = h->head.Cout;
                    a.Hout = H; a.Wout = W;
                    const float2* ab = pb.gn(x, h->out_norm);
                    void* scratch = pb.bump.take(head2_scratch_bytes(B, x.C));
                    const float* wf = h->head_w_f32;
                    const double macs = (double)B * H * W * h->head.Cout * 9.0 * h->head.Cin;
                    Launch Lh{F_HEAD, 2.0 * macs, ((double)B * H * W * x.C + 9.0 * x.C * h->head.Cout) * h->elem + 3.0 * (double)B * H * W * h->head.Cout * 4, nullptr};
                    Lh.fn = [=](hipStream_t s, const StepCtx& sc) -> hipError_t {
                        ConvArgs k = *ap;
                        k.x_state = sc.x_state; k.eps_out = sc.eps_out; k.do_ddim = sc.do_ddim;
                        k.c0 = sc.c[0]; k.c1 = sc.c[1]; k.c2 = sc.c[2]; k.c3 = sc.c[3];
                        k.noise = sc.noise; k.sigma = sc.sigma;
                        return launch_head2(k, ab, wf, scratch, sc.step, s);
                    };
                    plan->ops.push_back(std::move(Lh));
                    break;
                }
                const float2* ab = pb.gn(x, h->out_norm);
                TensorRef none;
                pb.conv(h->head, F_HEAD, x, none, ab, -1, nullptr, false, false, true);
                break;
            }
        }
    }
    plan->bytes = align_up(pb.bump.off, 256);
    return CCN_OK;
}

// A Plan owns hipGraphExec objects whose replays may still be running on caller streams (two batches in flight in cli.eval /
// bench.py): drain the device before destroying any of them.
void drop_plans(ccn_handle_s* h, size_t keep_newest)
{
    if (h->plans.size() <= keep_newest) return;
    (void)hipDeviceSynchronize();
    while (h->plans.size() > keep_newest) h->plans.erase(h->plans.begin());
}


int get_plan(ccn_handle_s* h, int B, int H, int W, int steps, void* ws, size_t ws_bytes, Plan** out)
{
    if (B <= 0 || H <= 0 || W <= 0 || steps <= 0) return fail(CCN_EINVAL, "B, H, W, steps must be positive");
    const int div = 1 << h->cfg.n_mult;
    if (H % div || W % div) return fail(CCN_EINVAL, "H and W must be divisible by 2^len(ch_mult)");
    if (!ws || ((uintptr_t)ws & 255)) return fail(CCN_EWORKSPACE, "workspace must be non-null and 256-byte aligned");
    for (size_t i = 0; i < h->plans.size(); ++i) {
        Plan* p = h->plans[i].get();
        if (p->B == B && p->H == H && p->W == W && p->steps == steps && p->ws == ws) {
            if (ws_bytes < p->bytes) return fail(CCN_EWORKSPACE, "workspace too small");
            // least recently used first: a hit moves the plan to the back, eviction (below) takes the front
            std::rotate(h->plans.begin() + (long)i, h->plans.begin() + (long)i + 1, h->plans.end());
            *out = p;
            return CCN_OK;
        }
    }
    std::unique_ptr<Plan> p(new Plan);
    p->B = B; p->H = H; p->W = W; p->steps = steps; p->ws = ws;
    int rc = build_plan(h, p.get(), ws, false);
    if (rc) return rc;
    if (ws_bytes < p->bytes) return fail(CCN_EWORKSPACE, "workspace too small: need " + std::to_string(p->bytes));
    HIPCHK(hipMemset(p->counters, 0, (size_t)kMaxNorms * B * 4));          // arrival counters start (and are left) at zero
    for (auto& z : p->zero_once) HIPCHK(hipMemset(z.first, 0, z.second));
    if (h->plans.size() >= 8) drop_plans(h, 7);
    *out = p.get();
    h->plans.push_back(std::move(p));
    return CCN_OK;
}

}  // namespace
int ccn_release_workspace(ccn_handle_t h, void* workspace_dev)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    bool any = false;
    for (auto& p : h->plans) any = any || p->ws == workspace_dev;
    if (!any) return CCN_OK;
    HIPCHK(hipDeviceSynchronize());                                 // replays on caller streams may still be using it
    h->plans.erase(std::remove_if(h->plans.begin(), h->plans.end(), [&](const std::unique_ptr<Plan>& p) { return p->ws == workspace_dev; }),
                   h->plans.end());
    return CCN_OK;
}
namespace {

// ---- running a plan -------------------------------------------------------------------------------------------
int ensure_events(ccn_handle_s* h, size_t n)
{
    while (h->ev_pool.size() < n) {
        hipEvent_t e;
        HIPCHK(hipEventCreate(&e));
        h->ev_pool.push_back(e);
    }
    return CCN_OK;
}

int run_launch(ccn_handle_s* h, const Launch& L, hipStream_t s, const StepCtx& c)
{
    if (h->profiling) {
        int rc = ensure_events(h, h->ev_used + 2);
        if (rc) return rc;
        const int e0 = (int)h->ev_used, e1 = e0 + 1;
        h->ev_used += 2;
        HIPCHK(hipEventRecord(h->ev_pool[e0], s));
        HIPCHK(L.fn(s, c));
        HIPCHK(hipEventRecord(h->ev_pool[e1], s));
        h->recs.push_back({L.family, L.flops, L.bytes, e0, e1});
        return CCN_OK;
    }
    HIPCHK(L.fn(s, c));
    return CCN_OK;
}

// conditioning for `TR` time rows and `FR` FiLM rows; sample: FR = S*B rows (s,b); forward: FR = B rows
int run_conditioning(ccn_handle_s* h, Plan* p, hipStream_t s, const int64_t* t_i64, int TR, int FR, int a_div)
{
    const ccn_config_t& c = h->cfg;
    const int td = c.time_dim, B = p->B;
    std::vector<Launch> v;
    const float* z = p->zbuf;
    v.push_back({F_COND, 0, 0, [=](hipStream_t st, const StepCtx&) {
        return t_i64 ? launch_temb_i64(t_i64, p->temb, TR, td, st) : launch_temb_i32(p->ts_dev, p->temb, TR, td, st); }});
    v.push_back({F_COND, 2.0 * TR * td * 4.0 * td, 0, [=](hipStream_t st, const StepCtx&) {
        return launch_linear(p->temb, nullptr, 1, TR, h->tp0_w, h->tp0_b, p->t1, TR, td, 4 * td, 1, st); }});
    v.push_back({F_COND, 2.0 * TR * td * 4.0 * td, 0, [=](hipStream_t st, const StepCtx&) {
        return launch_linear(p->t1, nullptr, 1, TR, h->tp2_w, h->tp2_b, p->tp, TR, 4 * td, td, 0, st); }});
    v.push_back({F_COND, 2.0 * B * c.z_dim * td, 0, [=](hipStream_t st, const StepCtx&) {
        return launch_linear(z, nullptr, 1, B, h->zp_w, h->zp_b, p->zp, B, c.z_dim, td, 1, st); }});
    v.push_back({F_COND, 2.0 * FR * td * (double)h->F, 0, [=](hipStream_t st, const StepCtx&) {
        return launch_linear(p->tp, p->zp, a_div, B, h->film_w, h->film_b, p->film, FR, td, h->F, 0, st); }});
    StepCtx c0;
    for (auto& L : v) { int rc = run_launch(h, L, s, c0); if (rc) return rc; }
    return CCN_OK;
}

int run_steps(ccn_handle_s* h, Plan* p, hipStream_t s, int steps, const float* coef, const float* sigma = nullptr, const float* noise = nullptr)
{
    const size_t img = (size_t)p->B * h->cfg.img_ch * p->H * p->W;
    for (int i = 0; i < steps; ++i) {
        StepCtx c;
        c.step = i; c.x_in = p->xstate; c.x_state = p->xstate; c.eps_out = nullptr; c.do_ddim = 1;
        for (int k = 0; k < 4; ++k) c.c[k] = coef[i * 4 + k];
        if (sigma && noise && sigma[i] > 0.f) { c.noise = noise + (size_t)i * img; c.sigma = sigma[i]; }
        for (auto& L : p->ops) { int rc = run_launch(h, L, s, c); if (rc) return rc; }
    }
    return CCN_OK;
}

// sticky device-side failure (a kernel ORed into the error word since the last check): report once, then clear
int check_device_errors(ccn_handle_s* h)
{
    if (!h->err_host) return CCN_OK;
    const unsigned e = __atomic_exchange_n(h->err_host, 0u, __ATOMIC_RELAXED);
    if (e & 1u)
        return fail(CCN_EHIP, "device-side hand-off timeout: a split-K partial tile never arrived; the results of the launches enqueued "
                              "since the last successful check are invalid");
    if (e & 2u)
        return fail(CCN_EHIP, "split-K partners ran on different XCDs (the hand-off took the agent-scope path: results are valid, but the "
                              "placement assumption of launch_conv_pr does not hold on this device / partition mode)");
    if (e) return fail(CCN_EHIP, "device-side error word " + std::to_string(e));
    return CCN_OK;
}

int check_ready(ccn_handle_s* h)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    if (!h->committed) return fail(CCN_ESTATE, "ccn_commit_params has not succeeded on this handle");
    return check_device_errors(h);
}

}  // namespace

// =================================================================================================================
extern "C" {

const char* ccn_last_error(void) { return g_err.c_str(); }
#ifdef CCN_DIAG
const char* ccn_version(void) { return "ccn_hip 0.3 (gfx950, diagnostics build)"; }
#else
const char* ccn_version(void) { return "ccn_hip 0.3 (gfx950)"; }
#endif

int ccn_create(const ccn_config_t* cfg, ccn_handle_t* out)
{
    if (!cfg || !out) return fail(CCN_EINVAL, "null argument");
    if (cfg->n_mult < 1 || cfg->n_mult > CCN_MAX_MULT) return fail(CCN_EINVAL, "n_mult out of range");
    if (cfg->dtype != CCN_DTYPE_F32 && cfg->dtype != CCN_DTYPE_BF16) return fail(CCN_EINVAL, "unknown dtype");
    if (cfg->base <= 0 || cfg->base % 8) return fail(CCN_EINVAL, "base must be a positive multiple of 8");
    if (cfg->img_ch < 1 || cfg->img_ch * 9 > 32) return fail(CCN_EINVAL, "img_ch must be 1..3");
    if (cfg->time_dim <= 0 || cfg->z_dim <= 0 || cfg->groups <= 0) return fail(CCN_EINVAL, "bad dims");
    int ch = cfg->base;
    for (int i = 0; i < cfg->n_mult; ++i) {
        if (cfg->ch_mult[i] < 1) return fail(CCN_EINVAL, "ch_mult entries must be >= 1");
        if ((long)ch * cfg->ch_mult[i] > (1 << 16)) return fail(CCN_EINVAL, "channel width (running product of ch_mult, models/unet.py:64) above 65536");
        ch *= cfg->ch_mult[i];
    }
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return fail(CCN_EHIP, "no HIP device");
    std::unique_ptr<ccn_handle_s> h(new ccn_handle_s);
    h->cfg = *cfg;
    h->elem = cfg->dtype == CCN_DTYPE_BF16 ? 2 : 4;
    h->G = cfg->groups;
    build_arch(h.get());
    for (auto& r : h->res)
        if (r.C % (r.C < h->G ? r.C : h->G)) return fail(CCN_EINVAL, "channel count not divisible by GroupNorm groups");
    HIPCHK(conv_prepare());
    HIPCHK(hipStreamCreateWithFlags(&h->cap_stream, hipStreamNonBlocking));
    HIPCHK(hipHostMalloc((void**)&h->err_host, 64, hipHostMallocMapped));
    *h->err_host = 0u;
    HIPCHK(hipHostGetDevicePointer((void**)&h->err_dev, h->err_host, 0));
    for (int i = 0; i < F_COUNT; ++i) h->fam_names.push_back(kFamilies[i]);
    *out = h.release();
    return CCN_OK;
}

int ccn_destroy(ccn_handle_t h)
{
    if (!h) return CCN_OK;
    (void)hipDeviceSynchronize();
    h->plans.clear();
    for (void* p : h->dev_allocs) (void)hipFree(p);
    for (auto e : h->ev_pool) (void)hipEventDestroy(e);
    if (h->cap_stream) (void)hipStreamDestroy(h->cap_stream);
    if (h->err_host) (void)hipHostFree(h->err_host);
    delete h;
    return CCN_OK;
}

int ccn_num_params(ccn_handle_t h, int32_t* n)
{
    if (!h || !n) return fail(CCN_EINVAL, "null argument");
    *n = (int32_t)h->params.size();
    return CCN_OK;
}

int ccn_param_info(ccn_handle_t h, int32_t i, const char** name, int64_t shape[4], int32_t* ndim)
{
    if (!h || !name || !shape || !ndim) return fail(CCN_EINVAL, "null argument");
    if (i < 0 || i >= (int32_t)h->params.size()) return fail(CCN_EINVAL, "parameter index out of range");
    const ParamInfo& p = h->params[i];
    *name = p.name.c_str();
    *ndim = (int32_t)p.shape.size();
    for (size_t k = 0; k < 4; ++k) shape[k] = k < p.shape.size() ? p.shape[k] : 1;
    return CCN_OK;
}

int ccn_load_param(ccn_handle_t h, const char* name, const float* data, const int64_t* shape, int32_t ndim)
{
    if (!h || !name || !data || !shape) return fail(CCN_EINVAL, "null argument");
    const ParamInfo* pi = nullptr;
    for (auto& p : h->params) if (p.name == name) { pi = &p; break; }
    if (!pi) return fail(CCN_EWEIGHTS, std::string("Unexpected key in state_dict: ") + name);
    if ((size_t)ndim != pi->shape.size()) return fail(CCN_EWEIGHTS, std::string("size mismatch for ") + name);
    for (int k = 0; k < ndim; ++k)
        if (shape[k] != pi->shape[k]) return fail(CCN_EWEIGHTS, std::string("size mismatch for ") + name);
    std::vector<float> buf(pi->numel());
    hipPointerAttribute_t attr;
    const hipError_t pe = hipPointerGetAttributes(&attr, data);
    if (pe == hipSuccess && attr.type == hipMemoryTypeDevice) {
        HIPCHK(hipMemcpy(buf.data(), data, buf.size() * 4, hipMemcpyDeviceToHost));
    } else {
        (void)hipGetLastError();
        std::memcpy(buf.data(), data, buf.size() * 4);
    }
    h->host[name] = std::move(buf);
    h->committed = false;
    return CCN_OK;
}

int ccn_set_weight_rounding(ccn_handle_t h, int32_t mode)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    if (mode != CCN_ROUND_NEAREST && mode != CCN_ROUND_DIFFUSED && mode != CCN_ROUND_DIFFUSED_PHASES) return fail(CCN_EINVAL, "unknown weight rounding mode");
    h->weight_rounding = mode;
    return CCN_OK;
}

int ccn_commit_params(ccn_handle_t h)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    std::string missing;
    for (auto& p : h->params)
        if (!h->host.count(p.name)) missing += (missing.empty() ? "" : ", ") + p.name;
    if (!missing.empty()) return fail(CCN_EWEIGHTS, "Missing key(s) in state_dict: " + missing);
    drop_plans(h, 0);                                             // drains the device first: replays may still read the old weights
    for (void* p : h->dev_allocs) (void)hipFree(p);
    h->dev_allocs.clear();
    const ccn_config_t& c = h->cfg;
    int rc;
    for (auto& m : h->host_ph) m.clear();
    const bool phases = c.dtype == CCN_DTYPE_BF16 && h->weight_rounding == CCN_ROUND_DIFFUSED_PHASES;
    h->nphase = phases ? phases_for(h) : 1;
    if (c.dtype == CCN_DTYPE_BF16 && h->weight_rounding != CCN_ROUND_NEAREST) {
        // (the head keeps fp32 weights: head_prep_kernel scales them per sample and rounds them with a carry along the steps)
        for (auto& p : h->params) {
            if (p.shape.size() != 4 || p.name == "out.weight") continue;
            if (phases) { diffuse_round_phases(h, p); continue; }
            auto& w = h->host.at(p.name);
            const bool convT = p.shape[2] == 4;                               // up.N.weight: (Cin, Cout, 4, 4)
            if (convT) diffuse_round_convT(w, (int)p.shape[0], (int)p.shape[1]);
            else diffuse_round_conv(w, (int)p.shape[0], (int)p.shape[1], (int)(p.shape[2] * p.shape[3]));
        }
    }
    // pack version k >= 1 of a conv's weights with the packer that made version 0 (a scratch ConvW receives the device pointers)
    auto pack_phases = [&](ConvW& cw, const std::string& name, int (*packer)(ccn_handle_s*, ConvW&, const std::string&)) -> int {
        cw.nphase = 1;
        if (!phases) return CCN_OK;
        for (int k = 1; k < h->nphase; ++k) {
            std::swap(h->host.at(name + ".weight"), h->host_ph[k - 1].at(name + ".weight"));
            ConvW t = cw;
            const int rc = packer(h, t, name);
            std::swap(h->host.at(name + ".weight"), h->host_ph[k - 1].at(name + ".weight"));
            if (rc) return rc;
            cw.w_ph[k - 1] = t.w; cw.wfrag_ph[k - 1] = t.wfrag;
        }
        cw.nphase = h->nphase;
        return CCN_OK;
    };
    h->stem = ConvW(); h->stem.kind = KIND_STEM; h->stem.Cin = c.img_ch; h->stem.Cout = c.base;
    if ((rc = pack_stem(h, h->stem, "in_conv"))) return rc;
    if ((rc = pack_phases(h->stem, "in_conv", pack_stem))) return rc;
    for (auto& r : h->res) {
        r.c1 = ConvW(); r.c1.kind = KIND_C3S1; r.c1.Cin = r.C; r.c1.Cout = r.C;
        r.c2 = r.c1;
        if ((rc = pack_conv3(h, r.c1, r.prefix + ".conv1"))) return rc;
        if ((rc = pack_conv3(h, r.c2, r.prefix + ".conv2"))) return rc;
        if ((rc = pack_phases(r.c1, r.prefix + ".conv1", pack_conv3))) return rc;
        if ((rc = pack_phases(r.c2, r.prefix + ".conv2", pack_conv3))) return rc;
        r.n1.C = r.n2.C = r.C;
        if ((rc = upload_f32(h, r.prefix + ".norm1.weight", &r.n1.gamma))) return rc;
        if ((rc = upload_f32(h, r.prefix + ".norm1.bias", &r.n1.beta))) return rc;
        if ((rc = upload_f32(h, r.prefix + ".norm2.weight", &r.n2.gamma))) return rc;
        if ((rc = upload_f32(h, r.prefix + ".norm2.bias", &r.n2.beta))) return rc;
    }
    for (size_t i = 0; i < h->downs.size(); ++i) {
        if ((rc = pack_conv3(h, h->downs[i], "down." + std::to_string(3 * i + 2)))) return rc;
        if ((rc = pack_phases(h->downs[i], "down." + std::to_string(3 * i + 2), pack_conv3))) return rc;
    }
    for (size_t i = 0; i < h->ups.size(); ++i) {
        if ((rc = pack_convT(h, h->ups[i], "up." + std::to_string(3 * i + 2)))) return rc;
        if ((rc = pack_phases(h->ups[i], "up." + std::to_string(3 * i + 2), pack_convT))) return rc;
    }
    h->head = ConvW(); h->head.kind = KIND_HEAD; h->head.Cin = c.base; h->head.Cout = c.img_ch;
    if ((rc = pack_conv3(h, h->head, "out"))) return rc;
    if ((rc = upload_f32(h, "out.weight", &h->head_w_f32))) return rc;
    h->out_norm.C = c.base;
    if ((rc = upload_f32(h, "out_norm.weight", &h->out_norm.gamma))) return rc;
    if ((rc = upload_f32(h, "out_norm.bias", &h->out_norm.beta))) return rc;
    if ((rc = upload_f32(h, "time_proj.0.weight", &h->tp0_w))) return rc;
    if ((rc = upload_f32(h, "time_proj.0.bias", &h->tp0_b))) return rc;
    if ((rc = upload_f32(h, "time_proj.2.weight", &h->tp2_w))) return rc;
    if ((rc = upload_f32(h, "time_proj.2.bias", &h->tp2_b))) return rc;
    if ((rc = upload_f32(h, "z_proj.0.weight", &h->zp_w))) return rc;
    if ((rc = upload_f32(h, "z_proj.0.bias", &h->zp_b))) return rc;
    // all FiLM linears as one (F x time_dim) matrix: per block [to_scale rows | to_shift rows]
    std::vector<float> fw((size_t)h->F * c.time_dim), fb((size_t)h->F);
    for (auto& r : h->res) {
        const char* parts[2] = {".film.to_scale", ".film.to_shift"};
        for (int q = 0; q < 2; ++q) {
            const auto& w = h->host.at(r.prefix + parts[q] + ".weight");
            const auto& b = h->host.at(r.prefix + parts[q] + ".bias");
            std::memcpy(&fw[(size_t)(r.film_off + q * r.C) * c.time_dim], w.data(), w.size() * 4);
            std::memcpy(&fb[(size_t)(r.film_off + q * r.C)], b.data(), b.size() * 4);
        }
    }
    if ((rc = upload(h, fw.data(), fw.size() * 4, (void**)&h->film_w))) return rc;
    if ((rc = upload(h, fb.data(), fb.size() * 4, (void**)&h->film_b))) return rc;
    HIPCHK(hipDeviceSynchronize());
    h->host.clear();
    for (auto& m : h->host_ph) m.clear();
    h->committed = true;
    return CCN_OK;
}

int ccn_workspace_bytes(ccn_handle_t h, int32_t B, int32_t H, int32_t W, int32_t steps, size_t* bytes)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!bytes) return fail(CCN_EINVAL, "null argument");
    if (B <= 0 || H <= 0 || W <= 0 || steps <= 0) return fail(CCN_EINVAL, "B, H, W, steps must be positive");
    const int div = 1 << h->cfg.n_mult;
    if (H % div || W % div) return fail(CCN_EINVAL, "H and W must be divisible by 2^len(ch_mult)");
    Plan p; p.B = B; p.H = H; p.W = W; p.steps = steps;
    rc = build_plan(h, &p, nullptr, true);
    if (rc) return rc;
    *bytes = p.bytes;
    return CCN_OK;
}

int ccn_forward(ccn_handle_t h, const float* x_dev, const float* z_dev, const int64_t* t_dev, float* eps_dev,
                int32_t B, int32_t H, int32_t W, void* workspace_dev, size_t workspace_bytes, void* stream)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!x_dev || !z_dev || !t_dev || !eps_dev) return fail(CCN_EINVAL, "null tensor pointer");
    Plan* p = nullptr;
    if ((rc = get_plan(h, B, H, W, 1, workspace_dev, workspace_bytes, &p))) return rc;
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipMemcpyAsync(p->zbuf, z_dev, (size_t)B * h->cfg.z_dim * 4, hipMemcpyDeviceToDevice, s));
    if ((rc = run_conditioning(h, p, s, t_dev, B, B, 1))) return rc;
    StepCtx c;
    c.step = 0; c.x_in = x_dev; c.x_state = nullptr; c.eps_out = eps_dev; c.do_ddim = 0;
    for (auto& L : p->ops) if ((rc = run_launch(h, L, s, c))) return rc;
    return CCN_OK;
}

static int sample_impl(ccn_handle_t h, const float* z_dev, const float* x_T_dev, float* x_out_dev, int32_t B, int32_t H, int32_t W,
                       int32_t steps, const int32_t* ts_host, const float* coef_host, const float* sigma_host, const float* noise_dev,
                       void* workspace_dev, size_t workspace_bytes, void* stream, int32_t use_graph)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!z_dev || !x_T_dev || !x_out_dev || !ts_host || !coef_host) return fail(CCN_EINVAL, "null pointer");
    Plan* p = nullptr;
    if ((rc = get_plan(h, B, H, W, steps, workspace_dev, workspace_bytes, &p))) return rc;
    hipStream_t s = (hipStream_t)stream;
    const size_t img_bytes = (size_t)B * h->cfg.img_ch * H * W * 4;
    HIPCHK(hipMemcpyAsync(p->zbuf, z_dev, (size_t)B * h->cfg.z_dim * 4, hipMemcpyDeviceToDevice, s));
    if (x_T_dev != p->xstate) HIPCHK(hipMemcpyAsync(p->xstate, x_T_dev, img_bytes, hipMemcpyDeviceToDevice, s));

    // the timestep table of a plan almost never changes: upload it only when it does, and then synchronously after draining
    // the stream (an async copy out of a host vector that the next call rewrites could still be pending)
    if (p->ts_keep.size() != (size_t)steps || std::memcmp(p->ts_keep.data(), ts_host, (size_t)steps * 4)) {
        HIPCHK(hipStreamSynchronize(s));
        p->ts_keep.assign(ts_host, ts_host + steps);
        HIPCHK(hipMemcpy(p->ts_dev, p->ts_keep.data(), (size_t)steps * 4, hipMemcpyHostToDevice));
    }
    if (use_graph && !h->profiling) {
        GraphEntry* ge = nullptr;
        for (auto& g : p->graphs)
            if (g.steps == steps && !std::memcmp(g.ts.data(), ts_host, (size_t)steps * 4) &&
                !std::memcmp(g.coef.data(), coef_host, (size_t)steps * 16) && g.noise == noise_dev &&
                g.sigma.size() == (sigma_host ? (size_t)steps : 0) &&
                (!sigma_host || !std::memcmp(g.sigma.data(), sigma_host, (size_t)steps * 4))) { ge = &g; break; }
        if (!ge) {
            GraphEntry g;
            g.steps = steps; g.ts.assign(ts_host, ts_host + steps); g.coef.assign(coef_host, coef_host + steps * 4);
            if (sigma_host) g.sigma.assign(sigma_host, sigma_host + steps);
            g.noise = noise_dev;
            HIPCHK(hipStreamBeginCapture(h->cap_stream, hipStreamCaptureModeRelaxed));
            rc = run_conditioning(h, p, h->cap_stream, nullptr, steps, steps * B, B);
            if (!rc) rc = run_steps(h, p, h->cap_stream, steps, coef_host, sigma_host, noise_dev);
            hipGraph_t graph = nullptr;
            const hipError_t ce = hipStreamEndCapture(h->cap_stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            if (ce != hipSuccess) return fail(CCN_EHIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            g.graph = graph;
            HIPCHK(hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
            // at most kMaxGraphsPerPlan captured graphs per plan, least recently used first (a caller that passes a fresh noise
            // tensor / coefficient table every call would otherwise grow the list without bound); an exec may still be replaying
            // on a caller stream, so the device is drained before one is destroyed
            if (p->graphs.size() >= kMaxGraphsPerPlan) {
                HIPCHK(hipDeviceSynchronize());
                GraphEntry& old = p->graphs.front();
                if (old.exec) (void)hipGraphExecDestroy(old.exec);
                if (old.graph) (void)hipGraphDestroy(old.graph);
                p->graphs.erase(p->graphs.begin());
            }
            p->graphs.push_back(g);
            ge = &p->graphs.back();
        } else if (ge != &p->graphs.back()) {
            std::rotate(p->graphs.begin() + (ge - p->graphs.data()), p->graphs.begin() + (ge - p->graphs.data()) + 1, p->graphs.end());
            ge = &p->graphs.back();
        }
        HIPCHK(hipGraphLaunch(ge->exec, s));
    } else {
        if ((rc = run_conditioning(h, p, s, nullptr, steps, steps * B, B))) return rc;
        if ((rc = run_steps(h, p, s, steps, coef_host, sigma_host, noise_dev))) return rc;
    }
    if (x_out_dev != p->xstate) HIPCHK(hipMemcpyAsync(x_out_dev, p->xstate, img_bytes, hipMemcpyDeviceToDevice, s));
    return CCN_OK;
}

int ccn_sample(ccn_handle_t h, const float* z_dev, const float* x_T_dev, float* x_out_dev, int32_t B, int32_t H, int32_t W,
               int32_t steps, const int32_t* ts_host, const float* coef_host, void* workspace_dev, size_t workspace_bytes,
               void* stream, int32_t use_graph)
{
    return sample_impl(h, z_dev, x_T_dev, x_out_dev, B, H, W, steps, ts_host, coef_host, nullptr, nullptr, workspace_dev, workspace_bytes, stream, use_graph);
}

int ccn_sample_eta(ccn_handle_t h, const float* z_dev, const float* x_T_dev, float* x_out_dev, int32_t B, int32_t H, int32_t W,
                   int32_t steps, const int32_t* ts_host, const float* coef_host, const float* sigma_host, const float* noise_dev,
                   void* workspace_dev, size_t workspace_bytes, void* stream, int32_t use_graph)
{
    if (!sigma_host || !noise_dev) return fail(CCN_EINVAL, "null sigma / noise");
    return sample_impl(h, z_dev, x_T_dev, x_out_dev, B, H, W, steps, ts_host, coef_host, sigma_host, noise_dev, workspace_dev, workspace_bytes, stream, use_graph);
}

int ccn_ddim_step(float* x_dev, const float* eps_dev, const float* noise_dev, float c0, float c1, float c2, float c3,
                  float sigma, int64_t n, void* stream)
{
    if (!x_dev || !eps_dev || n < 0) return fail(CCN_EINVAL, "bad argument");
    if (n == 0) return CCN_OK;
    HIPCHK(launch_ddim_step(x_dev, eps_dev, noise_dev, c0, c1, c2, c3, sigma, n, (hipStream_t)stream));
    return CCN_OK;
}

int ccn_q_sample(float* out_dev, const float* x0_dev, const float* noise_dev, const float* a_dev, const float* s_dev,
                 int32_t B, int64_t per_sample, void* stream)
{
    if (!out_dev || !x0_dev || !noise_dev || !a_dev || !s_dev || B <= 0 || per_sample <= 0) return fail(CCN_EINVAL, "bad argument");
    HIPCHK(launch_q_sample(out_dev, x0_dev, noise_dev, a_dev, s_dev, B, per_sample, (hipStream_t)stream));
    return CCN_OK;
}

int ccn_predict_x0(float* out_dev, const float* x_t_dev, const float* eps_dev, const float* a_dev, const float* s_dev,
                   int32_t B, int64_t per_sample, void* stream)
{
    if (!out_dev || !x_t_dev || !eps_dev || !a_dev || !s_dev || B <= 0 || per_sample <= 0) return fail(CCN_EINVAL, "bad argument");
    HIPCHK(launch_predict_x0(out_dev, x_t_dev, eps_dev, a_dev, s_dev, B, per_sample, (hipStream_t)stream));
    return CCN_OK;
}

int ccn_timestep_embedding(const int64_t* t_dev, float* out_dev, int32_t n, int32_t dim, void* stream)
{
    if (!t_dev || !out_dev || n <= 0 || dim <= 0) return fail(CCN_EINVAL, "bad argument");
    HIPCHK(launch_temb_i64(t_dev, out_dev, n, dim, (hipStream_t)stream));
    return CCN_OK;
}

int ccn_film_forward(const float* x_dev, const float* h_dev, const float* ws_dev, const float* bs_dev, const float* wh_dev,
                     const float* bh_dev, float* y_dev, int32_t B, int32_t C, int32_t H, int32_t W, int32_t D,
                     float* scratch_dev, void* stream)
{
    if (!x_dev || !h_dev || !ws_dev || !bs_dev || !wh_dev || !bh_dev || !y_dev || !scratch_dev) return fail(CCN_EINVAL, "null pointer");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || D <= 0) return fail(CCN_EINVAL, "bad shape");
    hipStream_t s = (hipStream_t)stream;
    float* sc = scratch_dev; float* sh = scratch_dev + (size_t)B * C;
    HIPCHK(launch_linear(h_dev, nullptr, 1, B, ws_dev, bs_dev, sc, B, D, C, 0, s));
    HIPCHK(launch_linear(h_dev, nullptr, 1, B, wh_dev, bh_dev, sh, B, D, C, 0, s));
    HIPCHK(launch_film_nchw(x_dev, sc, sh, y_dev, B, C, (int64_t)H * W, s));
    return CCN_OK;
}

int ccn_resblock_forward(ccn_handle_t h, const char* prefix, const float* x_dev, const float* cond_dev, float* y_dev,
                         int32_t B, int32_t H, int32_t W, void* workspace_dev, size_t workspace_bytes, void* stream)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!prefix || !x_dev || !cond_dev || !y_dev || !workspace_dev) return fail(CCN_EINVAL, "null pointer");
    if (B <= 0 || H <= 0 || W <= 0) return fail(CCN_EINVAL, "bad shape");
    const ResW* r = nullptr;
    for (auto& q : h->res) if (q.prefix == prefix) { r = &q; break; }
    if (!r) return fail(CCN_EINVAL, std::string("no ResBlock with prefix ") + prefix);
    if ((uintptr_t)workspace_dev & 255) return fail(CCN_EWORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int G = (r->C < h->G ? r->C : h->G), cpg = r->C / G;
    int nslot = (H * W + 1023) / 1024; if (nslot < 1) nslot = 1;
    Plan tmp; tmp.B = B; tmp.H = H; tmp.W = W; tmp.steps = 1;
    TensorRef x, o;
    // pass 0 measures (nothing is written before the size is known to fit), pass 1 places the buffers
    for (int pass = 0; pass < 2; ++pass) {
        tmp.ops.clear(); tmp.named.clear(); tmp.zero_once.clear();
        PlanBuilder pb(h, &tmp, workspace_dev, pass == 0);
        pb.film_stride = 2 * r->C;                                // dense (B, 2C) table: [scale | shift] of this block only
        tmp.film = (float*)pb.bump.take((size_t)B * 2 * r->C * 4);
        tmp.counters = (unsigned*)pb.bump.take((size_t)8 * B * 4);
        tmp.n_counters = 0;
        x = pb.new_tensor(r->C, H, W);
        x.part = (float2*)pb.bump.take((size_t)B * G * nslot * sizeof(float2));
        x.n_sp = nslot; x.n_nt = 1; x.bn = 1 << 30;
        o = pb.resblock(*r, x, false, 0);
        if (pb.bump.off > workspace_bytes) return fail(CCN_EWORKSPACE, "workspace too small: need " + std::to_string(pb.bump.off));
    }
    HIPCHK(hipMemsetAsync(tmp.counters, 0, (size_t)8 * B * 4, s));
    for (auto& z : tmp.zero_once) HIPCHK(hipMemsetAsync(z.first, 0, z.second, s));   // split-K hand-off flags of a fresh workspace
    HIPCHK(launch_nchw_to_nhwc(h->cfg.dtype, x_dev, x.p, B, r->C, H, W, s));
    HIPCHK(launch_gn_partials(h->cfg.dtype, x.p, x.part, B, H * W, r->C, cpg, G, nslot, s));
    HIPCHK(launch_linear(cond_dev, nullptr, 1, B, h->film_w + (size_t)r->film_off * h->cfg.time_dim, h->film_b + r->film_off,
                         tmp.film, B, h->cfg.time_dim, 2 * r->C, 0, s));
    StepCtx c;
    for (auto& L : tmp.ops) HIPCHK(L.fn(s, c));
    HIPCHK(launch_nhwc_to_nchw(h->cfg.dtype, o.p, y_dev, B, r->C, H, W, s));
    return CCN_OK;
}

int ccn_read_activation(ccn_handle_t h, const char* name, float* out_dev, size_t out_elems, void* stream)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!name || !out_dev) return fail(CCN_EINVAL, "null pointer");
    if (h->plans.empty()) return fail(CCN_ESTATE, "no forward has run yet");
    Plan* p = h->plans.back().get();
    auto it = p->named.find(name);
    if (it == p->named.end()) return fail(CCN_EINVAL, std::string("unknown activation ") + name);
    const TensorRef& t = it->second;
    if (out_elems < (size_t)p->B * t.C * t.H * t.W) return fail(CCN_EINVAL, "output buffer too small");
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(launch_nhwc_to_nchw(h->cfg.dtype, t.p, out_dev, p->B, t.C, t.H, t.W, s));
    HIPCHK(hipStreamSynchronize(s));
    return CCN_OK;
}

int ccn_poll_errors(ccn_handle_t h)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    return check_device_errors(h);
}

int ccn_profile_enable(ccn_handle_t h, int32_t on)
{
    if (!h) return fail(CCN_EINVAL, "null handle");
    h->profiling = on != 0;
    h->recs.clear();
    h->ev_used = 0;
    return CCN_OK;
}

int ccn_profile_read(ccn_handle_t h, const char** names, float* ms, int32_t* calls, double* flops, double* bytes,
                     int32_t cap, int32_t* n)
{
    if (!h || !names || !ms || !calls || !flops || !bytes || !n) return fail(CCN_EINVAL, "null pointer");
    HIPCHK(hipDeviceSynchronize());
    std::vector<double> t(F_COUNT, 0.0), fl(F_COUNT, 0.0), by(F_COUNT, 0.0);
    std::vector<int> cnt(F_COUNT, 0);
    for (auto& r : h->recs) {
        float e = 0.f;
        HIPCHK(hipEventElapsedTime(&e, h->ev_pool[r.e0], h->ev_pool[r.e1]));
        t[r.family] += e; fl[r.family] += r.flops; by[r.family] += r.bytes; cnt[r.family]++;
    }
    int k = 0;
    for (int f = 0; f < F_COUNT && k < cap; ++f) {
        if (!cnt[f]) continue;
        names[k] = h->fam_names[f].c_str(); ms[k] = (float)t[f]; calls[k] = cnt[f]; flops[k] = fl[f]; bytes[k] = by[f];
        ++k;
    }
    *n = k;
    h->recs.clear();
    h->ev_used = 0;
    return CCN_OK;
}

int ccn_algorithmic_work(ccn_handle_t h, int32_t B, int32_t H, int32_t W, double* flops, double* bytes)
{
    int rc = check_ready(h);
    if (rc) return rc;
    if (!flops || !bytes) return fail(CCN_EINVAL, "null pointer");
    Plan p; p.B = B; p.H = H; p.W = W; p.steps = 1;
    if ((rc = build_plan(h, &p, nullptr, true))) return rc;
    double f = 0, b = 0;
    for (auto& L : p.ops) if (L.family <= F_HEAD) { f += L.flops; b += L.bytes; }
    const ccn_config_t& c = h->cfg;
    f += 2.0 * B * ((double)c.time_dim * 4 * c.time_dim * 2 + (double)c.z_dim * c.time_dim + (double)c.time_dim * h->F);
    *flops = f; *bytes = b;
    return CCN_OK;
}

}  // extern "C"
