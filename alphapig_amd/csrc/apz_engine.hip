// libalphapig_hip.so -- C ABI (include/alphapig_hip.h) over the gfx950 kernels.
// Host-side glue only: parameter table, BatchNorm folding + weight packing, buffer ownership,
// launch sequencing on one HIP stream, HIP-event timing hooks.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "alphapig_hip.h"
#include "conv3x3_mfma.h"
#include "heads.h"
#include "trunk15_ring.h"
#include "trunk15_wino3.h"
#include "trunk15_wino3s.h"
#include "trunk15_wino3b.h"
#include "trunk15_wino3h.h"
#include "conv8_split.h"
#include "conv8_small.h"
#include "wgrad_wino3.h"
#include "sampler.h"
#include "conv_train.h"
#include "heads_train.h"
#include "weights_pack.h"

namespace {

typedef std::lock_guard<std::recursive_mutex> EngineLock;

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(APZ_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));                  \
    } while (0)

constexpr double BN_EPS = 1e-3;   // MXNet BatchNorm default (policy_value_net_mxnet.py:54,78,81)

struct Param {
    std::string name;
    int64_t size;
};

struct ConvLayer {
    std::string name;       // conv parameter prefix
    std::string bn;         // BN parameter prefix
    std::string mean_sfx, var_sfx;
    bool fix_gamma;
    int cin, cin_pad, cout;
    bool residual;          // add the block input before ReLU
    float* wpk = nullptr;
    float* wpk12 = nullptr; // 8x8 boards (conv8_kernel): [cot][c4][lane][12]
    float* upk2 = nullptr;  // trunk15_wino3_kernel: transformed weights G g G^T, [cot][row half][c4][lane][20] (wino_common.h)
    float* upk3s = nullptr; // trunk15_wino3s_kernel: the same values, [cot][wave][c4][piece][lane][4] (WinoPackSmall)
    void* upk3b = nullptr;  // trunk15_wino3b_kernel (apz_set_trunk_arith(APZ_ARITH_BF16X3) only): U as three bf16 terms (Wino3B)
    void* upk3h = nullptr;  // trunk15_wino3h_kernel (APZ_ARITH_F16X2 only): U S[co] as two fp16 terms (Wino3H)
    float* bias3h = nullptr;   // ... and its [128 biases][128 x 1 / S[co]]
    void* wpk8h = nullptr;  // conv8h_kernel (8x8 boards, APZ_ARITH_F16X2, C_in a multiple of 64): w S[co] as two fp16 terms (Conv8H)
    float* bias8h = nullptr;   // ... and its [cout biases][cout x 1 / S[co]]
    float* bias = nullptr;
};

struct Pending {
    int cls;
    hipEvent_t a, b;
    int launches;   // kernel launches bracketed by the pair
};

}  // namespace

struct apz_engine {
    apz_config cfg;
    int hw = 0, code_stride = 0, num_cu = 256, cmax = 0, clast = 0;
    hipStream_t stream = nullptr;
    std::vector<Param> params;
    std::vector<ConvLayer> convs;
    bool loaded = false;
    // heads
    float *w6 = nullptr, *b6 = nullptr, *wfc_pk = nullptr, *bfc = nullptr, *wv = nullptr, *bv = nullptr;
    // device buffers
    float* act[3] = {nullptr, nullptr, nullptr};
    float *planes = nullptr, *featp = nullptr, *featv = nullptr, *probs = nullptr, *values = nullptr, *fc_logits = nullptr;
    unsigned char* codes = nullptr;
    int *perm_s = nullptr, *perm_p = nullptr;
    // host pinned staging (apz_forward_host)
    float *h_planes = nullptr, *h_probs = nullptr, *h_values = nullptr;
    unsigned char* h_codes = nullptr;
    int last_n = 0;
    // stream-ordered submission slots (apz_submit_codes / apz_wait)
    struct Slot {
        unsigned char* h_codes = nullptr;
        float *h_probs = nullptr, *h_values = nullptr;
        float *d_probs = nullptr, *d_values = nullptr;
        unsigned char* d_codes = nullptr;
        hipEvent_t done = nullptr;
        int n = 0;
        bool busy = false;
    } slots[APZ_MAX_SLOTS];
    // One lock for everything that touches the engine's buffers, stream, weights or timing state.  Recursive: the
    // public entry points call each other (apz_forward_codes_host -> _async -> apz_encode_planes).  The pipeline
    // workers of SelfPlayEngine submit from their own threads while the main thread may call policy_value_fn,
    // set_params or the arena -- every such call now queues behind the submissions instead of racing them.
    // apz_submit_codes: the forward of a small batch is tens of launches (one board on the 10-block net: 45), each a few
    // microseconds of host time -- the third submission of a (slot, batch size) pair captures the launch sequence into a
    // HIP graph (the slot's code / result buffers and the engine's activation buffers are fixed addresses) and later ones
    // replay it with one call.  Dropped whenever the weights or the kernel selection change.  OFF by default
    // (apz_set_forward_graphs): measured on ROCm 7.2 / MI355X (tools/graph_ab.py, profiles/r04_graph_ab.log) the replay is
    // SLOWER than the plain launches -- BASELINE config 2 (seven launches per 32-board forward) 337 k -> 310 k leaf
    // evaluations/s, one-board policy_value_fn (45 launches) 0.332 -> 0.338 ms.
    std::map<long, hipGraphExec_t> fwd_graphs;
    std::map<long, int> fwd_seen;
    bool use_graphs = false;
    std::recursive_mutex submit_lock;
    bool ring = false;      // 15x15 / 128-filter resnet: trunk activations in rows16 layout (trunk15_ring.h)
    bool small8 = false;    // 8x8 boards: conv8_kernel / head8_kernel (conv8_small.h)
    float* wfc_raw = nullptr;   // head8_kernel: the policy FullyConnected weight as stored, [hw][4 hw]
    int act_ps = 0, act_rs = 0;
    bool lds_attr_set[40] = {false};   // hipFuncSetAttribute(MaxDynamicSharedMemorySize) done, per kernel variant
    int conv_lds_set[16] = {0};
    // persistent sampler staging (apz_sample_moves_host)
    int32_t* smp_vis = nullptr;
    float* smp_pi = nullptr;
    int32_t* smp_mv = nullptr;
    uint64_t* smp_keys = nullptr;
    size_t smp_cap = 0;
    float* zeros256 = nullptr;          // bias stand-in for bias-free convolutions
    double* bn_part = nullptr;                     // apz_bn_fwd / _bwd: per-(channel, batch split) partial sums [256 * BN_SPLITS][2]
    float* wgw_scratch = nullptr;                  // apz_wgrad_wino: partial dU per batch slice
    double* fold_ws = nullptr;                     // apz_load_weights_dev: scale / shift of one layer (2 x 256 doubles)
    float* head_scratch = nullptr;                 // apz_conv1x1_bwd / apz_pv_loss: per-board partial sums
    size_t head_scratch_floats = 0;
    size_t wgw_floats = 0;                         // capacity of wgw_scratch
    void* adam_tab = nullptr;                      // apz_adam_step: device copy of the tensor table
    size_t adam_cap = 0;
    float* wino_scratch[2] = {nullptr, nullptr};   // apz_wino_conv: rows16 input / output copies
    size_t wino_scratch_boards = 0;
    bool wgrad_attr_set[2] = {false, false};
    hipStream_t scratch_stream = nullptr;          // the stream of the last entry point that may have used the scratch buffers
    bool scratch_stream_valid = false;
    float* w3s_slabs = nullptr;              // trunk15_wino3s_kernel: row partials of the position halves
    unsigned* w3s_tickets = nullptr;         // ... and the pairs' ticket words (each launch exchanges its epoch in: trunk15_wino3s.h)
    unsigned w3s_epoch = 0;                  // last epoch handed out; never 0, never repeated between two memsets of the words
    bool no_small_trunk = false;             // apz_test_select_trunk(APZ_TRUNK_WINOGRAD_BATCHED): tests compare the two forms
    bool no_quarter_trunk = false;           // apz_test_select_trunk(APZ_TRUNK_WINOGRAD_NO_QUARTER): 64-channel items for every batch
    int trunk_arith = APZ_ARITH_F32;         // apz_set_trunk_arith: APZ_ARITH_BF16X3 / _F16X2 = the split kernels for batches > 32
    // APZ_ARITH_F16X2: trunk15_wino3h_kernel raises a word when an activation left the fp16 range (a non-finite output);
    // the words live in pinned host memory, one per submission slot + one for the synchronous entry points; the entry
    // point that collects a forward's results looks at its word and repeats the forward on the exact-fp32 kernel.
    unsigned* ovf_host = nullptr;            // [APZ_MAX_SLOTS + 1]
    unsigned* ovf_dev = nullptr;             // the same words as the device sees them
    unsigned* ovf_cur = nullptr;             // (device pointer) the word of the forward being queued
    bool force_f32 = false;                  // the repeat of an overflowed forward
    long ovf_repeats = 0;                    // forwards repeated so far (apz_trunk_overflows)
    int trunk_kernel = APZ_TRUNK_WINOGRAD;   // or APZ_TRUNK_DIRECT (trunk15_ring_kernel): apz_test_select_trunk, tests only
    // profiling
    bool profiling = false;
    int prof_stride = 1, prof_phase = 0;   // time every prof_stride-th forward only
    bool prof_now = false;
    std::vector<Pending> pending;
    std::vector<hipEvent_t> free_events;
    double k_ms[APZ_K_COUNT] = {0};
    long k_cnt[APZ_K_COUNT] = {0};
};

namespace {

void add_param(apz_engine* e, const std::string& n, int64_t sz) { e->params.push_back({n, sz}); }

void build_tables(apz_engine* e) {
    const apz_config& c = e->cfg;
    const int hw = e->hw;
    auto conv_act = [&](const std::string& name, int cin, int cout, int k) {
        add_param(e, name + "_weight", (int64_t)cout * cin * k * k);
        add_param(e, name + "_bias", cout);
        add_param(e, name + "_gamma", cout);
        add_param(e, name + "_beta", cout);
        add_param(e, name + "_mean", cout);
        add_param(e, name + "_var", cout);
    };
    int last = 0;
    if (c.net_kind == APZ_NET_RESNET) {
        const int F = c.n_filter;
        conv_act("res_conv1", c.c_in, F, 3);
        e->convs.push_back({"res_conv1", "res_conv1", "_mean", "_var", true, c.c_in, (c.c_in + 3) / 4 * 4, F, false});
        for (int i = 1; i <= c.n_blocks; i++) {
            for (const char* ab : {"A", "B"}) {
                const std::string cn = std::string("conv") + ab + std::to_string(i);
                const std::string bn = std::string("bn") + ab + std::to_string(i);
                add_param(e, cn + "_weight", (int64_t)F * F * 9);
                add_param(e, cn + "_bias", F);
                add_param(e, bn + "_gamma", F);
                add_param(e, bn + "_beta", F);
                add_param(e, bn + "_moving_mean", F);
                add_param(e, bn + "_moving_var", F);
                e->convs.push_back({cn, bn, "_moving_mean", "_moving_var", false, F, F, F, ab[0] == 'B'});
            }
        }
        last = F;
    } else {
        static const char* names[6] = {"conv1", "conv2", "conv3", "conv4", "conv5", "conv_final"};
        static const int widths[6] = {64, 64, 128, 128, 256, 256};
        int prev = c.c_in;
        for (int i = 0; i < 6; i++) {
            conv_act(names[i], prev, widths[i], 3);
            e->convs.push_back({names[i], names[i], "_mean", "_var", true, prev, (prev + 3) / 4 * 4, widths[i], false});
            prev = widths[i];
        }
        last = prev;
    }
    conv_act("conv3_1_1", last, 4, 1);
    add_param(e, "fc_3_1_1_weight", (int64_t)hw * 4 * hw);
    add_param(e, "fc_3_1_1_bias", hw);
    conv_act("conv3_2_1", last, 2, 1);
    add_param(e, "fc_3_2_1_weight", 2 * hw);
    add_param(e, "fc_3_2_1_bias", 1);
    e->clast = last;
    e->cmax = 0;
    for (auto& l : e->convs) e->cmax = std::max(e->cmax, l.cout);
}

// scale/shift of an inference BatchNorm folded behind a conv with bias
void fold_bn(const std::map<std::string, const float*>& P, const std::string& conv, const std::string& bn,
             const std::string& mean_sfx, const std::string& var_sfx, bool fix_gamma, int cout,
             std::vector<double>& scale, std::vector<double>& shift) {
    const float* bias = P.at(conv + "_bias");
    const float* gamma = P.at(bn + "_gamma");
    const float* beta = P.at(bn + "_beta");
    const float* mean = P.at(bn + mean_sfx);
    const float* var = P.at(bn + var_sfx);
    scale.resize(cout);
    shift.resize(cout);
    for (int o = 0; o < cout; o++) {
        const double g = fix_gamma ? 1.0 : (double)gamma[o];
        const double s = g / std::sqrt((double)var[o] + BN_EPS);
        scale[o] = s;
        shift[o] = ((double)bias[o] - (double)mean[o]) * s + (double)beta[o];
    }
}

template <typename T>
int upload(T** dst, const std::vector<T>& src) {
    if (!*dst) HIP_TRY(hipMalloc((void**)dst, src.size() * sizeof(T)));
    HIP_TRY(hipMemcpy(*dst, src.data(), src.size() * sizeof(T), hipMemcpyHostToDevice));
    return APZ_OK;
}

hipEvent_t get_event(apz_engine* e) {
    if (!e->free_events.empty()) {
        hipEvent_t ev = e->free_events.back();
        e->free_events.pop_back();
        return ev;
    }
    hipEvent_t ev;
    // timing events only order work on this one device: no system-scope fence (cache write-back) when they fire
    if (hipEventCreateWithFlags(&ev, hipEventDisableSystemFence) != hipSuccess) hipEventCreate(&ev);
    return ev;
}

// Brackets one or MORE consecutive launches of one kernel class with a single event pair: an event between
// two kernels costs a barrier packet + cache maintenance (~0.1 ms each under load), so the 20 trunk launches of
// a sampled forward share one pair and their average duration is elapsed / launches.
struct Timed {
    apz_engine* e;
    int cls;
    int launches = 1;
    hipEvent_t a = nullptr, b = nullptr;
    Timed(apz_engine* e_, int cls_, bool always = false) : e(e_), cls(cls_) {
        if (e->profiling && (e->prof_now || always)) {
            a = get_event(e);
            b = get_event(e);
            hipEventRecord(a, e->stream);
        }
    }
    ~Timed() {
        if (a) {
            hipEventRecord(b, e->stream);
            e->pending.push_back({cls, a, b, launches});
        }
    }
};

// Recycle the pairs whose closing event has already fired (oldest first: one stream, events complete in order), so
// that a long profiled run keeps a handful of events alive instead of creating thousands.
void resolve_ready(apz_engine* e) {
    size_t k = 0;
    while (k < e->pending.size() && hipEventQuery(e->pending[k].b) == hipSuccess) {
        Pending& p = e->pending[k++];
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            e->k_ms[p.cls] += ms;
            e->k_cnt[p.cls] += p.launches;
        }
        e->free_events.push_back(p.a);
        e->free_events.push_back(p.b);
    }
    if (k) e->pending.erase(e->pending.begin(), e->pending.begin() + k);
}

void resolve_pending(apz_engine* e) {
    for (auto& p : e->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            e->k_ms[p.cls] += ms;
            e->k_cnt[p.cls] += p.launches;
        }
        e->free_events.push_back(p.a);
        e->free_events.push_back(p.b);
    }
    e->pending.clear();
}

template <int H, int W, int CT, bool RESID>
int launch_conv_r(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n,
                  int out_ps, int out_rs, int relu = 1, int cout_groups = 1) {
    using G = apz::ConvGeo<H, W>;
    // keep channel chunks a power-of-two-ish split of Cin: 256 ch at 15x15 -> 2 x 128
    int cchunk = L.cin_pad;
    while (cchunk > G::max_chunk()) cchunk = ((cchunk / 2) + 3) & ~3;
    const int lds = G::lds_bytes(cchunk);
    auto kern = apz::conv3x3_mfma_kernel<H, W, CT, RESID>;
    int& configured_lds = e->conv_lds_set[(H == 15 ? 0 : 8) + (CT == 1 ? 0 : CT == 2 ? 2 : 4) + (RESID ? 1 : 0)];
    if (lds > configured_lds) {
        HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        configured_lds = lds;
    }
    // persistent grid: as many workgroups as fit at once, each loops over boards
    int per_cu = std::max(1, std::min(4, (160 * 1024) / std::max(lds, 1)));
    int grid = std::min(n, e->num_cu * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid, cout_groups), dim3(256), lds, e->stream, in, L.wpk, L.bias, resid, out, n, L.cin,
                       L.cin_pad, cchunk, relu, out_ps, out_rs, L.cout);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

template <int H, int W, int CT>
int launch_conv_t(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    // the stem of the ring path hands its output to trunk15_ring.h in rows16 layout
    const bool to16 = e->ring && &L == &e->convs[0];
    const int ps = to16 ? e->act_ps : H * W, rs = to16 ? e->act_rs : W;
    // Small batches (BASELINE config 2: 64 concurrent games = 32 boards per launch): one workgroup per board leaves most
    // CUs idle and every workgroup streams the layer's whole weight set (2.4 MB at 256 x 256) through one CU's L2 port.
    // Then each board goes to CT workgroups of 64 output channels (gridDim.y): same arithmetic per output element, in
    // the same order -- only which wave owns which channel tile changes.
    if (CT > 1 && n * CT <= 2 * e->num_cu) {
        if (resid) return launch_conv_r<H, W, 1, true>(e, L, in, resid, out, n, ps, rs, 1, CT);
        return launch_conv_r<H, W, 1, false>(e, L, in, resid, out, n, ps, rs, 1, CT);
    }
    if (resid) return launch_conv_r<H, W, CT, true>(e, L, in, resid, out, n, ps, rs);
    return launch_conv_r<H, W, CT, false>(e, L, in, resid, out, n, ps, rs);
}

template <int NW>
int launch_trunk_ring_t(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    using T = apz::Trunk15;
    bool& configured = e->lds_attr_set[NW == 8 ? 1 : 0];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_ring_kernel<true, NW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_ring_kernel<false, NW>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        configured = true;
    }
    const int grid = std::min(n, e->num_cu);   // one persistent workgroup per CU (LDS-bound)
    if (resid)
        hipLaunchKernelGGL((apz::trunk15_ring_kernel<true, NW>), dim3(grid), dim3(64 * NW), T::LDS_BYTES, e->stream, in,
                           L.wpk, L.bias, resid, out, n);
    else
        hipLaunchKernelGGL((apz::trunk15_ring_kernel<false, NW>), dim3(grid), dim3(64 * NW), T::LDS_BYTES, e->stream, in,
                           L.wpk, L.bias, resid, out, n);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// trunk15_wino3_kernel addresses a launch's activations through 32-bit buffer offsets (and parks out-of-range lanes at
// offset 2^31), so one launch takes at most WINO3_MAX_BOARDS boards; larger batches go out as several launches on
// offset pointers -- a board's bits do not depend on the launch shape (tests/test_gpu_net.py).
constexpr int WINO3_MAX_BOARDS = 16384;   // even (board pairs), 16384 * 128 planes * 960 B = 2^31 - 2^27
static_assert((long long)WINO3_MAX_BOARDS * 128 * 960 < (1ll << 31), "wino3 buffer offsets");

template <bool RESID, bool RELU>
int launch_wino3_t(apz_engine* e, int attr_slot, const float* in, const float* upk, const float* bias, const float* resid,
                   float* out, int n, const float* upk_small = nullptr) {
    using T = apz::Wino3;
    bool& configured = e->lds_attr_set[attr_slot];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<RESID, RELU>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<RESID, RELU, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3s_kernel<RESID, RELU>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, apz::Wino3S::LDS_BYTES));
        configured = true;
    }
    if (upk_small && n <= apz::Wino3S::MAX_BOARDS && !e->no_small_trunk) {
        // the latency path: sixteen workgroups per board (16 output channels x half the positions), the same bits
        // (csrc/trunk15_wino3s.h); the halves meet through global slabs + ticket words keyed by a per-launch epoch
        if (!e->w3s_slabs) {
            HIP_TRY(hipMalloc((void**)&e->w3s_slabs, apz::Wino3S::slab_floats() * sizeof(float)));
            HIP_TRY(hipMalloc((void**)&e->w3s_tickets, apz::Wino3S::counters() * sizeof(unsigned)));
            e->w3s_epoch = 0;
        }
        if (++e->w3s_epoch == 0 || e->w3s_epoch == 1) {
            // first launch, or the 32-bit epoch wrapped: zero the words ON THE LAUNCH STREAM (ordered before the kernel)
            e->w3s_epoch = 1;
            HIP_TRY(hipMemsetAsync(e->w3s_tickets, 0, apz::Wino3S::counters() * sizeof(unsigned), e->stream));
        }
        hipLaunchKernelGGL((apz::trunk15_wino3s_kernel<RESID, RELU>), dim3(n * 16), dim3(256), apz::Wino3S::LDS_BYTES, e->stream, in,
                           upk_small, bias, RESID ? resid : nullptr, out, n, e->w3s_slabs, e->w3s_tickets, e->w3s_epoch);
        HIP_TRY(hipGetLastError());
        return APZ_OK;
    }
    for (int b0 = 0; b0 < n; b0 += WINO3_MAX_BOARDS) {
        const int nb = std::min(n - b0, WINO3_MAX_BOARDS);
        const size_t off = (size_t)b0 * T::C * T::GPLANE;
        bool quarter = false;                                 // few pairs (training batch 128, arena): four workgroups per pair
        const int grid = apz::wino3_grid(nb, e->num_cu, e->no_quarter_trunk ? nullptr : &quarter);   // persistent workgroups; item = board pair x channel half (quarter)
        if (quarter)
            hipLaunchKernelGGL((apz::trunk15_wino3_kernel<RESID, RELU, true>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream,
                               in + off, upk, bias, RESID ? resid + off : nullptr, out + off, nb);
        else
            hipLaunchKernelGGL((apz::trunk15_wino3_kernel<RESID, RELU>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream, in + off,
                               upk, bias, RESID ? resid + off : nullptr, out + off, nb);
    }
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// The 3 x bf16 split kernel (opt-in): same layouts, same grids, 512 threads
template <bool RESID>
int launch_wino3b_t(apz_engine* e, int attr_slot, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    using T = apz::Wino3B;
    bool& configured = e->lds_attr_set[attr_slot];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3b_kernel<RESID, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    T::LDS_BYTES));
        configured = true;
    }
    for (int b0 = 0; b0 < n; b0 += WINO3_MAX_BOARDS) {
        const int nb = std::min(n - b0, WINO3_MAX_BOARDS);
        const size_t off = (size_t)b0 * T::C * T::GPLANE;
        const int grid = apz::wino3_grid(nb, e->num_cu);
        hipLaunchKernelGGL((apz::trunk15_wino3b_kernel<RESID, true>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream, in + off,
                           (const void*)L.upk3b, L.bias, RESID ? resid + off : nullptr, out + off, nb);
    }
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// The 2 x fp16 split kernel: same layouts, same grids, 512 threads
template <bool RESID>
int launch_wino3h_t(apz_engine* e, int attr_slot, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    using T = apz::Wino3H;
    bool& configured = e->lds_attr_set[attr_slot];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3h_kernel<RESID, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    T::LDS_BYTES));
        configured = true;
    }
    for (int b0 = 0; b0 < n; b0 += WINO3_MAX_BOARDS) {
        const int nb = std::min(n - b0, WINO3_MAX_BOARDS);
        const size_t off = (size_t)b0 * T::C * T::GPLANE;
        const int grid = apz::wino3_grid(nb, e->num_cu);
        hipLaunchKernelGGL((apz::trunk15_wino3h_kernel<RESID, true>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream, in + off,
                           (const void*)L.upk3h, L.bias3h, RESID ? resid + off : nullptr, out + off, nb, e->ovf_cur);
    }
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int launch_trunk_wino3(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    if (e->trunk_arith == APZ_ARITH_F16X2 && !e->force_f32 && L.upk3h && e->ovf_cur &&
        (n > apz::Wino3S::MAX_BOARDS || e->no_small_trunk)) {
        if (resid) return launch_wino3h_t<true>(e, 32, L, in, resid, out, n);
        return launch_wino3h_t<false>(e, 33, L, in, resid, out, n);
    }
    if (e->trunk_arith == APZ_ARITH_BF16X3 && L.upk3b && (n > apz::Wino3S::MAX_BOARDS || e->no_small_trunk)) {
        if (resid) return launch_wino3b_t<true>(e, 26, L, in, resid, out, n);
        return launch_wino3b_t<false>(e, 27, L, in, resid, out, n);
    }
    if (resid) return launch_wino3_t<true, true>(e, 6, in, L.upk2, L.bias, resid, out, n, L.upk3s);
    return launch_wino3_t<false, true>(e, 7, in, L.upk2, L.bias, nullptr, out, n, L.upk3s);
}

int launch_trunk_ring(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    if (e->trunk_kernel == APZ_TRUNK_WINOGRAD && L.upk2) return launch_trunk_wino3(e, L, in, resid, out, n);
    return launch_trunk_ring_t<4>(e, L, in, resid, out, n);   // the direct convolution: in-tree cross-check of the Winograd kernel
}

template <int C4, int CIN, bool CODES>
int launch_stem15_t(apz_engine* e, const ConvLayer& L, const float* in, float* out, int n) {
    constexpr int lds = apz::stem15_lds_bytes<C4>();
    bool& configured = e->lds_attr_set[(C4 == 1 ? 2 : 3) + (CODES ? 22 : 0)];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::stem15_kernel<C4, CIN, CODES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    lds));
        configured = true;
    }
    // two resident workgroups per CU; at C_in = 4 twice as many workgroups as fit, so that the dispatcher evens out the tail
    const int grid = std::min(n, e->num_cu * (C4 == 1 ? 4 : 2));
    hipLaunchKernelGGL((apz::stem15_kernel<C4, CIN, CODES>), dim3(grid), dim3(256), lds, e->stream, in, L.wpk, L.bias, out, n,
                       L.cin, (int)e->code_stride);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// codes != nullptr: the stem reads the position codes itself (no planes buffer)
int launch_stem15(apz_engine* e, const ConvLayer& L, const float* in, float* out, int n, const unsigned char* codes = nullptr) {
    if (codes) {
        if (L.cin == 4) return launch_stem15_t<1, 4, true>(e, L, (const float*)codes, out, n);
        if (L.cin == 9) return launch_stem15_t<3, 9, true>(e, L, (const float*)codes, out, n);
    } else {
        if (L.cin == 4) return launch_stem15_t<1, 4, false>(e, L, in, out, n);
        if (L.cin == 9) return launch_stem15_t<3, 9, false>(e, L, in, out, n);
    }
    return fail(APZ_E_UNSUPPORTED, "stem15: C_in must be 4 or 9");
}

// 8x8 boards: work item = (board, 16 output channels), the contraction split over the four waves (conv8_small.h)
template <bool RESID, bool CODES>
int launch_conv8_t(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    using T = apz::Conv8;
    bool& configured = e->lds_attr_set[28 + (RESID ? 1 : 0) + (CODES ? 2 : 0)];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::conv8_kernel<RESID, CODES>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    T::LDS_BYTES));
        configured = true;
    }
    const long items = (long)n * (L.cout / 16);
    const int grid = (int)std::min<long>(items, 2L * e->num_cu);       // two workgroups per CU (LDS), persistent over items
    hipLaunchKernelGGL((apz::conv8_kernel<RESID, CODES>), dim3(grid), dim3(256), T::LDS_BYTES, e->stream, in, L.wpk12, L.bias,
                       resid, out, n, L.cin, L.cin_pad / 4, L.cout, 1, (int)e->code_stride);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// ... with split operands on the fp16 matrix pipe (conv8_split.h): same items, same grids
template <bool RESID>
int launch_conv8h_t(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    using T = apz::Conv8H;
    bool& configured = e->lds_attr_set[34 + (RESID ? 1 : 0)];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::conv8h_kernel<RESID>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        configured = true;
    }
    const long items = (long)n * (L.cout / 16);
    const int grid = (int)std::min<long>(items, 2L * e->num_cu);
    hipLaunchKernelGGL((apz::conv8h_kernel<RESID>), dim3(grid), dim3(256), T::LDS_BYTES, e->stream, in, (const void*)L.wpk8h,
                       L.bias8h, resid, out, n, L.cin, L.cout, 1, e->ovf_cur);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int launch_conv8(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n,
                 const unsigned char* codes = nullptr) {
    if (L.cout % 16 || !L.wpk12) return fail(APZ_E_UNSUPPORTED, "conv8: C_out must be a multiple of 16");
    if (!codes && e->trunk_arith == APZ_ARITH_F16X2 && !e->force_f32 && L.wpk8h && e->ovf_cur) {
        if (resid) return launch_conv8h_t<true>(e, L, in, resid, out, n);
        return launch_conv8h_t<false>(e, L, in, nullptr, out, n);
    }
    if (codes) return launch_conv8_t<false, true>(e, L, (const float*)codes, nullptr, out, n);
    if (resid) return launch_conv8_t<true, false>(e, L, in, resid, out, n);
    return launch_conv8_t<false, false>(e, L, in, nullptr, out, n);
}

int launch_conv(apz_engine* e, const ConvLayer& L, const float* in, const float* resid, float* out, int n) {
    const int H = e->cfg.height, W = e->cfg.width, ct = L.cout / 64;
    if (e->small8) return launch_conv8(e, L, in, resid, out, n);
    if (e->ring && &L != &e->convs[0]) return launch_trunk_ring(e, L, in, resid, out, n);
    if (e->ring) return launch_stem15(e, L, in, out, n);
    if (H == 15 && W == 15) {
        if (ct == 2) return launch_conv_t<15, 15, 2>(e, L, in, resid, out, n);
        if (ct == 1) return launch_conv_t<15, 15, 1>(e, L, in, resid, out, n);
        if (ct == 4) return launch_conv_t<15, 15, 4>(e, L, in, resid, out, n);
    } else if (H == 8 && W == 8) {
        if (ct == 2) return launch_conv_t<8, 8, 2>(e, L, in, resid, out, n);
        if (ct == 1) return launch_conv_t<8, 8, 1>(e, L, in, resid, out, n);
        if (ct == 4) return launch_conv_t<8, 8, 4>(e, L, in, resid, out, n);
    }
    return fail(APZ_E_UNSUPPORTED, "conv3x3: unsupported board size / channel count");
}

// runs conv layers [0, upto] on e->planes; returns the buffer holding layer `upto`'s output
int run_trunk(apz_engine* e, const float* planes, int n, int upto, float** result, const unsigned char* codes = nullptr) {
    float *x = e->act[0], *t = e->act[1], *y = e->act[2];
    const float* cur = planes;
    const int nl = (int)e->convs.size();
    const int last = std::min(upto, nl - 1);
    {
        Timed tm(e, APZ_K_STEM);
        int rc;
        if (codes && e->small8)
            rc = launch_conv8(e, e->convs[0], nullptr, nullptr, x, n, codes);       // the first layer decodes the codes itself
        else if (codes && e->ring && e->cfg.net_kind == APZ_NET_RESNET)
            rc = launch_stem15(e, e->convs[0], nullptr, x, n, codes);
        else
            rc = launch_conv(e, e->convs[0], cur, nullptr, x, n);
        if (rc) return rc;
        cur = x;
    }
    if (last >= 1) {
        Timed tm(e, APZ_K_TRUNK);       // all trunk launches of this forward under one event pair
        tm.launches = last;
        for (int li = 1; li <= last; li++) {
            const ConvLayer& L = e->convs[li];
            if (e->cfg.net_kind == APZ_NET_RESNET) {
                if (!L.residual) {          // convA: x -> t
                    int rc = launch_conv(e, L, x, nullptr, t, n);
                    if (rc) return rc;
                    cur = t;
                } else {                    // convB: t (+x) -> y ; then y becomes the block output
                    int rc = launch_conv(e, L, t, x, y, n);
                    if (rc) return rc;
                    std::swap(x, y);
                    cur = x;
                }
            } else {
                float* dst = (cur == x) ? t : x;
                int rc = launch_conv(e, L, cur, nullptr, dst, n);
                if (rc) return rc;
                cur = dst;
            }
        }
    }
    *result = const_cast<float*>(cur);
    return APZ_OK;
}

void drop_forward_graphs(apz_engine* e) {
    for (auto& kv : e->fwd_graphs) hipGraphExecDestroy(kv.second);
    e->fwd_graphs.clear();
    e->fwd_seen.clear();
}

// codes_dev != nullptr (stem_takes_codes(e) only): the position codes instead of `planes`
bool stem_takes_codes(const apz_engine* e) { return (e->ring && e->cfg.net_kind == APZ_NET_RESNET) || e->small8; }

int forward_dev(apz_engine* e, const float* planes, int n, float* probs, float* values, float* logits,
                float* vlogits, const unsigned char* codes_dev = nullptr) {
    if (!e->loaded) return fail(APZ_E_STATE, "weights not loaded");
    if (n < 0 || n > e->cfg.max_batch) return fail(APZ_E_ARG, "batch exceeds max_batch");
    if (n == 0) return APZ_OK;
    e->prof_now = e->profiling && (e->prof_phase++ % e->prof_stride == 0);
    if (e->profiling) resolve_ready(e);
    // the whole forward under one pair, EVERY forward while profiling is on: sum / wall clock = the GPU-busy fraction of a
    // measured window (two records per ~2 ms forward, at the forward's edges where the slot's `done` event sits anyway)
    Timed whole(e, APZ_K_FORWARD, true);
    float* trunk = nullptr;
    int rc = run_trunk(e, planes, n, (int)e->convs.size() - 1, &trunk, codes_dev);
    if (rc) return rc;
    const int hw = e->hw;
    if (e->small8) {                    // both 1x1 convolutions, both FullyConnected layers, softmax and tanh in one launch
        Timed tm(e, APZ_K_HEAD_FC);
        hipLaunchKernelGGL(apz::head8_kernel, dim3(std::min(n, e->num_cu * 8)), dim3(256), 0, e->stream, trunk, e->w6, e->b6,
                           e->wfc_raw, e->bfc, e->wv, e->bv, probs, values, logits, vlogits, n, e->clast);
        HIP_TRY(hipGetLastError());
        e->last_n = n;
        return APZ_OK;
    }
    {
        Timed tm(e, APZ_K_HEAD_CONV);
        if (e->ring && e->clast % 32 == 0)
            hipLaunchKernelGGL(apz::head_conv1x1_r16_kernel, dim3(n), dim3(256), 0, e->stream, trunk, e->w6, e->b6,
                               e->featp, e->featv, n, e->clast);
        else
            hipLaunchKernelGGL(apz::head_conv1x1_kernel, dim3(std::min(n, e->num_cu * 8)), dim3(256), 0, e->stream, trunk,
                               e->w6, e->b6, e->featp, e->featv, n, e->clast, hw, e->cfg.width,
                               e->ring ? e->act_ps : hw, e->ring ? e->act_rs : e->cfg.width);
        HIP_TRY(hipGetLastError());
    }
    {
        Timed tm(e, APZ_K_HEAD_FC);
        const int ntile = (hw + 15) / 16;
        const int lds = 16 * std::max(4 * hw + 1, ntile * 16) * (int)sizeof(float);
        const int grid = (n + 15) / 16;
        const int tpw = (ntile + 3) / 4;
        if (tpw <= 1) {
            hipLaunchKernelGGL(apz::head_fc_kernel<1>, dim3(grid), dim3(256), lds, e->stream, e->featp, e->featv,
                               e->wfc_pk, e->bfc, e->wv, e->bv, probs, values, logits, vlogits, n, hw);
        } else if (tpw <= 4) {
            // the n-tiles over four workgroups per 16 boards, then one wavefront per board for softmax + value head
            const int lds1 = 16 * (4 * hw + 1) * (int)sizeof(float);
            hipLaunchKernelGGL((apz::head_fc_kernel<1, true>), dim3(grid, tpw), dim3(256), lds1, e->stream, e->featp, e->featv,
                               e->wfc_pk, e->bfc, e->wv, e->bv, probs, values, logits, vlogits, n, hw, e->fc_logits);
            hipLaunchKernelGGL(apz::head_softmax_value_kernel, dim3((n + 3) / 4), dim3(256), 0, e->stream, e->fc_logits, e->featv,
                               e->wv, e->bv, probs, values, logits, vlogits, n, hw, ntile * 16);
        } else {
            return fail(APZ_E_UNSUPPORTED, "policy head: board too large");
        }
        HIP_TRY(hipGetLastError());
    }
    e->last_n = n;
    return APZ_OK;
}

// APZ_ARITH_F16X2: arms the synchronous overflow word for the scope (entry points that run trunk layers without collecting a
// forward's results -- prewarm, the layer bench, apz_layer_io: the word is raised and nobody reads it)
struct ArmOverflowWord {
    apz_engine* e;
    explicit ArmOverflowWord(apz_engine* e_) : e(e_) {
        if (e->trunk_arith == APZ_ARITH_F16X2 && e->ovf_dev) e->ovf_cur = e->ovf_dev + APZ_MAX_SLOTS;
    }
    ~ArmOverflowWord() { e->ovf_cur = nullptr; }
};

// APZ_ARITH_F16X2: forward_dev with the overflow word `idx` armed.  `again` != nullptr: the forward is collected here --
// wait for the stream, look at the word and, if an activation left the fp16 range, run `again` (the same forward, which
// then takes the exact-fp32 trunk kernel).  Other arithmetics: plain forward_dev.
template <class F>
int forward_guarded(apz_engine* e, int idx, F run, bool collect) {
    if (e->trunk_arith != APZ_ARITH_F16X2 || !e->ovf_host) return run();
    e->ovf_host[idx] = 0;
    e->ovf_cur = e->ovf_dev + idx;
    int rc = run();
    e->ovf_cur = nullptr;
    if (rc || !collect) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    if (e->ovf_host[idx]) {
        e->ovf_host[idx] = 0;
        e->force_f32 = true;
        rc = run();
        e->force_f32 = false;
        e->ovf_repeats++;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    return APZ_OK;
}

// dihedral index tables (train_mxnet.py:115-135) on square boards
void build_perms(int N, std::vector<int>& ps, std::vector<int>& pp) {
    typedef std::vector<int> Grid;
    auto rot90 = [N](const Grid& a) { Grid r(N * N); for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) r[i * N + j] = a[j * N + (N - 1 - i)]; return r; };
    auto fliplr = [N](const Grid& a) { Grid r(N * N); for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) r[i * N + j] = a[i * N + (N - 1 - j)]; return r; };
    auto flipud = [N](const Grid& a) { Grid r(N * N); for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) r[i * N + j] = a[(N - 1 - i) * N + j]; return r; };
    Grid id(N * N);
    for (int i = 0; i < N * N; i++) id[i] = i;
    ps.clear();
    pp.clear();
    Grid s = id, p = flipud(id);
    for (int k = 1; k <= 4; k++) {
        s = rot90(s);
        p = rot90(p);
        Grid po = flipud(p);
        ps.insert(ps.end(), s.begin(), s.end());
        pp.insert(pp.end(), po.begin(), po.end());
        Grid sf = fliplr(s), pf = flipud(fliplr(p));
        ps.insert(ps.end(), sf.begin(), sf.end());
        pp.insert(pp.end(), pf.begin(), pf.end());
    }
}

}  // namespace

extern "C" {

const char* apz_last_error(void) { return g_err.c_str(); }
int apz_version(void) { return 1; }

int apz_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void apz_destroy(apz_engine* e) {
    if (!e) return;
    hipSetDevice(e->cfg.device);
    if (e->stream) hipStreamSynchronize(e->stream);
    resolve_pending(e);
    drop_forward_graphs(e);
    for (auto ev : e->free_events) hipEventDestroy(ev);
    for (auto& l : e->convs) {
        if (l.wpk) hipFree(l.wpk);
        if (l.upk2) hipFree(l.upk2);
        if (l.upk3s) hipFree(l.upk3s);
        if (l.upk3b) hipFree(l.upk3b);
        if (l.upk3h) hipFree(l.upk3h);
        if (l.bias3h) hipFree(l.bias3h);
        if (l.wpk8h) hipFree(l.wpk8h);
        if (l.bias8h) hipFree(l.bias8h);
        if (l.wpk12) hipFree(l.wpk12);
        if (l.bias) hipFree(l.bias);
    }
    void* dev[] = {e->w6, e->b6, e->wfc_pk, e->bfc, e->wv, e->bv, e->act[0], e->act[1], e->act[2], e->planes,
                   e->featp, e->featv, e->probs, e->values, e->codes, e->perm_s, e->perm_p, e->smp_vis, e->smp_pi, e->smp_mv, e->zeros256,
                   e->wino_scratch[0], e->wino_scratch[1], e->bn_part, e->adam_tab, e->wgw_scratch, e->head_scratch, e->fc_logits, e->fold_ws,
                   e->wfc_raw, e->w3s_slabs, e->w3s_tickets};
    for (void* p : dev)
        if (p) hipFree(p);
    if (e->ovf_host) hipHostFree(e->ovf_host);
    for (auto& sl : e->slots) {
        if (sl.h_codes) hipHostFree(sl.h_codes);
        if (sl.h_probs) hipHostFree(sl.h_probs);
        if (sl.h_values) hipHostFree(sl.h_values);
        if (sl.done) hipEventDestroy(sl.done);
    }
    void* host[] = {e->h_planes, e->h_probs, e->h_values, e->h_codes};
    for (void* p : host)
        if (p) hipHostFree(p);
    if (e->stream) hipStreamDestroy(e->stream);
    delete e;
}

apz_engine* apz_create(const apz_config* cfg) {
    if (!cfg || cfg->height < 1 || cfg->width < 1 || cfg->max_batch < 1 || (cfg->c_in != 9 && cfg->c_in != 4)) {
        fail(APZ_E_ARG, "bad config (c_in must be 9 or 4)");
        return nullptr;
    }
    if (!((cfg->height == 15 && cfg->width == 15) || (cfg->height == 8 && cfg->width == 8))) {
        fail(APZ_E_UNSUPPORTED, "HIP kernels are instantiated for 15x15 and 8x8 boards");
        return nullptr;
    }
    if (cfg->net_kind == APZ_NET_RESNET &&
        ((cfg->n_filter != 64 && cfg->n_filter != 128 && cfg->n_filter != 256) || cfg->n_blocks < 0)) {
        fail(APZ_E_UNSUPPORTED, "n_filter must be 64, 128 or 256");
        return nullptr;
    }
    if (cfg->net_kind != APZ_NET_RESNET && cfg->net_kind != APZ_NET_SIMPLE) {
        fail(APZ_E_ARG, "unknown net_kind");
        return nullptr;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) {
        fail(APZ_E_HIP, "no HIP device: libalphapig_hip needs an AMD GPU (there is no CPU fallback)");
        return nullptr;
    }
    if (cfg->device < 0 || cfg->device >= ndev) {
        fail(APZ_E_ARG, "device ordinal out of range");
        return nullptr;
    }
    apz_engine* e = new apz_engine();
    e->cfg = *cfg;
    e->hw = cfg->height * cfg->width;
    e->code_stride = (e->hw + 1 + 15) / 16 * 16;
    build_tables(e);
    auto bail = [&](const char* what, hipError_t err) -> apz_engine* {
        fail(APZ_E_HIP, std::string(what) + ": " + hipGetErrorString(err));
        apz_destroy(e);
        return nullptr;
    };
    hipError_t err;
    if ((err = hipSetDevice(cfg->device)) != hipSuccess) return bail("hipSetDevice", err);
    hipDeviceProp_t prop;
    if ((err = hipGetDeviceProperties(&prop, cfg->device)) != hipSuccess) return bail("hipGetDeviceProperties", err);
    e->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((err = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking)) != hipSuccess)
        return bail("hipStreamCreate", err);
    e->ring = cfg->net_kind == APZ_NET_RESNET && cfg->height == 15 && cfg->width == 15 && cfg->n_filter == 128;
    e->small8 = cfg->height == 8 && cfg->width == 8;
    e->act_ps = e->ring ? apz::Trunk15::GPLANE : e->hw;
    e->act_rs = e->ring ? apz::Trunk15::GROW : cfg->width;
    const size_t B = cfg->max_batch, hw = e->hw;
    const size_t act_bytes = B * e->cmax * (size_t)e->act_ps * sizeof(float);
    for (int i = 0; i < 3; i++)
    {
        if ((err = hipMalloc((void**)&e->act[i], act_bytes)) != hipSuccess) return bail("hipMalloc(act)", err);
        // rows16 pad columns must read as zero; they are never written with anything else
        if ((err = hipMemset(e->act[i], 0, act_bytes)) != hipSuccess) return bail("hipMemset(act)", err);
    }
    if ((err = hipMalloc((void**)&e->planes, B * 9 * hw * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->featp, B * 4 * hw * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->featv, B * 2 * hw * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->probs, B * hw * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->fc_logits, B * ((hw + 15) / 16 * 16) * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->values, B * sizeof(float))) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMalloc((void**)&e->codes, B * e->code_stride)) != hipSuccess) return bail("hipMalloc", err);
    if ((err = hipMemset(e->codes, 0, B * e->code_stride)) != hipSuccess) return bail("hipMemset(codes)", err);   // empty boards: apz_prewarm's input
    if ((err = hipHostMalloc((void**)&e->h_planes, B * 9 * hw * sizeof(float))) != hipSuccess) return bail("hipHostMalloc", err);
    if ((err = hipHostMalloc((void**)&e->h_probs, B * hw * sizeof(float))) != hipSuccess) return bail("hipHostMalloc", err);
    if ((err = hipHostMalloc((void**)&e->h_values, B * sizeof(float))) != hipSuccess) return bail("hipHostMalloc", err);
    if ((err = hipHostMalloc((void**)&e->h_codes, B * e->code_stride)) != hipSuccess) return bail("hipHostMalloc", err);
    if (cfg->height == cfg->width) {
        std::vector<int> ps, pp;
        build_perms(cfg->height, ps, pp);
        if (upload(&e->perm_s, ps) || upload(&e->perm_p, pp)) {
            apz_destroy(e);
            return nullptr;
        }
    }
    return e;
}

int apz_param_count(apz_engine* e) { return e ? (int)e->params.size() : fail(APZ_E_ARG, "null engine"); }

const char* apz_param_name(apz_engine* e, int i) {
    if (!e || i < 0 || i >= (int)e->params.size()) return nullptr;
    return e->params[i].name.c_str();
}

int64_t apz_param_size(apz_engine* e, int i) {
    if (!e || i < 0 || i >= (int)e->params.size()) return -1;
    return e->params[i].size;
}

int apz_load_weights(apz_engine* e, const char* const* names, const float* const* ptrs, const int64_t* sizes, int n) {
    if (!e || !names || !ptrs || !sizes) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    std::map<std::string, const float*> P;
    std::map<std::string, int64_t> S;
    for (int i = 0; i < n; i++) {
        P[names[i]] = ptrs[i];
        S[names[i]] = sizes[i];
    }
    for (auto& p : e->params) {
        auto it = S.find(p.name);
        if (it == S.end()) return fail(APZ_E_ARG, "missing parameter " + p.name);
        if (it->second != p.size)
            return fail(APZ_E_ARG, "parameter " + p.name + " has " + std::to_string(it->second) + " elements, expected " +
                                       std::to_string(p.size));
    }
    HIP_TRY(hipStreamSynchronize(e->stream));
    drop_forward_graphs(e);              // (weight buffers may move)
    std::vector<double> scale, shift;
    for (auto& L : e->convs) {
        fold_bn(P, L.name, L.bn, L.mean_sfx, L.var_sfx, L.fix_gamma, L.cout, scale, shift);
        const float* w = P.at(L.name + "_weight");   // [cout][cin][3][3]
        const int n4 = L.cin_pad / 4, ncot = L.cout / 16;
        // trunk15_ring_kernel layers: [cot][c4][lane][12] (three 16-byte loads per lane and ci4 step);
        // everything else: [cot][c4][tap][lane]
        const bool x4 = e->ring && &L != &e->convs[0];
        std::vector<float> pk((size_t)ncot * n4 * 64 * (x4 ? 12 : 9), 0.f), bias(L.cout);
        for (int cot = 0; cot < ncot; cot++)
            for (int c4 = 0; c4 < n4; c4++)
                for (int tap = 0; tap < 9; tap++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int co = cot * 16 + (lane & 15), ci = c4 * 4 + (lane >> 4);
                        float v = 0.f;
                        if (ci < L.cin) v = (float)((double)w[((size_t)co * L.cin + ci) * 9 + tap] * scale[co]);
                        if (x4)
                            pk[(((size_t)cot * n4 + c4) * 64 + lane) * 12 + tap] = v;
                        else
                            pk[(((size_t)cot * n4 + c4) * 9 + tap) * 64 + lane] = v;
                    }
        for (int o = 0; o < L.cout; o++) bias[o] = (float)shift[o];
        int rc = upload(&L.wpk, pk);
        if (rc) return rc;
        rc = upload(&L.bias, bias);
        if (rc) return rc;
        if (e->small8) {                // conv8_kernel: the nine taps of a lane side by side, [cot][c4][lane][12]
            std::vector<float> pk12((size_t)ncot * n4 * 64 * 12, 0.f);
            for (int cot = 0; cot < ncot; cot++)
                for (int c4 = 0; c4 < n4; c4++)
                    for (int lane = 0; lane < 64; lane++) {
                        const int co = cot * 16 + (lane & 15), ci = c4 * 4 + (lane >> 4);
                        if (ci >= L.cin) continue;
                        for (int tap = 0; tap < 9; tap++)
                            pk12[(((size_t)cot * n4 + c4) * 64 + lane) * 12 + tap] =
                                (float)((double)w[((size_t)co * L.cin + ci) * 9 + tap] * scale[co]);
                    }
            rc = upload(&L.wpk12, pk12);
            if (rc) return rc;
            if (e->trunk_arith == APZ_ARITH_F16X2 && apz::Conv8H::supports(L.cin, L.cout)) {
                // the same folded weights, per output channel times a power of two, as two fp16 terms: conv8_split.h
                std::vector<uint16_t> pkh;
                std::vector<float> b8;
                apz::conv8h_pack_host(w, scale.data(), shift.data(), L.cin, L.cout, pkh, b8);
                if (!L.wpk8h) HIP_TRY(hipMalloc(&L.wpk8h, apz::Conv8H::pk_bytes(L.cin, L.cout)));
                HIP_TRY(hipMemcpy(L.wpk8h, pkh.data(), apz::Conv8H::pk_bytes(L.cin, L.cout), hipMemcpyHostToDevice));
                rc = upload(&L.bias8h, b8);
                if (rc) return rc;
            }
        }
        if (x4) {
            // F(4x4,3x3) Winograd weights U[pos = 6i+k][co][ci] = (G g G^T)[i][k] of the BN-folded kernel g, in double,
            // rounded once; packed [cot 8][row half 2][c4 32][lane 64][20] (wino_common.h)
            static const double G[6][3] = {{1.0 / 4, 0, 0},           {-1.0 / 6, -1.0 / 6, -1.0 / 6},
                                           {-1.0 / 6, 1.0 / 6, -1.0 / 6}, {1.0 / 24, 1.0 / 12, 1.0 / 6},
                                           {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
            std::vector<float> up2(apz::WinoPack::UPK_FLOATS, 0.f), up3s(apz::WinoPackSmall::UPK_FLOATS, 0.f);
            for (int co = 0; co < 128; co++)
                for (int ci = 0; ci < 128; ci++) {
                    double g[3][3], t[6][3];
                    for (int a = 0; a < 3; a++)
                        for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
                    for (int i = 0; i < 6; i++)
                        for (int b = 0; b < 3; b++) t[i][b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
                    const int cot = co >> 4, jj = co & 15, qq = ci & 3, c4 = ci >> 2;
                    for (int i = 0; i < 6; i++)
                        for (int k = 0; k < 6; k++) {
                            const double u = t[i][0] * G[k][0] + t[i][1] * G[k][1] + t[i][2] * G[k][2];
                            const int half = i / 3;
                            up2[((((size_t)cot * 2 + half) * 32 + c4) * 64 + (qq * 16 + jj)) * 20 + (i - 3 * half) * 6 + k] = (float)u;
                            up3s[apz::WinoPackSmall::index(co, ci, 6 * i + k)] = (float)u;
                        }
                }
            rc = upload(&L.upk2, up2);
            if (rc) return rc;
            rc = upload(&L.upk3s, up3s);
            if (rc) return rc;
            if (e->trunk_arith == APZ_ARITH_BF16X3) {
                // the same U as three bf16 terms (round to nearest even, the remainder taken in double): trunk15_wino3b.h
                std::vector<uint16_t> ub;
                apz::wino3b_pack_host(
                    [&](int co, int ci, int pos) {
                        double g[3][3], t[3];
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
                        const int i = pos / 6, k = pos % 6;
                        for (int b = 0; b < 3; b++) t[b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
                        return t[0] * G[k][0] + t[1] * G[k][1] + t[2] * G[k][2];
                    },
                    ub);
                if (!L.upk3b) HIP_TRY(hipMalloc(&L.upk3b, apz::Wino3B::UPK_BYTES));
                HIP_TRY(hipMemcpy(L.upk3b, ub.data(), apz::Wino3B::UPK_BYTES, hipMemcpyHostToDevice));
            }
            if (e->trunk_arith == APZ_ARITH_F16X2) {
                // the same U, per output channel times a power of two, as two fp16 terms: trunk15_wino3h.h
                std::vector<uint16_t> uh;
                std::vector<float> b3(apz::Wino3H::BIAS_FLOATS);
                for (int o = 0; o < 128; o++) b3[o] = (float)shift[o];
                apz::wino3h_pack_host(
                    [&](int co, int ci, int pos) {
                        double g[3][3], t[3];
                        for (int a = 0; a < 3; a++)
                            for (int b = 0; b < 3; b++) g[a][b] = (double)w[((size_t)co * 128 + ci) * 9 + a * 3 + b] * scale[co];
                        const int i = pos / 6, k = pos % 6;
                        for (int b = 0; b < 3; b++) t[b] = G[i][0] * g[0][b] + G[i][1] * g[1][b] + G[i][2] * g[2][b];
                        return t[0] * G[k][0] + t[1] * G[k][1] + t[2] * G[k][2];
                    },
                    uh, b3.data() + 128);
                if (!L.upk3h) HIP_TRY(hipMalloc(&L.upk3h, apz::Wino3H::UPK_BYTES));
                HIP_TRY(hipMemcpy(L.upk3h, uh.data(), apz::Wino3H::UPK_BYTES, hipMemcpyHostToDevice));
                rc = upload(&L.bias3h, b3);
                if (rc) return rc;
            }
        }
    }
    // heads: two 1x1 conv_act (fix_gamma default) folded into one [6][C] matrix
    {
        const int C = e->clast, hw = e->hw;
        std::vector<float> w6((size_t)6 * C), b6(6);
        fold_bn(P, "conv3_1_1", "conv3_1_1", "_mean", "_var", true, 4, scale, shift);
        const float* wp = P.at("conv3_1_1_weight");
        for (int o = 0; o < 4; o++) {
            for (int c = 0; c < C; c++) w6[(size_t)o * C + c] = (float)((double)wp[o * C + c] * scale[o]);
            b6[o] = (float)shift[o];
        }
        fold_bn(P, "conv3_2_1", "conv3_2_1", "_mean", "_var", true, 2, scale, shift);
        const float* wq = P.at("conv3_2_1_weight");
        for (int o = 0; o < 2; o++) {
            for (int c = 0; c < C; c++) w6[(size_t)(4 + o) * C + c] = (float)((double)wq[o * C + c] * scale[o]);
            b6[4 + o] = (float)shift[o];
        }
        const float* wfc = P.at("fc_3_1_1_weight");   // [hw][4*hw]
        const int K = 4 * hw, KS = hw, ntile = (hw + 15) / 16;
        // head_fc_kernel: [tile][trip of 8 k-steps][lane][8], zero past KS and past hw outputs
        const int KG = (KS + 7) / 8;
        std::vector<float> pk((size_t)ntile * KG * 64 * 8, 0.f);
        for (int nt = 0; nt < ntile; nt++)
            for (int s = 0; s < KS; s++)
                for (int lane = 0; lane < 64; lane++) {
                    const int o = nt * 16 + (lane & 15), k = 4 * s + (lane >> 4);
                    if (o < hw) pk[((((size_t)nt * KG + (s >> 3)) * 64 + lane) << 3) + (s & 7)] = wfc[(size_t)o * K + k];
                }
        std::vector<float> bfc(P.at("fc_3_1_1_bias"), P.at("fc_3_1_1_bias") + hw);
        std::vector<float> wv(P.at("fc_3_2_1_weight"), P.at("fc_3_2_1_weight") + 2 * hw);
        std::vector<float> bv(P.at("fc_3_2_1_bias"), P.at("fc_3_2_1_bias") + 1);
        int rc;
        if (e->small8) {
            std::vector<float> raw(wfc, wfc + (size_t)hw * K);
            if ((rc = upload(&e->wfc_raw, raw))) return rc;
        }
        if ((rc = upload(&e->w6, w6)) || (rc = upload(&e->b6, b6)) || (rc = upload(&e->wfc_pk, pk)) ||
            (rc = upload(&e->bfc, bfc)) || (rc = upload(&e->wv, wv)) || (rc = upload(&e->bv, bv)))
            return rc;
    }
    e->loaded = true;
    return APZ_OK;
}

int apz_forward(apz_engine* e, const void* planes_dev, int n, void* probs_dev, void* values_dev, void* logits_dev,
                void* vlogits_dev) {
    if (!e || !planes_dev || !probs_dev || !values_dev) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    // (APZ_ARITH_F16X2: returns with the stream drained -- the overflow word has to be read before the results are used)
    return forward_guarded(e, APZ_MAX_SLOTS, [&]() {
        return forward_dev(e, (const float*)planes_dev, n, (float*)probs_dev, (float*)values_dev, (float*)logits_dev,
                           (float*)vlogits_dev);
    }, true);
}

int apz_forward_host(apz_engine* e, const float* planes_host, int n, float* probs_host, float* values_host) {
    if (!e || !planes_host || !probs_host || !values_host) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (n < 0 || n > e->cfg.max_batch) return fail(APZ_E_ARG, "batch exceeds max_batch");
    if (n == 0) return APZ_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t hw = e->hw, pin = (size_t)n * e->cfg.c_in * hw * sizeof(float);
    std::memcpy(e->h_planes, planes_host, pin);
    HIP_TRY(hipMemcpyAsync(e->planes, e->h_planes, pin, hipMemcpyHostToDevice, e->stream));
    int rc = forward_guarded(e, APZ_MAX_SLOTS, [&]() { return forward_dev(e, e->planes, n, e->probs, e->values, nullptr, nullptr); }, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(e->h_probs, e->probs, n * hw * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_values, e->values, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    resolve_pending(e);
    std::memcpy(probs_host, e->h_probs, n * hw * sizeof(float));
    std::memcpy(values_host, e->h_values, n * sizeof(float));
    return APZ_OK;
}

int apz_forward_codes_async(apz_engine* e, const uint8_t* codes_pinned, int n, float* probs_pinned,
                            float* values_pinned) {
    if (!e || !codes_pinned || !probs_pinned || !values_pinned) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (n < 0 || n > e->cfg.max_batch) return fail(APZ_E_ARG, "batch exceeds max_batch");
    if (e->cfg.c_in != 9 && e->cfg.c_in != 4) return fail(APZ_E_STATE, "bad c_in");
    if (n == 0) return APZ_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t hw = e->hw;
    HIP_TRY(hipMemcpyAsync(e->codes, codes_pinned, (size_t)n * e->code_stride, hipMemcpyHostToDevice, e->stream));
    // (APZ_ARITH_F16X2: the forward is collected here, see forward_guarded -- the copies below are queued behind it)
    int rc = forward_guarded(e, APZ_MAX_SLOTS, [&]() -> int {
        if (stem_takes_codes(e)) return forward_dev(e, nullptr, n, e->probs, e->values, nullptr, nullptr, e->codes);
        if (int r = apz_encode_planes(e, e->codes, n, e->cfg.c_in, e->planes)) return r;
        return forward_dev(e, e->planes, n, e->probs, e->values, nullptr, nullptr);
    }, true);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(probs_pinned, e->probs, n * hw * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(values_pinned, e->values, n * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    return APZ_OK;
}

int apz_forward_codes_host(apz_engine* e, const uint8_t* codes_host, int n, float* probs_host, float* values_host) {
    if (!e || !codes_host || !probs_host || !values_host) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (n < 0 || n > e->cfg.max_batch) return fail(APZ_E_ARG, "batch exceeds max_batch");
    if (n == 0) return APZ_OK;
    std::memcpy(e->h_codes, codes_host, (size_t)n * e->code_stride);
    int rc = apz_forward_codes_async(e, e->h_codes, n, e->h_probs, e->h_values);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(e->stream));
    resolve_pending(e);
    std::memcpy(probs_host, e->h_probs, (size_t)n * e->hw * sizeof(float));
    std::memcpy(values_host, e->h_values, n * sizeof(float));
    return APZ_OK;
}

// batches above this are a handful of long kernels: nothing to gain from a graph
constexpr int FWD_GRAPH_MAX_BOARDS = 64;

int apz_submit_codes(apz_engine* e, int slot, const uint8_t* codes_host, int n) {
    if (!e || !codes_host) return fail(APZ_E_ARG, "null argument");
    if (slot < 0 || slot >= APZ_MAX_SLOTS) return fail(APZ_E_ARG, "slot out of range");
    if (n < 1 || n > e->cfg.max_batch) return fail(APZ_E_ARG, "batch must be in [1, max_batch]");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    apz_engine::Slot& sl = e->slots[slot];
    if (sl.busy) return fail(APZ_E_STATE, "slot still in flight: call apz_wait first");
    const size_t B = e->cfg.max_batch, hw = e->hw;
    if (!sl.h_codes) {
        // Zero-copy slots: the 240-B/leaf codes and the 904-B/leaf results live in pinned,
        // device-mapped host memory.  The encoder reads the codes and the head kernel writes
        // probs/values straight over PCIe -- no blit kernels (and their ~25 us launch gaps)
        // around the forward.
        HIP_TRY(hipHostMalloc((void**)&sl.h_codes, B * e->code_stride, hipHostMallocMapped));
        HIP_TRY(hipHostMalloc((void**)&sl.h_probs, B * hw * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostMalloc((void**)&sl.h_values, B * sizeof(float), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void**)&sl.d_codes, sl.h_codes, 0));
        HIP_TRY(hipHostGetDevicePointer((void**)&sl.d_probs, sl.h_probs, 0));
        HIP_TRY(hipHostGetDevicePointer((void**)&sl.d_values, sl.h_values, 0));
        HIP_TRY(hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    }
    std::memcpy(sl.h_codes, codes_host, (size_t)n * e->code_stride);
    auto launch_all = [&]() -> int {
        if (stem_takes_codes(e))    // the stem decodes the codes itself (read straight from the pinned slot)
            return forward_dev(e, nullptr, n, sl.d_probs, sl.d_values, nullptr, nullptr, sl.d_codes);
        if (int rc = apz_encode_planes(e, sl.d_codes, n, e->cfg.c_in, e->planes)) return rc;
        return forward_dev(e, e->planes, n, sl.d_probs, sl.d_values, nullptr, nullptr);
    };
    int rc = APZ_OK;
    if (e->trunk_arith == APZ_ARITH_F16X2 && e->ovf_host) {   // the slot's overflow word: read by apz_wait
        e->ovf_host[slot] = 0;
        e->ovf_cur = e->ovf_dev + slot;
    }
    struct Disarm {
        apz_engine* e;
        ~Disarm() { e->ovf_cur = nullptr; }
    } disarm{e};
    const long key = ((long)slot << 32) | (long)n;
    const bool graphable = e->use_graphs && !e->profiling && n <= FWD_GRAPH_MAX_BOARDS && e->loaded;
    auto it = graphable ? e->fwd_graphs.find(key) : e->fwd_graphs.end();
    if (it != e->fwd_graphs.end()) {
        HIP_TRY(hipGraphLaunch(it->second, e->stream));
        e->last_n = n;
    } else if (graphable && ++e->fwd_seen[key] >= 3 && e->w3s_epoch < 0xFFFF0000u) {
        // (third use: every lazy allocation / attribute / ticket reset of this shape has happened outside the capture)
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        HIP_TRY(hipStreamBeginCapture(e->stream, hipStreamCaptureModeThreadLocal));
        rc = launch_all();
        const hipError_t ce = hipStreamEndCapture(e->stream, &graph);
        if (rc) {
            if (graph) hipGraphDestroy(graph);
            return rc;
        }
        if (ce != hipSuccess || !graph || hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) {
            if (graph) hipGraphDestroy(graph);
            (void)hipGetLastError();
            e->use_graphs = false;          // this runtime cannot capture the sequence: plain launches from here on
            rc = launch_all();
            if (rc) return rc;
        } else {
            hipGraphDestroy(graph);
            e->fwd_graphs[key] = exec;
            HIP_TRY(hipGraphLaunch(exec, e->stream));
        }
    } else {
        rc = launch_all();
        if (rc) return rc;
    }
    HIP_TRY(hipEventRecord(sl.done, e->stream));
    sl.n = n;
    sl.busy = true;
    return APZ_OK;
}

int apz_wait(apz_engine* e, int slot, float* probs_host, float* values_host) {
    if (!e || !probs_host || !values_host) return fail(APZ_E_ARG, "null argument");
    // (no engine lock: a slot is owned by the one thread that submitted into it, and holding the lock across the
    // event wait would stop the other pipeline groups from submitting behind this batch)
    if (slot < 0 || slot >= APZ_MAX_SLOTS) return fail(APZ_E_ARG, "slot out of range");
    apz_engine::Slot& sl = e->slots[slot];
    if (!sl.busy) return fail(APZ_E_STATE, "nothing submitted in this slot");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipEventSynchronize(sl.done));
    if (e->ovf_host && e->ovf_host[slot]) {
        // an activation of this batch left the fp16 range (trunk15_wino3h.h): the same batch again on the exact-fp32 kernel
        EngineLock guard(e->submit_lock);
        e->ovf_host[slot] = 0;
        e->force_f32 = true;
        int rc;
        if (stem_takes_codes(e)) {
            rc = forward_dev(e, nullptr, sl.n, sl.d_probs, sl.d_values, nullptr, nullptr, sl.d_codes);
        } else {
            rc = apz_encode_planes(e, sl.d_codes, sl.n, e->cfg.c_in, e->planes);
            if (!rc) rc = forward_dev(e, e->planes, sl.n, sl.d_probs, sl.d_values, nullptr, nullptr);
        }
        e->force_f32 = false;
        e->ovf_repeats++;
        if (rc) return rc;
        HIP_TRY(hipStreamSynchronize(e->stream));
    }
    std::memcpy(probs_host, sl.h_probs, (size_t)sl.n * e->hw * sizeof(float));
    std::memcpy(values_host, sl.h_values, (size_t)sl.n * sizeof(float));
    sl.busy = false;
    return sl.n;
}

void* apz_host_alloc(int64_t bytes) {
    void* p = nullptr;
    if (bytes <= 0 || hipHostMalloc(&p, (size_t)bytes) != hipSuccess) {
        fail(APZ_E_HIP, "hipHostMalloc failed");
        return nullptr;
    }
    return p;
}

void apz_host_free(void* p) {
    if (p) hipHostFree(p);
}

int apz_encode_planes(apz_engine* e, const void* codes_dev, int n, int n_planes, void* planes_dev) {
    if (!e || !codes_dev || !planes_dev) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (n_planes != 9 && n_planes != 4) return fail(APZ_E_ARG, "n_planes must be 9 or 4");
    if (n <= 0) return APZ_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    Timed tm(e, APZ_K_ENCODE);
    const int total = n * e->hw;
    hipLaunchKernelGGL(apz::encode_planes_kernel, dim3(std::min((total + 255) / 256, e->num_cu * 8)), dim3(256), 0,
                       e->stream, (const unsigned char*)codes_dev, (float*)planes_dev, n, e->cfg.height, e->cfg.width,
                       e->code_stride, n_planes);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_augment8(apz_engine* e, const void* planes_dev, const void* pi_dev, int n, int c, void* planes_out_dev,
                 void* pi_out_dev) {
    if (!e || !planes_dev || !pi_dev || !planes_out_dev || !pi_out_dev) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (!e->perm_s) return fail(APZ_E_UNSUPPORTED, "augmentation needs a square board");
    if (n <= 0) return APZ_OK;
    HIP_TRY(hipSetDevice(e->cfg.device));
    const long total = (long)n * 8 * (c + 1) * e->hw;
    hipLaunchKernelGGL(apz::augment8_kernel, dim3((int)std::min<long>((total + 255) / 256, e->num_cu * 16)), dim3(256),
                       0, e->stream, (const float*)planes_dev, (const float*)pi_dev, e->perm_s, e->perm_p,
                       (float*)planes_out_dev, (float*)pi_out_dev, n, c, e->hw);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_sample_moves_host(apz_engine* e, const int32_t* visits_host, int g, float temp, float alpha, float eps,
                          uint64_t seed, uint64_t step, float* pi_host, int32_t* moves_host) {
    return apz_sample_moves_keyed_host(e, visits_host, g, temp, alpha, eps, seed, step, nullptr, pi_host, moves_host);
}

int apz_sample_moves_keyed_host(apz_engine* e, const int32_t* visits_host, int g, float temp, float alpha, float eps,
                                uint64_t seed, uint64_t step, const uint64_t* keys_host, float* pi_host,
                                int32_t* moves_host) {
    if (!e || !visits_host || !pi_host || !moves_host) return fail(APZ_E_ARG, "null argument");
    if (g < 1 || e->hw > 256 || !(temp > 0.f) || !(alpha > 0.f) || eps < 0.f || eps > 1.f)
        return fail(APZ_E_ARG, "bad sampler arguments");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    const size_t hw = e->hw, vb = (size_t)g * hw * sizeof(int32_t);
    if ((size_t)g > e->smp_cap) {
        if (e->smp_vis) hipFree(e->smp_vis);
        if (e->smp_pi) hipFree(e->smp_pi);
        if (e->smp_mv) hipFree(e->smp_mv);
        if (e->smp_keys) hipFree(e->smp_keys);
        e->smp_vis = nullptr; e->smp_pi = nullptr; e->smp_mv = nullptr; e->smp_keys = nullptr; e->smp_cap = 0;
        HIP_TRY(hipMalloc((void**)&e->smp_keys, (size_t)g * sizeof(uint64_t)));
        HIP_TRY(hipMalloc((void**)&e->smp_vis, vb));
        HIP_TRY(hipMalloc((void**)&e->smp_pi, (size_t)g * hw * sizeof(float)));
        HIP_TRY(hipMalloc((void**)&e->smp_mv, (size_t)g * sizeof(int32_t)));
        e->smp_cap = g;
    }
    HIP_TRY(hipMemcpyAsync(e->smp_vis, visits_host, vb, hipMemcpyHostToDevice, e->stream));
    if (keys_host)
        HIP_TRY(hipMemcpyAsync(e->smp_keys, keys_host, (size_t)g * sizeof(uint64_t), hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(apz::root_sample_kernel, dim3(g), dim3(64), 0, e->stream, e->smp_vis, e->smp_pi, e->smp_mv, g,
                       (int)hw, 1.0f / temp, alpha, eps, (unsigned long long)seed, (unsigned long long)step,
                       keys_host ? (const unsigned long long*)e->smp_keys : nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpyAsync(pi_host, e->smp_pi, (size_t)g * hw * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipMemcpyAsync(moves_host, e->smp_mv, (size_t)g * sizeof(int32_t), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return APZ_OK;
}

int64_t apz_conv3x3_packed_size(int cin_p, int cout_p) {
    if (cin_p < 1 || cout_p < 16 || cout_p % 16) return -1;
    return (int64_t)(cout_p / 16) * ((cin_p + 3) / 4) * 9 * 64;
}

namespace {
struct StreamScope {      // run the engine's launch helpers on a caller-supplied stream
    apz_engine* e;
    hipStream_t saved;
    StreamScope(apz_engine* e_, void* s) : e(e_), saved(e_->stream) {
        if (s != APZ_ENGINE_STREAM) e->stream = (hipStream_t)s;   // NULL is a valid handle: the null stream
        // The training entry points share per-engine scratch (head_scratch, bn_part, fold_ws, wgw_scratch, wino_scratch,
        // adam_tab).  Calls on ONE stream are ordered by the stream; a caller that switches streams gets the new stream
        // ordered behind everything queued on the previous one, so two streams never work on the scratch at once.
        if (e->scratch_stream_valid && e->scratch_stream != e->stream) {
            hipEvent_t ev = get_event(e);
            if (hipEventRecord(ev, e->scratch_stream) == hipSuccess) hipStreamWaitEvent(e->stream, ev, 0);
            e->free_events.push_back(ev);
        }
        e->scratch_stream = e->stream;
        e->scratch_stream_valid = true;
    }
    ~StreamScope() { e->stream = saved; }
};
}  // namespace

// The same as apz_load_weights for tensors that already live in DEVICE memory (the trainer's): folding and packing run
// as kernels on `stream` (APZ_ENGINE_STREAM: the engine's own), and the engine's stream waits for them.
int apz_load_weights_dev(apz_engine* e, const char* const* names, const void* const* dev_ptrs, const int64_t* sizes, int n,
                         void* stream) {
    if (!e || !names || !dev_ptrs || !sizes) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    drop_forward_graphs(e);
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->loaded) return fail(APZ_E_STATE, "apz_load_weights_dev refreshes a loaded engine: call apz_load_weights first");
    std::map<std::string, const float*> P;
    std::map<std::string, int64_t> S;
    for (int i = 0; i < n; i++) {
        P[names[i]] = (const float*)dev_ptrs[i];
        S[names[i]] = sizes[i];
    }
    for (auto& p : e->params) {
        auto it = S.find(p.name);
        if (it == S.end()) return fail(APZ_E_ARG, "missing parameter " + p.name);
        if (it->second != p.size)
            return fail(APZ_E_ARG, "parameter " + p.name + " has " + std::to_string(it->second) + " elements, expected " +
                                       std::to_string(p.size));
    }
    // everything that can fail without having touched the weights comes first: shapes, scratch
    for (auto& L : e->convs)
        if (L.cout > 256) return fail(APZ_E_UNSUPPORTED, "load_weights_dev: more than 256 channels");
    if (!e->fold_ws) HIP_TRY(hipMalloc((void**)&e->fold_ws, 2 * 256 * sizeof(double)));
    hipStream_t engine_stream = e->stream;
    {
        StreamScope sc(e, stream);
        hipStream_t st = e->stream;
        if (st != engine_stream) {      // forwards already queued on the engine's stream still read the old weights
            hipEvent_t ev = get_event(e);
            HIP_TRY(hipEventRecord(ev, engine_stream));
            HIP_TRY(hipStreamWaitEvent(st, ev, 0));
            e->free_events.push_back(ev);
        }
        // From here on packing kernels are queued on `st`: whatever happens below, the engine's stream must end up
        // ordered behind them (a half-refreshed evaluator that ALSO races the packing kernels would be worse).
        struct Closing {
            apz_engine* e;
            hipStream_t st, engine_stream;
            ~Closing() {
                if (st == engine_stream) return;
                hipEvent_t ev = get_event(e);
                if (hipEventRecord(ev, st) == hipSuccess) hipStreamWaitEvent(engine_stream, ev, 0);
                e->free_events.push_back(ev);
            }
        } closing{e, st, engine_stream};
        double* scale = e->fold_ws;
        double* shift = e->fold_ws + 256;
        auto fold = [&](const std::string& conv, const std::string& bn, const std::string& mean_sfx, const std::string& var_sfx,
                        bool fix_gamma, int cout, float* bias_out) {
            hipLaunchKernelGGL(apz::fold_bn_kernel, dim3((cout + 63) / 64), dim3(64), 0, st, P.at(conv + "_bias"),
                               fix_gamma ? (const float*)nullptr : P.at(bn + "_gamma"), P.at(bn + "_beta"), P.at(bn + mean_sfx),
                               P.at(bn + var_sfx), scale, shift, bias_out, cout, (double)BN_EPS);
        };
        for (auto& L : e->convs) {
            fold(L.name, L.bn, L.mean_sfx, L.var_sfx, L.fix_gamma, L.cout, L.bias);
            const float* w = P.at(L.name + "_weight");
            const int n4 = L.cin_pad / 4, ncot = L.cout / 16;
            const bool x4 = e->ring && &L != &e->convs[0];
            const int total = ncot * n4 * 9 * 64;
            hipLaunchKernelGGL(apz::pack_direct_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, st, w, scale, L.wpk,
                               L.cin, n4, ncot, (int)x4);
            if (x4)
                hipLaunchKernelGGL(apz::pack_wino_folded_kernel, dim3(128 * 128 / 256), dim3(256), 0, st, w, scale, L.upk2, L.upk3s);
            if (x4 && e->trunk_arith == APZ_ARITH_BF16X3) {
                if (!L.upk3b) HIP_TRY(hipMalloc(&L.upk3b, apz::Wino3B::UPK_BYTES));
                hipLaunchKernelGGL(apz::pack_wino3b_folded_kernel, dim3(128 * 128 / 256), dim3(256), 0, st, w, scale,
                                   (unsigned short*)L.upk3b);
            }
            if (x4 && e->trunk_arith == APZ_ARITH_F16X2) {
                if (!L.upk3h) HIP_TRY(hipMalloc(&L.upk3h, apz::Wino3H::UPK_BYTES));
                if (!L.bias3h) HIP_TRY(hipMalloc(&L.bias3h, apz::Wino3H::BIAS_FLOATS * sizeof(float)));
                hipLaunchKernelGGL(apz::pack_wino3h_folded_kernel, dim3(128), dim3(128), 0, st, w, scale, shift,
                                   (unsigned short*)L.upk3h, L.bias3h);
            }
            if (e->small8 && L.wpk12)
                hipLaunchKernelGGL(apz::pack_direct_kernel, dim3(std::min((total + 255) / 256, 2048)), dim3(256), 0, st, w, scale,
                                   L.wpk12, L.cin, n4, ncot, 1);
            if (e->small8 && e->trunk_arith == APZ_ARITH_F16X2 && apz::Conv8H::supports(L.cin, L.cout)) {
                if (!L.wpk8h) HIP_TRY(hipMalloc(&L.wpk8h, apz::Conv8H::pk_bytes(L.cin, L.cout)));
                if (!L.bias8h) HIP_TRY(hipMalloc(&L.bias8h, 2 * (size_t)L.cout * sizeof(float)));
                hipLaunchKernelGGL(apz::pack_conv8h_kernel, dim3(L.cout), dim3(256), 0, st, w, scale, shift, (unsigned short*)L.wpk8h,
                                   L.bias8h, L.cin, L.cout);
            }
            HIP_TRY(hipGetLastError());
        }
        const int C = e->clast, hw = e->hw;
        fold("conv3_1_1", "conv3_1_1", "_mean", "_var", true, 4, nullptr);
        hipLaunchKernelGGL(apz::pack_head_conv_kernel, dim3((4 * C + 255) / 256), dim3(256), 0, st, P.at("conv3_1_1_weight"), scale,
                           shift, e->w6, e->b6, 4, 0, C);
        fold("conv3_2_1", "conv3_2_1", "_mean", "_var", true, 2, nullptr);
        hipLaunchKernelGGL(apz::pack_head_conv_kernel, dim3((2 * C + 255) / 256), dim3(256), 0, st, P.at("conv3_2_1_weight"), scale,
                           shift, e->w6, e->b6, 2, 4, C);
        const int ntile = (hw + 15) / 16, KG = (hw + 7) / 8;
        hipLaunchKernelGGL(apz::pack_fc_kernel, dim3(std::min((ntile * KG * 512 + 255) / 256, 2048)), dim3(256), 0, st,
                           P.at("fc_3_1_1_weight"), e->wfc_pk, hw, ntile, KG);
        HIP_TRY(hipGetLastError());
        if (e->small8 && e->wfc_raw)
            HIP_TRY(hipMemcpyAsync(e->wfc_raw, P.at("fc_3_1_1_weight"), (size_t)hw * 4 * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->bfc, P.at("fc_3_1_1_bias"), hw * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->wv, P.at("fc_3_2_1_weight"), 2 * hw * sizeof(float), hipMemcpyDeviceToDevice, st));
        HIP_TRY(hipMemcpyAsync(e->bv, P.at("fc_3_2_1_bias"), sizeof(float), hipMemcpyDeviceToDevice, st));
        // (`closing` orders the engine's stream behind all of the above: forwards queued from now on see the new weights)
    }
    return APZ_OK;
}

int apz_conv3x3_pack(apz_engine* e, const void* w_dev, int cin, int cout, int transpose_flip, void* wpk_dev,
                     void* stream) {
    if (!e || !w_dev || !wpk_dev || cin < 1 || cout < 1) return fail(APZ_E_ARG, "bad argument");
    const int co_p = transpose_flip ? cin : cout;
    if (co_p % 16) return fail(APZ_E_UNSUPPORTED, "packed C_out must be a multiple of 16");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const long total = apz_conv3x3_packed_size(transpose_flip ? cout : cin, co_p);
    hipLaunchKernelGGL(apz::pack_conv3x3_kernel, dim3((int)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0,
                       e->stream, (const float*)w_dev, (float*)wpk_dev, cin, cout, transpose_flip);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_conv3x3_fwd(apz_engine* e, const void* x_dev, const void* wpk_dev, const void* bias_dev, void* y_dev, int n,
                    int cin_p, int cout_p, int relu, void* stream) {
    if (!e || !x_dev || !wpk_dev || !y_dev || n < 1 || cin_p < 1) return fail(APZ_E_ARG, "bad argument");
    if (cout_p != 64 && cout_p != 128 && cout_p != 256)
        return fail(APZ_E_UNSUPPORTED, "conv3x3_fwd: C_out must be 64, 128 or 256");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->zeros256) {
        HIP_TRY(hipMalloc((void**)&e->zeros256, 256 * sizeof(float)));
        HIP_TRY(hipMemset(e->zeros256, 0, 256 * sizeof(float)));
    }
    StreamScope sc(e, stream);
    ConvLayer L;
    L.cin = cin_p;
    L.cin_pad = (cin_p + 3) / 4 * 4;
    L.cout = cout_p;
    L.residual = false;
    L.wpk = (float*)wpk_dev;
    L.bias = bias_dev ? (float*)bias_dev : e->zeros256;
    const int H = e->cfg.height, W = e->cfg.width;
    // small batches: split the output channels over 2 or 4 workgroups per board so that
    // boards x groups covers the CUs (each group stages the board's input itself)
    int groups = 1;
    while (groups * 64 < cout_p && n * groups * 2 <= e->num_cu * 2 && n * groups < e->num_cu) groups *= 2;
    const int ct = cout_p / 64 / groups;
    int rc = APZ_E_UNSUPPORTED;
#define APZ_DENSE(HH, WW, CT) rc = launch_conv_r<HH, WW, CT, false>(e, L, (const float*)x_dev, nullptr, (float*)y_dev, n, HH * WW, WW, relu, groups)
    if (H == 15 && W == 15) {
        if (ct == 1) APZ_DENSE(15, 15, 1); else if (ct == 2) APZ_DENSE(15, 15, 2); else APZ_DENSE(15, 15, 4);
    } else if (H == 8 && W == 8) {
        if (ct == 1) APZ_DENSE(8, 8, 1); else if (ct == 2) APZ_DENSE(8, 8, 2); else APZ_DENSE(8, 8, 4);
    }
#undef APZ_DENSE
    L.wpk = nullptr;
    L.bias = nullptr;
    if (rc == APZ_E_UNSUPPORTED) return fail(rc, "conv3x3_fwd: unsupported board size");
    return rc;
}

int64_t apz_wino_packed_size(void) { return (int64_t)apz::WinoPack::UPK_FLOATS; }

int apz_wino_pack(apz_engine* e, const void* w_dev, int transpose_flip, void* upk_dev, void* stream) {
    if (!e || !w_dev || !upk_dev) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    hipLaunchKernelGGL(apz::pack_wino2_kernel, dim3(8 * 2 * 32 * 64 / 256), dim3(256), 0, e->stream, (const float*)w_dev,
                       (float*)upk_dev, transpose_flip);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_wino_pack_many(apz_engine* e, const void* w_dev, int count, void* upk_dev, void* stream) {
    if (!e || !w_dev || !upk_dev || count < 1 || count > 16384) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    hipLaunchKernelGGL(apz::pack_wino2_many_kernel, dim3(8 * 2 * 32 * 64 / 256, 2 * count), dim3(256), 0, e->stream,
                       (const float*)w_dev, (float*)upk_dev);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_wino_conv_add(apz_engine* e, const void* x_dev, const void* upk_dev, const void* bias_dev, const void* resid_dev,
                      void* y_dev, int n, int relu, int layout, void* stream) {
    if (!e || !x_dev || !upk_dev || !y_dev || n < 1 || layout < 0 || layout > 1) return fail(APZ_E_ARG, "bad argument");
    if (e->cfg.height != 15 || e->cfg.width != 15) return fail(APZ_E_UNSUPPORTED, "wino_conv: 15x15 boards only");
    if (resid_dev && layout != APZ_LAYOUT_ROWS16) return fail(APZ_E_UNSUPPORTED, "wino_conv: residual input in the padded-row layout only");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->zeros256) {
        HIP_TRY(hipMalloc((void**)&e->zeros256, 256 * sizeof(float)));
        HIP_TRY(hipMemset(e->zeros256, 0, 256 * sizeof(float)));
    }
    StreamScope sc(e, stream);
    if (layout == APZ_LAYOUT_DENSE && (size_t)n > e->wino_scratch_boards) {   // grow the rows16 copies (previous users are ordered on their stream)
        HIP_TRY(hipDeviceSynchronize());
        for (int i = 0; i < 2; i++) {
            if (e->wino_scratch[i]) HIP_TRY(hipFree(e->wino_scratch[i]));
            e->wino_scratch[i] = nullptr;
            HIP_TRY(hipMalloc((void**)&e->wino_scratch[i], (size_t)n * 128 * 240 * sizeof(float)));
        }
        e->wino_scratch_boards = n;
    }
    const long planes = (long)n * 128;
    const int cgrid = (int)std::min<long>((planes * 240 + 255) / 256, 16384);
    const bool dense = layout == APZ_LAYOUT_DENSE;
    const float* xin = dense ? e->wino_scratch[0] : (const float*)x_dev;
    float* yout = dense ? e->wino_scratch[1] : (float*)y_dev;
    if (dense)
        hipLaunchKernelGGL(apz::rows16_from_dense_kernel, dim3(cgrid), dim3(256), 0, e->stream, (const float*)x_dev,
                           e->wino_scratch[0], planes);
    const float* b = bias_dev ? (const float*)bias_dev : e->zeros256;
    const float* rs = (const float*)resid_dev;
    // the self-play path's kernel (csrc/trunk15_wino3.h)
    int rc;
    if (rs && relu) rc = launch_wino3_t<true, true>(e, 6, xin, (const float*)upk_dev, b, rs, yout, n);
    else if (rs) rc = launch_wino3_t<true, false>(e, 17, xin, (const float*)upk_dev, b, rs, yout, n);
    else if (relu) rc = launch_wino3_t<false, true>(e, 7, xin, (const float*)upk_dev, b, nullptr, yout, n);
    else rc = launch_wino3_t<false, false>(e, 19, xin, (const float*)upk_dev, b, nullptr, yout, n);
    if (rc) return rc;
    if (dense)
        hipLaunchKernelGGL(apz::rows16_to_dense_kernel, dim3(cgrid), dim3(256), 0, e->stream, e->wino_scratch[1],
                           (float*)y_dev, planes);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_wino_conv_stats(apz_engine* e, const void* x_dev, const void* upk_dev, const void* bias_dev, void* y_dev,
                        void* stats_dev, int n, void* stream) {
    if (!e || !x_dev || !upk_dev || !y_dev || !stats_dev || n < 1) return fail(APZ_E_ARG, "bad argument");
    if (e->cfg.height != 15 || e->cfg.width != 15) return fail(APZ_E_UNSUPPORTED, "wino_conv: 15x15 boards only");
    if (n > WINO3_MAX_BOARDS) return fail(APZ_E_UNSUPPORTED, "wino_conv_stats: one launch (<= 16384 boards)");
    using T = apz::Wino3;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    if (!e->zeros256) {
        HIP_TRY(hipMalloc((void**)&e->zeros256, 256 * sizeof(float)));
        HIP_TRY(hipMemset(e->zeros256, 0, 256 * sizeof(float)));
    }
    StreamScope sc(e, stream);
    bool& configured = e->lds_attr_set[11];
    if (!configured) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<false, false, false, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)apz::trunk15_wino3_kernel<false, false, true, true>,
                                    hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
        configured = true;
    }
    const float* b = bias_dev ? (const float*)bias_dev : e->zeros256;
    bool quarter = false;
    const int grid = apz::wino3_grid(n, e->num_cu, e->no_quarter_trunk ? nullptr : &quarter);
    if (quarter)
        hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false, false, true, true>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream,
                           (const float*)x_dev, (const float*)upk_dev, b, (const float*)stats_dev, (float*)y_dev, n);
    else
        hipLaunchKernelGGL((apz::trunk15_wino3_kernel<false, false, false, true>), dim3(grid), dim3(512), T::LDS_BYTES, e->stream,
                           (const float*)x_dev, (const float*)upk_dev, b, (const float*)stats_dev, (float*)y_dev, n);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_wino_conv(apz_engine* e, const void* x_dev, const void* upk_dev, const void* bias_dev, void* y_dev, int n, int relu,
                  int layout, void* stream) {
    return apz_wino_conv_add(e, x_dev, upk_dev, bias_dev, nullptr, y_dev, n, relu, layout, stream);
}

namespace {
int wgrad_scratch(apz_engine* e, size_t floats) {   // per-slice partial weight gradients (both weight-gradient kernels)
    if (floats > e->wgw_floats) {
        HIP_TRY(hipDeviceSynchronize());
        if (e->wgw_scratch) HIP_TRY(hipFree(e->wgw_scratch));
        e->wgw_scratch = nullptr;
        e->wgw_floats = 0;
        HIP_TRY(hipMalloc((void**)&e->wgw_scratch, floats * sizeof(float)));
        e->wgw_floats = floats;
    }
    return APZ_OK;
}
}  // namespace

int apz_conv3x3_wgrad(apz_engine* e, const void* x_dev, const void* dy_dev, void* dw_dev, int n, int cin, int cout,
                      int layout, void* stream) {
    if (!e || !x_dev || !dy_dev || !dw_dev || n < 1 || cin < 1 || cout < 32 || cout % 32 || layout < 0 || layout > 1)
        return fail(APZ_E_ARG, "bad argument (C_out must be a multiple of 32)");
    if (layout == APZ_LAYOUT_ROWS16 && (e->cfg.height != 15 || e->cfg.width != 15))
        return fail(APZ_E_UNSUPPORTED, "padded-row layout: 15x15 boards only");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const int H = e->cfg.height, W = e->cfg.width;
    const int gx = cout / 32, gy = (cin + 63) / 64;
    int slices = std::max(1, std::min(n, (e->num_cu * 2) / std::max(1, gx * gy)));
    if (H == 15 && W == 15) {
        using G = apz::WgradGeo<15, 15>;
        if (!e->wgrad_attr_set[0]) {
            HIP_TRY(hipFuncSetAttribute((const void*)apz::conv3x3_wgrad_kernel<15, 15>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
            e->wgrad_attr_set[0] = true;
        }
        slices = std::max(1, std::min(n, e->num_cu / std::max(1, gx * gy)));   // 85 KB LDS: one workgroup per CU
        if (int rc = wgrad_scratch(e, (size_t)slices * cout * cin * 9)) return rc;
        if (layout == APZ_LAYOUT_ROWS16) {
            using G16 = apz::WgradGeo<15, 15, true>;
            bool& set16 = e->lds_attr_set[8];       // per engine (= per device), like every other attribute flag
            if (!set16) {
                HIP_TRY(hipFuncSetAttribute((const void*)apz::conv3x3_wgrad_kernel<15, 15, true>,
                                            hipFuncAttributeMaxDynamicSharedMemorySize, G16::LDS_BYTES));
                set16 = true;
            }
            hipLaunchKernelGGL((apz::conv3x3_wgrad_kernel<15, 15, true>), dim3(gx, gy, slices), dim3(256), G16::LDS_BYTES,
                               e->stream, (const float*)x_dev, (const float*)dy_dev, e->wgw_scratch, n, cin, cout);
        } else
            hipLaunchKernelGGL((apz::conv3x3_wgrad_kernel<15, 15>), dim3(gx, gy, slices), dim3(256), G::LDS_BYTES, e->stream,
                               (const float*)x_dev, (const float*)dy_dev, e->wgw_scratch, n, cin, cout);
    } else if (H == 8 && W == 8) {
        using G = apz::WgradGeo<8, 8>;
        if (int rc = wgrad_scratch(e, (size_t)slices * cout * cin * 9)) return rc;
        if (!e->wgrad_attr_set[1]) {
            HIP_TRY(hipFuncSetAttribute((const void*)apz::conv3x3_wgrad_kernel<8, 8>,
                                        hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS_BYTES));
            e->wgrad_attr_set[1] = true;
        }
        hipLaunchKernelGGL((apz::conv3x3_wgrad_kernel<8, 8>), dim3(gx, gy, slices), dim3(256), G::LDS_BYTES, e->stream,
                           (const float*)x_dev, (const float*)dy_dev, e->wgw_scratch, n, cin, cout);
    } else {
        return fail(APZ_E_UNSUPPORTED, "conv3x3_wgrad: unsupported board size");
    }
    // the slices' partial sums, added in index order
    hipLaunchKernelGGL(apz::colsum_kernel, dim3((cout * cin * 9 + 63) / 64), dim3(256), 0, e->stream, (const float*)e->wgw_scratch,
                       (float*)dw_dev, slices, cout * cin * 9, 1.0f);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

namespace {
int bn_geometry(apz_engine* e, int layout, int* ps, int* rs) {
    const int H = e->cfg.height, W = e->cfg.width;
    if (layout == APZ_LAYOUT_DENSE) {
        *ps = H * W, *rs = W;
    } else if (layout == APZ_LAYOUT_ROWS16 && H == 15 && W == 15) {
        *ps = 240, *rs = 16;
    } else {
        return fail(APZ_E_UNSUPPORTED, "padded-row layout: 15x15 boards only");
    }
    return APZ_OK;
}
constexpr int BN_SPLITS = 64;      // batch splits per channel at most (every consumer workgroup adds them)
int bn_scratch(apz_engine* e) {    // per-(channel, split) partial sums: overwritten by every statistics launch, nothing to zero
    if (!e->bn_part) HIP_TRY(hipMalloc((void**)&e->bn_part, (size_t)256 * BN_SPLITS * 2 * sizeof(double)));
    return APZ_OK;
}
// grid.y of the four BatchNorm launches: channels x splits ~ 8 workgroups per CU (padded rows: four boards per trip)
int bn_splits(const apz_engine* e, int n, int C, int layout) {
    const int per = layout == APZ_LAYOUT_ROWS16 ? (n + 3) / 4 : n;
    return std::max(1, std::min({per, (e->num_cu * (layout == APZ_LAYOUT_ROWS16 ? 8 : 4)) / C, BN_SPLITS}));
}
}  // namespace

int apz_bn_fwd(apz_engine* e, const void* x_dev, const void* resid_dev, const void* gamma_dev, const void* beta_dev,
               void* run_mean_dev, void* run_var_dev, void* y_dev, void* mean_dev, void* invstd_dev, int n, int C, int layout,
               int relu, float momentum, float eps, void* stream) {
    return apz_bn_fwd_stats(e, x_dev, resid_dev, gamma_dev, beta_dev, run_mean_dev, run_var_dev, y_dev, mean_dev, invstd_dev,
                            nullptr, nullptr, n, C, layout, relu, momentum, eps, stream);
}

int apz_bn_fwd_stats(apz_engine* e, const void* x_dev, const void* resid_dev, const void* gamma_dev, const void* beta_dev,
                     void* run_mean_dev, void* run_var_dev, void* y_dev, void* mean_dev, void* invstd_dev,
                     const void* stats_dev, void* mask_dev, int n, int C, int layout, int relu, float momentum, float eps,
                     void* stream) {
    if (!e || !x_dev || !beta_dev || !y_dev || !mean_dev || !invstd_dev || n < 1 || C < 1 || C > 256)
        return fail(APZ_E_ARG, "bad argument");
    if ((stats_dev || mask_dev) && layout != APZ_LAYOUT_ROWS16) return fail(APZ_E_UNSUPPORTED, "bn_fwd_stats: padded-row layout only");
    int ps, rs;
    if (int rc = bn_geometry(e, layout, &ps, &rs)) return rc;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    if (int rc = bn_scratch(e)) return rc;
    const int H = e->cfg.height, W = e->cfg.width;
    const int splits = bn_splits(e, n, C, layout);
    const apz::BnFinal fin{(float*)mean_dev, (float*)invstd_dev, (float*)run_mean_dev, (float*)run_var_dev, (double)n * H * W, eps,
                           momentum};
    if (layout == APZ_LAYOUT_ROWS16) {
        // statistics from the producer (apz_wino_conv_stats: one pair of sums per channel and board): no pass over x for them
        if (!stats_dev)
            hipLaunchKernelGGL(apz::bn_stats_r16_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)x_dev, e->bn_part, n, C);
        hipLaunchKernelGGL(apz::bn_apply_r16_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)x_dev,
                           (const float*)resid_dev, (const float*)gamma_dev, (const float*)beta_dev,
                           stats_dev ? (const double*)stats_dev : (const double*)e->bn_part, stats_dev ? n : splits, fin,
                           (float*)y_dev, (unsigned char*)mask_dev, n, C, relu);
    } else {
        hipLaunchKernelGGL(apz::bn_stats_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)x_dev, e->bn_part, n, C,
                           ps, rs, H, W);
        hipLaunchKernelGGL(apz::bn_apply_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)x_dev,
                           (const float*)resid_dev, (const float*)gamma_dev, (const float*)beta_dev, (const double*)e->bn_part,
                           splits, fin, (float*)y_dev, n, C, ps, rs, H, W, relu);
    }
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_bn_bwd_splits(apz_engine* e, int n, int C, int layout) {
    if (!e || n < 1 || C < 1 || C > 256 || layout < 0 || layout > 1) return fail(APZ_E_ARG, "bad argument");
    return bn_splits(e, n, C, layout);
}

int apz_bn_bwd(apz_engine* e, const void* dy_dev, const void* x_dev, const void* out_dev, const void* mask_dev,
               const void* gamma_dev, const void* mean_dev, const void* invstd_dev, void* dx_dev, void* dres_dev, void* dgamma_dev,
               void* dbeta_dev, void* dxsum_dev, int dxsum_ld, int n, int C, int layout, int relu, void* stream) {
    if (!e || !dy_dev || !x_dev || !mean_dev || !invstd_dev || !dx_dev || n < 1 || C < 1 || C > 256 ||
        (relu && !out_dev && !mask_dev) || (dxsum_dev && dxsum_ld < C) || (mask_dev && layout != APZ_LAYOUT_ROWS16))
        return fail(APZ_E_ARG, "bad argument");
    int ps, rs;
    if (int rc = bn_geometry(e, layout, &ps, &rs)) return rc;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    if (int rc = bn_scratch(e)) return rc;
    const int H = e->cfg.height, W = e->cfg.width;
    const int splits = bn_splits(e, n, C, layout);
    if (layout == APZ_LAYOUT_ROWS16) {
        hipLaunchKernelGGL(apz::bn_bwd_reduce_r16_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)dy_dev,
                           (const float*)x_dev, (const float*)out_dev, (const float*)mean_dev, (const float*)invstd_dev,
                           e->bn_part, (const unsigned char*)mask_dev, n, C, relu);
        hipLaunchKernelGGL(apz::bn_bwd_apply_r16_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)dy_dev,
                           (const float*)x_dev, (const float*)out_dev, (const float*)gamma_dev, (const float*)mean_dev,
                           (const float*)invstd_dev, (const double*)e->bn_part, splits, (float*)dx_dev, (float*)dres_dev,
                           (float*)dgamma_dev, (float*)dbeta_dev, (float*)dxsum_dev, dxsum_ld, (const unsigned char*)mask_dev, n, C,
                           relu, (double)n * H * W);
    } else {
        hipLaunchKernelGGL(apz::bn_bwd_reduce_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)dy_dev,
                           (const float*)x_dev, (const float*)out_dev, (const float*)mean_dev, (const float*)invstd_dev,
                           e->bn_part, n, C, ps, rs, H, W, relu);
        hipLaunchKernelGGL(apz::bn_bwd_apply_kernel, dim3(C, splits), dim3(256), 0, e->stream, (const float*)dy_dev,
                           (const float*)x_dev, (const float*)out_dev, (const float*)gamma_dev, (const float*)mean_dev,
                           (const float*)invstd_dev, (const double*)e->bn_part, splits, (float*)dx_dev, (float*)dres_dev,
                           (float*)dgamma_dev, (float*)dbeta_dev, (float*)dxsum_dev, dxsum_ld, n, C, ps, rs, H, W, relu,
                           (double)n * H * W);
    }
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_colsum(apz_engine* e, const void* in_dev, void* out_dev, int rows, int cols, float scale, void* stream) {
    if (!e || !in_dev || !out_dev || rows < 1 || cols < 1) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    hipLaunchKernelGGL(apz::colsum_kernel, dim3((cols + 63) / 64), dim3(256), 0, e->stream, (const float*)in_dev, (float*)out_dev,
                       rows, cols, scale);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_wgrad_wino(apz_engine* e, const void* x_dev, const void* dy_dev, void* dw_dev, int n, void* stream) {
    if (!e || !x_dev || !dy_dev || !dw_dev || n < 1) return fail(APZ_E_ARG, "bad argument");
    if (e->cfg.height != 15 || e->cfg.width != 15) return fail(APZ_E_UNSUPPORTED, "wgrad_wino: 15x15 boards only");
    if (n > 32768) return fail(APZ_E_UNSUPPORTED, "wgrad_wino: at most 32768 boards per call (32-bit buffer offsets)");
    using T = apz::WgradWino;
    using T3 = apz::WgradWino3;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    // Decomposition by channel blocks (csrc/wgrad_wino3.h: 16 blocks x slices = one workgroup per CU); the workgroups of a
    // slice sit on one XCD (see the kernel)
    const int spx = std::max(1, std::min((n + 7) / 8, e->num_cu / (8 * T3::BLOCKS)));
    const int slices = 8 * spx;
    if (int rc = wgrad_scratch(e, (size_t)slices * T::SCRATCH_FLOATS_PER_SLICE)) return rc;
    bool& attr = e->lds_attr_set[10];
    if (!attr) {
        HIP_TRY(hipFuncSetAttribute((const void*)apz::wgrad_wino3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
        HIP_TRY(hipFuncSetAttribute((const void*)apz::wgrad_wino3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T3::LDS_BYTES));
        attr = true;
    }
    // rows through raw buffer loads (hardware zero fill for rows off the board, scalar board offset) from 256 boards on:
    // 133.6 against 144.4 us at 512 boards, 38.7 against 37.8 us at 128 (same box, alternating rounds: tools/wgrad_kernel_bench.hip)
    if (n >= 256)
        hipLaunchKernelGGL(apz::wgrad_wino3_kernel<true>, dim3(T3::BLOCKS * slices), dim3(T3::THREADS), T3::LDS_BYTES, e->stream,
                           (const float*)x_dev, (const float*)dy_dev, e->wgw_scratch, n, spx);
    else
        hipLaunchKernelGGL(apz::wgrad_wino3_kernel<false>, dim3(T3::BLOCKS * slices), dim3(T3::THREADS), T3::LDS_BYTES, e->stream,
                           (const float*)x_dev, (const float*)dy_dev, e->wgw_scratch, n, spx);
    hipLaunchKernelGGL(apz::wgrad_wino_finish_kernel, dim3(128 * 128 * 9 / 4 / 256), dim3(256), 0, e->stream,
                       (const float*)e->wgw_scratch, slices, (float*)dw_dev);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_adam_step(apz_engine* e, const void* table_host, int ntensors, float lr_t, float b1, float b2, float eps,
                  float rescale, void* stream) {
    if (!e || !table_host || ntensors < 1 || ntensors > 4096) return fail(APZ_E_ARG, "bad argument");
    static_assert(sizeof(apz::AdamTensor) == 48, "apz_adam_tensor layout");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const size_t bytes = (size_t)ntensors * sizeof(apz::AdamTensor);
    if (bytes > e->adam_cap) {
        if (e->adam_tab) HIP_TRY(hipFree(e->adam_tab));
        e->adam_tab = nullptr;
        HIP_TRY(hipMalloc((void**)&e->adam_tab, bytes));
        e->adam_cap = bytes;
    }
    // Stream-ordered upload: the copy queues behind the previous step's kernel (which read the same device table), and
    // for pageable host memory hipMemcpyAsync returns once the 6 KB have been staged, so the caller's buffer need
    // not outlive the call.  (No stream synchronisation: the optimiser step no longer drains the GPU.)
    HIP_TRY(hipMemcpyAsync(e->adam_tab, table_host, bytes, hipMemcpyHostToDevice, e->stream));
    hipLaunchKernelGGL(apz::adam_step_kernel, dim3(32, ntensors), dim3(256), 0, e->stream,
                       (const apz::AdamTensor*)e->adam_tab, lr_t, b1, b2, eps, rescale);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

// ---- heads and loss of the training graph (csrc/heads_train.h)
namespace {
// C[M][N] = A B (+ bias): 64 x 64 tiles per workgroup when those fill the chip, else 16 x 16 tiles with the k-steps split over the waves
void launch_sgemm(apz_engine* e, const float* a, const float* b, const float* bias, float* c, int M, int N, int K, long a_rs,
                  long a_cs, long b_rs, long b_cs, int ldc) {
    const int my = (M + 63) / 64;
    if (((N + 63) / 64) * my >= e->num_cu / 2)
        hipLaunchKernelGGL(apz::sgemm_mfma_kernel<4>, dim3((N + 63) / 64, my), dim3(256), 0, e->stream, a, b, bias, c, M, N, K, a_rs,
                           a_cs, b_rs, b_cs, ldc);
    else
        hipLaunchKernelGGL(apz::sgemm_ksplit_kernel, dim3((N + 15) / 16, (M + 15) / 16), dim3(256), 0, e->stream, a, b, bias, c, M,
                           N, K, a_rs, a_cs, b_rs, b_cs, ldc);
}

int head_scratch(apz_engine* e, size_t floats) {
    if (floats > e->head_scratch_floats) {
        if (e->head_scratch) HIP_TRY(hipFree(e->head_scratch));
        e->head_scratch = nullptr;
        HIP_TRY(hipMalloc((void**)&e->head_scratch, floats * sizeof(float)));
        e->head_scratch_floats = floats;
    }
    return APZ_OK;
}
}  // namespace

int apz_conv1x1_fwd(apz_engine* e, const void* x_dev, const void* w_dev, const void* bias_dev, void* y_dev, int n, int C,
                    int CO, int layout, void* stream) {
    if (!e || !x_dev || !w_dev || !y_dev || n < 1 || C < 1 || C > 1024 || CO < 1 || CO > 8) return fail(APZ_E_ARG, "bad argument");
    int ps, rs;
    if (int rc = bn_geometry(e, layout, &ps, &rs)) return rc;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    hipLaunchKernelGGL(apz::conv1x1_fwd_kernel, dim3(n, (e->cfg.height * e->cfg.width + 63) / 64), dim3(256),
                       ((size_t)CO * C + 4 * 8 * 64) * sizeof(float), e->stream,
                       (const float*)x_dev, (const float*)w_dev, (const float*)bias_dev, (float*)y_dev, C, CO, e->cfg.height,
                       e->cfg.width, ps, rs);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_conv1x1_bwd(apz_engine* e, const void* x_dev, const void* w_dev, const void* dy_dev, void* dx_dev, void* dw_dev,
                    void* db_dev, int n, int C, int CO, int layout, int accumulate_dx, void* stream) {
    if (int rc = apz_conv1x1_bwd2(e, x_dev, w_dev, dy_dev, CO, nullptr, nullptr, 0, dx_dev, dw_dev, n, C, layout, accumulate_dx, stream))
        return rc;
    if (db_dev) return apz_bias_grad(e, dy_dev, db_dev, n, CO, APZ_LAYOUT_DENSE, stream);
    return APZ_OK;
}

int apz_conv1x1_bwd2(apz_engine* e, const void* x_dev, const void* w1_dev, const void* dy1_dev, int CO1, const void* w2_dev,
                     const void* dy2_dev, int CO2, void* dx_dev, void* dw_dev, int n, int C, int layout, int accumulate_dx,
                     void* stream) {
    if (!e || !x_dev || !w1_dev || !dy1_dev || !dw_dev || n < 1 || C < 1 || C > 1024 || CO1 < 1 || CO2 < 0 || CO1 + CO2 > 8 ||
        (CO2 > 0 && (!w2_dev || !dy2_dev)))
        return fail(APZ_E_ARG, "bad argument");
    int ps, rs;
    if (int rc = bn_geometry(e, layout, &ps, &rs)) return rc;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const int P = e->cfg.height * e->cfg.width, CO = CO1 + CO2;
    if (int rc = head_scratch(e, (size_t)n * CO * C)) return rc;
    const size_t lds = ((size_t)CO * 32 + (size_t)CO * P) * sizeof(float);
    hipLaunchKernelGGL(apz::conv1x1_bwd_kernel, dim3(n, (C + 31) / 32), dim3(256), lds, e->stream, (const float*)x_dev,
                       (const float*)w1_dev, (const float*)dy1_dev, (const float*)w2_dev, (const float*)dy2_dev, (float*)dx_dev,
                       e->head_scratch, C, CO1, CO2, e->cfg.height, e->cfg.width, ps, rs, accumulate_dx);
    hipLaunchKernelGGL(apz::colsum_kernel, dim3((CO * C + 63) / 64), dim3(256), 0, e->stream, (const float*)e->head_scratch,
                       (float*)dw_dev, n, CO * C, 1.0f);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_bias_grad(apz_engine* e, const void* dy_dev, void* db_dev, int n, int C, int layout, void* stream) {
    if (!e || !dy_dev || !db_dev || n < 1 || C < 1) return fail(APZ_E_ARG, "bad argument");
    int ps, rs;
    if (int rc = bn_geometry(e, layout, &ps, &rs)) return rc;
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const int slices = std::max(1, std::min(n, (e->num_cu * 8 + C - 1) / C));
    if (int rc = head_scratch(e, (size_t)slices * C)) return rc;
    hipLaunchKernelGGL(apz::bias_grad_kernel, dim3(C, slices), dim3(256), 0, e->stream, (const float*)dy_dev, e->head_scratch, n, C,
                       ps);
    hipLaunchKernelGGL(apz::colsum_kernel, dim3((C + 63) / 64), dim3(256), 0, e->stream, (const float*)e->head_scratch,
                       (float*)db_dev, slices, C, 1.0f);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_add(apz_engine* e, void* y_dev, const void* x_dev, int64_t count, void* stream) {
    if (!e || !y_dev || !x_dev || count < 1) return fail(APZ_E_ARG, "bad argument");
    if (((uintptr_t)y_dev | (uintptr_t)x_dev) & 15) return fail(APZ_E_ARG, "add: 16-byte aligned tensors");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const long n4 = count / 4;
    hipLaunchKernelGGL(apz::add_inplace_kernel, dim3((int)std::max<long>(1, std::min<long>((n4 + 255) / 256, 8192))), dim3(256), 0,
                       e->stream, (float*)y_dev, (const float*)x_dev, n4, (long)count);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_fc_fwd(apz_engine* e, const void* x_dev, const void* w_dev, const void* bias_dev, void* y_dev, int n, int K, int N,
               void* stream) {
    if (!e || !x_dev || !w_dev || !y_dev || n < 1 || K < 1 || N < 1) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    // y[n][N] = x[n][K] W[N][K]^T + b:  A = x (row stride K), B(k, j) = W[j][k]
    launch_sgemm(e, (const float*)x_dev, (const float*)w_dev, (const float*)bias_dev, (float*)y_dev, n, N, K, (long)K, 1L, 1L,
                 (long)K, N);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_fc_bwd(apz_engine* e, const void* x_dev, const void* w_dev, const void* dy_dev, void* dx_dev, void* dw_dev, void* db_dev,
               int n, int K, int N, void* stream) {
    if (!e || !x_dev || !w_dev || !dy_dev || n < 1 || K < 1 || N < 1) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    if (dx_dev)     // dx[n][K] = dy[n][N] W[N][K]
        launch_sgemm(e, (const float*)dy_dev, (const float*)w_dev, nullptr, (float*)dx_dev, n, K, N, (long)N, 1L, (long)K, 1L, K);
    if (dw_dev)     // dW[N][K] = dy^T[N][n] x[n][K]:  A(m, k) = dy[k][m]
        launch_sgemm(e, (const float*)dy_dev, (const float*)x_dev, nullptr, (float*)dw_dev, N, K, n, 1L, (long)N, (long)K, 1L, K);
    if (db_dev)
        hipLaunchKernelGGL(apz::colsum_kernel, dim3((N + 63) / 64), dim3(256), 0, e->stream, (const float*)dy_dev, (float*)db_dev,
                           n, N, 1.0f);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_dropout(apz_engine* e, const void* x_dev, void* y_dev, int64_t count, float keep, uint64_t seed, uint64_t step,
                void* stream) {
    if (!e || !x_dev || !y_dev || count < 1 || !(keep > 0.f) || keep > 1.f) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    hipLaunchKernelGGL(apz::dropout_kernel, dim3((int)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, e->stream,
                       (const float*)x_dev, (float*)y_dev, (long)count, keep, (unsigned long long)seed, (unsigned long long)step);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_pv_loss(apz_engine* e, const void* logits_dev, const void* vlogit_dev, const void* pi_dev, const void* z_dev, int n,
                void* loss3_dev, void* dlogits_dev, void* dvlogit_dev, void* probs_dev, void* values_dev, void* stream) {
    if (!e || !logits_dev || !vlogit_dev || n < 1) return fail(APZ_E_ARG, "bad argument");
    if ((loss3_dev || dlogits_dev || dvlogit_dev) && (!pi_dev || !z_dev)) return fail(APZ_E_ARG, "targets missing");
    if (e->hw > 4096) return fail(APZ_E_UNSUPPORTED, "board too large");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    if (int rc = head_scratch(e, (size_t)n * 3)) return rc;
    hipLaunchKernelGGL(apz::pv_loss_kernel, dim3((n + 3) / 4), dim3(256), 0, e->stream, (const float*)logits_dev,
                       (const float*)vlogit_dev, (const float*)pi_dev, (const float*)z_dev, n, e->hw, 1.0f / (float)n,
                       loss3_dev ? e->head_scratch : (float*)nullptr, (float*)dlogits_dev, (float*)dvlogit_dev, (float*)probs_dev,
                       (float*)values_dev);
    if (loss3_dev)  // (value loss, policy loss, entropy): batch means, samples summed in index order
        hipLaunchKernelGGL(apz::colsum_kernel, dim3(1), dim3(256), 0, e->stream, (const float*)e->head_scratch, (float*)loss3_dev, n,
                           3, 1.0f / (float)n);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_layout_convert(apz_engine* e, const void* src_dev, void* dst_dev, int64_t planes, int to_rows16, void* stream) {
    if (!e || !src_dev || !dst_dev || planes < 1) return fail(APZ_E_ARG, "bad argument");
    if (e->cfg.height != 15 || e->cfg.width != 15) return fail(APZ_E_UNSUPPORTED, "padded-row layout: 15x15 boards only");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    StreamScope sc(e, stream);
    const int grid = (int)std::min<int64_t>((planes * 240 + 255) / 256, 16384);
    if (to_rows16)
        hipLaunchKernelGGL(apz::rows16_from_dense_kernel, dim3(grid), dim3(256), 0, e->stream, (const float*)src_dev, (float*)dst_dev,
                           (long)planes);
    else
        hipLaunchKernelGGL(apz::rows16_to_dense_kernel, dim3(grid), dim3(256), 0, e->stream, (const float*)src_dev, (float*)dst_dev,
                           (long)planes);
    HIP_TRY(hipGetLastError());
    return APZ_OK;
}

int apz_prewarm(apz_engine* e, int n, int iters) {
    if (!e) return fail(APZ_E_ARG, "null engine");
    if (n < 1 || n > e->cfg.max_batch || iters < 0) return fail(APZ_E_ARG, "prewarm: bad batch / iteration count");
    EngineLock guard(e->submit_lock);
    if (!e->loaded) return fail(APZ_E_STATE, "weights not loaded");
    HIP_TRY(hipSetDevice(e->cfg.device));
    // e->codes is zero-filled at creation and only ever overwritten with valid codes: any content is a legal input
    ArmOverflowWord arm(e);
    for (int i = 0; i < iters; i++) {
        int rc;
        if (stem_takes_codes(e)) {
            rc = forward_dev(e, nullptr, n, e->probs, e->values, nullptr, nullptr, e->codes);
        } else {
            rc = apz_encode_planes(e, e->codes, n, e->cfg.c_in, e->planes);
            if (!rc) rc = forward_dev(e, e->planes, n, e->probs, e->values, nullptr, nullptr);
        }
        if (rc) return rc;
    }
    return APZ_OK;
}

int apz_sync(apz_engine* e) {
    if (!e) return fail(APZ_E_ARG, "null engine");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipStreamSynchronize(e->stream));
    resolve_pending(e);
    return APZ_OK;
}

void* apz_stream(apz_engine* e) { return e ? (void*)e->stream : nullptr; }

void* apz_device_alloc(apz_engine* e, int64_t bytes) {
    void* p = nullptr;
    if (!e || bytes <= 0 || hipSetDevice(e->cfg.device) != hipSuccess || hipMalloc(&p, (size_t)bytes) != hipSuccess) {
        fail(APZ_E_HIP, "hipMalloc failed");
        return nullptr;
    }
    return p;
}

void apz_device_free(apz_engine* e, void* p) {
    if (e && p) {
        hipSetDevice(e->cfg.device);
        hipFree(p);
    }
}

int apz_memcpy_h2d(apz_engine* e, void* dst_dev, const void* src_host, int64_t bytes) {
    if (!e || !dst_dev || !src_host || bytes < 0) return fail(APZ_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpyAsync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return APZ_OK;
}

int apz_memcpy_d2h(apz_engine* e, void* dst_host, const void* src_dev, int64_t bytes) {
    if (!e || !dst_host || !src_dev || bytes < 0) return fail(APZ_E_ARG, "bad argument");
    HIP_TRY(hipSetDevice(e->cfg.device));
    HIP_TRY(hipMemcpyAsync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    return APZ_OK;
}

int apz_conv3x3_bench(apz_engine* e, int layer, int n, int iters, int warmup, float* ms_out) {
    if (!e || !ms_out) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (!e->loaded) return fail(APZ_E_STATE, "weights not loaded");
    if (layer < 0 || layer >= (int)e->convs.size() || n < 1 || n > e->cfg.max_batch || iters < 1 || warmup < 0)
        return fail(APZ_E_ARG, "bad layer / batch / iteration count");
    HIP_TRY(hipSetDevice(e->cfg.device));
    const bool was = e->profiling;
    e->profiling = false;
    ArmOverflowWord arm(e);
    // produce this layer's real input from the planes resident in e->planes
    const float* in = e->planes;
    if (layer > 0) {
        float* prev = nullptr;
        int rc = run_trunk(e, e->planes, n, layer - 1, &prev);
        if (rc) { e->profiling = was; return rc; }
        in = prev;
    }
    const ConvLayer& L = e->convs[layer];
    float* out = nullptr;
    const float* resid = nullptr;
    for (int i = 0; i < 3; i++)
        if (e->act[i] != in && !out) out = e->act[i];
    if (L.residual)
        for (int i = 0; i < 3; i++)
            if (e->act[i] != in && e->act[i] != out) resid = e->act[i];
    hipEvent_t a = get_event(e), b = get_event(e);   // pooled, destroyed with the engine
    int rc = APZ_OK;
    for (int i = 0; i < warmup && !rc; i++) rc = launch_conv(e, L, in, resid, out, n);
    hipError_t he = hipEventRecord(a, e->stream);
    for (int i = 0; i < iters && !rc; i++) rc = launch_conv(e, L, in, resid, out, n);
    if (he == hipSuccess) he = hipEventRecord(b, e->stream);
    if (he == hipSuccess) he = hipStreamSynchronize(e->stream);
    float ms = 0.f;
    if (he == hipSuccess) he = hipEventElapsedTime(&ms, a, b);
    e->free_events.push_back(a);
    e->free_events.push_back(b);
    e->profiling = was;
    if (he != hipSuccess) return fail(APZ_E_HIP, std::string("conv bench timing: ") + hipGetErrorString(he));
    if (rc) return rc;
    ms_out[0] = ms / iters;
    return APZ_OK;
}

int apz_layer_io(apz_engine* e, int layer, float* host_out, int64_t count) {
    if (!e || !host_out) return fail(APZ_E_ARG, "null argument");
    EngineLock guard(e->submit_lock);
    if (!e->loaded || e->last_n < 1) return fail(APZ_E_STATE, "run a forward first");
    if (layer < 0 || layer >= (int)e->convs.size()) return fail(APZ_E_ARG, "bad layer");
    const int64_t need = (int64_t)e->last_n * e->convs[layer].cout * e->hw;
    if (count != need) return fail(APZ_E_ARG, "count must be n*C_out*H*W = " + std::to_string(need));
    HIP_TRY(hipSetDevice(e->cfg.device));
    float* buf = nullptr;
    const bool was = e->profiling;
    e->profiling = false;
    ArmOverflowWord arm(e);
    int rc = run_trunk(e, e->planes, e->last_n, layer, &buf);
    e->profiling = was;
    if (rc) return rc;
    if (!e->ring) {
        HIP_TRY(hipMemcpyAsync(host_out, buf, need * sizeof(float), hipMemcpyDeviceToHost, e->stream));
        HIP_TRY(hipStreamSynchronize(e->stream));
        return APZ_OK;
    }
    const int C = e->convs[layer].cout, H = e->cfg.height, W = e->cfg.width;
    std::vector<float> tmp((size_t)e->last_n * C * e->act_ps);
    HIP_TRY(hipMemcpyAsync(tmp.data(), buf, tmp.size() * sizeof(float), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    for (size_t pc = 0; pc < (size_t)e->last_n * C; pc++)
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) host_out[(pc * H + y) * W + x] = tmp[pc * e->act_ps + y * e->act_rs + x];
    return APZ_OK;
}

int apz_set_trunk_arith(apz_engine* e, int arith) {
    if (e) {
        EngineLock guard_(e->submit_lock);
        drop_forward_graphs(e);
    }
    if (!e || (arith != APZ_ARITH_F32 && arith != APZ_ARITH_BF16X3 && arith != APZ_ARITH_F16X2))
        return fail(APZ_E_ARG, "bad trunk arithmetic");
    EngineLock guard(e->submit_lock);
    if (arith == APZ_ARITH_BF16X3 && !e->ring)
        return fail(APZ_E_UNSUPPORTED, "the bf16 x 3 trunk kernel exists for the 15x15 / 128-filter residual net only");
    if (arith == APZ_ARITH_F16X2 && !e->ring && !e->small8)
        return fail(APZ_E_UNSUPPORTED, "the fp16 x 2 split kernels exist for the 15x15 / 128-filter residual net and for 8x8 boards");
    if (arith == APZ_ARITH_F16X2 && !e->ovf_host) {
        HIP_TRY(hipSetDevice(e->cfg.device));
        HIP_TRY(hipHostMalloc((void**)&e->ovf_host, (APZ_MAX_SLOTS + 1) * sizeof(unsigned), hipHostMallocMapped));
        std::memset(e->ovf_host, 0, (APZ_MAX_SLOTS + 1) * sizeof(unsigned));
        HIP_TRY(hipHostGetDevicePointer((void**)&e->ovf_dev, e->ovf_host, 0));
    }
    if (e->loaded && arith != e->trunk_arith)
        return fail(APZ_E_STATE, "apz_set_trunk_arith must be called before the weights are loaded");
    e->trunk_arith = arith;
    return APZ_OK;
}

long apz_trunk_overflows(apz_engine* e) { return e ? e->ovf_repeats : -1; }

int apz_test_select_trunk(apz_engine* e, int kind) {
    if (e) {
        EngineLock guard_(e->submit_lock);
        drop_forward_graphs(e);
    }
    if (!e || (kind != APZ_TRUNK_WINOGRAD && kind != APZ_TRUNK_DIRECT && kind != APZ_TRUNK_WINOGRAD_BATCHED &&
               kind != APZ_TRUNK_WINOGRAD_NO_QUARTER))
        return fail(APZ_E_ARG, "bad trunk kernel kind");
    EngineLock guard(e->submit_lock);
    if (!e->ring) return fail(APZ_E_UNSUPPORTED, "only the 15x15 / 128-filter residual net has two trunk kernels");
    e->trunk_kernel = kind == APZ_TRUNK_DIRECT ? APZ_TRUNK_DIRECT : APZ_TRUNK_WINOGRAD;
    e->no_small_trunk = kind == APZ_TRUNK_WINOGRAD_BATCHED || kind == APZ_TRUNK_WINOGRAD_NO_QUARTER;
    e->no_quarter_trunk = kind == APZ_TRUNK_WINOGRAD_NO_QUARTER;
    return APZ_OK;
}

int apz_set_forward_graphs(apz_engine* e, int on) {
    if (!e) return fail(APZ_E_ARG, "null engine");
    EngineLock guard(e->submit_lock);
    e->use_graphs = on != 0;
    if (!on) drop_forward_graphs(e);
    return APZ_OK;
}

int apz_set_profiling(apz_engine* e, int on) {
    if (!e) return fail(APZ_E_ARG, "null engine");
    EngineLock guard(e->submit_lock);
    HIP_TRY(hipStreamSynchronize(e->stream));
    resolve_pending(e);
    e->profiling = on != 0;
    e->prof_stride = on > 1 ? on : 1;   // on = k > 1: sample every k-th forward
    e->prof_phase = 0;
    for (int i = 0; i < APZ_K_COUNT; i++) {
        e->k_ms[i] = 0;
        e->k_cnt[i] = 0;
    }
    return APZ_OK;
}

int apz_kernel_time_ms(apz_engine* e, int kernel_class, float* out2) {
    if (!e || !out2 || kernel_class < 0 || kernel_class >= APZ_K_COUNT) return fail(APZ_E_ARG, "bad argument");
    EngineLock guard(e->submit_lock);
    out2[0] = (float)e->k_ms[kernel_class];
    out2[1] = (float)e->k_cnt[kernel_class];
    return APZ_OK;
}

}  // extern "C"
