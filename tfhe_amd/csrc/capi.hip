// C ABI of the library (declared in include/tfhe_amd.h): context management, workspace,
// stream handling and kernel timing around the launchers in kernels.hip.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/tfhe_amd.h"
#include "field.hpp"
#include "kernels.hpp"

using namespace tfhe_amd;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(e_ == hipErrorOutOfMemory ? TFHE_AMD_ERR_MEMORY : TFHE_AMD_ERR_HIP,     \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                    \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool ok = true;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) ok = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        int cur = -1;
        if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
    }
};

constexpr int kTimingRing = 256;

}  // namespace

struct tfhe_amd_ctx {
    tfhe_amd_params p{};
    int device = 0;
    NttUniform u{};
    d2* d_tw = nullptr;
    d2* d_bk_ntt = nullptr;
    uint32_t* d_ks = nullptr;
    hipStream_t stream = nullptr;
    // workspace, grown on demand
    uint32_t* d_c = nullptr;
    uint32_t* d_lv1 = nullptr;
    size_t ws_gates = 0;
    // staging for the host-pointer entry points
    uint32_t* d_io = nullptr;
    size_t io_gates = 0;
    // timing
    bool timing = false;
    std::vector<hipEvent_t> ev;  // kTimingRing * 3
    int ev_head = 0, ev_count = 0;
    double br_ms = 0.0, ks_ms = 0.0;
    int64_t calls = 0;
};

namespace {

bool params_sane(const tfhe_amd_params* p) {
    if (!p) return false;
    if (p->n < 1 || p->K < 1 || p->L < 1 || p->T < 1 || p->base_bit < 1) return false;
    if (p->N < 2 || (p->N & (p->N - 1))) return false;
    if (p->bg_bit != 8 && p->bg_bit != 16) return false;
    if (p->L * p->bg_bit > 31 || p->T * p->base_bit > 31) return false;
    return true;
}

bool blind_rotate_supported(const tfhe_amd_params* p) {
    return p->N == 512 && p->K == 2 && p->L == 2 && p->bg_bit == 8;
}
bool key_switch_supported(const tfhe_amd_params* p) { return key_switch_has_kernel(p->n + 1, p->T, p->base_bit); }

void build_uniform(NttUniform& u, const TwiddleSet& t) {
    u.p = (double)kP;
    u.pinv = 1.0 / (double)kP;
    u.zI = (double)(kZeta * kZeta);
    u.z = (double)kZeta;
    for (int i = 0; i < 8; i++) u.fw[i] = u.fwp[i] = u.iw[i] = u.iwp[i] = 0.0;
    for (int i = 1; i < 8; i++) {
        const TwPair f = tw_pair(t.fwd[i]), v = tw_pair(t.inv[i]);
        u.fw[i] = f.w; u.fwp[i] = f.wp;
        u.iw[i] = v.w; u.iwp[i] = v.wp;
    }
    const TwPair ni = tw_pair(t.n_inv);
    u.ninv = ni.w;
    u.ninvp = ni.wp;
}

// LDS twiddle block for N = 512 (section layout in ntt512.hpp)
std::vector<double> build_tw_block_512(const TwiddleSet& t) {
    std::vector<double> blk((size_t)kTwPairs * 2);
    auto put = [&](int pair, uint64_t v) {
        const TwPair tp = tw_pair(v);
        blk[(size_t)pair * 2] = tp.w;
        blk[(size_t)pair * 2 + 1] = tp.wp;
    };
    auto fill = [&](int base, int rows, int m1, const std::vector<uint64_t>& tab) {
        // row r: [0] = tab[m1 + r], [1..2] = tab[2 m1 + 2r + g], [3..6] = tab[4 m1 + 4r + g]
        for (int r = 0; r < rows; r++) {
            put(base + 7 * r, tab[m1 + r]);
            for (int g = 0; g < 2; g++) put(base + 7 * r + 1 + g, tab[2 * m1 + 2 * r + g]);
            for (int g = 0; g < 4; g++) put(base + 7 * r + 3 + g, tab[4 * m1 + 4 * r + g]);
        }
    };
    fill(kTwFwdB, 8, 8, t.fwd);
    fill(kTwFwdC, 64, 64, t.fwd);
    fill(kTwInvB, 8, 8, t.inv);
    fill(kTwInvC, 64, 64, t.inv);
    return blk;
}

int ensure_workspace(tfhe_amd_ctx* ctx, size_t n_gates) {
    if (n_gates <= ctx->ws_gates) return TFHE_AMD_OK;
    if (ctx->d_c) (void)hipFree(ctx->d_c);
    if (ctx->d_lv1) (void)hipFree(ctx->d_lv1);
    ctx->d_c = ctx->d_lv1 = nullptr;
    ctx->ws_gates = 0;
    const size_t n1 = (size_t)ctx->p.n + 1, l1 = (size_t)ctx->p.N * ctx->p.K + 1;
    HIP_TRY(hipMalloc(&ctx->d_c, n_gates * n1 * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&ctx->d_lv1, n_gates * l1 * sizeof(uint32_t)));
    ctx->ws_gates = n_gates;
    return TFHE_AMD_OK;
}

int ensure_io(tfhe_amd_ctx* ctx, size_t n_gates) {
    if (n_gates <= ctx->io_gates) return TFHE_AMD_OK;
    if (ctx->d_io) (void)hipFree(ctx->d_io);
    ctx->d_io = nullptr;
    ctx->io_gates = 0;
    HIP_TRY(hipMalloc(&ctx->d_io, 3 * n_gates * ((size_t)ctx->p.n + 1) * sizeof(uint32_t)));
    ctx->io_gates = n_gates;
    return TFHE_AMD_OK;
}

int drain_timing(tfhe_amd_ctx* ctx, int keep) {
    while (ctx->ev_count > keep) {
        hipEvent_t* e = &ctx->ev[(size_t)ctx->ev_head * 3];
        HIP_TRY(hipEventSynchronize(e[2]));
        float a = 0.f, b = 0.f;
        HIP_TRY(hipEventElapsedTime(&a, e[0], e[1]));
        HIP_TRY(hipEventElapsedTime(&b, e[1], e[2]));
        ctx->br_ms += a;
        ctx->ks_ms += b;
        ctx->calls++;
        ctx->ev_head = (ctx->ev_head + 1) % kTimingRing;
        ctx->ev_count--;
    }
    return TFHE_AMD_OK;
}

// bootstrap + key switch on device data; c may alias nothing else
int run_bootstrap(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates, hipStream_t s) {
    const tfhe_amd_params& p = ctx->p;
    hipEvent_t* e = nullptr;
    if (ctx->timing) {
        if (ctx->ev_count == kTimingRing) {
            int rc = drain_timing(ctx, kTimingRing - 1);
            if (rc) return rc;
        }
        e = &ctx->ev[(size_t)((ctx->ev_head + ctx->ev_count) % kTimingRing) * 3];
        HIP_TRY(hipEventRecord(e[0], s));
    }
    HIP_TRY(launch_blind_rotate_512(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c, ctx->d_lv1, (int)n_gates, p.n, p.K, p.L, s));
    if (e) HIP_TRY(hipEventRecord(e[1], s));
    HIP_TRY(launch_key_switch(ctx->d_lv1, ctx->d_ks, out, (int)n_gates, p.n + 1, p.N * p.K, p.T, p.base_bit, s));
    if (e) {
        HIP_TRY(hipEventRecord(e[2], s));
        ctx->ev_count++;
    }
    return TFHE_AMD_OK;
}

int create_common(const tfhe_amd_params* p, const uint32_t* bk, const uint32_t* ks, bool on_device, int device,
                  tfhe_amd_ctx** out) {
    if (!out) return fail(TFHE_AMD_ERR_ARGUMENT, "out is null");
    *out = nullptr;
    if (!params_sane(p) || !bk || !ks) return fail(TFHE_AMD_ERR_ARGUMENT, "bad parameters or null key pointer");
    int rc = tfhe_amd_params_supported(p);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");

    tfhe_amd_ctx* ctx = new (std::nothrow) tfhe_amd_ctx();
    if (!ctx) return fail(TFHE_AMD_ERR_MEMORY, "host allocation failed");
    ctx->p = *p;
    ctx->device = device;
    auto bail = [&](int code) {
        tfhe_amd_ctx_destroy(ctx);
        return code;
    };
#define CTX_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return bail(fail(e_ == hipErrorOutOfMemory ? TFHE_AMD_ERR_MEMORY : TFHE_AMD_ERR_HIP,    \
                             std::string(#expr) + ": " + hipGetErrorString(e_)));                  \
    } while (0)

    CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    const TwiddleSet tw = make_twiddles(p->N);
    build_uniform(ctx->u, tw);
    const std::vector<double> blk = build_tw_block_512(tw);
    CTX_TRY(hipMalloc(&ctx->d_tw, blk.size() * sizeof(double)));
    CTX_TRY(hipMemcpyAsync(ctx->d_tw, blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));

    const size_t bk_words = tfhe_amd_bk_words(p), ks_words = tfhe_amd_ks_words(p);
    const int n1 = p->n + 1, nlv1 = p->N * p->K;
    CTX_TRY(hipMalloc(&ctx->d_bk_ntt, bk_words * sizeof(double)));
    CTX_TRY(hipMalloc(&ctx->d_ks, key_switch_packed_words(n1, nlv1, p->T, p->base_bit) * sizeof(uint32_t)));
    // raw keys: uploaded to scratch buffers (host source) or used in place (device source)
    uint32_t *d_bk_raw = nullptr, *d_ks_raw = nullptr;
    const uint32_t *bk_src = bk, *ks_src = ks;
    hipError_t e2 = hipSuccess;
    if (!on_device) {
        e2 = hipMalloc(&d_bk_raw, bk_words * sizeof(uint32_t));
        if (e2 == hipSuccess) e2 = hipMalloc(&d_ks_raw, ks_words * sizeof(uint32_t));
        if (e2 == hipSuccess)
            e2 = hipMemcpyAsync(d_bk_raw, bk, bk_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        if (e2 == hipSuccess)
            e2 = hipMemcpyAsync(d_ks_raw, ks, ks_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream);
        bk_src = d_bk_raw;
        ks_src = d_ks_raw;
    }
    if (e2 == hipSuccess)
        e2 = launch_bk_transform_512(ctx->u, ctx->d_tw, bk_src, ctx->d_bk_ntt, (int)(bk_words / p->N), ctx->stream);
    if (e2 == hipSuccess) e2 = launch_ks_pack(ctx->d_ks, ks_src, n1, nlv1, p->T, p->base_bit, ctx->stream);
    if (e2 == hipSuccess) e2 = hipStreamSynchronize(ctx->stream);
    if (d_bk_raw) (void)hipFree(d_bk_raw);
    if (d_ks_raw) (void)hipFree(d_ks_raw);
    CTX_TRY(e2);
#undef CTX_TRY
    *out = ctx;
    return TFHE_AMD_OK;
}

}  // namespace

extern "C" {

void tfhe_amd_params_default(tfhe_amd_params* p) {
    if (!p) return;
    p->n = 636; p->N = 512; p->K = 2; p->bg_bit = 8; p->L = 2; p->T = 5; p->base_bit = 2;
}

int tfhe_amd_params_supported(const tfhe_amd_params* p) {
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    if (!blind_rotate_supported(p)) return fail(TFHE_AMD_ERR_UNSUPPORTED, "no blind-rotation kernel for this ring/gadget");
    if (!key_switch_supported(p)) return fail(TFHE_AMD_ERR_UNSUPPORTED, "no key-switch kernel for this base/size");
    return TFHE_AMD_OK;
}

size_t tfhe_amd_bk_words(const tfhe_amd_params* p) {
    return (size_t)p->n * (p->K + 1) * p->L * (p->K + 1) * p->N;
}
size_t tfhe_amd_ks_words(const tfhe_amd_params* p) {
    return (size_t)p->N * p->K * p->T * ((size_t)1 << p->base_bit) * (p->n + 1);
}

int tfhe_amd_ctx_create(const tfhe_amd_params* p, const uint32_t* bk_host, const uint32_t* ks_host, int device,
                        tfhe_amd_ctx** out) {
    return create_common(p, bk_host, ks_host, false, device, out);
}

int tfhe_amd_ctx_create_device(const tfhe_amd_params* p, const uint32_t* bk_dev, const uint32_t* ks_dev, int device,
                               tfhe_amd_ctx** out) {
    return create_common(p, bk_dev, ks_dev, true, device, out);
}

void tfhe_amd_ctx_destroy(tfhe_amd_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipEvent_t e : ctx->ev) (void)hipEventDestroy(e);
    if (ctx->d_tw) (void)hipFree(ctx->d_tw);
    if (ctx->d_bk_ntt) (void)hipFree(ctx->d_bk_ntt);
    if (ctx->d_ks) (void)hipFree(ctx->d_ks);
    if (ctx->d_c) (void)hipFree(ctx->d_c);
    if (ctx->d_lv1) (void)hipFree(ctx->d_lv1);
    if (ctx->d_io) (void)hipFree(ctx->d_io);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int tfhe_amd_bootstrap_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates,
                                    void* stream) {
    if (!ctx || (n_gates && (!c || !out))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    if (n_gates > (size_t)1 << 30) return fail(TFHE_AMD_ERR_ARGUMENT, "batch too large");
    DeviceGuard guard(ctx->device);
    int rc = ensure_workspace(ctx, n_gates);
    if (rc) return rc;
    return run_bootstrap(ctx, c, out, n_gates, stream ? (hipStream_t)stream : ctx->stream);
}

int tfhe_amd_nand_batch_device(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
                               size_t n_gates, void* stream) {
    if (!ctx || (n_gates && (!a || !b || !out))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    if (n_gates > (size_t)1 << 30) return fail(TFHE_AMD_ERR_ARGUMENT, "batch too large");
    DeviceGuard guard(ctx->device);
    int rc = ensure_workspace(ctx, n_gates);
    if (rc) return rc;
    hipStream_t s = stream ? (hipStream_t)stream : ctx->stream;
    HIP_TRY(launch_nand_combine(ctx->d_c, a, b, n_gates, ctx->p.n + 1, s));
    return run_bootstrap(ctx, ctx->d_c, out, n_gates, s);
}

int tfhe_amd_gate_bootstrapping_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* lv1, size_t n_gates,
                                             void* stream) {
    if (!ctx || (n_gates && (!c || !lv1))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    DeviceGuard guard(ctx->device);
    const tfhe_amd_params& p = ctx->p;
    HIP_TRY(launch_blind_rotate_512(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c, lv1, (int)n_gates, p.n, p.K, p.L,
                                    stream ? (hipStream_t)stream : ctx->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_identity_key_switch_batch_device(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out,
                                              size_t n_gates, void* stream) {
    if (!ctx || (n_gates && (!lv1 || !out))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    DeviceGuard guard(ctx->device);
    const tfhe_amd_params& p = ctx->p;
    HIP_TRY(launch_key_switch(lv1, ctx->d_ks, out, (int)n_gates, p.n + 1, p.N * p.K, p.T, p.base_bit,
                              stream ? (hipStream_t)stream : ctx->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_nand_batch(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n_gates) {
    if (!ctx || (n_gates && (!a || !b || !out))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    DeviceGuard guard(ctx->device);
    int rc = ensure_io(ctx, n_gates);
    if (rc) return rc;
    const size_t words = n_gates * ((size_t)ctx->p.n + 1), bytes = words * sizeof(uint32_t);
    uint32_t *da = ctx->d_io, *db = ctx->d_io + words, *dout = ctx->d_io + 2 * words;
    HIP_TRY(hipMemcpyAsync(da, a, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(db, b, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = tfhe_amd_nand_batch_device(ctx, da, db, dout, n_gates, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_bootstrap_batch(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates) {
    if (!ctx || (n_gates && (!c || !out))) return fail(TFHE_AMD_ERR_ARGUMENT, "null argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    DeviceGuard guard(ctx->device);
    int rc = ensure_io(ctx, n_gates);
    if (rc) return rc;
    const size_t words = n_gates * ((size_t)ctx->p.n + 1), bytes = words * sizeof(uint32_t);
    uint32_t *dc = ctx->d_io, *dout = ctx->d_io + 2 * words;
    HIP_TRY(hipMemcpyAsync(dc, c, bytes, hipMemcpyHostToDevice, ctx->stream));
    rc = tfhe_amd_bootstrap_batch_device(ctx, dc, dout, n_gates, nullptr);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, dout, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_ctx_synchronize(tfhe_amd_ctx* ctx) {
    if (!ctx) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    DeviceGuard guard(ctx->device);
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return TFHE_AMD_OK;
}

int tfhe_amd_ctx_enable_timing(tfhe_amd_ctx* ctx, int enable) {
    if (!ctx) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    DeviceGuard guard(ctx->device);
    if (enable && ctx->ev.empty()) {
        ctx->ev.resize((size_t)kTimingRing * 3);
        for (hipEvent_t& e : ctx->ev) {
            e = nullptr;
            HIP_TRY(hipEventCreate(&e));
        }
    }
    if (!enable) {
        int rc = drain_timing(ctx, 0);
        if (rc) return rc;
    }
    ctx->timing = enable != 0;
    return TFHE_AMD_OK;
}

int tfhe_amd_ctx_timing(tfhe_amd_ctx* ctx, double* blind_rotate_ms, double* key_switch_ms, int64_t* calls) {
    if (!ctx) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    DeviceGuard guard(ctx->device);
    int rc = drain_timing(ctx, 0);
    if (rc) return rc;
    if (blind_rotate_ms) *blind_rotate_ms = ctx->br_ms;
    if (key_switch_ms) *key_switch_ms = ctx->ks_ms;
    if (calls) *calls = ctx->calls;
    ctx->br_ms = ctx->ks_ms = 0.0;
    ctx->calls = 0;
    return TFHE_AMD_OK;
}

const char* tfhe_amd_last_error(void) { return g_last_error.c_str(); }
const char* tfhe_amd_version(void) { return "0.1.0"; }

}  // extern "C"
