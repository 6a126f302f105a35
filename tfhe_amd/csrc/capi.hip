// C ABI of the library (declared in include/tfhe_amd.h): context management, workspace,
// stream handling and kernel timing around the launchers in kernels.hip and
// generic_kernels.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <atomic>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <string>
#include <system_error>
#include <vector>

#include "../../include/tfhe_amd.h"
#include "field.hpp"
#include "kernels.hpp"
#include "fft256_wave.hpp"
#include "ntt512_half.hpp"
#include "ntt_wg.hpp"

using namespace tfhe_amd;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string& msg) {
    g_last_error = msg;
    return code;
}

int hip_code(hipError_t e) { return e == hipErrorOutOfMemory ? TFHE_AMD_ERR_MEMORY : TFHE_AMD_ERR_HIP; }

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail(hip_code(e_), std::string(#expr) + ": " + hipGetErrorString(e_)); \
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

// device scratch buffer freed on scope exit
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 4); }
    template <typename T> T* as() { return static_cast<T*>(p); }
};

constexpr int kTimingRing = 256;

// Workspace of tfhe_amd_nand_batch_device_noalloc: the combined inputs c [G][n+1] and the lv1
// results [G][N*K+1] of one call.  Every GPU the library runs on holds its own copy of this array
// for the life of the process; the name below is only a host-side handle, the address on a device
// comes from hipGetSymbolAddress with that device current (scratch_of).
constexpr size_t kNoAllocWords = TFHE_AMD_NOALLOC_SCRATCH_BYTES / sizeof(uint32_t);
__device__ __align__(256) uint32_t scratch[kNoAllocWords];

constexpr int kMaxDevices = 64;
// Per-device state of the static workspace: its address on that device, and the event and stream of
// the last call that used it, so that calls on different streams run one behind the other.
struct ScratchState {
    uint32_t* ptr = nullptr;
    hipEvent_t done = nullptr;    // recorded behind the last kernel that touches the workspace
    hipStream_t last = nullptr;   // stream of that call
    bool used = false;
    std::mutex mu;                // one call at a time enqueues on a device
};
ScratchState g_scratch[kMaxDevices];
std::mutex g_scratch_mutex;

// State of `scratch` on GPU `device`, which must be the current device.  The address and the event
// are resolved once per device (when the first context on it is created: no allocation later).
hipError_t scratch_of(int device, ScratchState** out) {
    if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    ScratchState& st = g_scratch[device];
    if (!st.ptr) {
        void* ptr = nullptr;
        hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!st.done) {
            e = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
            if (e != hipSuccess) return e;
        }
        st.ptr = static_cast<uint32_t*>(ptr);
    }
    *out = &st;
    return hipSuccess;
}

}  // namespace

namespace {
constexpr int kMaxLanes = 8;

struct Lane {
    hipStream_t stream = nullptr;  // private stream of the synchronous host-pointer calls
    // workspace, grown on demand
    uint32_t* d_c = nullptr;
    uint32_t* d_lv1 = nullptr;
    size_t ws_gates = 0;
    // Calls that use this lane on different streams are ordered through this event: every call
    // records it behind its last kernel, and a call on another stream waits for it before touching
    // the workspace.
    hipEvent_t ws_done = nullptr;
    hipStream_t ws_stream = nullptr;
    bool ws_used = false;
    // staging for the host-pointer entry points
    uint32_t* d_io = nullptr;
    size_t io_words = 0;
    bool busy = false;
};
}  // namespace

struct tfhe_amd_ctx {
    tfhe_amd_params p{};
    int device = 0;
    bool fast = false;             // specialised NTT kernels (N = 512, K = 1 or 2, L = 2, 8-bit gadget)
    bool wg = false;               // workgroup-cooperative NTT kernels (N = 2048 set)
    bool replica = false;          // products through a replay of the reference's FFT (verification)
    bool replica_generic = false;  // ... with the one-workgroup-per-gate kernel even at the default set
    d2* d_spec = nullptr;          // replica: key spectra
    int tiles = 0, shift = 0, pack_shift = 0, pack = 0;  // key split of the wg path (wg_kernels.hip)
    NttUniform u{};
    d2* d_tw = nullptr;
    d2* d_bk_ntt = nullptr;        // fast path
    uint32_t* d_bk_raw = nullptr;  // generic path
    uint32_t* d_ks = nullptr;      // packed
    bool has_bk = false, has_ks = false;
    int form = 0;                  // blind-rotation kernel form: 0 auto, 1 latency, 2 throughput
    hipStream_t stream = nullptr;
    // Workspaces.  A call that needs scratch memory (the combined inputs c, the lv1 results, the
    // staging of host-pointer calls) leases a lane for as long as it enqueues (device-pointer calls) or
    // runs (host-pointer calls); several host threads may therefore evaluate on one context at once,
    // as they may call the reference's nand() on one shared EvaluateKey (reference src/hom_nand.hpp:74-81
    // reads only its arguments).  Lanes are created on demand, at most kMaxLanes of them.
    std::mutex mu;
    std::condition_variable cv;
    std::vector<Lane*> lanes;
    ScratchState* scratch = nullptr;  // this device's copy of the static no-allocation workspace
    std::mutex timing_mu;
    // timing
    std::atomic<bool> timing{false};
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

bool fast_supported(const tfhe_amd_params* p) {
    return p->N == 512 && (p->K == 1 || p->K == 2) && p->L == 2 && p->bg_bit == 8;
}

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
    std::vector<double> blk((size_t)(kTwPairs + kHalfPairs) * 2);
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
    // table of the two-wave transforms (ntt512_half.hpp), appended.  Forward rows: group g starts at stage
    // 2g - 1 with m = 2^(2g-1) rows [tab[m + U], tab[2m + 2U], tab[2m + 2U + 1]]
    for (int g = 1; g <= 4; g++) {
        const int m = 1 << (2 * g - 1);
        for (int U = 0; U < m; U++) {
            const int pair = kTwPairs + 3 * (half_row_offset(g) + U);
            put(pair, t.fwd[m + U]);
            put(pair + 1, t.fwd[2 * m + 2 * U]);
            put(pair + 2, t.fwd[2 * m + 2 * U + 1]);
        }
    }
    // parity-split inverse: first-stage twiddles [k][lane], then rows of the same shape for m = 64, 16, 4, 1
    const int par = kTwPairs + 3 * kHalfRows;
    for (int k = 0; k < 4; k++)
        for (int l = 0; l < 64; l++) put(par + 64 * k + l, t.inv[256 + 4 * l + k]);
    for (int m = 64; m >= 1; m /= 4)
        for (int U = 0; U < m; U++) {
            const int pair = par + 256 + 3 * (parity_row_offset(m) + U);
            put(pair, t.inv[m + U]);
            put(pair + 1, t.inv[2 * m + 2 * U]);
            put(pair + 2, t.inv[2 * m + 2 * U + 1]);
        }
    return blk;
}

/* cos and sin of pi / 2^j for j = 3 .. 11, correctly rounded (computed in 80-digit arithmetic from the
 * double nearest to pi / 2^j; x86-64 glibc, whose std::polar the reference's recorded outputs come from,
 * returns the same values).  The power tables of the reference's FFT product (reference
 * src/types.hpp:257-268,276-283) start from these, so they do not depend on the math library of
 * the machine that runs this code. */
const double kUnitRoot[9][2] = {
    {0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2},  {0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60ap-3},
    {0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4},  {0x1.ff621e3796d7ep-1, 0x1.91f65f10dd814p-5},
    {0x1.ffd886084cd0dp-1, 0x1.92155f7a3667ep-6},  {0x1.fff62169b92dbp-1, 0x1.921d1fcdec784p-7},
    {0x1.fffd8858e8a92p-1, 0x1.921f0fe670071p-8},  {0x1.ffff621621d02p-1, 0x1.921f8becca4bap-9},
    {0x1.ffffd88586ee6p-1, 0x1.921faaee6472dp-10}};

// Power tables of the reference's FFT product for ring degree N (M = N / 2), as the reference builds
// them (src/types.hpp:257-268,276-283): [0] twist powers w^i, w = e^(i pi / N), by repeated
// multiplication; [1] omega^i, omega = e^(-2 pi i / M), by repeated multiplication; [2] untwist
// powers; [3] the omega table with entries 1 .. M-1 reversed.  The three starting values are what
// std::polar returns for angles pi / 2^j (kUnitRoot; libm outside its range).
std::vector<double> build_replica_tables(int N) {
    const int M = N / 2;
    std::vector<double> t((size_t)4 * M * 2);
    auto cmul = [](double xr, double xi, double yr, double yi, double& rr, double& ri) {
        const double ac = xr * yr, bd = xi * yi, ad = xr * yi, bc = xi * yr;
        rr = ac - bd;
        ri = ad + bc;
    };
    // std::polar(1.0, sign * pi / 2^j)
    auto polar = [](int j, int sign, double& r, double& i) {
        if (j >= 3 && j <= 11) {
            r = kUnitRoot[j - 3][0];
            i = sign * kUnitRoot[j - 3][1];
        } else {
            volatile double th = sign * M_PI / (double)(1u << j);
            r = 1.0 * std::cos(th);
            i = 1.0 * std::sin(th);
        }
    };
    int logM = 0;
    while ((1 << logM) < M) logM++;
    auto fill_powers = [&](int section, int j, int sign) {
        double wr, wi, pr = 1.0, pi = 0.0;
        polar(j, sign, wr, wi);
        for (int i = 0; i < M; i++) {
            t[((size_t)section * M + i) * 2] = pr;
            t[((size_t)section * M + i) * 2 + 1] = pi;
            double nr, ni;
            cmul(pr, pi, wr, wi, nr, ni);
            pr = nr;
            pi = ni;
        }
    };
    fill_powers(0, logM + 1, 1);   // M_PI / (M * 2)
    fill_powers(1, logM - 1, -1);  // -M_PI * 2 / M
    fill_powers(2, logM + 1, -1);  // -M_PI / (M * 2)
    for (int i = 0; i < M; i++) {
        const int src = i == 0 ? 0 : M - i;
        t[((size_t)3 * M + i) * 2] = t[((size_t)1 * M + src) * 2];
        t[((size_t)3 * M + i) * 2 + 1] = t[((size_t)1 * M + src) * 2 + 1];
    }
    if (M == 256) {
        // block of the wave-per-gate kernel (fft256_wave.hpp): per-lane twiddle rows of both directions
        t.resize((size_t)(4 * M + kFftFastPairs) * 2);
        auto copy = [&](size_t dst_pair, size_t src_pair) {
            t[dst_pair * 2] = t[src_pair * 2];
            t[dst_pair * 2 + 1] = t[src_pair * 2 + 1];
        };
        const size_t base = (size_t)4 * M;
        for (int dir = 0; dir < 2; dir++) {
            const size_t tab = (size_t)(dir ? 3 : 1) * M;
            for (int g = 1; g <= 4; g++)
                for (int U = 0; U < (1 << (2 * (g - 1))); U++) {
                    const size_t row = base + (dir ? kFftInvRows : kFftFwdRows) + 3 * (size_t)(fft_row_offset(g) + U);
                    copy(row, tab + bit_reverse(2 * U, 8));
                    copy(row + 1, tab + bit_reverse(4 * U, 8));
                    copy(row + 2, tab + bit_reverse(4 * U + 2, 8));
                }
        }
    }
    return t;
}

// twiddle table of the workgroup NTT: forward rows of groups 1.., then inverse rows (ntt_wg.hpp)
template <int LOGN>
std::vector<double> build_tw_table_wg(const TwiddleSet& t) {
    using W = WgNtt<LOGN>;
    std::vector<double> blk((size_t)W::TABLE_PAIRS * 2, 0.0);
    auto put = [&](int pair, uint64_t v) {
        const TwPair tp = tw_pair(v);
        blk[(size_t)pair * 2] = tp.w;
        blk[(size_t)pair * 2 + 1] = tp.wp;
    };
    for (int dir = 0; dir < 2; dir++) {
        const std::vector<uint64_t>& tab = dir ? t.inv : t.fwd;
        for (int g = 1; g < W::GROUPS; g++) {
            const int full = g < W::GROUPS - 1 || W::LAST == 3;
            const int m = full ? 1 << (3 * g) : 1 << (LOGN - 3);
            for (int U = 0; U < W::rows(g); U++) {
                const int row = dir * W::TABLE_ROWS + W::row_offset(g) + U;
                if (full || W::LAST == 3) put(7 * row, tab[m + U]);
                if (full || W::LAST >= 2)
                    for (int h = 0; h < 2; h++) put(7 * row + 1 + h, tab[2 * m + 2 * U + h]);
                for (int h = 0; h < 4; h++) put(7 * row + 3 + h, tab[4 * m + 4 * U + h]);
            }
        }
    }
    return blk;
}

size_t n1_of(const tfhe_amd_ctx* c) { return (size_t)c->p.n + 1; }
size_t lv1_of(const tfhe_amd_ctx* c) { return (size_t)c->p.N * c->p.K + 1; }
size_t trlwe_of(const tfhe_amd_ctx* c) { return (size_t)(c->p.K + 1) * c->p.N; }

int ensure_workspace(const tfhe_amd_ctx* ctx, Lane* lane, size_t n_gates) {
    if (n_gates <= lane->ws_gates) return TFHE_AMD_OK;
    // hipFree waits for the device, so no kernel still reads the old buffers
    if (lane->d_c) (void)hipFree(lane->d_c);
    if (lane->d_lv1) (void)hipFree(lane->d_lv1);
    lane->d_c = lane->d_lv1 = nullptr;
    lane->ws_gates = 0;
    HIP_TRY(hipMalloc(&lane->d_c, n_gates * n1_of(ctx) * sizeof(uint32_t)));
    HIP_TRY(hipMalloc(&lane->d_lv1, n_gates * lv1_of(ctx) * sizeof(uint32_t)));
    lane->ws_gates = n_gates;
    return TFHE_AMD_OK;
}

int ensure_io(Lane* lane, size_t words) {
    if (words <= lane->io_words) return TFHE_AMD_OK;
    if (lane->d_io) (void)hipFree(lane->d_io);
    lane->d_io = nullptr;
    lane->io_words = 0;
    HIP_TRY(hipMalloc(&lane->d_io, words * sizeof(uint32_t)));
    lane->io_words = words;
    return TFHE_AMD_OK;
}

void lane_free(Lane* lane) {
    if (!lane) return;
    if (lane->stream) (void)hipStreamSynchronize(lane->stream);
    if (lane->ws_done) (void)hipEventDestroy(lane->ws_done);
    void* bufs[] = {lane->d_c, lane->d_lv1, lane->d_io};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (lane->stream) (void)hipStreamDestroy(lane->stream);
    delete lane;
}

// Exclusive use of one lane of a context until the lease goes out of scope.  The context's device
// must be current.  `prefer`: the stream the call enqueues on; `own_stream`: the call runs on the
// lane's private stream (host-pointer calls).  Order of choice: a free lane whose workspace was last
// used on the call's stream (no cross-stream wait needed); a free lane whose last user has finished
// (or that was never used); a new lane while fewer than kMaxLanes exist, so that calls on different
// streams do not queue behind each other's workspace; any free lane; with all lanes busy it waits.
struct LaneLease {
    tfhe_amd_ctx* ctx;
    Lane* lane = nullptr;
    int rc = TFHE_AMD_OK;
    LaneLease(tfhe_amd_ctx* c, hipStream_t prefer, bool own_stream = false) : ctx(c) {
        std::unique_lock<std::mutex> lock(ctx->mu);
        for (;;) {
            Lane* free_lane = nullptr;
            Lane* idle_lane = nullptr;
            Lane* same_lane = nullptr;
            for (Lane* l : ctx->lanes) {
                if (l->busy) continue;
                if (!free_lane) free_lane = l;
                const hipStream_t want = own_stream ? l->stream : prefer;
                if (l->ws_used && l->ws_stream == want) { same_lane = l; break; }
                if (!idle_lane && (!l->ws_used || hipEventQuery(l->ws_done) == hipSuccess)) idle_lane = l;
            }
            (void)hipGetLastError();  // hipEventQuery reports "not ready" as an error
            if (same_lane) free_lane = same_lane;
            else if (idle_lane) free_lane = idle_lane;
            else if ((int)ctx->lanes.size() < kMaxLanes) free_lane = nullptr;
            if (free_lane) { lane = free_lane; break; }
            if ((int)ctx->lanes.size() < kMaxLanes) {
                Lane* l = new (std::nothrow) Lane();
                if (!l) { rc = fail(TFHE_AMD_ERR_MEMORY, "host allocation failed"); return; }
                hipError_t e = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&l->ws_done, hipEventDisableTiming);
                if (e != hipSuccess) {
                    lane_free(l);
                    rc = fail(hip_code(e), std::string("creating a workspace lane: ") + hipGetErrorString(e));
                    return;
                }
                ctx->lanes.push_back(l);
                lane = l;
                break;
            }
            ctx->cv.wait(lock);
        }
        lane->busy = true;
    }
    ~LaneLease() {
        if (!lane) return;
        {
            std::lock_guard<std::mutex> lock(ctx->mu);
            lane->busy = false;
        }
        ctx->cv.notify_one();
    }
    LaneLease(const LaneLease&) = delete;
    LaneLease& operator=(const LaneLease&) = delete;
};

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

hipStream_t pick_stream(const tfhe_amd_ctx* ctx, void* stream) {
    if (!stream) return ctx->stream;
    if (stream == TFHE_AMD_DEFAULT_STREAM) return nullptr;
    return (hipStream_t)stream;
}

int need_bk(const tfhe_amd_ctx* ctx) {
    return ctx->has_bk ? TFHE_AMD_OK : fail(TFHE_AMD_ERR_ARGUMENT, "context holds no bootstrapping key");
}
int need_ks(const tfhe_amd_ctx* ctx) {
    return ctx->has_ks ? TFHE_AMD_OK : fail(TFHE_AMD_ERR_ARGUMENT, "context holds no key-switching key");
}

int check_batch(const tfhe_amd_ctx* ctx, size_t n_gates, bool ptrs_ok) {
    if (!ctx) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    if (n_gates && !ptrs_ok) return fail(TFHE_AMD_ERR_ARGUMENT, "null array argument");
    if (n_gates > (size_t)1 << 30) return fail(TFHE_AMD_ERR_ARGUMENT, "batch too large");
    return TFHE_AMD_OK;
}

// blind rotation of device data: extract != 0 -> lv1 layout, else accumulators
int run_blind_rotate(tfhe_amd_ctx* ctx, const uint32_t* c, const uint32_t* tv, uint32_t* out, size_t n_gates,
                     int extract, hipStream_t s) {
    const tfhe_amd_params& p = ctx->p;
    if (ctx->replica)
        HIP_TRY(launch_blind_rotate_replica(ctx->d_spec, ctx->d_tw, c, tv, out, n_gates, p.n, p.N, p.K, p.L, p.bg_bit,
                                            extract, ctx->replica_generic ? 1 : 0, s));
    else if (ctx->fast) {
        // The wave-per-gate form keeps kResident gates in flight and a workgroup runs all n steps, so a
        // batch that is not a multiple of kResident pays a whole extra generation (about 7 ms at the
        // default set) for its remainder.  A remainder of up to two gates per CU goes through the
        // workgroup-per-gate form instead, behind the rest on the same stream (2.3 ms per 256 gates).
        // Identical results: the two forms compute the same words.  Measured alternatives: the remainder
        // ahead of the rest on a second stream costs more (the CUs it delays run their four generations
        // out of step with their neighbours and fetch every key row from the Infinity Cache on their own).
        constexpr size_t kResident = 2048, kTailMax = 512;
        const size_t tail = n_gates % kResident;
        if (ctx->form == 0 && n_gates > kResident && tail > 0 && tail <= kTailMax) {
            const size_t head = n_gates - tail;
            const size_t out_row = extract ? lv1_of(ctx) : trlwe_of(ctx);
            HIP_TRY(launch_blind_rotate_512(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c, tv, out, (int)head, p.n, p.K, p.L, extract, 2,
                                            s));
            HIP_TRY(launch_blind_rotate_512(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c + head * n1_of(ctx), tv, out + head * out_row,
                                            (int)tail, p.n, p.K, p.L, extract, 1, s));
            return TFHE_AMD_OK;
        }
        HIP_TRY(launch_blind_rotate_512(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c, tv, out, (int)n_gates, p.n, p.K, p.L,
                                        extract, ctx->form, s));
    }
    else if (ctx->wg)
        HIP_TRY(launch_blind_rotate_wg(ctx->u, ctx->d_tw, ctx->d_bk_ntt, c, tv, out, n_gates, p.n, p.N, p.K, p.L, p.bg_bit,
                                       ctx->shift, ctx->pack_shift, ctx->pack, extract, s));
    else {
        // rings beyond LDS keep their per-gate state in a workspace that lives for this launch
        // (stream-ordered allocation: no host synchronisation)
        const size_t ws_words = generic_big_workspace_words(n_gates, p.N, p.K, p.L);
        void* ws = nullptr;
        if (ws_words) HIP_TRY(hipMallocAsync(&ws, ws_words * sizeof(uint32_t), s));
        const hipError_t e = launch_blind_rotate_generic(ctx->d_bk_raw, c, tv, out, n_gates, p.n, p.N, p.K, p.L, p.bg_bit,
                                                         extract, static_cast<uint32_t*>(ws), s);
        if (ws) (void)hipFreeAsync(ws, s);
        HIP_TRY(e);
    }
    return TFHE_AMD_OK;
}

int run_key_switch(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out, size_t n_gates, hipStream_t s) {
    const tfhe_amd_params& p = ctx->p;
    HIP_TRY(launch_key_switch(lv1, ctx->d_ks, out, (int)n_gates, p.n + 1, p.N * p.K, p.T, p.base_bit, s));
    return TFHE_AMD_OK;
}

// bootstrap + key switch on device data, with optional event timing.  The last n_mux of the
// n_gates gates are MUXes: c holds n_gates + n_mux rows, and the lv1 result of row n_gates + j is
// added (plus 1/8) to that of row n_gates - n_mux + j before the key switch.
int run_bootstrap(tfhe_amd_ctx* ctx, Lane* lane, const uint32_t* c, uint32_t* out, size_t n_gates, hipStream_t s,
                  size_t n_mux = 0, uint32_t* lv1_ws = nullptr) {
    uint32_t* const d_lv1 = lv1_ws ? lv1_ws : lane->d_lv1;
    hipEvent_t* e = nullptr;
    std::unique_lock<std::mutex> tlock(ctx->timing_mu, std::defer_lock);
    // (the combination kernels that fill `c` ran on `s` already, after workspace_acquire)
    if (ctx->timing && !lv1_ws) {  // draining the event ring may block: not on the no-allocation path
        tlock.lock();  // one timed call at a time records into the ring
        if (ctx->ev_count == kTimingRing) {
            int rc = drain_timing(ctx, kTimingRing - 1);
            if (rc) return rc;
        }
        e = &ctx->ev[(size_t)((ctx->ev_head + ctx->ev_count) % kTimingRing) * 3];
        HIP_TRY(hipEventRecord(e[0], s));
    }
    int rc = run_blind_rotate(ctx, c, nullptr, d_lv1, n_gates + n_mux, 1, s);
    if (rc) return rc;
    if (n_mux)
        HIP_TRY(launch_mux_sum(d_lv1 + (n_gates - n_mux) * lv1_of(ctx), d_lv1 + n_gates * lv1_of(ctx), n_mux,
                               (int)lv1_of(ctx), s));
    if (e) HIP_TRY(hipEventRecord(e[1], s));
    rc = run_key_switch(ctx, d_lv1, out, n_gates, s);
    if (rc) return rc;
    if (e) {
        HIP_TRY(hipEventRecord(e[2], s));
        ctx->ev_count++;
    }
    if (lv1_ws) return TFHE_AMD_OK;  // the static workspace is not ordered through the context
    HIP_TRY(hipEventRecord(lane->ws_done, s));
    lane->ws_stream = s;
    lane->ws_used = true;
    return TFHE_AMD_OK;
}

// Orders a call on stream `s` behind the previous user of the lane's workspace when that one ran
// on another stream, and grows the workspace.
int workspace_acquire(const tfhe_amd_ctx* ctx, Lane* lane, size_t n_rows, hipStream_t s) {
    if (lane->ws_used && lane->ws_stream != s) HIP_TRY(hipStreamWaitEvent(s, lane->ws_done, 0));
    return ensure_workspace(ctx, lane, n_rows);
}

int check_gate_code(int op) {
    return (op >= 0 && op < TFHE_AMD_GATE_COUNT) ? TFHE_AMD_OK : fail(TFHE_AMD_ERR_ARGUMENT, "unknown gate code");
}

// Host-pointer wrapper: leases a lane, copies `n_in` input arrays to its staging buffer, runs `body`
// on the device copies on the lane's private stream, copies one output array back and waits.
// in_words[k] / out_words are per-array sizes.
template <typename Body>
int with_staging(tfhe_amd_ctx* ctx, const uint32_t* const* in, const size_t* in_words, int n_in, uint32_t* out,
                 size_t out_words, Body body) {
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    LaneLease lease(ctx, nullptr, true);
    if (lease.rc) return lease.rc;
    Lane* lane = lease.lane;
    size_t total = out_words;
    for (int k = 0; k < n_in; k++) total += in_words[k];
    int rc = ensure_io(lane, total);
    if (rc) return rc;
    const uint32_t* dev_in[4] = {nullptr, nullptr, nullptr, nullptr};
    // On failure nothing may stay in flight on the lane's stream: the copies read and write the
    // caller's host arrays, and the lane goes back to the pool.
    struct Drain {
        hipStream_t s;
        bool armed = true;
        ~Drain() { if (armed) (void)hipStreamSynchronize(s); }
    } drain{lane->stream};
    uint32_t* cur = lane->d_io;
    for (int k = 0; k < n_in; k++) {
        if (in[k]) {
            HIP_TRY(hipMemcpyAsync(cur, in[k], in_words[k] * sizeof(uint32_t), hipMemcpyHostToDevice, lane->stream));
            dev_in[k] = cur;
        }
        cur += in_words[k];
    }
    rc = body(lane, dev_in, cur);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(out, cur, out_words * sizeof(uint32_t), hipMemcpyDeviceToHost, lane->stream));
    HIP_TRY(hipStreamSynchronize(lane->stream));
    drain.armed = false;
    return TFHE_AMD_OK;
}

// ---- evaluation on device data with a leased lane; `s` is the stream everything is enqueued on ----

int bootstrap_on(tfhe_amd_ctx* ctx, Lane* lane, const uint32_t* c, uint32_t* out, size_t n_gates, hipStream_t s) {
    int rc = workspace_acquire(ctx, lane, n_gates, s);
    if (rc) return rc;
    return run_bootstrap(ctx, lane, c, out, n_gates, s);
}

// op < 0: NAND through its own kernel; otherwise gate code `op`, or per-gate codes `ops` when not null
int gate_on(tfhe_amd_ctx* ctx, Lane* lane, const uint8_t* ops, int op, const uint32_t* a, const uint32_t* b,
            uint32_t* out, size_t n_gates, hipStream_t s) {
    int rc = workspace_acquire(ctx, lane, n_gates, s);
    if (rc) return rc;
    if (op < 0) HIP_TRY(launch_nand_combine(lane->d_c, a, b, n_gates, ctx->p.n + 1, s));
    else        HIP_TRY(launch_gate_combine(lane->d_c, a, b, ops, op, n_gates, ctx->p.n + 1, s));
    return run_bootstrap(ctx, lane, lane->d_c, out, n_gates, s);
}

int level_on(tfhe_amd_ctx* ctx, Lane* lane, const uint8_t* ops, const uint32_t* a, const uint32_t* b,
             const uint32_t* x0, uint32_t* out, size_t n_gates, size_t n_mux, hipStream_t s) {
    const size_t n_two = n_gates - n_mux;
    int rc = workspace_acquire(ctx, lane, n_gates + n_mux, s);
    if (rc) return rc;
    const size_t n1 = n1_of(ctx);
    uint32_t* d_c = lane->d_c;
    HIP_TRY(launch_gate_combine(d_c, a, b, ops, 0, n_two, (int)n1, s));
    HIP_TRY(launch_gate_combine(d_c + n_two * n1, a + n_two * n1, b + n_two * n1, nullptr, TFHE_AMD_GATE_AND, n_mux,
                                (int)n1, s));
    HIP_TRY(launch_gate_combine(d_c + n_gates * n1, a + n_two * n1, x0, nullptr, TFHE_AMD_GATE_ANDNY, n_mux, (int)n1,
                                s));
    return run_bootstrap(ctx, lane, d_c, out, n_gates, s, n_mux);
}

// device-pointer entry points that need a workspace: lease a lane while enqueueing on the caller's stream
template <typename Body>
int with_lane(tfhe_amd_ctx* ctx, void* stream, Body body) {
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = pick_stream(ctx, stream);
    LaneLease lease(ctx, s);
    if (lease.rc) return lease.rc;
    return body(lease.lane, s);
}

// Per-device state of the stand-alone external product / cmux entry points, kept for the life of
// the process and grown on demand, so that repeated calls neither allocate nor rebuild tables: the
// staging buffers, the transformed TRGSW samples, the twiddle tables of every kernel family used.
struct EpScratch {
    std::mutex mu;  // one call at a time per device
    hipStream_t stream = nullptr;
    uint32_t* raw = nullptr;  // trgsw | x0 | x1 | out
    size_t raw_words = 0;
    d2* ntt = nullptr;        // transformed samples
    size_t ntt_doubles = 0;
    d2* tw[4] = {nullptr, nullptr, nullptr, nullptr};  // [0] wave kernels (N = 512), [1..3] workgroup kernels, log2 N = 9..11
    NttUniform u[4];
};
EpScratch g_ep[kMaxDevices];

int ep_twiddles(EpScratch& sc, int family, int N) {
    if (sc.tw[family]) return TFHE_AMD_OK;
    const TwiddleSet tw = make_twiddles(N);
    build_uniform(sc.u[family], tw);
    const std::vector<double> blk = family == 0 ? build_tw_block_512(tw)
                                    : family == 1 ? build_tw_table_wg<9>(tw)
                                    : family == 2 ? build_tw_table_wg<10>(tw)
                                                  : build_tw_table_wg<11>(tw);
    HIP_TRY(hipMalloc(&sc.tw[family], blk.size() * sizeof(double)));
    HIP_TRY(hipMemcpy(sc.tw[family], blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice));
    return TFHE_AMD_OK;
}

int external_product_common(const tfhe_amd_params* p, const uint32_t* trgsw, const uint32_t* x0, const uint32_t* x1,
                            uint32_t* out, size_t batch, int device, int mode) {
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    const bool fast = external_product_512_has_kernel(p->N, p->K, p->L, p->bg_bit);
    int tiles = 0, shift = 0, pack_shift = 0, pack = 0;
    const bool wg = !fast && wg_has_kernel(p->N, p->K, p->L, p->bg_bit) &&
                    wg_split(p->N, p->K, p->L, p->bg_bit, &tiles, &shift, &pack_shift, &pack);
    if (!fast && !wg && !generic_has_kernel(p->N, p->K, p->L))
        return fail(TFHE_AMD_ERR_UNSUPPORTED, "ring degree beyond 2^26");
    if (batch && (!trgsw || !x0 || !out || (mode && !x1))) return fail(TFHE_AMD_ERR_ARGUMENT, "null array argument");
    if (batch == 0) return TFHE_AMD_OK;
    if (batch > (size_t)1 << 24) return fail(TFHE_AMD_ERR_ARGUMENT, "batch too large");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev || device >= kMaxDevices) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    EpScratch& sc = g_ep[device];
    std::lock_guard<std::mutex> lock(sc.mu);
    if (!sc.stream) HIP_TRY(hipStreamCreateWithFlags(&sc.stream, hipStreamNonBlocking));
    const size_t K1 = (size_t)p->K + 1, R = K1 * p->L;
    const size_t W = K1 * p->N, GW = W * R, need = batch * (GW + 3 * W);
    if (need > sc.raw_words) {
        if (sc.raw) (void)hipFree(sc.raw);
        sc.raw = nullptr;
        sc.raw_words = 0;
        HIP_TRY(hipMalloc(&sc.raw, need * sizeof(uint32_t)));
        sc.raw_words = need;
    }
    uint32_t* d_g = sc.raw;
    uint32_t* d_0 = d_g + batch * GW;
    uint32_t* d_1 = d_0 + batch * W;
    uint32_t* d_out = d_1 + batch * W;
    HIP_TRY(hipMemcpyAsync(d_g, trgsw, batch * GW * 4, hipMemcpyHostToDevice, sc.stream));
    HIP_TRY(hipMemcpyAsync(d_0, x0, batch * W * 4, hipMemcpyHostToDevice, sc.stream));
    if (mode) HIP_TRY(hipMemcpyAsync(d_1, x1, batch * W * 4, hipMemcpyHostToDevice, sc.stream));
    if (fast || wg) {
        // the samples go through the key transform of the blind rotation (a "bootstrapping key" of
        // `batch` rows), then one step body per element
        const size_t ntt_doubles = fast ? batch * GW : batch * R * (size_t)tiles * p->N;
        if (ntt_doubles > sc.ntt_doubles) {
            if (sc.ntt) (void)hipFree(sc.ntt);
            sc.ntt = nullptr;
            sc.ntt_doubles = 0;
            HIP_TRY(hipMalloc(&sc.ntt, ntt_doubles * sizeof(double)));
            sc.ntt_doubles = ntt_doubles;
        }
        const int family = fast ? 0 : p->N == 512 ? 1 : p->N == 1024 ? 2 : 3;
        int rc = ep_twiddles(sc, family, p->N);
        if (rc) return rc;
        if (fast) {
            HIP_TRY(launch_bk_transform_512(sc.u[0], sc.tw[0], d_g, sc.ntt, (int)(batch * R * K1), sc.stream));
            HIP_TRY(launch_external_product_512(sc.u[0], sc.tw[0], sc.ntt, d_0, d_1, d_out, batch, p->K, p->L, mode,
                                                sc.stream));
        } else {
            HIP_TRY(launch_bk_transform_wg(sc.u[family], sc.tw[family], d_g, sc.ntt, batch * R, p->N, p->K, shift,
                                           pack_shift, pack, sc.stream));
            HIP_TRY(launch_external_product_wg(sc.u[family], sc.tw[family], sc.ntt, d_0, d_1, d_out, batch, p->N, p->K,
                                               p->L, p->bg_bit, shift, pack_shift, pack, mode, sc.stream));
        }
    } else {
        const size_t ws_words = generic_big_workspace_words(batch, p->N, p->K, p->L);
        void* ws = nullptr;
        if (ws_words) HIP_TRY(hipMallocAsync(&ws, ws_words * sizeof(uint32_t), sc.stream));
        const hipError_t e = launch_external_product_generic(d_g, d_0, d_1, d_out, batch, p->N, p->K, p->L, p->bg_bit, mode,
                                                             static_cast<uint32_t*>(ws), sc.stream);
        if (ws) (void)hipFreeAsync(ws, sc.stream);
        HIP_TRY(e);
    }
    HIP_TRY(hipMemcpyAsync(out, d_out, batch * W * 4, hipMemcpyDeviceToHost, sc.stream));
    HIP_TRY(hipStreamSynchronize(sc.stream));
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
    if (!fast_supported(p) && !generic_has_kernel(p->N, p->K, p->L))
        return fail(TFHE_AMD_ERR_UNSUPPORTED, "no blind-rotation kernel for this ring/gadget");
    if (!key_switch_has_kernel(p->n + 1, p->T, p->base_bit))
        return fail(TFHE_AMD_ERR_UNSUPPORTED, "no key-switch kernel for this base/size");
    return TFHE_AMD_OK;
}

size_t tfhe_amd_bk_words(const tfhe_amd_params* p) {
    return (size_t)p->n * (p->K + 1) * p->L * (p->K + 1) * p->N;
}
size_t tfhe_amd_ks_words(const tfhe_amd_params* p) {
    return (size_t)p->N * p->K * p->T * ((size_t)1 << p->base_bit) * (p->n + 1);
}

int tfhe_amd_ctx_create(const tfhe_amd_params* p, const uint32_t* bk, const uint32_t* ks, int device, unsigned flags,
                        tfhe_amd_ctx** out) {
    if (!out) return fail(TFHE_AMD_ERR_ARGUMENT, "out is null");
    *out = nullptr;
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    if (!bk && !ks) return fail(TFHE_AMD_ERR_ARGUMENT, "both key pointers are null");
    int rc = tfhe_amd_params_supported(p);
    if (rc) return rc;
    if ((flags & TFHE_AMD_CTX_REFERENCE_FFT) && !replica_has_kernel(p->N, p->K, p->L))
        return fail(TFHE_AMD_ERR_UNSUPPORTED, "no reference-FFT kernel for this ring (the reference multiplies "
                                              "exactly below degree 32; above 1024 the spectra exceed LDS)");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    const bool on_device = (flags & TFHE_AMD_CTX_KEYS_ON_DEVICE) != 0;

    tfhe_amd_ctx* ctx = new (std::nothrow) tfhe_amd_ctx();
    if (!ctx) return fail(TFHE_AMD_ERR_MEMORY, "host allocation failed");
    ctx->p = *p;
    ctx->device = device;
    ctx->replica = (flags & TFHE_AMD_CTX_REFERENCE_FFT) != 0;
    ctx->replica_generic = ctx->replica && (flags & TFHE_AMD_CTX_GENERIC_KERNELS);
    const bool special = !ctx->replica && !(flags & TFHE_AMD_CTX_GENERIC_KERNELS);
    ctx->fast = special && fast_supported(p);
    ctx->wg = special && !ctx->fast && wg_has_kernel(p->N, p->K, p->L, p->bg_bit) &&
              wg_split(p->N, p->K, p->L, p->bg_bit, &ctx->tiles, &ctx->shift, &ctx->pack_shift, &ctx->pack);
    ctx->form = (flags & TFHE_AMD_CTX_LATENCY_FORM) ? 1 : (flags & TFHE_AMD_CTX_THROUGHPUT_FORM) ? 2 : 0;
    auto bail = [&](hipError_t e, const char* what) {
        tfhe_amd_ctx_destroy(ctx);
        return fail(hip_code(e), std::string(what) + ": " + hipGetErrorString(e));
    };
#define CTX_TRY(expr)                                     \
    do {                                                  \
        hipError_t e_ = (expr);                           \
        if (e_ != hipSuccess) return bail(e_, #expr);     \
    } while (0)

    CTX_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    CTX_TRY(scratch_of(device, &ctx->scratch));
    const size_t bk_words = tfhe_amd_bk_words(p), ks_words = tfhe_amd_ks_words(p);
    const int n1 = p->n + 1, nlv1 = p->N * p->K;
    DevBuf up_bk, up_ks;  // upload scratch for host-resident keys
    if (bk) {
        const uint32_t* src = bk;
        if (ctx->replica) {
            const std::vector<double> tab = build_replica_tables(p->N);
            CTX_TRY(hipMalloc(&ctx->d_tw, tab.size() * sizeof(double)));
            CTX_TRY(hipMemcpy(ctx->d_tw, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
            CTX_TRY(hipMalloc(&ctx->d_spec, bk_words * sizeof(double)));  // N/2 complex doubles per polynomial
            if (!on_device) {
                CTX_TRY(up_bk.alloc(bk_words * sizeof(uint32_t)));
                CTX_TRY(hipMemcpyAsync(up_bk.p, bk, bk_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                src = up_bk.as<uint32_t>();
            }
            // the wave-per-gate kernel reads the spectra in the order its first transform leaves them
            const int wave_order = !ctx->replica_generic && replica_has_fast_kernel(p->N, p->K, p->L, p->bg_bit);
            CTX_TRY(launch_replica_key(src, ctx->d_tw, ctx->d_spec, bk_words / p->N, p->N, wave_order, ctx->stream));
        } else if (ctx->fast) {
            const TwiddleSet tw = make_twiddles(p->N);
            build_uniform(ctx->u, tw);
            const std::vector<double> blk = build_tw_block_512(tw);
            CTX_TRY(hipMalloc(&ctx->d_tw, blk.size() * sizeof(double)));
            CTX_TRY(hipMemcpy(ctx->d_tw, blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice));
            CTX_TRY(hipMalloc(&ctx->d_bk_ntt, bk_words * sizeof(double)));
            if (!on_device) {
                CTX_TRY(up_bk.alloc(bk_words * sizeof(uint32_t)));
                CTX_TRY(hipMemcpyAsync(up_bk.p, bk, bk_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                src = up_bk.as<uint32_t>();
            }
            CTX_TRY(launch_bk_transform_512(ctx->u, ctx->d_tw, src, ctx->d_bk_ntt, (int)(bk_words / p->N), ctx->stream));
        } else if (ctx->wg) {
            const TwiddleSet tw = make_twiddles(p->N);
            build_uniform(ctx->u, tw);
            const std::vector<double> blk = p->N == 2048   ? build_tw_table_wg<11>(tw)
                                            : p->N == 1024 ? build_tw_table_wg<10>(tw)
                                                           : build_tw_table_wg<9>(tw);
            CTX_TRY(hipMalloc(&ctx->d_tw, blk.size() * sizeof(double)));
            CTX_TRY(hipMemcpy(ctx->d_tw, blk.data(), blk.size() * sizeof(double), hipMemcpyHostToDevice));
            // [n][R][tiles] polynomials of N doubles
            CTX_TRY(hipMalloc(&ctx->d_bk_ntt, bk_words / (p->K + 1) * ctx->tiles * sizeof(double)));
            if (!on_device) {
                CTX_TRY(up_bk.alloc(bk_words * sizeof(uint32_t)));
                CTX_TRY(hipMemcpyAsync(up_bk.p, bk, bk_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
                src = up_bk.as<uint32_t>();
            }
            CTX_TRY(launch_bk_transform_wg(ctx->u, ctx->d_tw, src, ctx->d_bk_ntt, bk_words / p->N / (p->K + 1), p->N, p->K,
                                           ctx->shift, ctx->pack_shift, ctx->pack, ctx->stream));
        } else {
            CTX_TRY(hipMalloc(&ctx->d_bk_raw, bk_words * sizeof(uint32_t)));
            CTX_TRY(hipMemcpyAsync(ctx->d_bk_raw, bk, bk_words * sizeof(uint32_t),
                                   on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
        }
        ctx->has_bk = true;
    }
    if (ks) {
        const uint32_t* src = ks;
        CTX_TRY(hipMalloc(&ctx->d_ks, key_switch_packed_words(n1, nlv1, p->T, p->base_bit) * sizeof(uint32_t)));
        if (!on_device) {
            CTX_TRY(up_ks.alloc(ks_words * sizeof(uint32_t)));
            CTX_TRY(hipMemcpyAsync(up_ks.p, ks, ks_words * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->stream));
            src = up_ks.as<uint32_t>();
        }
        CTX_TRY(launch_ks_pack(ctx->d_ks, src, n1, nlv1, p->T, p->base_bit, ctx->stream));
        ctx->has_ks = true;
    }
    CTX_TRY(hipStreamSynchronize(ctx->stream));
#undef CTX_TRY
    *out = ctx;
    return TFHE_AMD_OK;
}

int tfhe_amd_keygen_device(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring, double alpha_lv0,
                           double alpha_lv1, const uint32_t* seed, int device, uint32_t* bk, uint32_t* ks, void* stream) {
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    if (!s_lv0 || !s_ring) return fail(TFHE_AMD_ERR_ARGUMENT, "null secret key");
    if (!seed) return fail(TFHE_AMD_ERR_ARGUMENT, "null seed");
    KeygenSeed key;
    std::memcpy(key.w, seed, sizeof(key.w));
    if (!bk && !ks) return fail(TFHE_AMD_ERR_ARGUMENT, "both key pointers are null");
    if (!(alpha_lv0 >= 0.0 && alpha_lv0 < 0.5 && alpha_lv1 >= 0.0 && alpha_lv1 < 0.5))
        return fail(TFHE_AMD_ERR_ARGUMENT, "noise parameters must lie in [0, 0.5)");
    if (p->N > 4096) return fail(TFHE_AMD_ERR_UNSUPPORTED, "ring too large for the key generation kernel");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    hipStream_t s = stream == TFHE_AMD_DEFAULT_STREAM ? nullptr : (hipStream_t)stream;
    const size_t ring = (size_t)p->K * p->N;
    DevBuf d_s0, d_sr;
    HIP_TRY(d_s0.alloc((size_t)p->n));
    HIP_TRY(d_sr.alloc(ring));
    HIP_TRY(hipMemcpyAsync(d_s0.p, s_lv0, (size_t)p->n, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(d_sr.p, s_ring, ring, hipMemcpyHostToDevice, s));
    if (bk)
        HIP_TRY(launch_keygen_bk(bk, d_s0.as<uint8_t>(), d_sr.as<uint8_t>(), p->n, p->N, p->K, p->L, p->bg_bit, alpha_lv1,
                                 key, s));
    if (ks)
        HIP_TRY(launch_keygen_ks(ks, d_s0.as<uint8_t>(), d_sr.as<uint8_t>(), p->n, p->N * p->K, p->T, p->base_bit,
                                 alpha_lv0, key, s));
    HIP_TRY(hipStreamSynchronize(s));  // the secret-key copies are freed on return
    return TFHE_AMD_OK;
}

int tfhe_amd_keygen(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring, double alpha_lv0,
                    double alpha_lv1, const uint32_t* seed, int device, uint32_t* bk, uint32_t* ks) {
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    if (!bk && !ks) return fail(TFHE_AMD_ERR_ARGUMENT, "both key pointers are null");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    const size_t bk_words = tfhe_amd_bk_words(p), ks_words = tfhe_amd_ks_words(p);
    DevBuf d_bk, d_ks;
    if (bk) HIP_TRY(d_bk.alloc(bk_words * sizeof(uint32_t)));
    if (ks) HIP_TRY(d_ks.alloc(ks_words * sizeof(uint32_t)));
    int rc = tfhe_amd_keygen_device(p, s_lv0, s_ring, alpha_lv0, alpha_lv1, seed, device, bk ? d_bk.as<uint32_t>() : nullptr,
                                    ks ? d_ks.as<uint32_t>() : nullptr, TFHE_AMD_DEFAULT_STREAM);
    if (rc) return rc;
    if (bk) HIP_TRY(hipMemcpy(bk, d_bk.p, bk_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    if (ks) HIP_TRY(hipMemcpy(ks, d_ks.p, ks_words * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TFHE_AMD_OK;
}

int tfhe_amd_ctx_create_from_secret(const tfhe_amd_params* p, const uint8_t* s_lv0, const uint8_t* s_ring,
                                    double alpha_lv0, double alpha_lv1, const uint32_t* seed, int device, unsigned flags,
                                    tfhe_amd_ctx** out) {
    if (!out) return fail(TFHE_AMD_ERR_ARGUMENT, "out is null");
    *out = nullptr;
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    int rc = tfhe_amd_params_supported(p);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
    DevBuf d_bk, d_ks;
    HIP_TRY(d_bk.alloc(tfhe_amd_bk_words(p) * sizeof(uint32_t)));
    HIP_TRY(d_ks.alloc(tfhe_amd_ks_words(p) * sizeof(uint32_t)));
    rc = tfhe_amd_keygen_device(p, s_lv0, s_ring, alpha_lv0, alpha_lv1, seed, device, d_bk.as<uint32_t>(),
                                d_ks.as<uint32_t>(), TFHE_AMD_DEFAULT_STREAM);
    if (rc) return rc;
    return tfhe_amd_ctx_create(p, d_bk.as<uint32_t>(), d_ks.as<uint32_t>(), device, flags | TFHE_AMD_CTX_KEYS_ON_DEVICE,
                               out);
}

void tfhe_amd_ctx_destroy(tfhe_amd_ctx* ctx) {
    if (!ctx) return;
    DeviceGuard guard(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (Lane* l : ctx->lanes) lane_free(l);
    for (hipEvent_t e : ctx->ev) if (e) (void)hipEventDestroy(e);
    void* bufs[] = {ctx->d_tw, ctx->d_bk_ntt, ctx->d_bk_raw, ctx->d_spec, ctx->d_ks};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

// ---- device-pointer entry points -----------------------------------------------------------

int tfhe_amd_bootstrap_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates,
                                    void* stream) {
    int rc = check_batch(ctx, n_gates, c && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    return with_lane(ctx, stream, [&](Lane* lane, hipStream_t s) { return bootstrap_on(ctx, lane, c, out, n_gates, s); });
}

int tfhe_amd_nand_batch_device(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
                               size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, a && b && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    return with_lane(ctx, stream,
                     [&](Lane* lane, hipStream_t s) { return gate_on(ctx, lane, nullptr, -1, a, b, out, n_gates, s); });
}

int tfhe_amd_nand_batch_device_noalloc(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out,
                                       size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, a && b && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    const size_t n1 = n1_of(ctx), per_gate = n1 + lv1_of(ctx);
    if (n_gates > kNoAllocWords / per_gate)
        return fail(TFHE_AMD_ERR_ARGUMENT, "batch of " + std::to_string(n_gates) + " gates exceeds the static workspace (" +
                                               std::to_string(kNoAllocWords / per_gate) +
                                               " gates at this parameter set); use tfhe_amd_nand_batch_device");
    if (!ctx->scratch || !ctx->scratch->ptr) return fail(TFHE_AMD_ERR_HIP, "static workspace unresolved on this device");
    // rings whose per-gate state does not fit LDS run on kernels that take a workspace per launch
    // (run_blind_rotate allocates it stream-ordered): not through this entry point
    if (!ctx->replica && !ctx->fast && !ctx->wg &&
        generic_big_workspace_words(n_gates, ctx->p.N, ctx->p.K, ctx->p.L) != 0)
        return fail(TFHE_AMD_ERR_UNSUPPORTED, "this parameter set's blind rotation needs a per-launch workspace; "
                                              "use tfhe_amd_nand_batch_device");
    DeviceGuard guard(ctx->device);
    hipStream_t s = pick_stream(ctx, stream);
    ScratchState& st = *ctx->scratch;
    std::lock_guard<std::mutex> lock(st.mu);
    // A stream under capture records a graph: an event from outside the capture may not enter it, and
    // the graph's launches are ordered by the streams the caller launches them on.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
    }
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (!capturing && st.used && st.last != s) HIP_TRY(hipStreamWaitEvent(s, st.done, 0));
    uint32_t* const d_c = st.ptr;
    uint32_t* const d_lv1 = st.ptr + n_gates * n1;
    HIP_TRY(launch_nand_combine(d_c, a, b, n_gates, (int)n1, s));
    rc = run_bootstrap(ctx, nullptr, d_c, out, n_gates, s, 0, d_lv1);
    if (rc) return rc;
    if (!capturing) {
        HIP_TRY(hipEventRecord(st.done, s));
        st.last = s;
        st.used = true;
    }
    return TFHE_AMD_OK;
}

int tfhe_amd_gate_batch_device(tfhe_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                               size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, a && b && out);
    if (rc || (rc = check_gate_code(op)) || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    return with_lane(ctx, stream,
                     [&](Lane* lane, hipStream_t s) { return gate_on(ctx, lane, nullptr, op, a, b, out, n_gates, s); });
}

int tfhe_amd_gates_batch_device(tfhe_amd_ctx* ctx, const uint8_t* ops, const uint32_t* a, const uint32_t* b,
                                uint32_t* out, size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, ops && a && b && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    return with_lane(ctx, stream,
                     [&](Lane* lane, hipStream_t s) { return gate_on(ctx, lane, ops, 0, a, b, out, n_gates, s); });
}

int tfhe_amd_not_batch_device(tfhe_amd_ctx* ctx, const uint32_t* a, uint32_t* out, size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, a && out);
    if (rc || n_gates == 0) return rc;
    DeviceGuard guard(ctx->device);
    HIP_TRY(launch_negate(out, a, n_gates * n1_of(ctx), pick_stream(ctx, stream)));
    return TFHE_AMD_OK;
}

int tfhe_amd_level_batch_device(tfhe_amd_ctx* ctx, const uint8_t* ops, const uint32_t* a, const uint32_t* b,
                                const uint32_t* x0, uint32_t* out, size_t n_gates, size_t n_mux, void* stream) {
    if (n_mux > n_gates) return fail(TFHE_AMD_ERR_ARGUMENT, "more MUX gates than gates");
    const size_t n_two = n_gates - n_mux;
    int rc = check_batch(ctx, n_gates + n_mux, a && b && out && (n_two == 0 || ops) && (n_mux == 0 || x0));
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    return with_lane(ctx, stream, [&](Lane* lane, hipStream_t s) {
        return level_on(ctx, lane, ops, a, b, x0, out, n_gates, n_mux, s);
    });
}

int tfhe_amd_mux_batch_device(tfhe_amd_ctx* ctx, const uint32_t* sel, const uint32_t* x1, const uint32_t* x0,
                              uint32_t* out, size_t n_gates, void* stream) {
    return tfhe_amd_level_batch_device(ctx, nullptr, sel, x1, x0, out, n_gates, n_gates, stream);
}

int tfhe_amd_gate_bootstrapping_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* lv1, size_t n_gates,
                                             void* stream) {
    int rc = check_batch(ctx, n_gates, c && lv1);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx))) return rc;
    DeviceGuard guard(ctx->device);
    return run_blind_rotate(ctx, c, nullptr, lv1, n_gates, 1, pick_stream(ctx, stream));
}

int tfhe_amd_blind_rotate_batch_device(tfhe_amd_ctx* ctx, const uint32_t* c, const uint32_t* test_vector,
                                       uint32_t* acc, size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, c && acc);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx))) return rc;
    DeviceGuard guard(ctx->device);
    return run_blind_rotate(ctx, c, test_vector, acc, n_gates, 0, pick_stream(ctx, stream));
}

int tfhe_amd_identity_key_switch_batch_device(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out,
                                              size_t n_gates, void* stream) {
    int rc = check_batch(ctx, n_gates, lv1 && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_ks(ctx))) return rc;
    DeviceGuard guard(ctx->device);
    return run_key_switch(ctx, lv1, out, n_gates, pick_stream(ctx, stream));
}

// ---- host-pointer entry points (synchronous; each call runs on a lane's private stream) -----------

int tfhe_amd_nand_batch(tfhe_amd_ctx* ctx, const uint32_t* a, const uint32_t* b, uint32_t* out, size_t n_gates) {
    int rc = check_batch(ctx, n_gates, a && b && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    const size_t w = n_gates * n1_of(ctx);
    const uint32_t* in[2] = {a, b};
    const size_t iw[2] = {w, w};
    return with_staging(ctx, in, iw, 2, out, w, [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return gate_on(ctx, lane, nullptr, -1, d[0], d[1], o, n_gates, lane->stream);
    });
}

int tfhe_amd_gate_batch(tfhe_amd_ctx* ctx, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                        size_t n_gates) {
    int rc = check_batch(ctx, n_gates, a && b && out);
    if (rc || (rc = check_gate_code(op)) || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    const size_t w = n_gates * n1_of(ctx);
    const uint32_t* in[2] = {a, b};
    const size_t iw[2] = {w, w};
    return with_staging(ctx, in, iw, 2, out, w, [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return gate_on(ctx, lane, nullptr, op, d[0], d[1], o, n_gates, lane->stream);
    });
}

int tfhe_amd_not_batch(tfhe_amd_ctx* ctx, const uint32_t* a, uint32_t* out, size_t n_gates) {
    int rc = check_batch(ctx, n_gates, a && out);
    if (rc || n_gates == 0) return rc;
    const size_t w = n_gates * n1_of(ctx);
    const uint32_t* in[1] = {a};
    const size_t iw[1] = {w};
    return with_staging(ctx, in, iw, 1, out, w, [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        HIP_TRY(launch_negate(o, d[0], w, lane->stream));
        return (int)TFHE_AMD_OK;
    });
}

int tfhe_amd_mux_batch(tfhe_amd_ctx* ctx, const uint32_t* sel, const uint32_t* x1, const uint32_t* x0, uint32_t* out,
                       size_t n_gates) {
    int rc = check_batch(ctx, 2 * n_gates, sel && x1 && x0 && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    const size_t w = n_gates * n1_of(ctx);
    const uint32_t* in[3] = {sel, x1, x0};
    const size_t iw[3] = {w, w, w};
    return with_staging(ctx, in, iw, 3, out, w, [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return level_on(ctx, lane, nullptr, d[0], d[1], d[2], o, n_gates, n_gates, lane->stream);
    });
}

int tfhe_amd_bootstrap_batch(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* out, size_t n_gates) {
    int rc = check_batch(ctx, n_gates, c && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx)) || (rc = need_ks(ctx))) return rc;
    const size_t w = n_gates * n1_of(ctx);
    const uint32_t* in[1] = {c};
    const size_t iw[1] = {w};
    return with_staging(ctx, in, iw, 1, out, w, [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return bootstrap_on(ctx, lane, d[0], o, n_gates, lane->stream);
    });
}

int tfhe_amd_gate_bootstrapping_batch(tfhe_amd_ctx* ctx, const uint32_t* c, uint32_t* lv1, size_t n_gates) {
    int rc = check_batch(ctx, n_gates, c && lv1);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx))) return rc;
    const uint32_t* in[1] = {c};
    const size_t iw[1] = {n_gates * n1_of(ctx)};
    return with_staging(ctx, in, iw, 1, lv1, n_gates * lv1_of(ctx), [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return run_blind_rotate(ctx, d[0], nullptr, o, n_gates, 1, lane->stream);
    });
}

int tfhe_amd_blind_rotate_batch(tfhe_amd_ctx* ctx, const uint32_t* c, const uint32_t* test_vector, uint32_t* acc,
                                size_t n_gates) {
    int rc = check_batch(ctx, n_gates, c && acc);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_bk(ctx))) return rc;
    const uint32_t* in[2] = {c, test_vector};
    const size_t iw[2] = {n_gates * n1_of(ctx), trlwe_of(ctx)};
    return with_staging(ctx, in, iw, 2, acc, n_gates * trlwe_of(ctx), [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return run_blind_rotate(ctx, d[0], d[1], o, n_gates, 0, lane->stream);
    });
}

int tfhe_amd_identity_key_switch_batch(tfhe_amd_ctx* ctx, const uint32_t* lv1, uint32_t* out, size_t n_gates) {
    int rc = check_batch(ctx, n_gates, lv1 && out);
    if (rc || n_gates == 0) return rc;
    if ((rc = need_ks(ctx))) return rc;
    const uint32_t* in[1] = {lv1};
    const size_t iw[1] = {n_gates * lv1_of(ctx)};
    return with_staging(ctx, in, iw, 1, out, n_gates * n1_of(ctx), [&](Lane* lane, const uint32_t* const* d, uint32_t* o) {
        return run_key_switch(ctx, d[0], o, n_gates, lane->stream);
    });
}

int tfhe_amd_external_product_batch(const tfhe_amd_params* p, const uint32_t* trgsw, const uint32_t* trlwe,
                                    uint32_t* out, size_t batch, int device) {
    return external_product_common(p, trgsw, trlwe, nullptr, out, batch, device, 0);
}

int tfhe_amd_cmux_batch(const tfhe_amd_params* p, const uint32_t* trgsw, const uint32_t* d0, const uint32_t* d1,
                        uint32_t* out, size_t batch, int device) {
    return external_product_common(p, trgsw, d0, d1, out, batch, device, 1);
}

// ---- housekeeping ----------------------------------------------------------------------------

const char* tfhe_amd_ctx_kernel_family(const tfhe_amd_ctx* ctx) {
    if (!ctx || !ctx->has_bk) return "none";
    if (ctx->replica)
        return (!ctx->replica_generic && replica_has_fast_kernel(ctx->p.N, ctx->p.K, ctx->p.L, ctx->p.bg_bit))
                   ? "reference-fft-wave"
                   : "reference-fft-workgroup";
    if (ctx->fast) return "ntt512";
    if (ctx->wg) return "ntt-workgroup";
    return "generic";
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
    std::lock_guard<std::mutex> tlock(ctx->timing_mu);
    if (enable && ctx->ev.empty()) {
        ctx->ev.assign((size_t)kTimingRing * 3, nullptr);
        for (hipEvent_t& e : ctx->ev) HIP_TRY(hipEventCreate(&e));
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
    std::lock_guard<std::mutex> tlock(ctx->timing_mu);
    int rc = drain_timing(ctx, 0);
    if (rc) return rc;
    if (blind_rotate_ms) *blind_rotate_ms = ctx->br_ms;
    if (key_switch_ms) *key_switch_ms = ctx->ks_ms;
    if (calls) *calls = ctx->calls;
    ctx->br_ms = ctx->ks_ms = 0.0;
    ctx->calls = 0;
    return TFHE_AMD_OK;
}

// ---- several GPUs behind one handle ------------------------------------------------------------

}  // extern "C"

struct tfhe_amd_multi_ctx {
    tfhe_amd_params p{};
    std::vector<tfhe_amd_ctx*> ctx;  // one per entry of the device list (a device may appear twice)
};

namespace {

// contiguous block split: device g of P owns gates [g G / P, (g + 1) G / P)
size_t block_begin(size_t n_gates, size_t g, size_t parts) { return n_gates * g / parts; }

// Runs body(g) for every device of the list on a host thread of its own and returns the first
// failure (with its message as this thread's last error).
template <typename Body>
int on_all_devices(tfhe_amd_multi_ctx* m, Body body) {
    const size_t P = m->ctx.size();
    std::vector<int> rc(P, TFHE_AMD_OK);
    std::vector<std::string> msg(P);
    auto run = [&](size_t g) {
        rc[g] = body(g);
        if (rc[g]) msg[g] = g_last_error;  // thread-local in the worker
    };
    std::vector<std::thread> workers;
    workers.reserve(P);
    for (size_t g = 1; g < P; g++) {
        try {
            workers.emplace_back(run, g);
        } catch (const std::system_error&) {
            run(g);  // no thread to spare: this device's share runs on the caller's thread
        }
    }
    run(0);
    for (std::thread& t : workers) t.join();
    for (size_t g = 0; g < P; g++)
        if (rc[g]) return fail(rc[g], "device " + std::to_string(m->ctx[g]->device) + ": " + msg[g]);
    return TFHE_AMD_OK;
}

}  // namespace

extern "C" {

int tfhe_amd_multi_ctx_create(const tfhe_amd_params* p, const uint32_t* bk, const uint32_t* ks, const int* device_ids,
                              int n_devices, unsigned flags, tfhe_amd_multi_ctx** out) {
    if (!out) return fail(TFHE_AMD_ERR_ARGUMENT, "out is null");
    *out = nullptr;
    if (!params_sane(p)) return fail(TFHE_AMD_ERR_ARGUMENT, "parameter set violates the reference's constraints");
    if (!device_ids || n_devices < 1 || n_devices > kMaxDevices)
        return fail(TFHE_AMD_ERR_ARGUMENT, "device list is empty or too long");
    if (!bk && !ks) return fail(TFHE_AMD_ERR_ARGUMENT, "both key pointers are null");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    for (int g = 0; g < n_devices; g++)
        if (device_ids[g] < 0 || device_ids[g] >= ndev) return fail(TFHE_AMD_ERR_ARGUMENT, "no such device");
    tfhe_amd_multi_ctx* m = new (std::nothrow) tfhe_amd_multi_ctx();
    if (!m) return fail(TFHE_AMD_ERR_MEMORY, "host allocation failed");
    m->p = *p;
    m->ctx.assign((size_t)n_devices, nullptr);
    auto bail = [&](int rc) {
        const std::string msg = g_last_error;
        tfhe_amd_multi_ctx_destroy(m);
        return fail(rc, msg);
    };
    // The raw keys reach the first device of the list once (from the host, or they are there already
    // with TFHE_AMD_CTX_KEYS_ON_DEVICE); the other devices get theirs by device-to-device copies over
    // xGMI, and every device transforms its own copy: the raw keys are half the size of the
    // transformed ones.
    const int dev0 = device_ids[0];
    const size_t bk_bytes = tfhe_amd_bk_words(p) * sizeof(uint32_t), ks_bytes = tfhe_amd_ks_words(p) * sizeof(uint32_t);
    const bool on_device = (flags & TFHE_AMD_CTX_KEYS_ON_DEVICE) != 0;
    DevBuf raw_bk, raw_ks;  // on dev0
    const uint32_t* d_bk = bk;
    const uint32_t* d_ks = ks;
    if (!on_device && n_devices > 1) {
        DeviceGuard guard(dev0);
        if (!guard.ok) return bail(fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed"));
        hipError_t e = hipSuccess;
        if (bk && (e = raw_bk.alloc(bk_bytes)) == hipSuccess) e = hipMemcpy(raw_bk.p, bk, bk_bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess && ks && (e = raw_ks.alloc(ks_bytes)) == hipSuccess)
            e = hipMemcpy(raw_ks.p, ks, ks_bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) return bail(fail(hip_code(e), std::string("uploading the keys: ") + hipGetErrorString(e)));
        d_bk = bk ? raw_bk.as<uint32_t>() : nullptr;
        d_ks = ks ? raw_ks.as<uint32_t>() : nullptr;
    }
    const unsigned dev_flags = flags | (n_devices > 1 ? (unsigned)TFHE_AMD_CTX_KEYS_ON_DEVICE : 0u);
    const int rc = on_all_devices(m, [&](size_t g) -> int {
        const int dev = device_ids[g];
        if (dev == dev0 || n_devices == 1) return tfhe_amd_ctx_create(p, d_bk, d_ks, dev, dev_flags, &m->ctx[g]);
        DeviceGuard guard(dev);
        if (!guard.ok) return fail(TFHE_AMD_ERR_HIP, "hipSetDevice failed");
        DevBuf peer_bk, peer_ks;
        if (d_bk) {
            HIP_TRY(peer_bk.alloc(bk_bytes));
            HIP_TRY(hipMemcpyPeer(peer_bk.p, dev, d_bk, dev0, bk_bytes));
        }
        if (d_ks) {
            HIP_TRY(peer_ks.alloc(ks_bytes));
            HIP_TRY(hipMemcpyPeer(peer_ks.p, dev, d_ks, dev0, ks_bytes));
        }
        return tfhe_amd_ctx_create(p, d_bk ? peer_bk.as<uint32_t>() : nullptr, d_ks ? peer_ks.as<uint32_t>() : nullptr, dev,
                                   dev_flags, &m->ctx[g]);
    });
    if (rc) return bail(rc);
    *out = m;
    return TFHE_AMD_OK;
}

void tfhe_amd_multi_ctx_destroy(tfhe_amd_multi_ctx* m) {
    if (!m) return;
    for (tfhe_amd_ctx* c : m->ctx) tfhe_amd_ctx_destroy(c);
    delete m;
}

int tfhe_amd_multi_ctx_device_count(const tfhe_amd_multi_ctx* m) { return m ? (int)m->ctx.size() : 0; }

tfhe_amd_ctx* tfhe_amd_multi_ctx_get(tfhe_amd_multi_ctx* m, int index) {
    return (m && index >= 0 && index < (int)m->ctx.size()) ? m->ctx[(size_t)index] : nullptr;
}

void tfhe_amd_multi_shard(const tfhe_amd_multi_ctx* m, size_t n_gates, int index, size_t* begin, size_t* end) {
    const size_t P = m ? m->ctx.size() : 1, g = (size_t)(index < 0 ? 0 : index);
    if (begin) *begin = block_begin(n_gates, g < P ? g : P, P);
    if (end) *end = block_begin(n_gates, g + 1 < P ? g + 1 : P, P);
}

int tfhe_amd_multi_gate_batch(tfhe_amd_multi_ctx* m, int op, const uint32_t* a, const uint32_t* b, uint32_t* out,
                              size_t n_gates) {
    if (!m) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    if (n_gates && !(a && b && out)) return fail(TFHE_AMD_ERR_ARGUMENT, "null array argument");
    int rc = check_gate_code(op);
    if (rc || n_gates == 0) return rc;
    const size_t P = m->ctx.size(), n1 = (size_t)m->p.n + 1;
    return on_all_devices(m, [&](size_t g) -> int {
        const size_t lo = block_begin(n_gates, g, P), hi = block_begin(n_gates, g + 1, P);
        if (hi == lo) return TFHE_AMD_OK;
        return tfhe_amd_gate_batch(m->ctx[g], op, a + lo * n1, b + lo * n1, out + lo * n1, hi - lo);
    });
}

int tfhe_amd_multi_nand_batch(tfhe_amd_multi_ctx* m, const uint32_t* a, const uint32_t* b, uint32_t* out,
                              size_t n_gates) {
    return tfhe_amd_multi_gate_batch(m, TFHE_AMD_GATE_NAND, a, b, out, n_gates);
}

int tfhe_amd_multi_bootstrap_batch(tfhe_amd_multi_ctx* m, const uint32_t* c, uint32_t* out, size_t n_gates) {
    if (!m) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    if (n_gates && !(c && out)) return fail(TFHE_AMD_ERR_ARGUMENT, "null array argument");
    if (n_gates == 0) return TFHE_AMD_OK;
    const size_t P = m->ctx.size(), n1 = (size_t)m->p.n + 1;
    return on_all_devices(m, [&](size_t g) -> int {
        const size_t lo = block_begin(n_gates, g, P), hi = block_begin(n_gates, g + 1, P);
        if (hi == lo) return TFHE_AMD_OK;
        return tfhe_amd_bootstrap_batch(m->ctx[g], c + lo * n1, out + lo * n1, hi - lo);
    });
}

int tfhe_amd_multi_nand_batch_device(tfhe_amd_multi_ctx* m, const uint32_t* const* a, const uint32_t* const* b,
                                     uint32_t* const* out, const size_t* n_gates, void* const* streams) {
    if (!m) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    if (!a || !b || !out || !n_gates) return fail(TFHE_AMD_ERR_ARGUMENT, "null array argument");
    // enqueueing is quick: one host thread walks the devices, the GPUs run side by side
    for (size_t g = 0; g < m->ctx.size(); g++) {
        const int rc = tfhe_amd_nand_batch_device(m->ctx[g], a[g], b[g], out[g], n_gates[g], streams ? streams[g] : nullptr);
        if (rc) return rc;
    }
    return TFHE_AMD_OK;
}

int tfhe_amd_multi_ctx_synchronize(tfhe_amd_multi_ctx* m) {
    if (!m) return fail(TFHE_AMD_ERR_ARGUMENT, "null context");
    for (tfhe_amd_ctx* c : m->ctx) {
        const int rc = tfhe_amd_ctx_synchronize(c);
        if (rc) return rc;
    }
    return TFHE_AMD_OK;
}

const char* tfhe_amd_last_error(void) { return g_last_error.c_str(); }
const char* tfhe_amd_version(void) { return "0.4.0"; }

}  // extern "C"
