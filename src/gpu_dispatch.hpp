// Glue between the drop-in template headers and the C ABI of libtfhe_amd (include/tfhe_amd.h):
// flattening of the reference's pointer-tree containers into the flat uint32_t layouts the
// library takes, and a small cache of device contexts keyed on the identity of the key
// objects (the reference's nand(a, b, evaluate_key) carries no context handle).
//
// Evaluation has no host implementation behind these calls: a missing library is a link
// error, and every run-time failure throws std::runtime_error with the library's message.
#pragma once

#include <cstdint>
#include <cstdlib>
#include <list>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "tfhe_amd.h"
#include "types.hpp"

namespace TFHE {
namespace gpu {

inline void check(int rc, const char* what) {
    if (rc != TFHE_AMD_OK)
        throw std::runtime_error(std::string(what) + ": tfhe_amd error " + std::to_string(rc) + ": " +
                                 tfhe_amd_last_error());
}

inline int& device() {  // GPU used by the header-level calls; change before the first evaluation
    static int dev = 0;
    return dev;
}

// GPUs that the batch calls (HomNAND::nand_batch, gate_batch) spread their gates over, in
// contiguous blocks; empty (the default) = device() alone.  Single-gate calls run on the first GPU
// of the list.  Initial value: the environment variable TFHE_AMD_DEVICES, a comma-separated list
// ("0,1,2,3").  Change before the first evaluation with a key, or call cache().clear().
inline std::vector<int>& devices() {
    static std::vector<int> devs = [] {
        std::vector<int> v;
        if (const char* e = std::getenv("TFHE_AMD_DEVICES")) {
            const char* q = e;
            while (*q) {
                char* end = nullptr;
                const long d = std::strtol(q, &end, 10);
                if (end == q) break;
                v.push_back((int)d);
                q = (*end == ',') ? end + 1 : end;
            }
        }
        return v;
    }();
    return devs;
}

// Context flags (TFHE_AMD_CTX_*) for the contexts the headers create.  Setting
// TFHE_AMD_CTX_REFERENCE_FFT makes nand() and the blind rotations return the ciphertexts of the
// reference built with g++ -O2 bit for bit (its FFT product replayed) instead of the exact ones;
// the ring must support it (degree 32 .. 1024).  Change before the first evaluation with a key, or
// call cache().clear().  Unmodified programs can be switched from outside with the environment
// variable TFHE_AMD_CONTEXT_FLAGS=16.
inline unsigned& context_flags() {
    // initial value: the environment variable TFHE_AMD_CONTEXT_FLAGS (a number), else the default
    static unsigned flags = [] {
        const char* e = std::getenv("TFHE_AMD_CONTEXT_FLAGS");
        return e ? (unsigned)std::strtoul(e, nullptr, 0) : (unsigned)TFHE_AMD_CTX_DEFAULT;
    }();
    return flags;
}

// ---- flatten / unflatten -------------------------------------------------------------------
template <int N>
void put(std::vector<uint32_t>& out, const Vector<TorusValue, N>& v) {
    for (int i = 0; i < N; i++) out.push_back(v[i].get_raw_value());
}
template <int N>
void put(std::vector<uint32_t>& out, const Polynomial<TorusValue, N>& v) {
    for (int i = 0; i < N; i++) out.push_back(v[i].get_raw_value());
}
template <int N, int K1>
void put(std::vector<uint32_t>& out, const Vector<Polynomial<TorusValue, N>, K1>& v) {
    for (int q = 0; q < K1; q++) put(out, v[q]);
}
template <int N, int R, int K1>
void put(std::vector<uint32_t>& out, const Matrix<Polynomial<TorusValue, N>, R, K1>& m) {
    for (int r = 0; r < R; r++) put(out, m[r]);
}
// bootstrapping key: [n][(K+1)L][K+1][N]
template <int N, int R, int K1, int TLWE_N>
std::vector<uint32_t> flatten_bk(const Vector<Matrix<Polynomial<TorusValue, N>, R, K1>, TLWE_N>& bk) {
    std::vector<uint32_t> out;
    out.reserve((size_t)TLWE_N * R * K1 * N);
    for (int i = 0; i < TLWE_N; i++) put(out, bk[i]);
    return out;
}
// key-switching key: [N_LV1][T][2^BASE_BIT][N_LV0 + 1]
template <int W, int B, int T, int N_LV1>
std::vector<uint32_t> flatten_ks(const Vector<Vector<Vector<Vector<TorusValue, W>, B>, T>, N_LV1>& ks) {
    std::vector<uint32_t> out;
    out.reserve((size_t)N_LV1 * T * B * W);
    for (int i = 0; i < N_LV1; i++)
        for (int m = 0; m < T; m++)
            for (int o = 0; o < B; o++) put(out, ks[i][m][o]);
    return out;
}
template <int N>
Vector<TorusValue, N> take_vector(const uint32_t* src) {
    Vector<TorusValue, N> v;
    for (int i = 0; i < N; i++) v[i] = TorusValue::from_raw_value(src[i]);
    return v;
}
template <int N, int K1>
Vector<Polynomial<TorusValue, N>, K1> take_trlwe(const uint32_t* src) {
    Vector<Polynomial<TorusValue, N>, K1> v;
    for (int q = 0; q < K1; q++)
        for (int j = 0; j < N; j++) v[q][j] = TorusValue::from_raw_value(src[q * N + j]);
    return v;
}

// ---- context cache ---------------------------------------------------------------------------
// The device copy of a key: one context on device(), or one per GPU of devices().  Handed out as
// shared_ptr so that an entry evicted by one thread stays alive while another thread evaluates on it.
struct DeviceKey {
    tfhe_amd_ctx* single = nullptr;
    tfhe_amd_multi_ctx* multi = nullptr;
    DeviceKey() = default;
    DeviceKey(const DeviceKey&) = delete;
    DeviceKey& operator=(const DeviceKey&) = delete;
    ~DeviceKey() {
        if (multi) tfhe_amd_multi_ctx_destroy(multi);
        if (single) tfhe_amd_ctx_destroy(single);
    }
    // the context single-gate and lower-level calls run on
    tfhe_amd_ctx* ctx() const { return multi ? tfhe_amd_multi_ctx_get(multi, 0) : single; }
};

// Keyed on the addresses of the caller's key objects, the device list and the context flags, plus a
// fingerprint of the key contents (words sampled across the whole key), so that a key destroyed and
// another built at the same address is detected.  A key modified in place between two evaluations
// is detected only if a sampled word changed: call invalidate() after editing a key.  get() may be
// called from several threads at once, as the reference's nand() may on one shared EvaluateKey.
class ContextCache {
    // The device copy of one key set, built at most once: the first thread to ask for it builds it
    // under the slot's own mutex while the list stays open to threads that evaluate on other keys.
    struct Slot {
        std::mutex mu;
        std::shared_ptr<DeviceKey> key;  // null until built
    };
    struct Entry {
        const void* bk_id;
        const void* ks_id;
        uint64_t print;
        tfhe_amd_params params;
        unsigned flags;
        std::vector<int> devs;
        std::shared_ptr<Slot> slot;
    };
    std::mutex mutex_;  // guards entries_ only: lookups, inserts, evictions
    std::list<Entry> entries_;
    static constexpr size_t kMaxEntries = 4;

    static bool same(const tfhe_amd_params& a, const tfhe_amd_params& b) {
        return a.n == b.n && a.N == b.N && a.K == b.K && a.bg_bit == b.bg_bit && a.L == b.L && a.T == b.T &&
               a.base_bit == b.base_bit;
    }
    static std::vector<int> device_list() {
        return devices().empty() ? std::vector<int>{device()} : devices();
    }

public:
    void clear() {
        std::lock_guard<std::mutex> lock(mutex_);
        entries_.clear();  // contexts still in use elsewhere die with their last user
    }
    // forget the device copy of the key objects at these addresses (after editing a key in place)
    void invalidate(const void* bk_id, const void* ks_id) {
        std::lock_guard<std::mutex> lock(mutex_);
        entries_.remove_if([&](const Entry& e) { return e.bk_id == bk_id && e.ks_id == ks_id; });
    }

    // `print`: cheap fingerprint of the key contents; `make_bk` / `make_ks`: produce the flat keys
    // (an empty vector = that key is absent); only called on a miss.  Threads that start together on
    // one key build it once (the others wait on that key's slot); threads on other, cached keys are
    // not held up by the flattening, upload and transform of a miss.
    template <typename MakeBk, typename MakeKs>
    std::shared_ptr<DeviceKey> get(const tfhe_amd_params& params, const void* bk_id, const void* ks_id, uint64_t print,
                                   MakeBk make_bk, MakeKs make_ks) {
        const std::vector<int> devs = device_list();
        const unsigned flags = context_flags();
        std::shared_ptr<Slot> slot;
        {
            std::lock_guard<std::mutex> lock(mutex_);
            for (auto it = entries_.begin(); it != entries_.end(); ++it)
                if (it->bk_id == bk_id && it->ks_id == ks_id && same(it->params, params) && it->flags == flags &&
                    it->devs == devs) {
                    if (it->print == print) {
                        entries_.splice(entries_.begin(), entries_, it);
                        slot = entries_.front().slot;
                    } else {
                        entries_.erase(it);  // same address, different contents
                    }
                    break;
                }
            if (!slot) {
                slot = std::make_shared<Slot>();
                entries_.push_front(Entry{bk_id, ks_id, print, params, flags, devs, slot});
                if (entries_.size() > kMaxEntries) entries_.pop_back();
            }
        }
        std::lock_guard<std::mutex> build_lock(slot->mu);
        if (slot->key) return slot->key;
        try {
            const std::vector<uint32_t> bk = make_bk(), ks = make_ks();
            auto key = std::make_shared<DeviceKey>();
            if (devs.size() > 1)
                check(tfhe_amd_multi_ctx_create(&params, bk.empty() ? nullptr : bk.data(), ks.empty() ? nullptr : ks.data(),
                                                devs.data(), (int)devs.size(), flags, &key->multi),
                      "tfhe_amd_multi_ctx_create");
            else
                check(tfhe_amd_ctx_create(&params, bk.empty() ? nullptr : bk.data(), ks.empty() ? nullptr : ks.data(),
                                          devs[0], flags, &key->single),
                      "tfhe_amd_ctx_create");
            slot->key = key;
        } catch (...) {
            // no entry without a context stays behind: the next call starts over
            std::lock_guard<std::mutex> lock(mutex_);
            entries_.remove_if([&](const Entry& e) { return e.slot == slot; });
            throw;
        }
        return slot->key;
    }
};

inline ContextCache& cache() {
    static ContextCache c;
    return c;
}

// fingerprint helpers: one word of every row of every TRGSW sample / of every (coefficient, level)
// block of the key-switching key, at a position that moves with the row (about 9,000 words of the
// default keys, a few tens of microseconds)
inline uint64_t mix(uint64_t h, uint32_t v) { return (h ^ v) * 0x100000001B3ULL; }
template <int N, int R, int K1, int TLWE_N>
uint64_t fingerprint(const Vector<Matrix<Polynomial<TorusValue, N>, R, K1>, TLWE_N>& bk) {
    uint64_t h = 0xCBF29CE484222325ULL;
    for (int i = 0; i < TLWE_N; i++)
        for (int r = 0; r < R; r++) h = mix(h, bk[i][r][(i + r) % K1][(i * 7 + r * 31) % N].get_raw_value());
    return h;
}
template <int W, int B, int T, int N_LV1>
uint64_t fingerprint(const Vector<Vector<Vector<Vector<TorusValue, W>, B>, T>, N_LV1>& ks) {
    uint64_t h = 0x84222325CBF29CE4ULL;
    for (int i = 0; i < N_LV1; i++)
        for (int m = 0; m < T; m++) h = mix(h, ks[i][m][1 + (i + m) % (B - 1)][(i * 5 + m) % W].get_raw_value());
    return h;
}

}  // namespace gpu
}  // namespace TFHE
