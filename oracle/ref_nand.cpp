// TEST INFRASTRUCTURE ONLY -- timing / differential harness around the REAL reference.
//
// Built only where /root/reference exists (oracle/Makefile target `ref`), output into
// oracle/_ref/.  It fills the reference's own containers from flat key / ciphertext
// blobs (layouts in oracle/tfhe_oracle.h), calls TFHE::HomNAND<>::nand (reference
// src/hom_nand.hpp:74-81) on independent gates from several threads sharing one
// read-only EvaluateKey, writes the outputs and prints one JSON timing line.
// Default parameters only: HomNAND cannot be instantiated with another ring
// (reference src/hom_nand.hpp:26,49).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hom_nand.hpp"

using H = TFHE::HomNAND<>;

static std::vector<uint32_t> slurp(const char* path) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(1); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<uint32_t> v(sz / 4);
    if (fread(v.data(), 4, v.size(), f) != v.size()) { perror("fread"); exit(1); }
    fclose(f);
    return v;
}

int main(int argc, char** argv) {
    if (argc < 6) { fprintf(stderr, "usage: %s bk.bin ks.bin in.bin out.bin threads\n", argv[0]); return 1; }
    constexpr int n = 636, N = 512, K = 2, L = 2, T = 5, BB = 2, n1 = n + 1;
    auto bkv = slurp(argv[1]);
    auto ksv = slurp(argv[2]);
    auto in = slurp(argv[3]);
    int threads = atoi(argv[5]);
    if (bkv.size() != (size_t)n * (K + 1) * L * (K + 1) * N) { fprintf(stderr, "bk size\n"); return 1; }
    if (ksv.size() != (size_t)N * K * T * (1 << BB) * n1) { fprintf(stderr, "ks size\n"); return 1; }
    size_t G = in.size() / (2 * n1);

    auto ek = std::make_unique<H::EvaluateKey>();
    size_t q = 0;
    for (int i = 0; i < n; i++)
        for (int r = 0; r < (K + 1) * L; r++)
            for (int c = 0; c < K + 1; c++)
                for (int j = 0; j < N; j++) ek->bk[i][r][c][j] = TFHE::TorusValue::from_raw_value(bkv[q++]);
    q = 0;
    for (int i = 0; i < N * K; i++)
        for (int m = 0; m < T; m++)
            for (int o = 0; o < (1 << BB); o++)
                for (int w = 0; w < n1; w++) ek->ks[i][m][o][w] = TFHE::TorusValue::from_raw_value(ksv[q++]);

    std::vector<H::Cipher> a(G), b(G), out(G);
    for (size_t g = 0; g < G; g++)
        for (int w = 0; w < n1; w++) {
            a[g][w] = TFHE::TorusValue::from_raw_value(in[g * n1 + w]);
            b[g][w] = TFHE::TorusValue::from_raw_value(in[(G + g) * n1 + w]);
        }

    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; t++)
        pool.emplace_back([&, t] {
            H h;  // one instance per thread: nand() only reads its arguments
            for (size_t g = t; g < G; g += threads) out[g] = h.nand(a[g], b[g], *ek);
        });
    for (auto& th : pool) th.join();
    double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    std::vector<uint32_t> ov;
    for (size_t g = 0; g < G; g++) for (int w = 0; w < n1; w++) ov.push_back(out[g][w].get_raw_value());
    FILE* f = fopen(argv[4], "wb");
    fwrite(ov.data(), 4, ov.size(), f);
    fclose(f);
    printf("{\"gates\": %zu, \"threads\": %d, \"seconds\": %.6f, \"gates_per_s\": %.6f}\n", G, threads, sec, G / sec);
    return 0;
}
