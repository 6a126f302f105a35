// TEST INFRASTRUCTURE ONLY -- the reference's OWN test driver (/root/reference/test/main.cpp,
// compiled where it lies, unmodified) built against THIS repository's drop-in headers and
// libtfhe_amd instead of the reference's headers.  One test per process: the reference's
// tests abort on failure.  Built only where /root/reference exists (oracle/Makefile, target
// `ref`), output into oracle/_ref/.
#include <cstdio>
#include <cstring>

#define main reference_main
#include "test/main.cpp"
#undef main

int main(int argc, char** argv) {
    const char* t = argc > 1 ? argv[1] : "HomNAND";
    if (!std::strcmp(t, "TLWE")) test_TLWE();
    else if (!std::strcmp(t, "TRLWE")) test_TRLWE();
    else if (!std::strcmp(t, "ExternalProduct")) test_ExternalProduct();
    else if (!std::strcmp(t, "CMUX")) test_CMUX();
    else if (!std::strcmp(t, "PolynomialXExp")) test_PolynomialXExp();
    else if (!std::strcmp(t, "Bootstrapping")) test_Bootstrapping();
    else if (!std::strcmp(t, "KeySwitching")) test_KeySwitching();
    else if (!std::strcmp(t, "HomNAND")) test_HomNAND();
    else if (!std::strcmp(t, "FFT")) test_FFT();
    else if (!std::strcmp(t, "main")) return reference_main();
    else { std::fprintf(stderr, "unknown test %s\n", t); return 2; }
    std::printf("reference test_%s passed on the drop-in headers\n", t);
    return 0;
}
