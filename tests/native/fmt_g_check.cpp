// Checks src/fmt_g.h against printf("%g") (CPU only): every text must be identical.
//   g++ -O2 -std=c++17 -I src tests/native/fmt_g_check.cpp -o fmt_g_check && ./fmt_g_check [millions]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "fmt_g.h"

template <typename F>
static long check(F v, long &bad)
{
    char a[64], b[64];
    fmtg::format_g6(a, sizeof a, v);
    std::snprintf(b, sizeof b, "%g", (double)v);
    if (std::strcmp(a, b) != 0) {
        if (bad < 20) std::printf("MISMATCH %a: got '%s' want '%s'\n", (double)v, a, b);
        ++bad;
    }
    return 1;
}

int main(int argc, char **argv)
{
    const long millions = argc > 1 ? std::atol(argv[1]) : 4;
    long bad = 0, n = 0;
    std::mt19937_64 rng(12345);
    // hand-picked: zeros, powers of ten and their neighbours, ties at six digits, range ends, denormals
    const double picks[] = {0.0, -0.0, 1.0, -1.0, 0.1, 0.5, 0.25, 100000.0, 999999.5, 999999.4999, 1e5 - 0.5, 1e6, 1e-4, 1e-5, 9.999995e-5,
                            9.9999949e-5, 123456.5, 123455.5, 0.1234565, 0.1234575, 1.000005, 1.0000049, 1e-38, 1e38, 3.4e38,
                            1.4e-45, 1e-300, 1e300, 1e-310, 1.7e308, 5e-324, 0.0001234565, 1234565.0, 12345650000.0, 2.5e-7,
                            0.099997, 0.0976405, 7.65472e-05, 0.000838482};
    for (double p : picks)
        for (double s : {1.0, -1.0}) {
            n += check<double>(s * p, bad);
            n += check<float>((float)(s * p), bad);
            n += check<double>(std::nextafter(s * p, 1e308), bad);
            n += check<double>(std::nextafter(s * p, -1e308), bad);
            n += check<float>(std::nextafterf((float)(s * p), 1e38f), bad);
            n += check<float>(std::nextafterf((float)(s * p), -1e38f), bad);
        }
    for (int k = -320; k <= 308; ++k) {
        const double p = std::pow(10.0, k);
        for (double f : {1.0, 0.9999995, 0.99999949, 1.0000005, 9.999995, 5.0, 1.5})
            n += check<double>(p * f, bad), n += check<float>((float)(p * f), bad);
    }
    // every tie and near-tie of six digits in one decade (floats and doubles)
    for (long i = 100000; i < 1000000; i += 7) {
        n += check<double>((double)i + 0.5, bad);
        n += check<double>(((double)i + 0.5) * 1e-9, bad);
        n += check<float>((float)(((double)i + 0.5) * 1e-3), bad);
    }
    // random bit patterns (all exponents), and values of a running simulation
    std::uniform_real_distribution<double> sim(-0.2, 0.2);
    for (long i = 0; i < millions * 1000000; ++i) {
        const uint64_t bits = rng();
        float f; uint32_t b32 = (uint32_t)bits; std::memcpy(&f, &b32, 4);
        double d; std::memcpy(&d, &bits, 8);
        n += check<float>(f, bad);
        if ((i & 3) == 0) n += check<double>(d, bad);
        n += check<float>((float)sim(rng), bad);
        if ((i & 3) == 1) n += check<double>(sim(rng) * 1e-3, bad);
    }
    // speed on simulation-like values
    std::vector<float> vals(4000000);
    for (float &x : vals) x = (float)sim(rng);
    char tmp[64];
    long len = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (float x : vals) len += fmtg::format_g6(tmp, sizeof tmp, x);
    auto t1 = std::chrono::steady_clock::now();
    for (float x : vals) len += std::snprintf(tmp, sizeof tmp, "%g", (double)x);
    auto t2 = std::chrono::steady_clock::now();
    for (float x : vals) len += fmtg::slow_path(tmp, sizeof tmp, x);
    auto t3 = std::chrono::steady_clock::now();
    std::printf("%ld values checked, %ld mismatches; %.1f ns per value against %.1f for snprintf and %.1f for std::to_chars (%ld)\n", n, bad,
                std::chrono::duration<double, std::nano>(t1 - t0).count() / vals.size(),
                std::chrono::duration<double, std::nano>(t2 - t1).count() / vals.size(),
                std::chrono::duration<double, std::nano>(t3 - t2).count() / vals.size(), len);
    return bad ? 1 : 0;
}
