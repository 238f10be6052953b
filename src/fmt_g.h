// "%g" at the default precision of six significant digits -- what the reference's `stream << value`
// prints for every number of its -omv / -ov / -ovf outputs (io.cpp:219, :245, :271) -- several times
// faster than snprintf or std::to_chars.  Plain C++17, no device code.
//
// The value is scaled by a power of ten so that its six leading digits become an integer, in double
// arithmetic: the product carries a relative error of a few 1e-16, so the integer part and the
// rounding direction is certain unless the fraction lies within 1e-7 of one half -- those two values
// in ten million (and infinities, NaNs and doubles beyond 1e+-280) take the library's own conversion.
// The text is identical to printf's in every case.
#pragma once
#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>

namespace fmtg {

// 10^k for k = -340 .. 340, filled once
struct Pow10Table {
    double v[681];
    Pow10Table()
    {
        for (int k = -340; k <= 340; ++k) v[k + 340] = std::pow(10.0, (double)k);
    }
    double operator()(int k) const { return v[k + 340]; }
};
inline const Pow10Table &pow10_table()
{
    static const Pow10Table t;
    return t;
}

template <typename F>
inline int slow_path(char *dst, size_t cap, F v)
{
    if (!(std::fabs(v) <= std::numeric_limits<F>::max())) return std::snprintf(dst, cap, "%g", (double)v);
    const std::to_chars_result r = std::to_chars(dst, dst + cap - 1, v, std::chars_format::general, 6);
    *r.ptr = '\0';
    return (int)(r.ptr - dst);
}

// Writes v as printf("%g") would (NUL-terminated) and returns the length.  cap >= 32.
template <typename F>
inline int format_g6(char *dst, size_t cap, F v)
{
    const double a = std::fabs((double)v);
    if (a == 0.0) {
        // printf keeps the sign of a negative zero
        if (std::signbit(v)) { std::memcpy(dst, "-0", 3); return 2; }
        std::memcpy(dst, "0", 2);
        return 1;
    }
    if (!(a <= 1e280) || a < 1e-280) return slow_path(dst, cap, v);
    const Pow10Table &p10 = pow10_table();
    // decimal exponent of the leading digit: from the binary exponent, corrected below if the scaled
    // value falls outside [1e5, 1e6)
    uint64_t bits;
    std::memcpy(&bits, &a, 8);
    const int e2 = (int)(bits >> 52) - 1023;                 // a is a normal double here: 2^e2 <= a < 2^(e2+1)
    int e10 = (e2 * 78913) >> 18;                            // floor(e2 * log10(2)), at most one too small
    uint64_t n = 0;
    for (int attempt = 0; attempt < 3; ++attempt) {
        const double m = a * p10(5 - e10);     // |5 - e10| <= 285: the power and the product are normal doubles
        if (m < 1e5) { --e10; continue; }
        if (m >= 1e6) { ++e10; continue; }
        const uint64_t whole = (uint64_t)m;    // truncation = floor for a positive value
        const double frac = m - (double)whole;
        // too close to a tie of the exact value to call (near a whole number either neighbour rounds to it)
        if (std::fabs(frac - 0.5) < 1e-7) return slow_path(dst, cap, v);
        n = whole + (frac > 0.5 ? 1u : 0u);
        if (n == 1000000u) { n = 100000u; ++e10; }
        break;
    }
    if (n < 100000u || n > 999999u) return slow_path(dst, cap, v);
    char digits[6];
    {
        static const char two[] = "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
                                  "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
                                  "8081828384858687888990919293949596979899";
        const unsigned hi = (unsigned)n / 10000u, rest = (unsigned)n % 10000u, mid = rest / 100u, lo = rest % 100u;
        std::memcpy(digits, two + 2 * hi, 2);
        std::memcpy(digits + 2, two + 2 * mid, 2);
        std::memcpy(digits + 4, two + 2 * lo, 2);
    }
    int nd = 6;
    while (nd > 1 && digits[nd - 1] == '0') --nd;     // %g drops trailing zeros
    char *p = dst;
    if (std::signbit(v)) *p++ = '-';
    if (e10 >= -4 && e10 < 6) {
        if (e10 >= 0) {
            // digits before the point: e10 + 1 of them, padded with zeros the stripping removed
            for (int i = 0; i <= e10; ++i) *p++ = i < nd ? digits[i] : '0';
            if (nd > e10 + 1) {
                *p++ = '.';
                for (int i = e10 + 1; i < nd; ++i) *p++ = digits[i];
            }
        } else {
            *p++ = '0';
            *p++ = '.';
            for (int i = -1; i > e10; --i) *p++ = '0';
            for (int i = 0; i < nd; ++i) *p++ = digits[i];
        }
    } else {
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            for (int i = 1; i < nd; ++i) *p++ = digits[i];
        }
        *p++ = 'e';
        int e = e10;
        if (e < 0) { *p++ = '-'; e = -e; } else { *p++ = '+'; }
        if (e >= 100) { *p++ = (char)('0' + e / 100); e %= 100; }
        *p++ = (char)('0' + e / 10);
        *p++ = (char)('0' + e % 10);
    }
    *p = '\0';
    return (int)(p - dst);
}

}  // namespace fmtg
