// CPU harness for cyclistsocialforce_amd/csrc/csf_math64.h (tests/test_math64.py): the header is compiled with g++, the two
// hardware approximations are libm results cut to 25 bits (v_rcp_f64 / v_rsq_f64 deliver at least that), and every function is
// compared with libm over the arguments the per-agent kernel has.  Prints one JSON object of worst errors.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#define CSF_HD static inline
#include "csf_math64.h"

static double cut(double v) {
    uint64_t b;
    std::memcpy(&b, &v, 8);
    b &= ~((1ull << 27) - 1);
    std::memcpy(&v, &b, 8);
    return v;
}
struct HwHost {
    static double rcp(double b) { return cut(1.0 / b); }
    static double rsq(double x) { return cut(1.0 / std::sqrt(x)); }
};
using namespace csf::m64;

static double ulps(double got, double want) {
    if (got == want) return 0.0;
    const double u = std::fabs(want) > 0 ? std::ldexp(1.0, std::ilogb(want) - 52) : 4.9e-324;
    return std::fabs(got - want) / u;
}

int main() {
    std::mt19937_64 rng(12345);
    std::uniform_real_distribution<double> U(0.0, 1.0);
    auto logu = [&](double lo, double hi) { return std::exp(std::log(lo) + U(rng) * (std::log(hi) - std::log(lo))); };
    auto sgn = [&]() { return U(rng) < 0.5 ? -1.0 : 1.0; };
    double e_sqrt = 0, e_div = 0, e_rcp = 0, e_rsq = 0, e_sin = 0, e_cos = 0, e_tan = 0, e_atan2 = 0, a_sin = 0, a_cos = 0, a_atan2 = 0;
    const int N = 2000000;
    for (int i = 0; i < N; i++) {
        const double x = logu(1e-30, 1e30), a = sgn() * logu(1e-12, 1e12), b = sgn() * logu(1e-12, 1e12);
        e_sqrt = std::fmax(e_sqrt, ulps(sqrt_s<HwHost>(x), std::sqrt(x)));
        e_rsq = std::fmax(e_rsq, ulps(rsqrt_s<HwHost>(x), 1.0 / std::sqrt(x)));
        e_rcp = std::fmax(e_rcp, ulps(rcp_s<HwHost>(b), 1.0 / b));
        e_div = std::fmax(e_div, ulps(div_s<HwHost>(a, b), a / b));
        // angles: wrapped ones, with many near the multiples of pi/2; a few unwrapped yaws of hundreds of turns
        double th = (U(rng) * 2 - 1) * 3.141592653589793;
        if (i % 4 == 1) th = std::rint(th / 1.5707963267948966) * 1.5707963267948966 + sgn() * logu(1e-18, 1e-3);
        if (i % 16 == 2) th = (U(rng) * 2 - 1) * 5000.0;
        double s, c;
        sincos_s(th, &s, &c);
        a_sin = std::fmax(a_sin, std::fabs(s - std::sin(th)));
        a_cos = std::fmax(a_cos, std::fabs(c - std::cos(th)));
        if (std::fabs(th) < 3.2) {
            e_sin = std::fmax(e_sin, ulps(s, std::sin(th)));
            e_cos = std::fmax(e_cos, ulps(c, std::cos(th)));
        }
        const double de = (U(rng) * 2 - 1) * 1.5;
        e_tan = std::fmax(e_tan, ulps(tan_s<HwHost>(de), std::tan(de)));
        double yy = sgn() * logu(1e-9, 1e3), xx = sgn() * logu(1e-9, 1e3);
        if (i % 8 == 3) yy = xx * (1.0 + sgn() * logu(1e-16, 1e-2));            // near the diagonals
        if (i % 8 == 4) yy = xx * (i % 16 == 4 ? 0.4375 : 0.6875) * (1.0 + sgn() * logu(1e-16, 1e-3));   // near the break points
        if (i % 64 == 5) yy = 0.0;
        if (i % 64 == 6) xx = 0.0;
        const double got = atan2_s<HwHost>(yy, xx), want = std::atan2(yy, xx);
        e_atan2 = std::fmax(e_atan2, ulps(got, want));
        a_atan2 = std::fmax(a_atan2, std::fabs(got - want));
    }
    const bool zeros = sqrt_s<HwHost>(0.0) == 0.0 && atan2_s<HwHost>(0.0, 0.0) == 0.0 && atan2_s<HwHost>(0.0, -2.0) == std::atan2(0.0, -2.0) &&
                       atan2_s<HwHost>(-0.0, -2.0) == std::atan2(-0.0, -2.0) && div_s<HwHost>(0.0, 3.0) == 0.0;
    std::printf("{\"ulp_sqrt\": %.3f, \"ulp_rsqrt\": %.3f, \"ulp_rcp\": %.3f, \"ulp_div\": %.3f, \"ulp_sin\": %.3f, \"ulp_cos\": %.3f, \"ulp_tan\": %.3f, "
                "\"ulp_atan2\": %.3f, \"abs_sin\": %.3e, \"abs_cos\": %.3e, \"abs_atan2\": %.3e, \"zeros\": %s}\n",
                e_sqrt, e_rsq, e_rcp, e_div, e_sin, e_cos, e_tan, e_atan2, a_sin, a_cos, a_atan2, zeros ? "true" : "false");
    return 0;
}
