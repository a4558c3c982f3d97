// csf_math64.h — short fp64 elementary functions for the per-agent kernel (csf_agent.hip).
//
// The per-agent kernel is ONE wave per CU running a dependent chain: its time is its instruction count.  The
// device library's fp64 sqrt / division / sincos / tan / atan2 are written for every argument a double can hold (scaling
// against over- and underflow, Payne - Hanek reduction, IEEE-exact quotients): 31, 19, ~200, ~220 and ~125 instructions.  The
// kernel's arguments are lengths of a traffic scene (1e-9 .. 1e6 m), angles that limitAngle has just wrapped, and steering
// angles below pi/2, so the same results to an ulp or two come out of a third of the instructions:
//   sqrt_s   v_rsq_f64 + one Goldschmidt step + two corrections (what the compiler emits, without the range scaling)
//   div_s    v_rcp_f64 + two Newton steps + one correction of the quotient
//   sincos_s two-constant Cody - Waite reduction by pi/2 (exact for |x| < 1e5) + the degree-13 / 14 kernels of fdlibm
//   tan_s    sin / cos of the same
//   atan2_s  min / max quotient folded into |t| <= 7/16 BEFORE the one division, fdlibm's atan polynomial
// The results feed state that the goldens pin at 1e-12 (201 steps of control + move) and 2e-7 (the planner); none of them
// decides a yes / no of the reference - the field-of-view chain (csf_dev.h: untracked_exact_xy, sign_phi_exact) keeps the
// library's atan2 / acos.
//
// HW is the policy that supplies the two hardware approximations (device: v_rcp_f64 / v_rsq_f64; tests/test_math64.py
// compiles this header with g++ and 1/x, 1/sqrt(x) rounded to ~26 bits, and compares with libm).  CSF_HD: the includer's
// function qualifiers.
#pragma once
#include <math.h>

namespace csf {
namespace m64 {

template <class HW>
CSF_HD double rcp_s(double b) {       // 1 / b to an ulp or two (b != 0, finite)
    double x = HW::rcp(b);
    x = fma(fma(-b, x, 1.0), x, x);
    x = fma(fma(-b, x, 1.0), x, x);
    return x;
}

template <class HW>
CSF_HD double div_s(double a, double b) {   // a / b (b != 0, finite); a = 0 gives 0
    const double r = rcp_s<HW>(b);
    const double q = a * r;
    return fma(fma(-b, q, a), r, q);
}

template <class HW>
CSF_HD double sqrt_s(double x) {      // x in {0} u [1e-280, 1e280]; negative: NaN
    const double y = HW::rsq(x);
    double g = x * y, h = 0.5 * y;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g);
    h = fma(h, r, h);
    g = fma(fma(-g, g, x), h, g);
    g = fma(fma(-g, g, x), h, g);
    return x == 0.0 ? 0.0 : g;
}

template <class HW>
CSF_HD double rsqrt_s(double x) {     // 1 / sqrt(x), x > 0
    double y = HW::rsq(x);
    double e = fma(-x * y, y, 1.0);   // 1 - x y^2
    y = fma(y * fma(0.375, e, 0.5), e, y);          // y (1 + e/2 + 3 e^2 / 8)
    e = fma(-x * y, y, 1.0);
    return fma(0.5 * y, e, y);
}

// sin and cos of x, |x| < 1e5 (an angle limitAngle has wrapped, a steering angle, an unwrapped yaw)
CSF_HD void sincos_s(double x, double *sn, double *cs) {
    const double fn = rint(x * 6.36619772367581382433e-01);
    double r = fma(-fn, 1.57079632673412561417e+00, x);       // exact: 33 bits of pi/2 times an integer below 2^17
    r = fma(-fn, 6.07710050650619224932e-11, r);
    const int n = (int)fn;
    const double z = r * r;
    const double ps = 8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 + z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10)));
    const double s = fma(z * r, fma(z, ps, -1.66666666666666324348e-01), r);
    const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 + z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
    const double hz = 0.5 * z, w = 1.0 - hz;
    const double c = w + (((1.0 - w) - hz) + z * z * pc);
    const bool swap = (n & 1) != 0;
    const double ss = swap ? c : s, cc = swap ? s : c;
    *sn = (n & 2) ? -ss : ss;
    *cs = ((n + 1) & 2) ? -cc : cc;
}

template <class HW>
CSF_HD double tan_s(double x) {       // |x| < pi/2 - 1e-3 (a steering angle)
    double s, c;
    sincos_s(x, &s, &c);
    return div_s<HW>(s, c);
}

// atan2(y, x) as numpy has it (atan2(0, 0) = 0), finite arguments
template <class HW>
CSF_HD double atan2_s(double y, double x) {
    const double ax = fabs(x), ay = fabs(y);
    const double mx = fmax(ax, ay), mn = fmin(ax, ay);
    // atan(mn / mx) with the argument folded into |t| <= 7/16 before the division (fdlibm's break points 7/16 and 11/16:
    // atan(a) = atan(c) + atan((a - c) / (1 + a c)), c = 1/2, 1)
    const bool lo = 16.0 * mn < 7.0 * mx, mid = 16.0 * mn < 11.0 * mx;
    const double num = lo ? mn : (mid ? 2.0 * mn - mx : mn - mx);
    const double den = lo ? mx : (mid ? fma(2.0, mx, mn) : mn + mx);
    const double hi = lo ? 0.0 : (mid ? 4.63647609000806093515e-01 : 7.85398163397448278999e-01);
    const double lw = lo ? 0.0 : (mid ? 2.26987774529616870924e-17 : 3.06161699786838301793e-17);
    const double t = div_s<HW>(num, den);
    const double z = t * t, w = z * z;
    const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 + w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
    const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 + w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
    double r = hi - ((t * (s1 + s2) - lw) - t);                // in [0, pi/4]
    if (ay > ax) r = 1.57079632679489655800e+00 - (r - 6.12323399573676603587e-17);
    if (x < 0.0) r = 3.14159265358979311600e+00 - (r - 1.22464679914735317720e-16);
    r = mx == 0.0 ? 0.0 : r;
    return copysign(r, y);
}

}  // namespace m64
}  // namespace csf
