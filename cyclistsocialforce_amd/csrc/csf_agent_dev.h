// csf_agent_dev.h — the per-agent tick as device functions (fp64, one lane per agent): destination queue and navigation
// state machine, destination force (straight line / spline planner), controller + kinematics of the five rider models,
// ring-buffer bookkeeping and the fp32 source record, for agent_kernel (csf_agent.hip).  Reference lines are cited at every
// function.
#pragma once
#include "csf_dev.h"
#define CSF_HD __device__ __forceinline__
#include "csf_math64.h"

namespace csf {


__device__ __forceinline__ double clampd(double x, double lo, double hi) { return fmax(fmin(x, hi), lo); }

// The kernel's elementary functions (csf_math64.h: why, and how exact): lengths, quotients with a divisor that is not zero,
// sin / cos / tan / atan2 of angles that are wrapped or small.  The field-of-view chain keeps the library's (csf_dev.h).
struct HwDev {
    static __device__ __forceinline__ double rcp(double b) { return __builtin_amdgcn_rcp(b); }
    static __device__ __forceinline__ double rsq(double x) { return __builtin_amdgcn_rsq(x); }
};
__device__ __forceinline__ double rcp_nr(double b) { return m64::rcp_s<HwDev>(b); }
__device__ __forceinline__ double qdiv(double a, double b) { return m64::div_s<HwDev>(a, b); }
__device__ __forceinline__ double qsqrt(double x) { return m64::sqrt_s<HwDev>(x); }
__device__ __forceinline__ double qrsqrt(double x) { return m64::rsqrt_s<HwDev>(x); }
__device__ __forceinline__ double qtan(double x) { return m64::tan_s<HwDev>(x); }
__device__ __forceinline__ double qatan2(double y, double x) { return m64::atan2_s<HwDev>(y, x); }
__device__ __forceinline__ void qsincos(double x, double *s, double *c) { m64::sincos_s(x, s, c); }   // a wrapped or clamped angle
__device__ __forceinline__ void qsincos_any(double x, double *s, double *c) {
    if (fabs(x) < 1e5) m64::sincos_s(x, s, c);
    else sincos(x, s, c);                                     // (an unwrapped yaw after 16 000 turns)
}

// utils.py:124-139 / 167-182 with the quotient th / 2 pi as a product: the same integer wherever th is not within an ulp of a
// multiple of 2 pi, and there both wrap to the same angle
__device__ __forceinline__ double limit_angle_m(double th) {
    th = fma(floor(th * 0.15915494309189533577), -2 * PI, th);
    if (th > PI) th -= 2 * PI;
    else if (th < -PI) th += 2 * PI;
    return th;
}
__device__ __forceinline__ double angle_diff_m(double a1, double a2) {
    double da = fabs(a1 - a2);
    if (da > PI) da = 2 * PI - da;
    double t1 = fabs(limit_angle_m(a1 - da) - a2), t2 = fabs(limit_angle_m(a1 + da) - a2);
    return t1 < t2 ? -da : da;
}

// Registers of one agent while it is being ticked.
struct Agent {
    int64_t a;
    double x, y, psi, v, delta, theta;
    double vdes;
    int64_t qb;   // first row of the destination queue
    int32_t K;    // rows
    int32_t ptr;
    int32_t zn;   // 0 cruise, 1 brake, 2 arrived
    double zv0, zd0, zd1;
    int32_t ti;
    uint32_t st;
    double cpsi, spsi;  // cos / sin of psi when the integrator has just computed them (cs_fresh), for the fp32 record
    bool cs_fresh;
    // The destination rows ptr .. ptr + 3 (beyond the queue's end: its last row) and the ring's last two positions, asked for
    // in ONE batch as soon as queue pointer and ring column are known (load_rows, load_ring): every later look at a row is a
    // register, where the reference's call tree (updateDestination -> updateNavState -> planner -> control) would be five
    // dependent round trips.  The rows are asked for again when the pointer moves - once per leg of a route.
    double rx[4], ry[4], rs[4];
    double h0x, h0y, h1x, h1y;   // ring[ti], ring[ti - 1]
    // InvPendulum / PlanarPoint / PlanarBicycle side-state, asked for with the agent's own scalars (agent_kernel): the LTI state
    // (vehicle.py:1728), the riding flag, the run of small steering angles; the unwrapped yaw (dynamics.py:943-966)
    double xl[5];
    int32_t dgood;
    bool riding;
    double ppsi;
    // the parameter set of this road user (vehicle.py:64-204: every vehicle owns one): the engine's only one in the kernel
    // arguments, or its row of the class table; pb: the PlanarBicycle step matrices that belong to it (Dev::pb)
    const csf_params *p;
    const double *pb;
};

// HET: the population holds more than one parameter set (csf_set_param_classes)
template <bool HET>
__device__ __forceinline__ void agent_params(const Dev &d, int64_t a, Agent &g) {
    if (HET) {
        const int c = d.cls[a];
        g.p = d.ptab + c;
        g.pb = d.pbtab + 7 * c;
    } else {
        g.p = &d.p;
        g.pb = d.pb;
    }
}

// the rows ptr .. ptr + 3 of the destination queue (vehicle.py:183-185, 606-647) into the registers
__device__ __forceinline__ void load_rows(const Dev &d, Agent &g) {
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int64_t r = 3 * (g.qb + max(min(g.ptr + k, g.K - 1), 0));
        g.rx[k] = d.q[r];
        g.ry[k] = d.q[r + 1];
        g.rs[k] = d.q[r + 2];
    }
}

// the ring's last two positions (vehicle.py:1474-1491 reads them from traj)
__device__ __forceinline__ void load_ring(const Dev &d, Agent &g) {
    const int hm = d.hist_len - 1;
    g.h1x = d.hx[(int64_t)((g.ti - 1) & hm) * d.cap + g.a], g.h1y = d.hy[(int64_t)((g.ti - 1) & hm) * d.cap + g.a];
    g.h0x = d.hx[(int64_t)(g.ti & hm) * d.cap + g.a], g.h0y = d.hy[(int64_t)(g.ti & hm) * d.cap + g.a];
}

__device__ __forceinline__ bool qstop(const Agent &g) { return g.rs[0] != 0.0; }   // stop flag of the current destination

// vehicle.py:596-604
__device__ __forceinline__ double dest_dist(const Agent &g) {
    double ex = g.rx[0] - g.x, ey = g.ry[0] - g.y;
    return qsqrt(ex * ex + ey * ey);
}

// vehicle.py:545-594
__device__ __forceinline__ void update_destination(const Dev &d, Agent &g) {
    if (g.zn != 0) return;                                    // :567-568
    double dnext = dest_dist(g);
    int adv = 0;                                              // rows the pointer moves: 0, 1 or 2
    if (dnext <= g.p->d_arrived_inter) adv = min(g.ptr + 1, g.K - 1) - g.ptr;   // :571-574
    if (g.ptr + adv < g.K - 1) {                              // :577-583
        double ex = (adv ? g.rx[2] : g.rx[1]) - g.x, ey = (adv ? g.ry[2] : g.ry[1]) - g.y;
        if (qsqrt(ex * ex + ey * ey) < dnext) adv += 1;
    }
    if (adv) {                                                // (once per leg of a road user's route)
        g.ptr += adv;
        load_rows(d, g);
    }
}

// vehicle.py:354-457.  Returns the desired speed; ddest through the reference argument.
__device__ __forceinline__ double update_nav(const Dev &d, Agent &g, double &ddest) {
    const csf_params &p = *g.p;
    const double k = 1.5;                                     // :377
    double d0, d1;
    if (g.zn == 0) {                                          // :379-386
        d0 = qdiv(0.5 * (p.v_max_harddecel * p.v_max_harddecel - g.v * g.v), p.a_desired_default[0]);
        d1 = qdiv(0.5 * -(p.v_max_harddecel * p.v_max_harddecel), p.a_max[0]);
    } else {
        d0 = g.zd0;
        d1 = g.zd1;
    }
    ddest = dest_dist(g);
    bool x0 = qstop(g);                                       // :397-400
    bool x1 = ddest <= k * (d0 + d1), x2 = ddest <= p.d_arrived_stop, x3 = g.v <= p.v_max_stop;
    bool z0 = g.zn == 0, z1 = g.zn == 1, z2 = g.zn == 2;
    bool n0 = !x0 || (x0 && !x1 && ((z0 && !x2) || z1));      // :404-406
    bool n1 = x0 && ((z0 && ((!x2 && x1) || (x2 && !x3))) || (z1 && x1 && (!x2 || !x3)));
    bool n2 = x0 && (((z0 || z1) && x2 && x3) || z2);         // :414
    if ((int)n0 + (int)n1 + (int)n2 != 1) g.st |= CSF_ST_NAVSTATE;  // reference only prints (:416-425)
    if (z0 && n1) {                                           // :428-430
        g.zv0 = g.v;
        g.zd0 = d0;
        g.zd1 = d1;
    }
    g.zn = n0 ? 0 : (n1 ? 1 : 2);
    if (n0) return g.vdes;                                    // :434-435
    if (n1) {                                                 // :436-450
        if (ddest < k * g.zd1) return qdiv(qdiv(p.v_max_harddecel, g.zd1) * ddest, k);
        return (g.zv0 - p.v_max_harddecel) / g.zd0 * (ddest - g.zd1) * 1 / k + p.v_max_harddecel;
    }
    return 0.0;                                               // :452-453
}

// vehicle.py:1150-1194 / 2078-2108
__device__ __forceinline__ void direct_approach(const Dev &d, Agent &g, double &fx, double &fy) {
    update_destination(d, g);
    double ddest, vd = update_nav(d, g, ddest);
    if (ddest > 0) {
        fx = qdiv(-vd * (g.x - g.rx[0]), ddest);
        fy = qdiv(-vd * (g.y - g.ry[0]), ddest);
    } else {
        fx = 0;
        fy = 0;
    }
}

// ---- cubic B-spline through M in {4,5,6} points (scipy splprep(s=0) + splev, vehicle.py:1496-1510) ----
// Chord-length parameter u, clamped knots with interior knots u[2..M-3] (FITPACK's rule for s = 0).  The end
// conditions make the first and last coefficient equal the end points, so only an (M-2)x(M-2) totally positive
// system remains; it is eliminated without pivoting.  Everything is templated on M and unrolled so that knots
// and coefficients live in registers: a span is chosen with selects, never with an indexed load (indexed
// per-lane arrays go to scratch memory, which made this kernel latency-bound).
template <int M>
struct Spline {
    double t[M + 4];
    double cx[M], cy[M];
};

// the six knots t[l-2..l+3] and four coefficients c[l-3..l] of the span l = 3 + si that contains u
struct Window {
    double k[6];
    double x[4], y[4];
};

template <int M, bool KREG>
__device__ __forceinline__ void window_at(const Spline<M> &s, double u, Window &w) {
    int si = 0;  // interior knots <= u; u = 1 falls into the last span (fpbspl / splev convention)
#pragma unroll
    for (int j = 0; j < M - 4; j++) si += (u >= s.t[4 + j]) ? 1 : 0;
    // The knots as opaque register values: left as members of `s`, the compiler turns the chain of selects below into ONE
    // load from a selected address - and keeps the knot vector in scratch memory for it (184 bytes per lane in every kernel
    // that holds this planner, a store and a dependent load of ~1 us on the destination force's chain).  (The allocation
    // itself costs a launch nothing - tools/graph_gap_ubench.hip, mode 8 - the traffic does.)  KREG: where the planner is on a
    // tick's critical path - the one-launch tick and the one-wave kernel: config 2 13.2 -> 13.0 us; the per-agent kernel of large
    // populations is one wave per CU with other latencies to hide behind and ran 0.3 us FASTER with the load (8.4 against 8.7).
    if constexpr (KREG) {
        double tk[M + 4];
#pragma unroll
        for (int j = 0; j < M + 4; j++) {
            tk[j] = s.t[j];
            asm volatile("" : "+v"(tk[j]));
        }
#pragma unroll
        for (int q = 0; q < 6; q++) {
            double v = tk[1 + q];
#pragma unroll
            for (int c = 1; c <= M - 4; c++) v = (si == c) ? tk[1 + q + c] : v;
            w.k[q] = v;
        }
    } else {
#pragma unroll
        for (int q = 0; q < 6; q++) {
            double v = s.t[1 + q];
#pragma unroll
            for (int c = 1; c <= M - 4; c++) v = (si == c) ? s.t[1 + q + c] : v;
            w.k[q] = v;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
        double vx = s.cx[q], vy = s.cy[q];
#pragma unroll
        for (int c = 1; c <= M - 4; c++) {
            vx = (si == c) ? s.cx[q + c] : vx;
            vy = (si == c) ? s.cy[q + c] : vy;
        }
        w.x[q] = vx;
        w.y[q] = vy;
    }
}

// non-zero cubic basis values N3[0..3] at u for the window's span (Cox - de Boor), plus the quadratic and
// linear ones that the derivatives use.  k[2] = t[l], k[3] = t[l+1].
__device__ __forceinline__ void basis(const double (&k)[6], double u, double N3[4], double N2[3], double N1[2]) {
    const double a1 = u - k[2], b1 = k[3] - u;
    const double w = rcp_nr(k[3] - k[2]);
    N1[0] = b1 * w;
    N1[1] = a1 * w;
    const double a2 = u - k[1], b2 = k[4] - u;
    const double w0 = N1[0] * rcp_nr(k[3] - k[1]), w1 = N1[1] * rcp_nr(k[4] - k[2]);
    N2[0] = b1 * w0;
    N2[1] = a2 * w0 + b2 * w1;
    N2[2] = a1 * w1;
    const double a3 = u - k[0], b3 = k[5] - u;
    const double v0 = N2[0] * rcp_nr(k[3] - k[0]), v1 = N2[1] * rcp_nr(k[4] - k[1]), v2 = N2[2] * rcp_nr(k[5] - k[2]);
    N3[0] = b1 * v0;
    N3[1] = a3 * v0 + b2 * v1;
    N3[2] = a2 * v1 + b3 * v2;
    N3[3] = a1 * v2;
}

template <int M>
__device__ __forceinline__ bool spline_fit(Spline<M> &s, const double (&px)[M], const double (&py)[M]) {
    double u[M];
    u[0] = 0;
    bool ok = true;
#pragma unroll
    for (int r = 1; r < M; r++) {
        const double ex = px[r] - px[r - 1], ey = py[r] - py[r - 1];
        const double dd = qsqrt(ex * ex + ey * ey);
        ok = ok && (dd > 0.0);  // splprep raises ValueError on duplicate consecutive points
        u[r] = u[r - 1] + dd;
    }
    if (!ok) return false;
    const double itot = rcp_nr(u[M - 1]);
#pragma unroll
    for (int r = 1; r < M - 1; r++) u[r] *= itot;
    u[M - 1] = 1.0;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        s.t[j] = 0.0;
        s.t[M + j] = 1.0;
    }
#pragma unroll
    for (int j = 0; j < M - 4; j++) s.t[4 + j] = u[2 + j];
    // collocation rows of the interior points u_1 .. u_{M-2}; the span of u_r is known statically:
    // u_1 is in the first span, u_r (2 <= r <= M-3) is the interior knot t[r+2], u_{M-2} is in the last span
    constexpr int Q = M - 2;
    double A[Q][Q], bx[Q], by[Q];
#pragma unroll
    for (int r = 0; r < Q; r++) {
#pragma unroll
        for (int c = 0; c < Q; c++) A[r][c] = 0.0;
        const int si = (r == 0) ? 0 : ((r < M - 4) ? r : M - 4);  // row r is point r+1
        double k[6];
#pragma unroll
        for (int q = 0; q < 6; q++) k[q] = s.t[1 + si + q];
        double N3[4], N2[3], N1[2];
        basis(k, u[r + 1], N3, N2, N1);
        double rx = px[r + 1], ry = py[r + 1];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int c = si + j;  // coefficient index
            if (c == 0) {
                rx -= N3[j] * px[0];
                ry -= N3[j] * py[0];
            } else if (c == M - 1) {
                rx -= N3[j] * px[M - 1];
                ry -= N3[j] * py[M - 1];
            } else {
                A[r][c - 1] = N3[j];
            }
        }
        bx[r] = rx;
        by[r] = ry;
    }
    double ipv[Q];
#pragma unroll
    for (int c = 0; c < Q; c++) {
        const double ip = rcp_nr(A[c][c]);
        ipv[c] = ip;
#pragma unroll
        for (int r = c + 1; r < Q; r++) {
            const double f = A[r][c] * ip;
#pragma unroll
            for (int j = c; j < Q; j++) A[r][j] -= f * A[c][j];
            bx[r] -= f * bx[c];
            by[r] -= f * by[c];
        }
    }
#pragma unroll
    for (int r = Q - 1; r >= 0; r--) {
        double sx = bx[r], sy = by[r];
#pragma unroll
        for (int j = r + 1; j < Q; j++) {
            sx -= A[r][j] * bx[j];
            sy -= A[r][j] * by[j];
        }
        bx[r] = sx * ipv[r];
        by[r] = sy * ipv[r];
    }
    s.cx[0] = px[0];
    s.cy[0] = py[0];
#pragma unroll
    for (int r = 0; r < Q; r++) {
        s.cx[r + 1] = bx[r];
        s.cy[r + 1] = by[r];
    }
    s.cx[M - 1] = px[M - 1];
    s.cy[M - 1] = py[M - 1];
    return true;
}

template <int M, bool KREG>
__device__ __forceinline__ void spline_pos(const Spline<M> &s, double u, double &X, double &Y) {
    Window w;
    window_at<M, KREG>(s, u, w);
    double N3[4], N2[3], N1[2];
    basis(w.k, u, N3, N2, N1);
    X = N3[0] * w.x[0] + N3[1] * w.x[1] + N3[2] * w.x[2] + N3[3] * w.x[3];
    Y = N3[0] * w.y[0] + N3[1] * w.y[1] + N3[2] * w.y[2] + N3[3] * w.y[3];
}

// position, first and second derivative at u (splev with der = 0, 1, 2)
template <int M, bool KREG>
__device__ __forceinline__ void spline_all(const Spline<M> &s, double u, double &X, double &Y, double &dX,
                                           double &dY, double &ddX, double &ddY) {
    Window w;
    window_at<M, KREG>(s, u, w);
    double N3[4], N2[3], N1[2];
    basis(w.k, u, N3, N2, N1);
    X = N3[0] * w.x[0] + N3[1] * w.x[1] + N3[2] * w.x[2] + N3[3] * w.x[3];
    Y = N3[0] * w.y[0] + N3[1] * w.y[1] + N3[2] * w.y[2] + N3[3] * w.y[3];
    double ex[3], ey[3];
#pragma unroll
    for (int q = 0; q < 3; q++) {  // c'_j = 3 (c_{j+1} - c_j) / (t_{j+4} - t_{j+1})
        const double f = 3.0 * rcp_nr(w.k[q + 3] - w.k[q]);
        ex[q] = f * (w.x[q + 1] - w.x[q]);
        ey[q] = f * (w.y[q + 1] - w.y[q]);
    }
    dX = N2[0] * ex[0] + N2[1] * ex[1] + N2[2] * ex[2];
    dY = N2[0] * ey[0] + N2[1] * ey[1] + N2[2] * ey[2];
    double gx[2], gy[2];
#pragma unroll
    for (int q = 0; q < 2; q++) {
        const double f = 2.0 * rcp_nr(w.k[q + 3] - w.k[q + 1]);
        gx[q] = f * (ex[q + 1] - ex[q]);
        gy[q] = f * (ey[q + 1] - ey[q]);
    }
    ddX = N1[0] * gx[0] + N1[1] * gx[1];
    ddY = N1[0] * gy[0] + N1[1] * gy[1];
}

// vehicle.py:1494-1558 once the M control points are known
template <int M, bool KREG>
__device__ __forceinline__ void spline_force(const Dev &d, Agent &g, const double (&px)[M], const double (&py)[M], bool last,
                             double vd, double &fx, double &fy) {
    const int nS = 20, ipred = 3, ipredlast = 5;              // :1446-1448
    Spline<M> s;
    if (!spline_fit(s, px, py)) {                             // :1495-1507 raises in the reference
        g.st |= CSF_ST_SPLINE;
        fx = 0;
        fy = 0;
        return;
    }
    int i = 1;                                                // :1516-1522
    if (last) {
        double best = INFINITY;
        for (int k = 0; k < nS; k++) {
            double X, Y;
            spline_pos<M, KREG>(s, k == nS - 1 ? 1.0 : (double)k / (nS - 1), X, Y);
            const double dd = (X - g.x) * (X - g.x) + (Y - g.y) * (Y - g.y);
            if (dd < best) {
                best = dd;
                i = k;
            }
        }
    }
    const int iprev = i + (qstop(g) ? ipredlast : ipred);     // :1523-1526
    if (iprev < nS) {                                         // :1529-1553
        const double ui = i == nS - 1 ? 1.0 : (double)i / (nS - 1);
        const double up = iprev == nS - 1 ? 1.0 : (double)iprev / (nS - 1);
        double X0, Y0, X1, Y1, dX, dY, ddX, ddY;
        spline_all<M, KREG>(s, ui, X0, Y0, dX, dY, ddX, ddY);
        spline_pos<M, KREG>(s, up, X1, Y1);
        const double sp = qsqrt(dX * dX + dY * dY);
        const double R = sp * sp * sp / fabs(dX * ddY - dY * ddX);  // :1532-1537
        const double thetacomf = 10 * (2 * PI / 360);         // :1541
        double v = fmax(2.5, sqrt(thetacomf * g.p->g * R));    // :1542-1544
        v = fmin(v, vd);                                      // :1545
        const double ex = X1 - X0, ey = Y1 - Y0;
        const double tmp = v * qrsqrt(ex * ex + ey * ey);     // :1548-1553
        fx = tmp * ex;
        fy = tmp * ey;
    } else {
        direct_approach(d, g, fx, fy);                        // :1555-1556 (second queue + nav update)
    }
}

// vehicle.py:1416-1558
template <bool KREG>
__device__ __forceinline__ void twod_dest(const Dev &d, Agent &g, double &fx, double &fy) {
    update_destination(d, g);                                 // :1451
    double ddest, vd = update_nav(d, g, ddest);               // :1452
    if (g.ti == 0) {                                          // :1455-1458
        double sp, cp;
        qsincos(g.psi, &sp, &cp);
        fx = vd * cp;
        fy = vd * sp;
        return;
    }
    if (g.zn == 2) {                                          // :1461-1462
        fx = 0;
        fy = 0;
        return;
    }
    const double h1x = g.h1x, h1y = g.h1y, h0x = g.h0x, h0y = g.h0y;
    const bool last = g.ptr + 1 >= g.K;                       // :537-543
    if (!last) {                                              // :1465-1479: two trajectory points + up to 4 destinations
        const int cnt = min(g.ptr + 4, g.K) - g.ptr;          // >= 2
        if (cnt == 2) {
            const double px[4] = {h1x, h0x, g.rx[0], g.rx[1]};
            const double py[4] = {h1y, h0y, g.ry[0], g.ry[1]};
            spline_force<4, KREG>(d, g, px, py, false, vd, fx, fy);
        } else if (cnt == 3) {
            const double px[5] = {h1x, h0x, g.rx[0], g.rx[1], g.rx[2]};
            const double py[5] = {h1y, h0y, g.ry[0], g.ry[1], g.ry[2]};
            spline_force<5, KREG>(d, g, px, py, false, vd, fx, fy);
        } else {
            const double px[6] = {h1x, h0x, g.rx[0], g.rx[1], g.rx[2], g.rx[3]};
            const double py[6] = {h1y, h0y, g.ry[0], g.ry[1], g.ry[2], g.ry[3]};
            spline_force<6, KREG>(d, g, px, py, false, vd, fx, fy);
        }
    } else {                                                  // :1486-1492: last leg, three trajectory points
        const int hm = d.hist_len - 1;
        const int back = max(0, g.ti - d.back);
        const double px[4] = {d.hx[(int64_t)(back & hm) * d.cap + g.a], h1x, h0x, g.rx[0]};
        const double py[4] = {d.hy[(int64_t)(back & hm) * d.cap + g.a], h1y, h0y, g.ry[0]};
        spline_force<4, KREG>(d, g, px, py, true, vd, fx, fy);
    }
}

template <int MODEL, bool KREG>
__device__ __forceinline__ void dest_force(const Dev &d, Agent &g, double &fx, double &fy) {
    if (MODEL == CSF_UNCONTROLLED) fx = 0, fy = 0;            // vehicle.py:987-988
    else if (MODEL == CSF_BICYCLE) direct_approach(d, g, fx, fy);  // vehicle.py:1189-1194
    else if (MODEL == CSF_BALANCINGRIDER) {                   // vehicle.py:295-297, then calc_direct_approach_dest_force (:1987-1988, 2078-2108)
        update_destination(d, g);
        direct_approach(d, g, fx, fy);
    } else {
        if (MODEL == CSF_PLANARPOINT || MODEL == CSF_PLANARBIKE) update_destination(d, g);  // Vehicle.calcDestinationForce :295-297
        twod_dest<KREG>(d, g, fx, fy);
    }
}

// vehicle.py:1218-1272 (Bicycle.control + Bicycle.move; PIDcontroller with ki = kd = 0, dynamics.py:33-54)
__device__ __forceinline__ void bike_control_move(const Dev &d, Agent &g, double Fx, double Fy) {
    const csf_params &p = *g.p;
    double theta = qatan2(Fy, Fx);                            // :1223
    double vd = qsqrt(Fx * Fx + Fy * Fy);                     // :1224
    double ddest = dest_dist(g);                              // :1226-1229
    if (ddest < 3 && g.ptr + 1 >= g.K) vd = (vd * (1.0 / 3)) * ddest; // :1231-1232
    double target = angle_diff_m(g.psi, theta);               // :1235
    double om = p.k_p_delta * angle_diff_m(g.delta, target);  // :1239-1242
    double acc = p.k_p_v * (vd - g.v);                        // :1240-1243
    acc = clampd(acc, p.a_max[0], p.a_max[1]);                // :1249
    double delta = limit_angle_m(g.delta + p.t_s * om);       // :1254
    double v = g.v + p.t_s * acc;                             // :1255
    delta = clampd(delta, -p.delta_max, p.delta_max);         // :1257
    v = clampd(v, p.v_max_riding[0], p.v_max_riding[1]);      // :1258
    double sd, cd;                                            // tan(delta) / l as one quotient
    qsincos(delta, &sd, &cd);
    double psi = limit_angle_m(g.psi + qdiv(p.t_s * v * sd, cd * p.l));  // :1260-1262
    qsincos(psi, &g.spsi, &g.cpsi);                           // once: the next tick's record needs the same two
    g.cs_fresh = true;
    g.y += p.t_s * v * g.spsi;                                // :1264
    g.x += p.t_s * v * g.cpsi;                                // :1265
    g.psi = psi;
    g.v = v;
    g.delta = delta;
}

// ---- InvPendulum: exact zero-order-hold step of the speed-dependent closed loop -------------------------
// vehicle.py:1738-1786, 1810-1848; parameters.py:1832-1892.  control.forced_response over [0, t_s] with
// constant input equals x+ = E11 x + E12 u with E = exp([[A h, B h],[0, 0]]).  E is formed by scaling and
// squaring of a degree-12 Taylor polynomial (||M/2^s||_1 <= 1/2); only the 5 non-trivial rows are carried.
__device__ __forceinline__ void invpend_step_yaw(const Dev &d, Agent &g, double *xl, double Fx, double Fy) {
    const csf_params &p = *g.p;
    const double v = g.v;
    const double iv = rcp_nr(v), iv2 = iv * iv, iv3 = iv2 * iv;
    const double kx0 = 3.48203226e02 - 5.12057324e03 * iv + 1.58364873e04 * iv2 - 1.98073306e04 * iv3;
    const double kx1 = -4.51700000e01;
    const double kx2 = -9.16379250e02 + 1.31769807e04 * iv - 6.57341643e04 * iv2 + 8.22163589e04 * iv3;
    const double kx3 = 3.20214069e02 - 4.69953797e03 * iv + 1.66378680e04 * iv2 - 2.43114309e04 * iv3;
    const double kx4 = 2.87549256e-08 - 2.27913445e03 * iv;
    const double ku = -3.38638984e-09 - 2.27913445e+03 * iv;
    const double igl = rcp_nr(p.g * p.l);
    const double Ktau2 = (v * p.l_2) * igl, K = (v * v) * igl;
    const double itau1sq = qdiv(p.m * p.g * p.h, p.i_bike_longlong + p.m * p.h * p.h);
    const double bI = rcp_nr(p.i_steer_vertvert), h = p.t_s;
    double M[5][6];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) M[r][c] = 0.0;
    M[0][1] = h;
    M[1][0] = -bI * kx0 * h;
    M[1][1] = (-p.c_steer * bI - bI * kx1) * h;
    M[1][2] = -bI * kx2 * h;
    M[1][3] = -bI * kx3 * h;
    M[1][4] = -bI * kx4 * h;
    M[1][5] = ku * bI * h;
    M[2][3] = h;
    M[3][0] = -K * itau1sq * h;
    M[3][1] = -Ktau2 * itau1sq * h;
    M[3][2] = itau1sq * h;
    M[4][0] = v * rcp_nr(p.l) * h;
    // Balancing: the steer-rate row holds entries of 20 .. 70 (gains over the steering inertia), its column 0.01 .. 0.03, and
    // the 1-norm of M - the number of squarings below - follows the row.  With x1 scaled by a power of two (exact: M' =
    // D^-1 M D, exp(M) = D exp(M') D^-1, D = diag(1, dd, 1, 1, 1, 1)) row and column meet in the middle: three or four
    // squarings instead of seven.  The state goes through as x' = D^-1 x and comes back as D x'.
    double dd = 1.0, idd = 1.0;
    {
        const double r1 = fabs(M[1][0]) + fabs(M[1][2]) + fabs(M[1][3]) + fabs(M[1][4]) + fabs(M[1][5]);
        const double c1 = fabs(M[0][1]) + fabs(M[3][1]);
        if (r1 > 0.0 && c1 > 0.0 && isfinite(r1) && isfinite(c1)) {
            const int eb = max(-20, min(20, (ilogb(r1) - ilogb(c1)) / 2));
            dd = ldexp(1.0, eb);
            idd = ldexp(1.0, -eb);
        }
        M[1][0] *= idd, M[1][2] *= idd, M[1][3] *= idd, M[1][4] *= idd, M[1][5] *= idd;
        M[0][1] *= dd, M[3][1] *= dd;
    }
    double nrm = 0;
#pragma unroll
    for (int c = 0; c < 6; c++) {
        double col = 0;
#pragma unroll
        for (int r = 0; r < 5; r++) col += fabs(M[r][c]);
        nrm = fmax(nrm, col);
    }
    int sq = 0;
    while (nrm > 0.5 && sq < 40) {
        nrm *= 0.5;
        sq++;
    }
    const double sc = ldexp(1.0, -sq);
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) M[r][c] *= sc;
    // Horner: E = I + M (I + M/2 (I + M/3 (... (I + M/12))))   — rows 0..4; row 5 of every term is e5
    double E[5][6];
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 6; c++) E[r][c] = (r == c ? 1.0 : 0.0) + M[r][c] * (1.0 / 12);
    // M has 12 non-zero entries (rows 0, 2, 4 one each, row 3 three, row 1 six): the products M E are written out
    const double m01 = M[0][1], m23 = M[2][3], m40 = M[4][0], m30 = M[3][0], m31 = M[3][1], m32 = M[3][2];
#pragma unroll
    for (int k = 11; k >= 1; k--) {
        double T[5][6];
        const double ik = 1.0 / k;                            // (unrolled: a constant)
#pragma unroll
        for (int c = 0; c < 6; c++) {
            double r1 = (c == 5) ? M[1][5] : 0.0;  // M[:,5] * E[5][c], E row 5 = e5
#pragma unroll
            for (int q = 0; q < 5; q++) r1 += M[1][q] * E[q][c];
            T[0][c] = (c == 0 ? 1.0 : 0.0) + (m01 * E[1][c]) * ik;
            T[1][c] = (c == 1 ? 1.0 : 0.0) + r1 * ik;
            T[2][c] = (c == 2 ? 1.0 : 0.0) + (m23 * E[3][c]) * ik;
            T[3][c] = (c == 3 ? 1.0 : 0.0) + (m30 * E[0][c] + m31 * E[1][c] + m32 * E[2][c]) * ik;
            T[4][c] = (c == 4 ? 1.0 : 0.0) + (m40 * E[0][c]) * ik;
        }
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) E[r][c] = T[r][c];
    }
    for (int it = 0; it < sq; it++) {
        double T[5][6];
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) {
                double acc = (c == 5) ? E[r][5] : 0.0;
#pragma unroll
                for (int q = 0; q < 5; q++) acc += E[r][q] * E[q][c];
                T[r][c] = acc;
            }
#pragma unroll
        for (int r = 0; r < 5; r++)
#pragma unroll
            for (int c = 0; c < 6; c++) E[r][c] = T[r][c];
    }
    const double psi_d = qatan2(Fy, Fx);                      // :1832
    double xn[5];
    xl[1] *= idd;                                             // x' = D^-1 x (the balancing above)
#pragma unroll
    for (int r = 0; r < 5; r++) {
        double acc = E[r][5] * psi_d;
#pragma unroll
        for (int c = 0; c < 5; c++) acc += E[r][c] * xl[c];
        xn[r] = acc;
    }
    xn[1] *= dd;
#pragma unroll
    for (int r = 0; r < 5; r++) xl[r] = xn[r];                // :1843
    g.psi = limit_angle_m(xn[4]);                             // :1844
    g.delta = limit_angle_m(xn[0]);                           // :1845
    g.theta = limit_angle_m(xn[2]);                           // :1846
}

// ---- BalancingRiderBicycle: the Whipple-Carvallo bicycle under full-state feedback (dynamics.py:261-705) -----------------
// x = (roll, steer, roll rate, steer rate, yaw) in the bike model's frame (y to the right, z down: steer, yaw and y mirrored
// against the engine's, :318-371), unwrapped, in g.xl; the speed the gains in use belong to in g.ppsi.
// the state matrix at speed v (:535-560) and the steer-torque input column
__device__ __forceinline__ void br_state_space(const csf_params &p, double v, double (&A)[5][5], double (&B)[5]) {
#pragma unroll
    for (int r = 0; r < 5; r++)
#pragma unroll
        for (int c = 0; c < 5; c++) A[r][c] = 0.0;
    A[0][2] = 1.0;
    A[1][3] = 1.0;
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 2; c++) {
            A[2 + r][c] = -(p.br_minv_k0g[2 * r + c] + v * v * p.br_minv_k2[2 * r + c]);
            A[2 + r][2 + c] = -v * p.br_minv_c1[2 * r + c];
        }
    A[4][1] = p.br_yaw[0] * v;                                // :549
    A[4][3] = p.br_yaw[1];                                    // :550
    B[0] = B[1] = B[4] = 0.0;
    B[2] = p.br_minv_steer[0];
    B[3] = p.br_minv_steer[1];
}

// M x = b for five unknowns, elimination with partial pivoting on registers (rows swapped by selects: an indexed per-lane array
// would go to scratch memory); false: singular
__device__ __forceinline__ bool solve5(double (&M)[5][5], double (&b)[5]) {
    bool ok = true;
#pragma unroll
    for (int k = 0; k < 5; k++) {
#pragma unroll
        for (int r = k + 1; r < 5; r++) {                     // bring the larger of rows k and r to k
            const bool sw = fabs(M[r][k]) > fabs(M[k][k]);
#pragma unroll
            for (int c = k; c < 5; c++) {
                const double t = M[k][c];
                M[k][c] = sw ? M[r][c] : t;
                M[r][c] = sw ? t : M[r][c];
            }
            const double t = b[k];
            b[k] = sw ? b[r] : t;
            b[r] = sw ? t : b[r];
        }
        ok = ok && M[k][k] != 0.0;
        const double ip = 1.0 / M[k][k];
#pragma unroll
        for (int r = k + 1; r < 5; r++) {
            const double f = M[r][k] * ip;
#pragma unroll
            for (int c = k; c < 5; c++) M[r][c] -= f * M[k][c];
            b[r] -= f * b[k];
        }
    }
#pragma unroll
    for (int k = 4; k >= 0; k--) {
        double acc = b[k];
#pragma unroll
        for (int c = k + 1; c < 5; c++) acc -= M[k][c] * b[c];
        b[k] = acc / M[k][k];
    }
    return ok;
}

// dynamics.py:600-615: the gains that put the closed loop's poles where the control model wants them at speed v - one input, so
// the placement has one solution (the reference calls control.place): Ackermann's K = e_5^T W^-1 p(A), W = (B, A B, ..., A^4 B),
// p the polynomial of the desired poles, applied to the row e_5^T W^-1 from the left as (A - p0) and (A^2 - 2 Re p A + |p|^2)
__device__ __forceinline__ bool br_gains(const csf_params &p, double v, double (&K)[5]) {
    if (p.br_mode == 2) {                                     // :604-605
#pragma unroll
        for (int k = 0; k < 5; k++) K[k] = p.br_gains[k];
        return true;
    }
    double f[5];
#pragma unroll
    for (int i = 0; i < 5; i++) f[i] = p.br_pole_fun[2 * i] + p.br_pole_fun[2 * i + 1] * v;   // parameters.py:1400-1409
    double A[5][5], B[5], W[5][5], col[5], y[5];
    br_state_space(p, v, A, B);
#pragma unroll
    for (int r = 0; r < 5; r++) col[r] = B[r];
#pragma unroll
    for (int k = 0; k < 5; k++) {                             // W^T: row k = (A^k B)^T
        double nxt[5];
#pragma unroll
        for (int r = 0; r < 5; r++) {
            W[k][r] = col[r];
            double acc = 0.0;
#pragma unroll
            for (int c = 0; c < 5; c++) acc += A[r][c] * col[c];
            nxt[r] = acc;
        }
#pragma unroll
        for (int r = 0; r < 5; r++) col[r] = nxt[r];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) y[k] = k == 4 ? 1.0 : 0.0;    // W^T y = e_5
    const bool ok = solve5(W, y);
    auto row_times_A = [&](const double (&in)[5], double (&out)[5]) {
#pragma unroll
        for (int c = 0; c < 5; c++) {
            double acc = 0.0;
#pragma unroll
            for (int r = 0; r < 5; r++) acc += in[r] * A[r][c];
            out[c] = acc;
        }
    };
    double t1[5], t2[5];
    row_times_A(y, t1);
#pragma unroll
    for (int c = 0; c < 5; c++) y[c] = t1[c] - f[0] * y[c];
#pragma unroll
    for (int pair = 0; pair < 2; pair++) {
        const double re = f[1 + 2 * pair], im = f[2 + 2 * pair];
        row_times_A(y, t1);
        row_times_A(t1, t2);
#pragma unroll
        for (int c = 0; c < 5; c++) y[c] = t2[c] - 2.0 * re * t1[c] + (re * re + im * im) * y[c];
    }
#pragma unroll
    for (int k = 0; k < 5; k++) K[k] = y[k];
    return ok;
}

// BalancingRiderDynamics.step (dynamics.py:664-705): speed, gains (renewed only when the speed has changed, :671-673), commanded
// yaw, and the implicit midpoint rule (:497-513) on x' = (A - B K) x + B K_psi psi_c - a linear system in x+ (the reference
// hands the seven equations to MINPACK); the position follows from the yaw at both ends of the step.  dr[2]: steer rate, roll rate
// in the engine's frame.
__device__ __forceinline__ void balancingrider_step(const Dev &d, Agent &g, double *xl, double Fx, double Fy, double (&dr)[2]) {
    const csf_params &p = *g.p;
    const double h = p.t_s, v_old = g.v;
    const double vd = qsqrt(Fx * Fx + Fy * Fy);                                   // :640
    const double acc = clampd(p.k_p_v * (vd - v_old), p.a_max[0], p.a_max[1]);   // :643-644
    const double v = clampd(v_old + h * acc, p.v_max_riding[0], p.v_max_riding[1]);   // :647
    if (v != v_old) g.ppsi = 0.5 * (v + v_old);                                   // :671-673
    double K[5];
    bool ok = br_gains(p, g.ppsi, K);
    const double psi_F = limit_angle(atan2(-Fy, Fx));                             // :656-658 (the lateral force mirrored)
    const double psi_c = xl[4] + angle_diff(xl[4], psi_F);                        // :660-663
    const double vbar = 0.5 * (v + v_old);                                        // :684, 687
    double A[5][5], B[5], L[5][5], rhs[5];
    br_state_space(p, vbar, A, B);
#pragma unroll
    for (int r = 0; r < 5; r++) {
        double accr = xl[r] + h * B[r] * K[4] * psi_c;
#pragma unroll
        for (int c = 0; c < 5; c++) {
            const double acl = A[r][c] - B[r] * K[c];                             // :472
            L[r][c] = (r == c ? 1.0 : 0.0) - 0.5 * h * acl;
            accr += 0.5 * h * acl * xl[c];
        }
        rhs[r] = accr;
    }
    ok = solve5(L, rhs) && ok;
    if (!ok) {
        g.st |= CSF_ST_NAN;                                   // (the reference: "System not controllable!" / a failed root find)
        dr[0] = -xl[3];
        dr[1] = xl[2];
        return;
    }
    double sm, cm;
    qsincos_any(0.5 * (xl[4] + rhs[4]), &sm, &cm);
    g.x += h * vbar * cm;                                     // :475-477; :337-348 back to the engine's frame
    g.y -= h * vbar * sm;
#pragma unroll
    for (int k = 0; k < 5; k++) xl[k] = rhs[k];
    g.psi = -limit_angle(xl[4]);
    g.v = v;
    g.delta = -limit_angle(xl[1]);
    g.theta = limit_angle(xl[0]);
    dr[0] = -xl[3];
    dr[1] = xl[2];
}

__device__ __forceinline__ int64_t a_idx(int c, int64_t cap, int64_t a) { return (int64_t)c * cap + a; }

template <int MODEL>
__device__ __forceinline__ void integrate(const Dev &d, Agent &g, double Fx, double Fy) {
    const csf_params &p = *g.p;
    const int64_t a = g.a, cap = d.cap;
    if (MODEL == CSF_UNCONTROLLED) {                          // vehicle.py:964-979: the next prescribed state, if there is one
        g.ti = g.ti < 2000000000 ? g.ti + 1 : g.ti;           // (:973: the counter never wraps)
        const int32_t rows = d.slen[a];
        if (rows > 0) {
            if (g.ti < rows) {
                const double *r = d.script + 4 * (d.sbeg[a] + g.ti);
                g.x = r[0], g.y = r[1], g.psi = r[2], g.v = r[3];
            }
        } else if (g.ti < p.traj_len) {                       // no trajectory given: the ring of zeros of Vehicle.__init__ (:158-160)
            g.x = g.y = g.psi = g.v = 0.0;
        }
        return;
    }
    if (MODEL == CSF_BICYCLE) {                               // vehicle.py:1274-1289
        bike_control_move(d, g, Fx, Fy);
    } else if (MODEL == CSF_TWOD) {                           // vehicle.py:1386-1414
        if (g.zn == 2) {
            g.v = 0;
            g.delta = 0;
        } else bike_control_move(d, g, Fx, Fy);
    } else if (MODEL == CSF_INVPEND) {                        // vehicle.py:1883-1950
        bool riding = g.riding;
        // updateRidingState (:1932-1950): the slice traj[4, imin:i+1] is all inside +-delta_max_walk iff the
        // run of good samples ending at column i is at least as long as the slice
        bool cvwalk = g.v < p.v_max_walk;
        int imin = max(0, (int)((double)g.ti - 1.0 / p.t_s));
        bool cdelta = g.dgood >= (g.ti - imin + 1);
        riding = !cvwalk && ((!riding && cdelta) || riding);
        d.zrid[a] = riding ? 1 : 0;
        double xl[5];
#pragma unroll
        for (int k = 0; k < 5; k++) xl[k] = g.xl[k];
        if (g.zn == 2) {                                      // :1898-1899
            g.v = 0;
            g.delta = 0;
            g.theta = 0;
        } else if (riding) {
            double vd = qsqrt(Fx * Fx + Fy * Fy);             // step_pos :1850-1881 (old psi)
            double acc = clampd(p.k_p_v * (vd - g.v), p.a_max[0], p.a_max[1]);
            double v = clampd(g.v + p.t_s * acc, p.v_max_riding[0], p.v_max_riding[1]);
            double so, co;
            qsincos(g.psi, &so, &co);
            g.y += p.t_s * v * so;
            g.x += p.t_s * v * co;
            g.v = v;
            invpend_step_yaw(d, g, xl, Fx, Fy);               // uses the new speed (:1902-1903)
        } else {                                              // walking :1905-1916
            g.v = p.v_max_walk;
            g.theta = 0;
            bike_control_move(d, g, Fx, Fy);
            xl[0] = g.delta;
            xl[1] = 0;
            xl[2] = g.theta;
            xl[3] = 0;
            xl[4] = g.psi;
        }
#pragma unroll
        for (int k = 0; k < 5; k++) d.lti[k * cap + a] = xl[k];
    } else if (MODEL == CSF_BALANCINGRIDER) {                 // vehicle.py:301-328 with BalancingRiderDynamics.step
        double xl[5], dr[2];
#pragma unroll
        for (int k = 0; k < 5; k++) xl[k] = g.xl[k];
        balancingrider_step(d, g, xl, Fx, Fy, dr);
#pragma unroll
        for (int k = 0; k < 5; k++) d.lti[k * cap + a] = xl[k];
        d.ppsi[a] = g.ppsi;
        d.s[6 * cap + a] = dr[0];                             // steer rate, roll rate: columns 6, 7 of vehicle.s (vehicle.py:1960)
        d.s[7 * cap + a] = dr[1];
    } else if (MODEL == CSF_PLANARBIKE) {                     // PlanarTwoWheelerDynamics.step: dynamics.py:225-258
        // x = (delta, psi) follows x' = (A - B K_x) x + B K_u psi_d with A = [[0, 0], [v / l, 0]], B = (1, 0)^T, K_x placed
        // for the class's two poles and K_u from a simulated step response, both re-derived for the current speed every
        // step (dynamics.py:203-223, 1167-1226).  In z = (v delta / l, psi) the closed loop is the same for every speed
        // (csf_engine.hip: derive_planarbike), so one precomputed exact step does it.
        const double a = g.v / p.l;                           // the speed BEFORE the speed update (:228)
        double del = g.xl[0], psu = g.ppsi;                   // unwrapped (dynamics.py:195-197, 244)
        const double psi_d = qatan2(Fy, Fx), v_d = qsqrt(Fy * Fy + Fx * Fx);   // :231-232
        if (a > 0.0) {
            const double z0 = a * del, z1 = psu;
            del = qdiv(g.pb[0] * z0 + g.pb[1] * z1 + g.pb[4] * psi_d, a);   // :235-244
            psu = g.pb[2] * z0 + g.pb[3] * z1 + g.pb[5] * psi_d;
        } else {
            g.st |= CSF_ST_UNCONTROLLABLE;                    // dynamics.py:1212-1214 asserts; here the yaw loop holds still
        }
        d.lti[a_idx(0, cap, g.a)] = del;
        d.ppsi[g.a] = psu;
        g.psi = limit_angle_m(psu);                           // :246-247
        g.delta = limit_angle_m(del);
        g.v = v_d + (g.v - v_d) * g.pb[6];                    // PPointSpeedDynamics: dynamics.py:156, 175
        qsincos(g.psi, &g.spsi, &g.cpsi);
        g.cs_fresh = true;
        g.y += p.t_s * g.v * g.spsi;                          // :251-258
        g.x += p.t_s * g.v * g.cpsi;
    } else {                                                  // PlanarPoint: dynamics.py:996-1079
        double vd = qsqrt(Fx * Fx + Fy * Fy);                 // :1018
        double acc = clampd(p.k_p_v * (vd - g.v), p.a_max[0], p.a_max[1]);
        double v = clampd(g.v + p.t_s * acc, p.v_max_riding[0], p.v_max_riding[1]);
        double psi_c = limit_angle_m(qatan2(Fy, Fx));         // dynamics.py:112-121
        double vbar = 0.5 * (v + g.v);                        // :1065
        // implicit midpoint of psi' = -k (psi - psi_c), x' = v cos psi, y' = v sin psi in closed form
        double hk = p.t_s * p.k_psi, pu = g.ppsi;
        double pn = qdiv(pu * (1 - 0.5 * hk) + hk * psi_c, 1 + 0.5 * hk);
        double pm = 0.5 * (pu + pn);
        double sm, cm;
        qsincos_any(pm, &sm, &cm);
        g.x += p.t_s * vbar * cm;
        g.y += p.t_s * vbar * sm;
        d.ppsi[a] = pn;
        g.psi = limit_angle_m(pn);                            // :959-964
        g.v = v;
    }
    // ring-buffer bookkeeping — vehicle.py:1279-1282, 1407-1410, 1923-1926
    g.ti = (g.ti + 1) % p.traj_len;
    const int slot = g.ti & (d.hist_len - 1);
    d.hx[(int64_t)slot * cap + a] = g.x;
    d.hy[(int64_t)slot * cap + a] = g.y;
    if (MODEL == CSF_INVPEND) {
        bool good = (-p.delta_max_walk < g.delta) && (p.delta_max_walk > g.delta);
        int run = g.dgood;
        d.dgood[a] = good ? min(run + 1, 1 << 30) : 0;
    }
}

// fp32 source record of the next tick — the (x, y, psi) snapshot of intersection.py:660-677.  The position is stored as
// an offset from `o`, the road user's own origin (csf_dev.h: rorg), formed in fp64: one rounding, of a few metres.
__device__ __forceinline__ void write_record(const Dev &d, const csf_params &p, int64_t a, const float2 o, double x, double y,
                                             double psi, double v, bool cs_fresh = false, double c = 0.0, double s = 0.0,
                                             int32_t place = -1) {
    if (!cs_fresh) qsincos(psi, &s, &c);
    const double px = (x - d.ox) - (double)o.x, py = (y - d.oy) - (double)o.y;
    const float4 q = make_float4((float)px, (float)py, (float)c, (float)s);
    d.rec_w[a] = q;
    float2 lo = make_float2(0.f, 0.f);
    if (d.keep_lo) {
        lo = make_float2((float)(px - (double)q.x), (float)(py - (double)q.y));   // (what fp32 left over: csf_dev.h reclo)
        d.reclo[a] = lo;
    }
    // scene coordinates = offset + origin, in fp32 (the same sum a rank forms from the records it has gathered, so that
    // every path sees the same numbers): by slot (the receivers), and the copy in binned order (csf_bin.hip) that the
    // pair kernel's tiles are filled with
    const float4 g = make_float4(q.x + o.x, q.y + o.y, q.z, q.w);
    d.recg_w[a] = g;
    if (d.recs_valid) {
        const int32_t p = place >= 0 ? place : d.pos[a];      // (the per-agent kernel asks for it with its first loads)
        d.recs_w[p] = g;
        if (d.recv_binned) {                                   // (large populations only: csf_dev.h recb)
            const float2 bo = d.borg[p >> 6];
            d.recb[p] = make_float4(q.x + (o.x - bo.x), q.y + (o.y - bo.y), q.z, q.w);
        }
    }
    float2 q2 = make_float2(0.0f, 1.0f);
    if (d.has_bike) {                                         // vehicle.py:1062-1064 (v <= 0: e := 0); other classes: unused
        double e = 0.0;
        if (p.model == CSF_BICYCLE && v > 0.0) e = fmin(pow(v / p.v_max_riding[1], 0.1), 0.7);
        q2 = make_float2((float)e, (float)qrsqrt(1.0 - e * e));
        d.rec2_w[a] = q2;
        if (d.recs_valid) d.recs2[d.pos[a]] = q2;
    }
    if (d.xbuf != nullptr) {                                  // the entry that travels to the other ranks (csf_dev.h: xbuf)
        d.xbuf[2 * a] = q;
        d.xbuf[2 * a + 1] = make_float4(lo.x, lo.y, q2.x, q2.y);
    }
}


// One road user's tick.  FUSED (small_tick_kernel, below): the repulsive sum (frx, fry) comes from the caller - no pair kernel
// has run, so there are no partial sums to read and no pairs handed over.
// MID (csf_mid.hip: the tick of a mid-size population in one launch, one workgroup per receiver group): 1 - this call is the
// destination-force phase, run by the group's first wave WHILE the other waves form the pair sums (they may be setting
// CSF_ST_EDGE in the status word: its bits go in with an atomic OR); 2 - this call is the rest, by the same wave behind the
// workgroup's barrier: the repulsive sum comes from the caller (frx, fry) and what the other waves left in memory - status bits,
// hand-overs - comes in through agent-scope loads, which bypass this CU's L1 (csf_dev.h: ld_pub).
template <bool PUB>
__device__ __forceinline__ EdgeRec load_edge(const EdgeRec *p) {
    if (!PUB) return *p;
    EdgeRec r;
    r.xi = ld_pub<true>(&p->xi); r.yi = ld_pub<true>(&p->yi); r.psi = ld_pub<true>(&p->psi); r.hfov = ld_pub<true>(&p->hfov);
    r.fx = ld_pub<true>(&p->fx); r.fy = ld_pub<true>(&p->fy); r.fx2 = ld_pub<true>(&p->fx2); r.fy2 = ld_pub<true>(&p->fy2);
    r.recv = ld_pub<true>(&p->recv); r.next = ld_pub<true>(&p->next); r.stamp = ld_pub<true>(&p->stamp); r.flags = ld_pub<true>(&p->flags);
    return r;
}

template <int MODEL, bool HET, bool FUSED, int MID = 0>
__device__ __forceinline__ void agent_body(const Dev &d, const int phases, const int64_t a, uint64_t *const tr, const uint32_t ka_lines,
                                           const double frx, const double fry) {
    auto stamp = [&](int k) {
        if (tr != nullptr) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            if ((threadIdx.x & 63) == 0) tr[k] = wall_clock64();
        }
    };
    if (a >= d.hi) return;
    // a slot left behind by csf_remove_agents; asked only when there is one (n_live != n, uniform): the answer is a round
    // trip to memory that every other load of the kernel would wait behind
    if (d.n_live != d.n && !d.alive[a]) return;
    const int64_t cap = d.cap;
    constexpr bool CHASE = MID == 3;                          // beside the pair launch (csf_dev.h: part4): the sums are waited for
    constexpr bool IN = MID == 2 || CHASE, OUT = MID == 1, SUMS_GIVEN = FUSED || MID == 2;
    constexpr bool PLANNER = MODEL != CSF_BICYCLE && MODEL != CSF_UNCONTROLLED;   // (the models whose planner reads the ring)
    constexpr int PRE = 16;
    Agent g;
    float2 pp[PRE];
    g.a = a;
    g.x = d.s[a];
    g.y = d.s[cap + a];
    g.psi = d.s[2 * cap + a];
    g.v = d.s[3 * cap + a];
    g.delta = d.s[4 * cap + a];
    g.theta = d.s[5 * cap + a];
    g.vdes = d.vdes[a];
    g.qb = d.qbeg[a];
    g.K = d.qlen[a];
    g.ptr = d.ptr[a];
    g.zn = d.znav[a] & 3;
    g.zv0 = d.znp[a];
    g.zd0 = d.znp[cap + a];
    g.zd1 = d.znp[2 * cap + a];
    g.ti = d.ti[a];
    g.st = ld_pub<IN>(&d.status[a]);
    g.cs_fresh = false;
    kernarg_touched(ka_lines);
    const int32_t place = d.recs_valid ? d.pos[a] : -1;        // (for the record written at the end)
    agent_params<HET>(d, a, g);
    if (HET && g.p->model != MODEL) return;                    // a mixed population: one launch per vehicle class
    // The partial sums of the pair kernel (the first 16 chunks: the usual split) and the road term are requested here, with
    // the agent's own scalars, so that their round trip runs beside the destination-force phase.  Unconditional loads from
    // clamped addresses: a guarded load is a branch with its own wait.
#pragma unroll
    for (int c = 0; c < PRE; c++) pp[c] = make_float2(0.f, 0.f);
    if (!SUMS_GIVEN && !CHASE) {                               // (eight chunks is the single-device split: the other eight loads
#pragma unroll                                                 //  would only queue in front of everything asked for after them)
        for (int c = 0; c < PRE / 2; c++) pp[c] = d.part[(int64_t)min(c, d.n_split - 1) * cap + a];
        if (d.n_split > PRE / 2) {
#pragma unroll
            for (int c = PRE / 2; c < PRE; c++) pp[c] = d.part[(int64_t)min(c, d.n_split - 1) * cap + a];
        }
    }
    const float2 froad = d.froad[a];
    const float2 rorg = d.rorg[a];                             // (for the record written at the end)
    if (MODEL == CSF_INVPEND || MODEL == CSF_BALANCINGRIDER) {   // (side-state of the model: with the first round trip, not the fourth)
#pragma unroll
        for (int k = 0; k < 5; k++) g.xl[k] = d.lti[(int64_t)k * cap + a];
    }
    if (MODEL == CSF_INVPEND) {
        g.riding = d.zrid[a] != 0;
        g.dgood = d.dgood[a];
    }
    if (MODEL == CSF_PLANARBIKE) g.xl[0] = d.lti[a];
    if (MODEL == CSF_PLANARPOINT || MODEL == CSF_PLANARBIKE || MODEL == CSF_BALANCINGRIDER) g.ppsi = d.ppsi[a];
    load_rows(d, g);                                           // the second and last round trip of the common path
    if (PLANNER && (phases & PH_DEST)) load_ring(d, g);
    // pairs of this receiver that the pair kernel could not decide within fp32 rounding of a field-of-view edge wait in a ring
    // (csf_dev.h: EdgeRec): bit 31 of the status word says so - a few dozen road users of a large population, per tick
    bool edge_pending = !FUSED && !CHASE && (phases & PH_COMBINE) && (g.st & CSF_ST_EDGE) != 0u;   // (CHASE: asked behind the wait)
    if (edge_pending) g.st &= ~CSF_ST_EDGE;
    stamp(1);
    double fdx, fdy;
    if (phases & PH_DEST) {
        dest_force<MODEL, FUSED || MID == 1 || MID == 2>(d, g, fdx, fdy);
        d.F[2 * cap + a] = fdx;
        d.F[3 * cap + a] = fdy;
        d.ptr[a] = g.ptr;
        d.znav[a] = (uint8_t)g.zn;
        d.znp[a] = g.zv0;
        d.znp[cap + a] = g.zd0;
        d.znp[2 * cap + a] = g.zd1;
    } else {
        // (unconditional - a guarded load is a branch with its own wait -, from a row the view has: csf_replay_forces hands a view of
        // two rows, and this phase's result is not used there)
        fdx = d.F[(int64_t)min(2, d.F_rows - 1) * cap + a];
        fdy = d.F[(int64_t)min(3, d.F_rows - 1) * cap + a];
    }
    stamp(2);
    if (CHASE) {
        // The pair launch that forms this wave's sums is still running.  Every pair workgroup ends with ONE write-through store per
        // receiver - its partial sum and the tick's tag in one 16-byte granule - and neither waits nor signals; here a poll loop over
        // the lanes' own granules (agent-scope loads, an s_sleep between polls), then the status words and the hand-over entries.  A
        // wait that does not end gives up after CHASE_SPIN_LIMIT polls and says so (csf_sync turns that into an error): the grid always drains.
        // every lane polls the granules of ITS road user - (x, y, tag, 0), one per source chunk, each written by one store of the pair
        // workgroup that formed it - until every lane of the wave has all of this tick's
        const uint32_t tag = d.chase_tag;
        const int nsp = d.n_split;
        unsigned spins = 0;
        for (;;) {
            bool mine = true;
            v4f_t g8[8];
            ld_granules16_x8(&d.part4[a], cap, nsp < 8 ? nsp : 8, g8);
#pragma unroll
            for (int c = 0; c < 8; c++) {
                pp[c] = make_float2(g8[c].x, g8[c].y);
                mine = mine && (c >= nsp || __float_as_uint(g8[c].z) == tag);
            }
            if (nsp > 8) {                                             // (more than eight source chunks: a second batch)
                ld_granules16_x8(&d.part4[(int64_t)8 * cap + a], cap, nsp - 8, g8);
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    pp[8 + c] = make_float2(g8[c].x, g8[c].y);
                    mine = mine && (8 + c >= nsp || __float_as_uint(g8[c].z) == tag);
                }
            }
            if (__builtin_amdgcn_ballot_w64(!mine) == 0ull) break;
            if (++spins > CHASE_SPIN_LIMIT) {
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&d.chase_misc[1], 1u);
                    if (d.chase_err != nullptr) __hip_atomic_store(d.chase_err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                break;
            }
            __builtin_amdgcn_s_sleep(16);
        }
        stamp(3);
        if (d.chase_clock != nullptr && (threadIdx.x & 63) == 0) atomicMax(d.chase_clock + 8 * d.chase_slot + 5, (unsigned long long)wall_clock64());
        g.st |= ld_pub<true>(&d.status[a]) & CSF_ST_EDGE;      // (set by pair workgroups since the first load: an atomic OR)
        edge_pending = (g.st & CSF_ST_EDGE) != 0u;
        g.st &= ~CSF_ST_EDGE;
    }
    double Fx, Fy;
    if (phases & PH_COMBINE) {
        double rx = 0, ry = 0;
        Fx = fdx;
        Fy = fdy;
        if (d.n_live > 1) {                                   // intersection.py:813, 825, 849-851
            // the first 16 chunks (the usual split) requested together: one round trip, not one per group of four.  Added
            // in chunk order: reproducible
            int c = 0;
#pragma unroll
            for (; c < PRE; c++) {
                if (!SUMS_GIVEN && c < d.n_split) {
                    rx += (double)pp[c].x;
                    ry += (double)pp[c].y;
                }
            }
            for (; !SUMS_GIVEN && c < d.n_split; c++) {
                const float2 pr = d.part[(int64_t)c * cap + a];
                rx += (double)pr.x;
                ry += (double)pr.y;
            }
            if (SUMS_GIVEN) rx = frx, ry = fry;
            if (edge_pending) {     // rare: decide those pairs as the reference does (intersection.py:711-741) and put the sum right
                double cx = 0, cy = 0;   // (fp64: the order in which the entries were appended does not show)
                int32_t at = ld_pub<IN>(&d.edge_head[a]);
                // (a receiver's chain is at most the ring; a walk that long without reaching its end is counted like an overflow)
                for (int guard = 0; at != 0; guard++) {
                    if (guard == (int)EDGE_CAP) {
                        atomicAdd(d.near_dropped, 1u);
                        break;
                    }
                    const EdgeRec er = load_edge<IN>(&d.edge[(unsigned)(at - 1) % EDGE_CAP]);
                    if (er.recv != (int32_t)a || er.stamp != d.edge_stamp) {   // left over from another launch, or a ring that overflowed
                        if (er.stamp == d.edge_stamp) atomicAdd(d.near_dropped, 1u);
                        break;
                    }
                    const bool seen = !untracked_exact_xy(er.xi, er.yi, g.x, g.y, g.psi, er.hfov, d.p.priority_rule == CSF_P2R);
                    double wx = 0, wy = 0;                       // what the reference adds for this pair ...
                    // (road users that coincide in fp64 contribute nothing - deviation D2 - though their records may be an ulp apart)
                    if (seen && (g.x != er.xi || g.y != er.yi)) {
                        wx = (double)er.fx;
                        wy = (double)er.fy;
                        if (er.flags & EDGE_SIDE) {              // np.sign(phi) by the reference's own chain (vehicle.py:1617-1625)
                            double psii = er.psi;
                            if (er.flags & EDGE_HEADING_REC)
                                psii = atan2((double)__int_as_float(__double2hiint(er.psi)), (double)__int_as_float(__double2loint(er.psi)));
                            const int sg = sign_phi_exact(er.xi, er.yi, psii, g.x, g.y);
                            if (sg < 0) {
                                wx = (double)er.fx2;
                                wy = (double)er.fy2;
                            } else if (sg == 0) {                // phi = 0 exactly: no tangential part, |F| = P along the line
                                const double ex = g.x - er.xi, ey = g.y - er.yi, P = sqrt(wx * wx + wy * wy);
                                const double il = 1.0 / sqrt(ex * ex + ey * ey);
                                wx = P * ex * il;
                                wy = P * ey * il;
                            }
                        }
                    }
                    if (er.flags & EDGE_SEEN) {                  // ... minus what the pair kernel added
                        wx -= (double)er.fx;
                        wy -= (double)er.fy;
                    }
                    cx += wx;
                    cy += wy;
                    at = er.next;
                }
                d.edge_head[a] = 0;
                rx += cx;
                ry += cy;
            }
            double rin = qsqrt(rx * rx + ry * ry), lim = qsqrt(fdx * fdx + fdy * fdy);
            if (rin > lim) {                                  // utils.py:79-84
                const double irin = rcp_nr(rin);
                rx = rx * lim * irin;
                ry = ry * lim * irin;
            }
            Fx = rx + fdx;                                    // :847-848
            Fy = ry + fdy;
        }
        if (d.nv > 0) {                                       // :854-857
            Fx += (double)froad.x;
            Fy += (double)froad.y;
        }
        d.F[a] = Fx;                                          // :860-861
        d.F[cap + a] = Fy;
        d.F[4 * cap + a] = rx;
        d.F[5 * cap + a] = ry;
        if (!(isfinite(Fx) && isfinite(Fy))) g.st |= CSF_ST_NAN;
    } else {
        Fx = d.F[a];
        Fy = d.F[cap + a];
    }
    stamp(4);
    if (phases & PH_INTEGRATE) {
        const bool frozen = d.replay_len != nullptr && d.replay_tick >= d.replay_len[a];  // replay sequence ended
        if (!frozen) {
            if (phases & PH_FIXSPEED) g.v = qsqrt(Fx * Fx + Fy * Fy);  // calibration.py:454-458
            integrate<MODEL>(d, g, Fx, Fy);
        }
        stamp(5);
        d.s[a] = g.x;
        d.s[cap + a] = g.y;
        d.s[2 * cap + a] = g.psi;
        d.s[3 * cap + a] = g.v;
        d.s[4 * cap + a] = g.delta;
        d.s[5 * cap + a] = g.theta;
        d.ti[a] = g.ti;
        write_record(d, *g.p, a, rorg, g.x, g.y, g.psi, g.v, g.cs_fresh, g.cpsi, g.spsi, place);
        if (d.src64_w != nullptr) {                           // (fused tick, csf_mid.hip: the next tick's snapshot)
            d.src64_w[a] = g.x;
            d.src64_w[cap + a] = g.y;
            d.src64_w[2 * cap + a] = g.psi;
        }
        if (d.hist != nullptr) {
            int64_t t1 = d.tick + 1;
            if (t1 % d.hist_stride == 0) {
                int64_t smp = (t1 / d.hist_stride - 1) % d.hist_cap;
                double *o = d.hist + (smp * d.n + a) * d.ns;
                o[0] = g.x;
                o[1] = g.y;
                o[2] = g.psi;
                o[3] = g.v;
                if (d.ns > 4) o[4] = g.delta;
                if (d.ns > 5) o[5] = g.theta;
            }
        }
    }
    if (OUT) {
        // (fused tick, csf_mid.hip: the pair workgroups of this group may be setting CSF_ST_EDGE in the same word right now)
        if ((g.st & ~CSF_ST_EDGE) != 0u) atomicOr(&d.status[a], g.st & ~CSF_ST_EDGE);
    } else {
        d.status[a] = g.st;
    }
    if (tr != nullptr && (threadIdx.x & 63) == 0) tr[6] = wall_clock64();
    stamp(7);
}

}  // namespace csf
