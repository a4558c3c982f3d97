/*
 * csf_oracle.c — CPU restatement (fp64, plain C) of the reference's per-tick hot path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (cyclistsocialforce_amd/, include/csf.h) never links, imports or
 * falls back to it.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/src/cyclistsocialforce/).  The restatement is deliberately literal — polar
 * coordinates, arccos/atan2/sin/cos exactly where the reference has them — so that it is an
 * independent check of the trig-free HIP kernels.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against every golden vector in
 * tests/golden/ (captured by running the literal reference, see tests/golden/make_golden.py).
 * The InvPendulum rows are pinned against the reference's matrix assembly + state machine running
 * on a scipy.linalg.expm shim of python-control (absent here): "parity unpinned against
 * python-control" for that row.
 *
 * Documented deviations from the literal reference (all in places where the reference yields NaN or
 * raises):
 *   D1  vehicle.py:1644-1646  P*Fx/F is evaluated as P*unit(g) with g P-free: identical where the
 *       reference is finite, 0 where exp() underflow makes the reference 0/0 = NaN (SURVEY finding 4).
 *   D2  coincident agents (rho == 0): 0 instead of NaN.
 *   D3  vehicle.py:1062-1064  v < 0 makes pow() NaN in the reference; here e := 0 for v <= 0.
 *   D4  vehicle.py:1496       splprep raises ValueError on duplicate consecutive points; here the
 *       agent gets status bit CSFO_ST_SPLINE and a zero destination force for that tick.
 *   D5  vehicle.py:320-321    Vehicle.step (PlanarPoint) never wraps i (IndexError at tick 3000);
 *       here i wraps like Bicycle.step (vehicle.py:1279-1280).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define CSFO_BICYCLE 0
#define CSFO_TWOD 1
#define CSFO_INVPEND 2
#define CSFO_PLANARPOINT 3
#define CSFO_PLANARBIKE 4
#define CSFO_UNCONTROLLED 5   /* vehicle.py:920-988: follows a prescribed trajectory, exerts the TwoD field, feels nothing */
#define CSFO_BALANCINGRIDER 6 /* vehicle.py:1953-1990; dynamics.py:261-705: Whipple-Carvallo bicycle under full-state feedback */
#define CSFO_NS_MAX 8         /* columns of vehicle.s at most (BalancingRiderBicycle: + steer rate, roll rate) */

#define CSFO_ST_SPLINE 1u
#define CSFO_ST_NAN 2u
#define CSFO_ST_NAVSTATE 4u

#define PI 3.141592653589793238462643383279502884

/* Field order mirrors csf_params of include/csf.h so that the test harness fills both from one table. */
typedef struct csfo_params {
    /* VehicleParameters — parameters.py:430-508 */
    double t_s, d_arrived_inter, d_arrived_stop, v_max_stop, v_max_harddecel, hfov;
    double f_0, e_0, e_1, sigma_0, sigma_1, sigma_2, sigma_3;
    /* BicycleParameters — parameters.py:780-800 */
    double v_max_riding[2], p_decay, p_0, l, l_2, delta_max, a_max[2], a_desired_default[2];
    double k_p_v, k_p_delta, g;
    /* InvPendulumBicycleParameters — parameters.py:1429-1472, 1641-1643 */
    double h, m, i_bike_longlong, i_steer_vertvert, c_steer, v_max_walk, delta_max_walk;
    /* PlanarPointBicycleParameters — parameters.py:1180-1201 (gain = -Re(pole), dynamics.py:933-940) */
    double k_psi;
    /* PlanarBicycleParameters — parameters.py:1203-1211: the two desired poles (re, im, re, im) */
    double pb_poles[4];
    /* BalancingRiderBicycleParameters — parameters.py:1214-1411; dynamics.py:261-705.  The linearised Whipple-Carvallo
     * bicycle (Meijaard, Papadopoulos, Ruina & Schwab 2007: M q'' + v C1 q' + (g K0 + v^2 K2) q = f, q = (roll, steer)) as the
     * 2 x 2 blocks of its state matrix (row-major), the steer-torque column of M^-1, the yaw row (dynamics.py:301-303), and
     * the control model: desired poles as straight lines over speed (parameters.py:1400-1409; (intercept, slope) of p0_real,
     * p1_real, p1_imag, p2_real, p2_imag), constant poles (slopes 0), or constant gains (br_mode 2, dynamics.py:604-605) */
    double br_minv_k0g[4], br_minv_k2[4], br_minv_c1[4], br_minv_steer[2], br_yaw[2], br_pole_fun[10], br_gains[5];
    int32_t model, priority_rule /* 0 unregulated, 1 p2r */, traj_len /* int(30/t_s) */, br_mode;
} csfo_params;

typedef struct csfo {
    csfo_params p;
    /* per-vehicle parameter sets (every reference vehicle owns a params object: vehicle.py:64-204): class table and the
     * class of every agent; NULL = everyone uses p */
    int n_classes;
    csfo_params *ptab;
    uint8_t *pcls;
    int n, ns;
    double *s;        /* [n][CSFO_NS_MAX] */
    double *vdes;     /* [n]  params.v_desired_default per agent (demoCSFstandalone.py:104-113) */
    int64_t *qoff;    /* [n+1] CSR offsets into dq */
    double *dq;       /* [sum][3] x, y, stop */
    int32_t *ptr;     /* destpointer */
    uint8_t *znav;    /* [n][3] */
    double *znavp;    /* [n][4] v0, d0, d1, i */
    int32_t *i;       /* ring-buffer column */
    double *traj;     /* [n][3][L]: x, y, delta */
    double *xlti;     /* [n][5] delta, ddelta, theta, dtheta, psi (InvPendulum) */
    uint8_t *zrid;    /* [n][2] */
    double *xdyn;     /* [n][3] psi (unwrapped), x, y (PlanarPoint) */
    double *vdyn;     /* [n] */
    double *sx, *sy, *spsi, *sv; /* snapshot: intersection.py:660-677 (+ source speed for A2') */
    double *Fx, *Fy, *Fdx, *Fdy, *Frx, *Fry;
    int64_t nv;       /* road vertices, per-vertex F0 / sigma */
    double *vx, *vy, *vF0, *vsig;
    uint32_t *status;
    int64_t tick;
    /* UncontrolledVehicle (vehicle.py:920-988): the prescribed trajectory of every agent, rows of (x, y, psi, v); CSR */
    int64_t *soff;    /* [n+1] or NULL */
    double *script;   /* [sum][4] */
} csfo_t;

/* ------------------------------------------------------------------ utils.py ---- */

/* utils.py:124-139 */
double csfo_limit_angle(double th) {
    th = floor(th / (2 * PI)) * (-2 * PI) + th;
    if (th > PI) th = th - 2 * PI;
    else if (th < -PI) th = th + 2 * PI;
    return th;
}

/* utils.py:151-182 (scalar branch) */
double csfo_angle_difference(double a1, double a2) {
    double da = (a1 > a2) ? a1 - a2 : a2 - a1;
    if (da > PI) da = 2 * PI - da;
    double t1 = fabs(csfo_limit_angle(a1 - da) - a2);
    double t2 = fabs(csfo_limit_angle(a1 + da) - a2);
    return (t1 < t2) ? -da : da;
}

/* utils.py:204-227 */
static double thresh(double x, double lo, double hi) { return fmax(fmin(x, hi), lo); }

/* utils.py:56-86, one vector */
void csfo_limit_magnitude(double *x, double *y, double r) {
    double rin = sqrt(*x * *x + *y * *y);
    if (rin > r) {
        *x = *x * r / rin;
        *y = *y * r / rin;
    }
}

/* utils.py:185-194 */
static void cart2polar(double x, double y, double *rho, double *phi) {
    *rho = sqrt(x * x + y * y);
    double p = acos(x / *rho);
    if (y < 0) p = -p;
    *phi = p;
}

/* ------------------------------------------------------- the force fields (A2, A2') ---- */

/* vehicle.py:1560-1648 — field of source (x0,y0,psi0) at one receiver (x,y,psi). Deviations D1, D2. */
void csfo_pair_twod(const csfo_params *p, double x0, double y0, double psi0, double x, double y,
                    double psi, double *Fx, double *Fy) {
    *Fx = 0.0;
    *Fy = 0.0;
    if (p->f_0 == 0.0) return;                                     /* :1592-1593 */
    double psi_rel = psi0 - psi;                                   /* :1595 */
    double s2 = sin(psi_rel) * sin(psi_rel);
    double sig_a = p->sigma_0 + p->sigma_1 * s2;                   /* :1604-1606 */
    double sig_b = p->sigma_2 + p->sigma_3 * s2;                   /* :1607-1609 */
    double e = p->e_0 - p->e_1 * s2;                               /* :1612 */
    double dx = x - x0, dy = y - y0;                               /* :1615-1616 */
    double rho, phi1;
    cart2polar(dx, dy, &rho, &phi1);                               /* :1617 */
    if (!(rho > 0.0)) return;                                      /* D2 */
    double phi = csfo_limit_angle(phi1 - psi0);                    /* :1618 */
    double cosphi = cos(phi), sinphi = sin(phi);
    double sgn = (phi > 0) - (phi < 0);
    double sigma = sig_a - sig_b * sqrt((1 - cosphi) / 2);         /* :1624 */
    double dsigm = -sig_b * sqrt((1 + cosphi) / 2) * sgn / 2;      /* :1625 */
    double q2 = 1 - (e * cosphi) * (e * cosphi);
    double q = sqrt(q2);
    double P = p->f_0 * exp(-rho * q / sigma);                     /* :1628 */
    /* :1631-1639 with the common factor P taken out (D1) */
    double frho = q / sigma;
    double fphi = -(q2 * dsigm - e * e * sinphi * cosphi * sigma) / (sigma * sigma * q);
    double gx = frho * cos(phi1) - fphi * sin(phi1);               /* :1641 */
    double gy = frho * sin(phi1) + fphi * cos(phi1);               /* :1642 */
    double G = sqrt(gx * gx + gy * gy);                            /* :1644 */
    if (!(G > 0.0)) return;
    *Fx = P * gx / G;                                              /* :1645 */
    *Fy = P * gy / G;                                              /* :1646 */
}

/* vehicle.py:1054-1147 — older elliptic field of base Bicycle; needs the source's speed. D2, D3. */
void csfo_pair_bicycle(const csfo_params *p, double x0, double y0, double psi0, double v0, double x,
                       double y, double *Fx, double *Fy) {
    *Fx = 0.0;
    *Fy = 0.0;
    double dx = x - x0, dy = y - y0;                               /* :1123-1124 */
    double rho, phi;
    cart2polar(dx, dy, &rho, &phi);                                /* :1126 */
    if (!(rho > 0.0)) return;                                      /* D2 */
    double phi0 = phi - psi0;                                      /* :1129 */
    double e = 0.0;                                                /* :1062-1064, D3 */
    if (v0 > 0.0) {
        double pw = pow(v0 / p->v_max_riding[1], 0.1);
        e = (0.7 < pw) ? 0.7 : pw;
    }
    double se = sqrt(1 - e * e);
    double b = (1 / (se * p->p_decay)) * rho * (1 - e * cos(phi0));   /* :1095-1099 */
    double P = p->p_0 * exp(-b) / p->p_decay;                      /* :1101, :1132 */
    double Frho0 = P * ((1 - e * cos(phi0)) / se);                 /* :1135-1137 */
    double Fphi0 = P * ((e * sin(phi0)) / se);                     /* :1138-1140 */
    *Fx = Frho0 * cos(phi) - Fphi0 * sin(phi);                     /* :1144 */
    *Fy = Frho0 * sin(phi) + Fphi0 * cos(phi);                     /* :1145 */
}

/* intersection.py:690-745 — is source i ignored by receiver j?  (row = source, column = receiver) */
int csfo_untracked(double hfov_i, int rule, int i, int j, double xi, double yi, double xj, double yj,
                   double psij) {
    if (i == j) return 1;                                          /* :729-730 */
    double az = csfo_limit_angle(atan2(yi - yj, xi - xj));         /* :711-718 */
    double rel = csfo_angle_difference(psij, az);                  /* :724-726 */
    if (fabs(rel) > hfov_i / 2) return 1;                          /* :733-736 */
    if (rule == 1 && rel > 0) return 1;                            /* :739-741 */
    return 0;
}

/* intersection.py:690-745 — the whole matrix (row = source i with ITS hfov, :733-735; column = receiver j) */
void csfo_untracked_matrix(const double *hfov, int rule, int64_t n, const double *x, const double *y, const double *psi,
                           uint8_t *out) {
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; i++)
        for (int64_t j = 0; j < n; j++)
            out[i * n + j] = (uint8_t)csfo_untracked(hfov[i], rule, (int)(i != j), 0, x[i], y[i], x[j], y[j], psi[j]);
}

/* intersection.py:226-242 — one vertex list with per-vertex F0 and sigma (summed over edges: 36-48, 81-94) */
void csfo_road_force(int64_t nv, const double *vx, const double *vy, const double *vF0,
                     const double *vsig, double x, double y, double *Fx, double *Fy) {
    double fx = 0, fy = 0;
    for (int64_t k = 0; k < nv; k++) {
        double ex = vx[k] - x, ey = vy[k] - y;
        double r = sqrt(ex * ex + ey * ey);                        /* :231-234 */
        double F = -vF0[k] * pow(r, -vsig[k]);                     /* :238 */
        fx += F * (ex / r);                                        /* :235, :239 */
        fy += F * (ey / r);                                        /* :236, :240 */
    }
    *Fx = fx;
    *Fy = fy;
}

/* intersection.py:814-843 for a SAMPLE of receivers of a large population: the column sums (before the clamp of
 * :841-845) of receivers recv[0..m) over all n sources, every pair through csfo_untracked and the field function.
 * O(n) memory: the full-size parity tests (262 144 and 1 048 576 agents) cannot afford a csfo_t with its 30-s
 * trajectory rings.  v is read for the Bicycle field only. */
void csfo_column_sums(const csfo_params *p, int64_t n, const double *x, const double *y, const double *psi,
                      const double *v, int64_t m, const int64_t *recv, double *rx, double *ry) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < m; k++) {
        const int64_t j = recv[k];
        double sx = 0, sy = 0;
        for (int64_t i = 0; i < n; i++) {
            if (csfo_untracked(p->hfov, p->priority_rule, i == j ? 0 : 1, 0, x[i], y[i], x[j], y[j], psi[j])) continue;
            double gx, gy;
            if (p->model == CSFO_BICYCLE) csfo_pair_bicycle(p, x[i], y[i], psi[i], v[i], x[j], y[j], &gx, &gy);
            else csfo_pair_twod(p, x[i], y[i], psi[i], x[j], y[j], psi[j], &gx, &gy);
            sx += gx;                                              /* :842-843 */
            sy += gy;
        }
        rx[k] = sx;
        ry[k] = sy;
    }
}

/* the same for a population whose vehicles own different parameter sets / are of different classes (csfo_set_classes):
 * source i with the field, the hfov and the class of ITS set tab[cls[i]] */
void csfo_column_sums_classes(const csfo_params *tab, const uint8_t *cls, int rule, int64_t n, const double *x, const double *y,
                              const double *psi, const double *v, int64_t m, const int64_t *recv, double *rx, double *ry) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < m; k++) {
        const int64_t j = recv[k];
        double sx = 0, sy = 0;
        for (int64_t i = 0; i < n; i++) {
            const csfo_params *p = &tab[cls[i]];
            if (csfo_untracked(p->hfov, rule, i == j ? 0 : 1, 0, x[i], y[i], x[j], y[j], psi[j])) continue;
            double gx, gy;
            if (p->model == CSFO_BICYCLE) csfo_pair_bicycle(p, x[i], y[i], psi[i], v[i], x[j], y[j], &gx, &gy);
            else csfo_pair_twod(p, x[i], y[i], psi[i], x[j], y[j], psi[j], &gx, &gy);
            sx += gx;
            sy += gy;
        }
        rx[k] = sx;
        ry[k] = sy;
    }
}

/* intersection.py:226-242, 854-857 for m receivers (OpenMP over the receivers) */
void csfo_road_forces(int64_t nv, const double *vx, const double *vy, const double *vF0, const double *vsig,
                      int64_t m, const double *x, const double *y, double *fx, double *fy) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t k = 0; k < m; k++) csfo_road_force(nv, vx, vy, vF0, vsig, x[k], y[k], &fx[k], &fy[k]);
}

/* ----------------------------------------------- destination queue + nav state machine ---- */

static inline double *S(csfo_t *o, int a) { return o->s + CSFO_NS_MAX * (size_t)a; }
static inline const csfo_params *PA(const csfo_t *o, int a) { return o->pcls ? &o->ptab[o->pcls[a]] : &o->p; }
static inline int qlen(csfo_t *o, int a) { return (int)(o->qoff[a + 1] - o->qoff[a]); }
static inline double *qrow(csfo_t *o, int a, int k) { return o->dq + 3 * (o->qoff[a] + k); }
static inline double *trj(csfo_t *o, int a, int row) {
    return o->traj + ((size_t)a * 3 + row) * (size_t)o->p.traj_len;
}

/* vehicle.py:596-604 */
static double dest_distance(csfo_t *o, int a) {
    double *d = qrow(o, a, o->ptr[a]), *s = S(o, a);
    return sqrt(pow(d[0] - s[0], 2) + pow(d[1] - s[1], 2));
}

/* vehicle.py:537-543 */
static int is_last_dest(csfo_t *o, int a) { return o->ptr[a] + 1 >= qlen(o, a); }

/* vehicle.py:545-594 */
static void update_destination(csfo_t *o, int a) {
    double dnext = dest_distance(o, a);                            /* :560 */
    uint8_t *z = o->znav + 3 * a;
    if (z[1] || z[2]) return;                                      /* :567-568 */
    int K = qlen(o, a);
    if (dnext <= PA(o, a)->d_arrived_inter) {                           /* :571-574 */
        int np1 = o->ptr[a] + 1;
        o->ptr[a] = np1 < K - 1 ? np1 : K - 1;
    }
    if (o->ptr[a] < K - 1) {                                       /* :577-583 */
        double *d = qrow(o, a, o->ptr[a] + 1), *s = S(o, a);
        double dnn = sqrt((d[0] - s[0]) * (d[0] - s[0]) + (d[1] - s[1]) * (d[1] - s[1]));
        if (dnn < dnext) o->ptr[a] += 1;
    }
}

/* vehicle.py:354-457 */
static void update_nav_state(csfo_t *o, int a, double *vd_out, double *ddest_out) {
    const csfo_params *p = PA(o, a);
    uint8_t *zn = o->znav + 3 * a;
    double *zp = o->znavp + 4 * a, *s = S(o, a);
    int stop = qrow(o, a, o->ptr[a])[2] != 0.0;
    const double k = 1.5;                                          /* :377 */
    double d0, d1;
    if (zn[0]) {                                                   /* :379-386 */
        d0 = 0.5 * (p->v_max_harddecel * p->v_max_harddecel - s[3] * s[3]) / p->a_desired_default[0];
        d1 = 0.5 * -(p->v_max_harddecel * p->v_max_harddecel) / p->a_max[0];
    } else {                                                       /* :388-389 */
        d0 = zp[1];
        d1 = zp[2];
    }
    double ddest = dest_distance(o, a);                            /* :392 */
    int x0 = stop, x1 = ddest <= k * (d0 + d1), x2 = ddest <= p->d_arrived_stop,
        x3 = s[3] <= p->v_max_stop;                                /* :397-400 */
    int z0 = zn[0], z1 = zn[1], z2 = zn[2];
    int n0 = (!x0) || (x0 && !x1 && ((z0 && !x2) || z1));          /* :404-406 */
    int n1 = x0 && ((z0 && ((!x2 && x1) || (x2 && !x3))) || (z1 && x1 && (!x2 || !x3))); /* :407-413 */
    int n2 = x0 && (((z0 || z1) && x2 && x3) || z2);               /* :414 */
    zn[0] = (uint8_t)n0;
    zn[1] = (uint8_t)n1;
    zn[2] = (uint8_t)n2;
    if (n0 + n1 + n2 != 1) o->status[a] |= CSFO_ST_NAVSTATE;      /* :416-425 only prints */
    if (z0 && n1) {                                                /* :428-430 */
        zp[0] = s[3];
        zp[1] = d0;
        zp[2] = d1;
        zp[3] = (double)o->i[a];
    }
    double vd;
    if (n0) vd = o->vdes[a];                                       /* :434-435 */
    else if (n1) {                                                 /* :436-450 */
        if (ddest < k * zp[2]) vd = p->v_max_harddecel / zp[2] * ddest * 1 / k;
        else vd = (zp[0] - p->v_max_harddecel) / zp[1] * (ddest - zp[2]) * 1 / k + p->v_max_harddecel;
    } else vd = 0;                                                 /* :452-453 (invalid state: reference raises) */
    *vd_out = vd;
    *ddest_out = ddest;
}

/* vehicle.py:1150-1194 and 2078-2108 — straight-line destination force */
static void direct_approach(csfo_t *o, int a, double *Fx, double *Fy) {
    update_destination(o, a);                                      /* :1168-1169 */
    double vd, ddest;
    update_nav_state(o, a, &vd, &ddest);                           /* :1171 */
    double *d = qrow(o, a, o->ptr[a]), *s = S(o, a);
    if (ddest > 0) {                                               /* :1173-1178 */
        *Fx = -vd * (s[0] - d[0]) / ddest;
        *Fy = -vd * (s[1] - d[1]) / ddest;
    } else {
        *Fx = 0;
        *Fy = 0;
    }
}

/* ------------------------------------------------------------------ cubic B-spline ---- */
/* scipy.interpolate.splprep(s=0, k=3) / splev (FITPACK parcur, fpbspl, splev, splder; SciPy 1.15.3
 * here, reference requires >= 1.10).  Published algorithm: chord-length parameter u in [0,1];
 * knots = 4 x 0, u[2..m-3], 4 x 1; interpolation conditions B(u_r) c = p_r; de Boor evaluation. */

static int find_span(const double *t, int n, int k, double u) {
    int l = k;
    while (u >= t[l + 1] && l != n - 1) l++;
    return l;
}

/* values of the k+1 non-zero degree-k B-splines at u (Cox–de Boor), span l */
static void basis_funs(const double *t, int l, int k, double u, double *N) {
    double left[4], right[4];
    N[0] = 1.0;
    for (int j = 1; j <= k; j++) {
        left[j] = u - t[l + 1 - j];
        right[j] = t[l + j] - u;
        double saved = 0.0;
        for (int r = 0; r < j; r++) {
            double tmp = N[r] / (right[r + 1] + left[j - r]);
            N[r] = saved + right[r + 1] * tmp;
            saved = left[j - r] * tmp;
        }
        N[j] = saved;
    }
}

static double bspl_eval(const double *t, const double *c, int n, int k, double u) {
    double N[4];
    int l = find_span(t, n, k, u);
    basis_funs(t, l, k, u, N);
    double v = 0;
    for (int j = 0; j <= k; j++) v += N[j] * c[l - k + j];
    return v;
}

/* returns 0 on success, 1 if the input is rejected (duplicate consecutive points) */
int csfo_spline20(int m, const double *px, const double *py, double out[20][6]) {
    double u[6], t[10], A[6][6], cx[6], cy[6];
    u[0] = 0;
    for (int r = 1; r < m; r++) {
        double d = sqrt((px[r] - px[r - 1]) * (px[r] - px[r - 1]) + (py[r] - py[r - 1]) * (py[r] - py[r - 1]));
        if (!(d > 0.0)) return 1;
        u[r] = u[r - 1] + d;
    }
    for (int r = 1; r < m; r++) u[r] /= u[m - 1];
    u[m - 1] = 1.0;
    int n = m, k = 3;
    for (int j = 0; j < 4; j++) {
        t[j] = 0.0;
        t[n + j] = 1.0;
    }
    for (int j = 0; j < m - 4; j++) t[4 + j] = u[2 + j];
    memset(A, 0, sizeof A);
    for (int r = 0; r < m; r++) {
        double N[4];
        int l = find_span(t, n, k, u[r]);
        basis_funs(t, l, k, u[r], N);
        for (int j = 0; j <= k; j++) A[r][l - k + j] = N[j];
        cx[r] = px[r];
        cy[r] = py[r];
    }
    for (int c = 0; c < m; c++) { /* Gaussian elimination, partial pivoting */
        int piv = c;
        for (int r = c + 1; r < m; r++)
            if (fabs(A[r][c]) > fabs(A[piv][c])) piv = r;
        if (A[piv][c] == 0.0) return 1;
        if (piv != c) {
            for (int j = 0; j < m; j++) {
                double tmp = A[c][j];
                A[c][j] = A[piv][j];
                A[piv][j] = tmp;
            }
            double tmp = cx[c];
            cx[c] = cx[piv];
            cx[piv] = tmp;
            tmp = cy[c];
            cy[c] = cy[piv];
            cy[piv] = tmp;
        }
        for (int r = c + 1; r < m; r++) {
            double f = A[r][c] / A[c][c];
            if (f == 0.0) continue;
            for (int j = c; j < m; j++) A[r][j] -= f * A[c][j];
            cx[r] -= f * cx[c];
            cy[r] -= f * cy[c];
        }
    }
    for (int r = m - 1; r >= 0; r--) {
        double sx = cx[r], sy = cy[r];
        for (int j = r + 1; j < m; j++) {
            sx -= A[r][j] * cx[j];
            sy -= A[r][j] * cy[j];
        }
        cx[r] = sx / A[r][r];
        cy[r] = sy / A[r][r];
    }
    /* derivative splines: c'_j = k (c_{j+1}-c_j)/(t_{j+k+1}-t_{j+1}) on knots t[1..] */
    double c1x[6], c1y[6], c2x[6], c2y[6];
    for (int j = 0; j < n - 1; j++) {
        double den = t[j + 4] - t[j + 1];
        c1x[j] = den > 0 ? 3 * (cx[j + 1] - cx[j]) / den : 0.0;
        c1y[j] = den > 0 ? 3 * (cy[j + 1] - cy[j]) / den : 0.0;
    }
    const double *t1 = t + 1; /* degree 2, n-1 coefficients */
    for (int j = 0; j < n - 2; j++) {
        double den = t1[j + 3] - t1[j + 1];
        c2x[j] = den > 0 ? 2 * (c1x[j + 1] - c1x[j]) / den : 0.0;
        c2y[j] = den > 0 ? 2 * (c1y[j + 1] - c1y[j]) / den : 0.0;
    }
    const double *t2 = t + 2; /* degree 1, n-2 coefficients */
    for (int q = 0; q < 20; q++) {
        double uu = (double)q / 19.0; /* np.linspace(0,1,20) — vehicle.py:1508 */
        if (q == 19) uu = 1.0;
        out[q][0] = bspl_eval(t, cx, n, 3, uu);
        out[q][1] = bspl_eval(t, cy, n, 3, uu);
        out[q][2] = bspl_eval(t1, c1x, n - 1, 2, uu);
        out[q][3] = bspl_eval(t1, c1y, n - 1, 2, uu);
        out[q][4] = bspl_eval(t2, c2x, n - 2, 1, uu);
        out[q][5] = bspl_eval(t2, c2y, n - 2, 1, uu);
    }
    return 0;
}

/* vehicle.py:1416-1558 */
static void twod_dest_force(csfo_t *o, int a, double *Fx, double *Fy) {
    const csfo_params *p = PA(o, a);
    const int nSplV = 4, nSplpnts = 20, ipred = 3, ipredlast = 5;  /* :1444-1448 */
    double *s = S(o, a);
    update_destination(o, a);                                      /* :1451 */
    double vd, ddest;
    update_nav_state(o, a, &vd, &ddest);                           /* :1452 */
    int i = o->i[a];
    if (i == 0) {                                                  /* :1455-1458 */
        *Fx = vd * cos(s[2]);
        *Fy = vd * sin(s[2]);
        return;
    }
    if (o->znav[3 * a + 2]) {                                      /* :1461-1462 */
        *Fx = 0;
        *Fy = 0;
        return;
    }
    double px[6], py[6];
    int m;
    double *tx = trj(o, a, 0), *ty = trj(o, a, 1);
    int last = is_last_dest(o, a);
    double *dest = qrow(o, a, o->ptr[a]);
    if (!last) {                                                   /* :1465-1479 */
        int K = qlen(o, a), hi = o->ptr[a] + nSplV;
        if (hi > K) hi = K;
        px[0] = tx[i - 1];
        py[0] = ty[i - 1];
        px[1] = tx[i];
        py[1] = ty[i];
        m = 2;
        for (int k = o->ptr[a]; k < hi; k++, m++) {
            px[m] = qrow(o, a, k)[0];
            py[m] = qrow(o, a, k)[1];
        }
    } else {                                                       /* :1486-1492 */
        int back = i - (int)(1 / p->t_s);
        if (back < 0) back = 0;
        px[0] = tx[back];
        py[0] = ty[back];
        px[1] = tx[i - 1];
        py[1] = ty[i - 1];
        px[2] = tx[i];
        py[2] = ty[i];
        px[3] = dest[0];
        py[3] = dest[1];
        m = 4;
    }
    double sp[20][6];
    if (csfo_spline20(m, px, py, sp)) {                            /* :1495-1507, D4 */
        o->status[a] |= CSFO_ST_SPLINE;
        *Fx = 0;
        *Fy = 0;
        return;
    }
    int ii = 1;                                                    /* :1516-1522 */
    if (last) {
        double best = INFINITY;
        for (int q = 0; q < nSplpnts; q++) {
            double d = (sp[q][0] - s[0]) * (sp[q][0] - s[0]) + (sp[q][1] - s[1]) * (sp[q][1] - s[1]);
            if (d < best) {
                best = d;
                ii = q;
            }
        }
    }
    int iprev = ii + (dest[2] != 0.0 ? ipredlast : ipred);         /* :1523-1526 */
    if (iprev < nSplpnts) {                                        /* :1529-1553 */
        double R = pow(sqrt(sp[ii][2] * sp[ii][2] + sp[ii][3] * sp[ii][3]), 3) /
                   fabs(sp[ii][2] * sp[ii][5] - sp[ii][3] * sp[ii][4]);
        double thetacomf = 10 * (2 * PI / 360);
        double v = fmax(2.5, sqrt(thetacomf * p->g * R));
        v = fmin(v, vd);
        double ex = sp[iprev][0] - sp[ii][0], ey = sp[iprev][1] - sp[ii][1];
        double tmp = v / sqrt(ex * ex + ey * ey);
        *Fx = tmp * ex;
        *Fy = tmp * ey;
    } else {
        direct_approach(o, a, Fx, Fy);                             /* :1555-1556 */
    }
}

static void dest_force(csfo_t *o, int a, double *Fx, double *Fy) {
    switch (PA(o, a)->model) {
    case CSFO_UNCONTROLLED:                                        /* vehicle.py:987-988 */
        *Fx = 0;
        *Fy = 0;
        break;
    case CSFO_BICYCLE:
        direct_approach(o, a, Fx, Fy);                             /* vehicle.py:1189-1194 */
        break;
    case CSFO_BALANCINGRIDER:
        update_destination(o, a);                                  /* vehicle.py:295-297 */
        direct_approach(o, a, Fx, Fy);                             /* vehicle.py:1987-1988, 2078-2108 */
        break;
    case CSFO_PLANARPOINT:
    case CSFO_PLANARBIKE:
        update_destination(o, a);                                  /* vehicle.py:295-297 */
        twod_dest_force(o, a, Fx, Fy);                             /* vehicle.py:2025, 2070 */
        break;
    default:
        twod_dest_force(o, a, Fx, Fy);
    }
}

/* --------------------------------------------------------- controllers + integrators ---- */

/* vehicle.py:1218-1245 with dynamics.py:33-54 (ki = kd = 0) */
static void bike_control(csfo_t *o, int a, double Fx, double Fy, double *acc, double *omega) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a), *dest = qrow(o, a, o->ptr[a]);
    double theta = atan2(Fy, Fx);                                  /* :1223 */
    double v = sqrt(pow(Fx, 2) + pow(Fy, 2));                      /* :1224 */
    double ddest = sqrt(pow(dest[0] - s[0], 2) + pow(dest[1] - s[1], 2)); /* :1226-1229 */
    if (ddest < 3 && is_last_dest(o, a)) v = (v / 3) * ddest;      /* :1231-1232 */
    double target = csfo_angle_difference(s[2], theta);            /* :1235 */
    double ddelta = csfo_angle_difference(s[4], target);           /* :1239 */
    double dv = v - s[3];                                          /* :1240 */
    *omega = p->k_p_delta * ddelta;                                /* :1242 */
    *acc = p->k_p_v * dv;                                          /* :1243 */
}

/* vehicle.py:1247-1272 */
static void bike_move(csfo_t *o, int a, double acc, double omega) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a);
    acc = thresh(acc, p->a_max[0], p->a_max[1]);                   /* :1249 */
    double delta = csfo_limit_angle(s[4] + p->t_s * omega);        /* :1254 */
    double v = s[3] + p->t_s * acc;                                /* :1255 */
    delta = thresh(delta, -p->delta_max, p->delta_max);            /* :1257 */
    v = thresh(v, p->v_max_riding[0], p->v_max_riding[1]);         /* :1258 */
    double theta = s[2] + p->t_s * v * tan(delta) / p->l;          /* :1260 */
    theta = csfo_limit_angle(theta);                               /* :1262 */
    s[1] = s[1] + p->t_s * v * sin(theta);                         /* :1264 */
    s[0] = s[0] + p->t_s * v * cos(theta);                         /* :1265 */
    s[2] = theta;
    s[3] = v;
    s[4] = delta;
}

/* single-vehicle entry for known-answer tests: Bicycle.step / TwoDBicycle.step on a free agent */
void csfo_control_move(const csfo_params *p, const double *s_in, const double *dest, int is_last,
                       double Fx, double Fy, double *s_out) {
    csfo_t o;
    memset(&o, 0, sizeof o);
    o.p = *p;
    double s[6] = {0}, dq[6];
    int64_t qoff[2] = {0, is_last ? 1 : 2};
    int32_t ptr = 0;
    memcpy(s, s_in, 5 * sizeof(double));
    memcpy(dq, dest, 3 * sizeof(double));
    memcpy(dq + 3, dest, 3 * sizeof(double));
    o.s = s;
    o.dq = dq;
    o.qoff = qoff;
    o.ptr = &ptr;
    double acc, om;
    bike_control(&o, 0, Fx, Fy, &acc, &om);
    bike_move(&o, 0, acc, om);
    memcpy(s_out, s, 5 * sizeof(double));
}

/* ---- small dense matrix exponential: scaling and squaring with the [13/13] Pade approximant
 * (Higham 2005), the published algorithm behind scipy.linalg.expm which the python-control shim of
 * the golden generator uses.  n <= 6. */
#define MN 6
static void mmul(int n, const double *A, const double *B, double *C) {
    double T[MN * MN];
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double acc = 0;
            for (int k = 0; k < n; k++) acc += A[i * n + k] * B[k * n + j];
            T[i * n + j] = acc;
        }
    memcpy(C, T, sizeof(double) * n * n);
}

static int lu_solve(int n, double *A, double *B) { /* A X = B, in place, B is n x n */
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++)
            if (fabs(A[r * n + c]) > fabs(A[piv * n + c])) piv = r;
        if (A[piv * n + c] == 0.0) return 1;
        if (piv != c)
            for (int j = 0; j < n; j++) {
                double t = A[c * n + j];
                A[c * n + j] = A[piv * n + j];
                A[piv * n + j] = t;
                t = B[c * n + j];
                B[c * n + j] = B[piv * n + j];
                B[piv * n + j] = t;
            }
        for (int r = c + 1; r < n; r++) {
            double f = A[r * n + c] / A[c * n + c];
            for (int j = c; j < n; j++) A[r * n + j] -= f * A[c * n + j];
            for (int j = 0; j < n; j++) B[r * n + j] -= f * B[c * n + j];
        }
    }
    for (int r = n - 1; r >= 0; r--)
        for (int j = 0; j < n; j++) {
            double acc = B[r * n + j];
            for (int k = r + 1; k < n; k++) acc -= A[r * n + k] * B[k * n + j];
            B[r * n + j] = acc / A[r * n + r];
        }
    return 0;
}

void csfo_expm(int n, const double *Ain, double *E) {
    static const double b[14] = {64764752532480000., 32382376266240000., 7771770303897600.,
                                 1187353796428800.,  129060195264000.,   10559470521600.,
                                 670442572800.,      33522128640.,       1323241920.,
                                 40840800.,          960960.,            16380.,
                                 182.,               1.};
    double A[MN * MN], A2[MN * MN], A4[MN * MN], A6[MN * MN], U[MN * MN], V[MN * MN], T[MN * MN];
    double nrm = 0;
    for (int j = 0; j < n; j++) {
        double c = 0;
        for (int i = 0; i < n; i++) c += fabs(Ain[i * n + j]);
        if (c > nrm) nrm = c;
    }
    int sq = 0;
    if (nrm > 5.371920351148152) {
        sq = (int)ceil(log2(nrm / 5.371920351148152));
        if (sq < 0) sq = 0;
    }
    double sc = ldexp(1.0, -sq);
    for (int i = 0; i < n * n; i++) A[i] = Ain[i] * sc;
    mmul(n, A, A, A2);
    mmul(n, A2, A2, A4);
    mmul(n, A4, A2, A6);
    for (int i = 0; i < n * n; i++) T[i] = b[13] * A6[i] + b[11] * A4[i] + b[9] * A2[i];
    mmul(n, A6, T, T);
    for (int i = 0; i < n * n; i++) T[i] += b[7] * A6[i] + b[5] * A4[i] + b[3] * A2[i];
    for (int i = 0; i < n; i++) T[i * n + i] += b[1];
    mmul(n, A, T, U);
    for (int i = 0; i < n * n; i++) T[i] = b[12] * A6[i] + b[10] * A4[i] + b[8] * A2[i];
    mmul(n, A6, T, V);
    for (int i = 0; i < n * n; i++) V[i] += b[6] * A6[i] + b[4] * A4[i] + b[2] * A2[i];
    for (int i = 0; i < n; i++) V[i * n + i] += b[0];
    double P[MN * MN], Q[MN * MN];
    for (int i = 0; i < n * n; i++) {
        P[i] = V[i] + U[i];
        Q[i] = V[i] - U[i];
    }
    lu_solve(n, Q, P);
    for (int k = 0; k < sq; k++) mmul(n, P, P, P);
    memcpy(E, P, sizeof(double) * n * n);
}

/* vehicle.py:1738-1786 + parameters.py:1832-1892: closed loop A - B K_x, K_u B at speed v */
static void invpend_closed_loop(const csfo_params *p, double v, double A[25], double B[5]) {
    static const double kx[5][4] = {
        {3.48203226e02, -5.12057324e03, 1.58364873e04, -1.98073306e04},
        {-4.51700000e01, 0.0, 0.0, 0.0},
        {-9.16379250e02, 1.31769807e04, -6.57341643e04, 8.22163589e04},
        {3.20214069e02, -4.69953797e03, 1.66378680e04, -2.43114309e04},
        {2.87549256e-08, -2.27913445e03, 0.0, 0.0}};
    static const double ku[4] = {-3.38638984e-09, -2.27913445e+03, 0.0, 0.0};
    double vd[4] = {1, pow(v, -1), pow(v, -2), pow(v, -3)};        /* parameters.py:1885 */
    double Kx[5], Ku = 0;
    for (int r = 0; r < 5; r++) {
        Kx[r] = 0;
        for (int c = 0; c < 4; c++) Kx[r] += kx[r][c] * vd[c];
    }
    for (int c = 0; c < 4; c++) Ku += ku[c] * vd[c];
    double Ktau2 = (v * p->l_2) / (p->g * p->l);                   /* parameters.py:1850 */
    double K = (v * v) / (p->g * p->l);                            /* :1851 */
    double tau3 = p->l / v;                                        /* :1853 */
    double tau1sq = (p->i_bike_longlong + p->m * p->h * p->h) / (p->m * p->g * p->h); /* :1641-1643 */
    memset(A, 0, 25 * sizeof(double));
    A[0 * 5 + 1] = 1;                                              /* vehicle.py:1740-1760 */
    A[1 * 5 + 1] = -p->c_steer / p->i_steer_vertvert;
    A[2 * 5 + 3] = 1;
    A[3 * 5 + 0] = -K / tau1sq;
    A[3 * 5 + 1] = -Ktau2 / tau1sq;
    A[3 * 5 + 2] = 1 / tau1sq;
    A[4 * 5 + 0] = 1 / tau3;
    double Bo[5] = {0, 1 / p->i_steer_vertvert, 0, 0, 0};          /* :1762 */
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 5; c++) A[r * 5 + c] -= Bo[r] * Kx[c]; /* :1785 */
    for (int r = 0; r < 5; r++) B[r] = Ku * Bo[r];                 /* :1786 */
}

/* vehicle.py:1810-1848: one exact zero-order-hold step (control.forced_response over [0, t_s] with
 * constant input = the block matrix exponential [[A h, B h], [0, 0]]) */
static void invpend_step_yaw(csfo_t *o, int a, double Fx, double Fy, double *psi, double *delta,
                             double *theta) {
    const csfo_params *p = PA(o, a);
    double A[25], B[5], M[36], E[36];
    invpend_closed_loop(p, S(o, a)[3], A, B);                      /* :1829 (speed already updated) */
    double psi_d = atan2(Fy, Fx);                                  /* :1832 */
    memset(M, 0, sizeof M);
    for (int r = 0; r < 5; r++) {
        for (int c = 0; c < 5; c++) M[r * 6 + c] = A[r * 5 + c] * p->t_s;
        M[r * 6 + 5] = B[r] * p->t_s;
    }
    csfo_expm(6, M, E);
    double *x = o->xlti + 5 * a, xn[5];
    for (int r = 0; r < 5; r++) {
        double acc = E[r * 6 + 5] * psi_d;
        for (int c = 0; c < 5; c++) acc += E[r * 6 + c] * x[c];
        xn[r] = acc;
    }
    memcpy(x, xn, sizeof xn);                                      /* :1843 */
    *psi = csfo_limit_angle(xn[4]);                                /* :1844 (C = [0,0,0,0,1]) */
    *delta = csfo_limit_angle(xn[0]);                              /* :1845 */
    *theta = csfo_limit_angle(xn[2]);                              /* :1846 */
}

/* vehicle.py:1932-1950 */
static void invpend_update_riding_state(csfo_t *o, int a) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a);
    uint8_t *zr = o->zrid + 2 * a;
    int cvwalk = s[3] < p->v_max_walk;                             /* :1939 */
    int i = o->i[a];
    int imin = (int)(i - 1 / p->t_s);                              /* :1941 */
    if (imin < 0) imin = 0;
    int cdelta = 1;
    double *td = trj(o, a, 2);
    for (int k = imin; k <= i; k++)                                /* :1943-1947 */
        if (!(-p->delta_max_walk < td[k] && p->delta_max_walk > td[k])) cdelta = 0;
    int z0 = !cvwalk && ((zr[1] && cdelta) || zr[0]);              /* :1949 */
    zr[0] = (uint8_t)z0;
    zr[1] = (uint8_t)!z0;                                          /* :1950 */
}

/* dynamics.py:996-1079; the implicit-midpoint system is solved in closed form (yaw equation linear,
 * position explicit given yaw) where the reference runs MINPACK lm to ~1.5e-8 */
static void planarpoint_step(csfo_t *o, int a, double Fx, double Fy) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a), *x = o->xdyn + 3 * a;
    double vd = sqrt(Fx * Fx + Fy * Fy);                           /* :1018 */
    double acc = thresh(p->k_p_v * (vd - o->vdyn[a]), p->a_max[0], p->a_max[1]); /* :1021-1022 */
    double v = thresh(o->vdyn[a] + p->t_s * acc, p->v_max_riding[0], p->v_max_riding[1]); /* :1025 */
    double psi_c = csfo_limit_angle(atan2(Fy, Fx));                /* dynamics.py:115 */
    double vbar = (v + s[3]) / 2;                                  /* :1065 */
    double h = p->t_s, k = p->k_psi;
    double psi_n = (x[0] * (1 - h * k / 2) + h * k * psi_c) / (1 + h * k / 2);
    double pm = (x[0] + psi_n) / 2;
    x[1] = x[1] + h * vbar * cos(pm);
    x[2] = x[2] + h * vbar * sin(pm);
    x[0] = psi_n;
    o->vdyn[a] = v;                                                /* :1075-1076 */
    s[0] = x[1];                                                   /* :959-964 */
    s[1] = x[2];
    s[2] = csfo_limit_angle(x[0]);
    s[3] = v;
}

/* ------------------------------------------------------------- PlanarBicycle (f)4 ---- */

/* first-order-hold discretisation python-control's forced_response applies to a continuous system (the block matrix
 * exponential): x+ = Ad x + Bd0 u_k + Bd1 u_{k+1} for a single-input system with n = 2 states */
static void foh2(const double Acl[4], const double Bcl[2], double dt, double Ad[4], double Bd0[2], double Bd1[2]) {
    double M[16] = {0}, E[16];
    M[0] = Acl[0] * dt; M[1] = Acl[1] * dt; M[2] = Bcl[0] * dt;
    M[4] = Acl[2] * dt; M[5] = Acl[3] * dt; M[6] = Bcl[1] * dt;
    M[11] = 1.0;
    csfo_expm(4, M, E);
    Ad[0] = E[0]; Ad[1] = E[1]; Ad[2] = E[4]; Ad[3] = E[5];
    Bd1[0] = E[3]; Bd1[1] = E[7];
    Bd0[0] = E[2] - Bd1[0]; Bd0[1] = E[6] - Bd1[1];
}

/* PlanarTwoWheelerDynamics.update (dynamics.py:203-223) -> from_pole_placement (dynamics.py:1167-1226) for
 * A = [[0, 0], [v / w, 0]], B = [1, 0]^T, C = [0, 1]: K_x by pole placement (a single-input system has exactly one
 * solution, found here by matching the coefficients of the characteristic polynomial), K_u as the reciprocal of the
 * output at the end of a simulated response (T = 0, 0.01, ..., 9.99; input 0 for the first ten samples, then 1). */
void csfo_planarbike_gains(const csfo_params *p, double v, double Kx[2], double *Ku) {
    const double a10 = v / p->l;
    const double sum = p->pb_poles[0] + p->pb_poles[2];                                   /* p1 + p2 (real) */
    const double prod = p->pb_poles[0] * p->pb_poles[2] - p->pb_poles[1] * p->pb_poles[3]; /* Re(p1 p2) */
    Kx[0] = -sum;
    Kx[1] = prod / a10;
    const double Acl[4] = {-Kx[0], -Kx[1], a10, 0.0}, Bcl[2] = {1.0, 0.0};
    double Ad[4], Bd0[2], Bd1[2];
    foh2(Acl, Bcl, 0.01, Ad, Bd0, Bd1);
    double x0 = 0, x1 = 0;
    for (int i = 1; i < 1000; i++) {                               /* np.arange(10.0, step=0.01): 1000 samples */
        const double u0 = (i - 1) >= 10 ? 1.0 : 0.0, u1 = i >= 10 ? 1.0 : 0.0;
        const double n0 = Ad[0] * x0 + Ad[1] * x1 + Bd0[0] * u0 + Bd1[0] * u1;
        const double n1 = Ad[2] * x0 + Ad[3] * x1 + Bd0[1] * u0 + Bd1[1] * u1;
        x0 = n0;
        x1 = n1;
    }
    *Ku = 1.0 / x1;                                                /* dynamics.py:1222-1224 */
}

/* PlanarTwoWheelerDynamics.step (dynamics.py:225-258) + PPointSpeedDynamics.step (dynamics.py:160-175) */
static void planarbike_step(csfo_t *o, int a, double Fx, double Fy) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a), *x = o->xdyn + 3 * a;                      /* x = (delta, psi), unwrapped */
    double Kx[2], Ku;
    csfo_planarbike_gains(p, s[3], Kx, &Ku);                       /* :228: gains for the current speed */
    const double psi_d = atan2(Fy, Fx), v_d = sqrt(Fy * Fy + Fx * Fx);   /* :231-232 */
    const double Acl[4] = {-Kx[0], -Kx[1], s[3] / p->l, 0.0}, Bcl[2] = {Ku, 0.0};
    double Ad[4], Bd0[2], Bd1[2];
    foh2(Acl, Bcl, p->t_s, Ad, Bd0, Bd1);                          /* :235-243: input psi_d at both ends of the step */
    const double n0 = Ad[0] * x[0] + Ad[1] * x[1] + (Bd0[0] + Bd1[0]) * psi_d;
    const double n1 = Ad[2] * x[0] + Ad[3] * x[1] + (Bd0[1] + Bd1[1]) * psi_d;
    x[0] = n0;
    x[1] = n1;
    s[2] = csfo_limit_angle(x[1]);                                 /* :246-247 */
    s[4] = csfo_limit_angle(x[0]);
    s[3] = v_d + (s[3] - v_d) * exp(-p->k_p_v * p->t_s);           /* :156, 175 */
    const double y = s[1] + p->t_s * s[3] * sin(s[2]);             /* :251-258 */
    const double xx = s[0] + p->t_s * s[3] * cos(s[2]);
    s[0] = xx;
    s[1] = y;
}

/* ---------------------------------------------------------- BalancingRiderBicycle (f)4 ---- */

/* the bicycle-rider state matrix at speed v (dynamics.py:535-560; x = (roll, steer, roll rate, steer rate, yaw) in the bike
 * model's frame) and the steer-torque input column */
static void br_state_space(const csfo_params *p, double v, double A[25], double B[5]) {
    memset(A, 0, 25 * sizeof(double));
    A[0 * 5 + 2] = 1;
    A[1 * 5 + 3] = 1;
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) {
            A[(2 + r) * 5 + c] = -(p->br_minv_k0g[2 * r + c] + v * v * p->br_minv_k2[2 * r + c]);
            A[(2 + r) * 5 + 2 + c] = -v * p->br_minv_c1[2 * r + c];
        }
    A[4 * 5 + 1] = p->br_yaw[0] * v;                               /* :549 */
    A[4 * 5 + 3] = p->br_yaw[1];                                   /* :550 */
    B[0] = B[1] = B[4] = 0;
    B[2] = p->br_minv_steer[0];
    B[3] = p->br_minv_steer[1];
}

/* solve M x = b for n <= 5 by elimination with partial pivoting (M row-major, destroyed) */
static int solve_small(int n, double *M, double *b) {
    for (int k = 0; k < n; k++) {
        int piv = k;
        for (int r = k + 1; r < n; r++)
            if (fabs(M[r * n + k]) > fabs(M[piv * n + k])) piv = r;
        if (M[piv * n + k] == 0) return -1;
        if (piv != k) {
            for (int c = 0; c < n; c++) {
                double t = M[k * n + c];
                M[k * n + c] = M[piv * n + c];
                M[piv * n + c] = t;
            }
            double t = b[k];
            b[k] = b[piv];
            b[piv] = t;
        }
        for (int r = k + 1; r < n; r++) {
            double f = M[r * n + k] / M[k * n + k];
            for (int c = k; c < n; c++) M[r * n + c] -= f * M[k * n + c];
            b[r] -= f * b[k];
        }
    }
    for (int k = n - 1; k >= 0; k--) {
        double acc = b[k];
        for (int c = k + 1; c < n; c++) acc -= M[k * n + c] * b[c];
        b[k] = acc / M[k * n + k];
    }
    return 0;
}

/* dynamics.py:600-615: the gains that put the closed loop's poles where the control model wants them at speed v
 * (from_pole_placement -> control.place, dynamics.py:1167-1209).  One input: the placement has exactly one solution,
 * Ackermann's K = e_n^T W^-1 p(A), W = (B, A B, ..., A^4 B), p the polynomial of the desired poles - formed as products of
 * (A - p0) and (A^2 - 2 Re p A + |p|^2) applied to the row e_n^T W^-1 from the left. */
void csfo_balancingrider_gains(const csfo_params *p, double v, double K[5]) {
    if (p->br_mode == 2) {                                         /* :604-605 */
        memcpy(K, p->br_gains, 5 * sizeof(double));
        return;
    }
    double f[5];
    for (int i = 0; i < 5; i++) f[i] = p->br_pole_fun[2 * i] + p->br_pole_fun[2 * i + 1] * v;   /* parameters.py:1400-1409 */
    double A[25], B[5], W[25], col[5], y[5];
    br_state_space(p, v, A, B);
    memcpy(col, B, sizeof col);
    for (int k = 0; k < 5; k++) {                                  /* W^T, row k = (A^k B)^T */
        for (int r = 0; r < 5; r++) W[k * 5 + r] = col[r];
        double nxt[5];
        for (int r = 0; r < 5; r++) {
            double acc = 0;
            for (int c = 0; c < 5; c++) acc += A[r * 5 + c] * col[c];
            nxt[r] = acc;
        }
        memcpy(col, nxt, sizeof col);
    }
    for (int k = 0; k < 5; k++) y[k] = k == 4 ? 1.0 : 0.0;         /* W^T y = e_5 */
    if (solve_small(5, W, y) != 0) {
        for (int k = 0; k < 5; k++) K[k] = NAN;                    /* not controllable: the reference's assertion (:1205-1207) */
        return;
    }
#define ROW_TIMES_A(out, in)                                   \
    for (int c = 0; c < 5; c++) {                              \
        double acc = 0;                                        \
        for (int r = 0; r < 5; r++) acc += (in)[r] * A[r * 5 + c]; \
        (out)[c] = acc;                                        \
    }
    double t1[5], t2[5];
    ROW_TIMES_A(t1, y);                                            /* y (A - p0) */
    for (int c = 0; c < 5; c++) y[c] = t1[c] - f[0] * y[c];
    for (int pair = 0; pair < 2; pair++) {                         /* y (A^2 - 2 re A + (re^2 + im^2)) */
        const double re = f[1 + 2 * pair], im = f[2 + 2 * pair];
        ROW_TIMES_A(t1, y);
        ROW_TIMES_A(t2, t1);
        for (int c = 0; c < 5; c++) y[c] = t2[c] - 2 * re * t1[c] + (re * re + im * im) * y[c];
    }
#undef ROW_TIMES_A
    memcpy(K, y, 5 * sizeof(double));
}

/* BalancingRiderDynamics.step (dynamics.py:664-705).  xlti = (roll, steer, roll rate, steer rate, yaw) in the bike model's
 * frame (y to the right, z down: steer, yaw and y mirrored, :318-371), unwrapped; vdyn = the speed the gains in use were
 * computed for (they are renewed only when the speed changes, :671-673).  The implicit midpoint rule (:497-513) on
 * x' = (A - B K) x + B K_psi psi_c is a linear system in x+; the position follows from the yaw at both ends.  (The reference
 * solves the same seven equations with MINPACK's lm to its default tolerance, ~1e-8.) */
static void balancingrider_step(csfo_t *o, int a, double Fx, double Fy) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a), *x = o->xlti + 5 * a;
    const double h = p->t_s, v_old = s[3];
    double vd = sqrt(Fx * Fx + Fy * Fy);                           /* :640 */
    double acc = thresh(p->k_p_v * (vd - v_old), p->a_max[0], p->a_max[1]);   /* :643-644 */
    double v = thresh(v_old + h * acc, p->v_max_riding[0], p->v_max_riding[1]);   /* :647 */
    if (v != v_old) o->vdyn[a] = (v + v_old) / 2;                  /* :671-673 */
    double K[5];
    csfo_balancingrider_gains(p, o->vdyn[a], K);
    double psi_F = csfo_limit_angle(atan2(-Fy, Fx));               /* :656-658 (the lateral force mirrored) */
    double psi_c = x[4] + csfo_angle_difference(x[4], psi_F);      /* :660-663 */
    const double vbar = (v + v_old) / 2;                           /* :684, 687 */
    double A[25], B[5], L[25], rhs[5];
    br_state_space(p, vbar, A, B);
    for (int r = 0; r < 5; r++)
        for (int c = 0; c < 5; c++) A[r * 5 + c] -= B[r] * K[c];   /* :472 */
    for (int r = 0; r < 5; r++) {
        double accr = x[r] + h * B[r] * K[4] * psi_c;
        for (int c = 0; c < 5; c++) {
            L[r * 5 + c] = (r == c ? 1.0 : 0.0) - 0.5 * h * A[r * 5 + c];
            accr += 0.5 * h * A[r * 5 + c] * x[c];
        }
        rhs[r] = accr;
    }
    if (solve_small(5, L, rhs) != 0) {
        o->status[a] |= CSFO_ST_NAN;
        return;
    }
    const double pm = 0.5 * (x[4] + rhs[4]);
    const double px = s[0] + h * vbar * cos(pm), py = -s[1] + h * vbar * sin(pm);   /* :475-477 */
    memcpy(x, rhs, 5 * sizeof(double));
    s[0] = px;                                                     /* :337-348 */
    s[1] = -py;
    s[2] = -csfo_limit_angle(x[4]);
    s[3] = v;
    s[4] = -csfo_limit_angle(x[1]);
    s[5] = csfo_limit_angle(x[0]);
    s[6] = -x[3];
    s[7] = x[2];
}

/* --------------------------------------------------------------- population tick ---- */

/* intersection.py:747-864 for receivers [lo, hi) */
void csfo_calc_forces_range(csfo_t *o, int lo, int hi) {
    const csfo_params *p = &o->p;
    int n = o->n;
#pragma omp parallel for schedule(dynamic, 16)
    for (int j = lo; j < hi; j++) {
        double fdx, fdy;
        dest_force(o, j, &fdx, &fdy);                              /* :799 */
        double fx = fdx, fy = fdy;
        double rx = 0, ry = 0;
        if (n > 1) {                                               /* :813, :825 */
            for (int i = 0; i < n; i++) {
                const csfo_params *pi = PA(o, i);                  /* the field and the hfov of vehicle i: :733-735, 815 */
                if (csfo_untracked(pi->hfov, p->priority_rule, i, j, o->sx[i], o->sy[i], o->sx[j],
                                   o->sy[j], o->spsi[j]))
                    continue;                                      /* :815-823 */
                double gx, gy;
                if (pi->model == CSFO_BICYCLE)                    /* the field of vehicle i's class */
                    csfo_pair_bicycle(pi, o->sx[i], o->sy[i], o->spsi[i], o->sv[i], o->sx[j], o->sy[j], &gx, &gy);
                else
                    csfo_pair_twod(pi, o->sx[i], o->sy[i], o->spsi[i], o->sx[j], o->sy[j], o->spsi[j], &gx, &gy);
                rx += gx;                                          /* :842-843 column sum */
                ry += gy;
            }
            csfo_limit_magnitude(&rx, &ry, sqrt(fdx * fdx + fdy * fdy)); /* :841-845 */
            fx = rx + fdx;                                         /* :847-848 */
            fy = ry + fdy;
        }
        if (o->nv > 0) {                                           /* :854-857 */
            double ex, ey;
            csfo_road_force(o->nv, o->vx, o->vy, o->vF0, o->vsig, o->sx[j], o->sy[j], &ex, &ey);
            fx += ex;
            fy += ey;
        }
        o->Fdx[j] = fdx;
        o->Fdy[j] = fdy;
        o->Frx[j] = rx;
        o->Fry[j] = ry;
        o->Fx[j] = fx;                                             /* :860-861 */
        o->Fy[j] = fy;
        if (isnan(fx) || isnan(fy)) o->status[j] |= CSFO_ST_NAN;
    }
}

/* vehicle.*.step for agents [lo, hi) with the forces of csfo_calc_forces_range — intersection.py:891-892 */
void csfo_integrate_range(csfo_t *o, int lo, int hi) {
    int L = o->p.traj_len;
#pragma omp parallel for schedule(static)
    for (int a = lo; a < hi; a++) {
        const csfo_params *p = PA(o, a);
        double *s = S(o, a), Fx = o->Fx[a], Fy = o->Fy[a];
        double acc, om;
        switch (p->model) {
        case CSFO_BICYCLE:                                         /* vehicle.py:1274-1289 */
            bike_control(o, a, Fx, Fy, &acc, &om);
            bike_move(o, a, acc, om);
            break;
        case CSFO_TWOD:                                            /* vehicle.py:1386-1414 */
            if (o->znav[3 * a + 2]) {
                s[3] = 0;
                s[4] = 0;
            } else {
                bike_control(o, a, Fx, Fy, &acc, &om);
                bike_move(o, a, acc, om);
            }
            break;
        case CSFO_INVPEND:                                         /* vehicle.py:1883-1930 */
            invpend_update_riding_state(o, a);
            if (o->znav[3 * a + 2]) {
                s[3] = 0;
                s[4] = 0;
                s[5] = 0;
            } else if (o->zrid[2 * a]) {
                /* step_pos — vehicle.py:1850-1881 (old psi) */
                double vd = sqrt(Fx * Fx + Fy * Fy);
                double a_ = thresh(p->k_p_v * (vd - s[3]), p->a_max[0], p->a_max[1]);
                double v = thresh(s[3] + p->t_s * a_, p->v_max_riding[0], p->v_max_riding[1]);
                s[1] = s[1] + p->t_s * v * sin(s[2]);
                s[0] = s[0] + p->t_s * v * cos(s[2]);
                s[3] = v;
                double psi, de, th;
                invpend_step_yaw(o, a, Fx, Fy, &psi, &de, &th);
                s[2] = psi;
                s[4] = de;
                s[5] = th;
            } else {                                               /* :1905-1916 walking */
                s[3] = p->v_max_walk;
                s[5] = 0;
                bike_control(o, a, Fx, Fy, &acc, &om);
                bike_move(o, a, acc, om);
                double *x = o->xlti + 5 * a;
                x[0] = s[4];
                x[1] = 0;
                x[2] = s[5];
                x[3] = 0;
                x[4] = s[2];
            }
            break;
        case CSFO_PLANARPOINT:                                     /* vehicle.py:301-328 */
            planarpoint_step(o, a, Fx, Fy);
            break;
        case CSFO_PLANARBIKE:
            planarbike_step(o, a, Fx, Fy);
            break;
        case CSFO_BALANCINGRIDER:
            balancingrider_step(o, a, Fx, Fy);
            break;
        case CSFO_UNCONTROLLED: {                                  /* vehicle.py:964-979: the next prescribed state, if any */
            int64_t i = (int64_t)o->i[a] + 1;                      /* (:973: the counter never wraps) */
            if (i > 2000000000) i = 2000000000;
            o->i[a] = (int32_t)i;
            int64_t cols = o->soff ? o->soff[a + 1] - o->soff[a] : 0;
            if (cols > 0) {
                if (i < cols) memcpy(s, o->script + 4 * (o->soff[a] + i), 4 * sizeof(double));
            } else if (i < L) {                                    /* no trajectory given: Vehicle.__init__'s ring of zeros (:158-160) */
                s[0] = s[1] = s[2] = s[3] = 0;
            }
            continue;
        }
        }
        int i = (o->i[a] + 1) % L;                                 /* vehicle.py:1279-1282, D5 */
        o->i[a] = i;
        trj(o, a, 0)[i] = s[0];
        trj(o, a, 1)[i] = s[1];
        trj(o, a, 2)[i] = s[4];
    }
}

/* intersection.py:660-677 */
void csfo_update_snapshot_range(csfo_t *o, int lo, int hi) {
    for (int a = lo; a < hi; a++) {
        o->sx[a] = S(o, a)[0];
        o->sy[a] = S(o, a)[1];
        o->spsi[a] = S(o, a)[2];
        o->sv[a] = S(o, a)[3];
    }
}

/* intersection.py:866-896 */
void csfo_step(csfo_t *o, int nticks) {
    for (int t = 0; t < nticks; t++) {
        if (o->n > 0) {
            csfo_calc_forces_range(o, 0, o->n);
            csfo_integrate_range(o, 0, o->n);
            csfo_update_snapshot_range(o, 0, o->n);
        }
        o->tick++;
    }
}

/* ----------------------------------------------------------------- construction ---- */

static const int NS[7] = {5, 5, 6, 4, 5, 4, 8};

/* what the constructors of the rider classes derive from the start state (vehicle.py:1728-1736; dynamics.py:195-197, 828,
 * 987-993), by the class of agent a */
static void init_side_state(csfo_t *o, int a) {
    const csfo_params *p = PA(o, a);
    double *s = S(o, a);
    o->zrid[2 * a] = o->zrid[2 * a + 1] = 0;
    if (p->model == CSFO_INVPEND) {                                /* vehicle.py:1728-1736 */
        double *x = o->xlti + 5 * a;
        x[0] = s[4];
        x[1] = 0;
        x[2] = s[5];
        x[3] = 0;
        x[4] = s[2];
        if (s[3] < p->v_max_walk) o->zrid[2 * a + 1] = 1;
        else o->zrid[2 * a] = 1;
    }
    if (p->model == CSFO_BALANCINGRIDER) {                         /* dynamics.py:306-307, 350-371 */
        double *x = o->xlti + 5 * a;
        x[0] = s[5];
        x[1] = -s[4];
        x[2] = s[7];
        x[3] = -s[6];
        x[4] = -s[2];
        o->vdyn[a] = s[3];                                         /* :307: the gains of the start speed */
    }
    if (p->model == CSFO_PLANARBIKE) {                             /* dynamics.py:195-197 */
        o->xdyn[3 * a] = s[4];
        o->xdyn[3 * a + 1] = s[2];
    }
    if (p->model == CSFO_PLANARPOINT) {                            /* dynamics.py:828, 987-993 */
        o->xdyn[3 * a] = s[2];
        o->xdyn[3 * a + 1] = s[0];
        o->xdyn[3 * a + 2] = s[1];
        o->vdyn[a] = s[3];
    }
}

/* ns: columns of s0 / of csfo_get_state (a population of several vehicle classes - csfo_set_classes - uses 6, the unused
 * states of a class stay 0) */
csfo_t *csfo_create_ns(const csfo_params *p, int n, int ns, const double *s0, const double *vdes,
                       const int64_t *qoff, const double *dq) {
    csfo_t *o = (csfo_t *)calloc(1, sizeof *o);
    o->p = *p;
    o->n = n;
    o->ns = ns;
    size_t N = (size_t)(n > 0 ? n : 1), L = (size_t)p->traj_len;
    o->s = (double *)calloc(N * CSFO_NS_MAX, sizeof(double));
    o->vdes = (double *)calloc(N, sizeof(double));
    o->qoff = (int64_t *)calloc(N + 1, sizeof(int64_t));
    memcpy(o->qoff, qoff, sizeof(int64_t) * (size_t)(n + 1));
    size_t Q = (size_t)qoff[n];
    o->dq = (double *)calloc((Q ? Q : 1) * 3, sizeof(double));
    memcpy(o->dq, dq, sizeof(double) * Q * 3);
    o->ptr = (int32_t *)calloc(N, sizeof(int32_t));
    o->znav = (uint8_t *)calloc(N * 3, 1);
    o->znavp = (double *)calloc(N * 4, sizeof(double));
    o->i = (int32_t *)calloc(N, sizeof(int32_t));
    o->traj = (double *)calloc(N * 3 * L, sizeof(double));
    o->xlti = (double *)calloc(N * 5, sizeof(double));
    o->zrid = (uint8_t *)calloc(N * 2, 1);
    o->xdyn = (double *)calloc(N * 3, sizeof(double));
    o->vdyn = (double *)calloc(N, sizeof(double));
    o->sx = (double *)calloc(N, sizeof(double));
    o->sy = (double *)calloc(N, sizeof(double));
    o->spsi = (double *)calloc(N, sizeof(double));
    o->sv = (double *)calloc(N, sizeof(double));
    o->Fx = (double *)calloc(N, sizeof(double));
    o->Fy = (double *)calloc(N, sizeof(double));
    o->Fdx = (double *)calloc(N, sizeof(double));
    o->Fdy = (double *)calloc(N, sizeof(double));
    o->Frx = (double *)calloc(N, sizeof(double));
    o->Fry = (double *)calloc(N, sizeof(double));
    o->status = (uint32_t *)calloc(N, sizeof(uint32_t));
    for (int a = 0; a < n; a++) {
        double *s = S(o, a);
        for (int k = 0; k < o->ns; k++) s[k] = s0[(size_t)a * o->ns + k];
        s[2] = csfo_limit_angle(s[2]);                             /* vehicle.py:154-155 */
        o->vdes[a] = vdes[a];
        o->znav[3 * a] = 1;                                        /* vehicle.py:188 */
        trj(o, a, 0)[0] = s[0];                                    /* vehicle.py:159-160 */
        trj(o, a, 1)[0] = s[1];
        trj(o, a, 2)[0] = s[4];
        init_side_state(o, a);
    }
    csfo_update_snapshot_range(o, 0, n);                           /* intersection.py:320 */
    return o;
}

csfo_t *csfo_create(const csfo_params *p, int n, const double *s0, const double *vdes,
                    const int64_t *qoff, const double *dq) {
    return csfo_create_ns(p, n, NS[p->model], s0, vdes, qoff, dq);
}

/* Test aid (no reference counterpart): re-anchor the oracle on a state produced elsewhere - what csf_push_state does for
 * the engine (calibration.py:455-460 edits vehicle.s the same way): s [n][ns] replaces vehicle.s, the current column of
 * the trajectory ring and the position snapshot; ptr / znav / col (optional) replace the destination pointer, the
 * navigation state and the ring column vehicle.i.  The dynamics of a chaotic population cannot be compared point by point over thousands of ticks;
 * the long-run parity tests re-anchor every 100 ticks and compare the segments in between. */
void csfo_push_state(csfo_t *o, const double *s_in, const int32_t *ptr, const uint8_t *znav, const int32_t *col) {
    for (int a = 0; a < o->n; a++) {
        double *s = S(o, a);
        for (int k = 0; k < o->ns; k++) s[k] = s_in[(size_t)a * o->ns + k];
        if (col) o->i[a] = col[a];                                  /* the ring column vehicle.i (vehicle.py:1279-1280) */
        int i = o->i[a];
        trj(o, a, 0)[i] = s[0];
        trj(o, a, 1)[i] = s[1];
        trj(o, a, 2)[i] = s[4];
        /* the integrators keep angles UNWRAPPED (vehicle.py:1844-1846, dynamics.py:943-966) while s holds them wrapped: keep
         * the winding number the oracle has, i.e. the multiple of 2 pi nearest to its own unwrapped value */
#define REWIND(own, wrapped) ((wrapped) + 2 * PI * nearbyint(((own) - (wrapped)) / (2 * PI)))
        if (PA(o, a)->model == CSFO_INVPEND) {
            double *x = o->xlti + 5 * a;
            x[0] = REWIND(x[0], s[4]);
            x[2] = REWIND(x[2], s[5]);
            x[4] = REWIND(x[4], s[2]);
        }
        if (PA(o, a)->model == CSFO_BALANCINGRIDER) {             /* (dynamics.py:350-371; the rates are states of vehicle.s here) */
            double *x = o->xlti + 5 * a;
            x[0] = REWIND(x[0], s[5]);
            x[1] = REWIND(x[1], -s[4]);
            x[2] = s[7];
            x[3] = -s[6];
            x[4] = REWIND(x[4], -s[2]);
        }
        if (PA(o, a)->model == CSFO_PLANARBIKE) {
            o->xdyn[3 * a] = REWIND(o->xdyn[3 * a], s[4]);
            o->xdyn[3 * a + 1] = REWIND(o->xdyn[3 * a + 1], s[2]);
        }
        if (PA(o, a)->model == CSFO_PLANARPOINT) {
            o->xdyn[3 * a] = REWIND(o->xdyn[3 * a], s[2]);
            o->xdyn[3 * a + 1] = s[0];
            o->xdyn[3 * a + 2] = s[1];
            o->vdyn[a] = s[3];
        }
#undef REWIND
        if (ptr) o->ptr[a] = ptr[a];
        if (znav)
            for (int k = 0; k < 3; k++) o->znav[3 * a + k] = znav[3 * a + k];
    }
    csfo_update_snapshot_range(o, 0, o->n);
}

/* the prescribed trajectories of UncontrolledVehicle agents (vehicle.py:958-960): CSR over all n agents, rows (x, y, psi, v) */
void csfo_set_script(csfo_t *o, const int64_t *off, const double *rows) {
    free(o->soff);
    free(o->script);
    o->soff = (int64_t *)malloc((size_t)(o->n + 1) * sizeof(int64_t));
    memcpy(o->soff, off, (size_t)(o->n + 1) * sizeof(int64_t));
    o->script = (double *)malloc((size_t)(4 * off[o->n] + 1) * sizeof(double));
    memcpy(o->script, rows, (size_t)(4 * off[o->n]) * sizeof(double));
}

void csfo_destroy(csfo_t *o) {
    if (!o) return;
    free(o->s); free(o->vdes); free(o->qoff); free(o->dq); free(o->ptr); free(o->znav);
    free(o->znavp); free(o->i); free(o->traj); free(o->xlti); free(o->zrid); free(o->xdyn);
    free(o->vdyn); free(o->sx); free(o->sy); free(o->spsi); free(o->sv); free(o->Fx); free(o->Fy);
    free(o->Fdx); free(o->Fdy); free(o->Frx); free(o->Fry); free(o->status); free(o->soff); free(o->script);
    free(o->vx); free(o->vy); free(o->vF0); free(o->vsig);
    free(o->ptab); free(o->pcls);
    free(o);
}

/* every reference vehicle owns its params object (vehicle.py:64-204) and any vehicle classes may share an intersection
 * (intersection.py:797-823 calls each vehicle's own methods): a table of parameter sets - each with its model - and the set
 * of every agent.  t_s and traj_len are those of the population (the sets of one Scenario share the clock). */
void csfo_set_classes(csfo_t *o, int n_classes, const csfo_params *tab, const uint8_t *cls) {
    free(o->ptab);
    free(o->pcls);
    o->ptab = NULL;
    o->pcls = NULL;
    o->n_classes = 0;
    if (n_classes <= 0) return;
    o->n_classes = n_classes;
    o->ptab = (csfo_params *)malloc(sizeof(csfo_params) * (size_t)n_classes);
    memcpy(o->ptab, tab, sizeof(csfo_params) * (size_t)n_classes);
    o->pcls = (uint8_t *)malloc((size_t)(o->n > 0 ? o->n : 1));
    memcpy(o->pcls, cls, (size_t)o->n);
    for (int a = 0; a < o->n; a++) init_side_state(o, a);         /* (the sets may be of other vehicle classes: before any tick) */
}

/* edges as CSR over vertices with one (F0, sigma) per edge — intersection.py:222-224 */
void csfo_set_road(csfo_t *o, int n_edges, const int64_t *off, const double *xy, const double *F0,
                   const double *sigma) {
    free(o->vx); free(o->vy); free(o->vF0); free(o->vsig);
    int64_t nv = n_edges > 0 ? off[n_edges] : 0;
    o->nv = nv;
    o->vx = (double *)calloc((size_t)(nv ? nv : 1), sizeof(double));
    o->vy = (double *)calloc((size_t)(nv ? nv : 1), sizeof(double));
    o->vF0 = (double *)calloc((size_t)(nv ? nv : 1), sizeof(double));
    o->vsig = (double *)calloc((size_t)(nv ? nv : 1), sizeof(double));
    for (int e = 0; e < n_edges; e++)
        for (int64_t k = off[e]; k < off[e + 1]; k++) {
            o->vx[k] = xy[2 * k];
            o->vy[k] = xy[2 * k + 1];
            o->vF0[k] = F0[e];
            o->vsig[k] = sigma[e];
        }
}

/* --------------------------------------------------------------------- accessors ---- */

void csfo_get_state(csfo_t *o, double *s_out) {
    for (int a = 0; a < o->n; a++)
        for (int k = 0; k < o->ns; k++) s_out[(size_t)a * o->ns + k] = S(o, a)[k];
}
void csfo_get_forces(csfo_t *o, double *Fx, double *Fy) {
    memcpy(Fx, o->Fx, sizeof(double) * (size_t)o->n);
    memcpy(Fy, o->Fy, sizeof(double) * (size_t)o->n);
}
void csfo_get_force_parts(csfo_t *o, double *Fdx, double *Fdy, double *Frx, double *Fry) {
    memcpy(Fdx, o->Fdx, sizeof(double) * (size_t)o->n);
    memcpy(Fdy, o->Fdy, sizeof(double) * (size_t)o->n);
    memcpy(Frx, o->Frx, sizeof(double) * (size_t)o->n);
    memcpy(Fry, o->Fry, sizeof(double) * (size_t)o->n);
}
void csfo_get_nav(csfo_t *o, int32_t *ptr, uint8_t *znav, int32_t *i, uint32_t *status) {
    memcpy(ptr, o->ptr, sizeof(int32_t) * (size_t)o->n);
    memcpy(znav, o->znav, 3 * (size_t)o->n);
    memcpy(i, o->i, sizeof(int32_t) * (size_t)o->n);
    memcpy(status, o->status, sizeof(uint32_t) * (size_t)o->n);
}
/* snapshot exchange for sharded (multi-rank) tests: (x, y, psi, v) per agent */
void csfo_get_snapshot(csfo_t *o, int lo, int hi, double *out) {
    for (int a = lo; a < hi; a++) {
        out[4 * (a - lo) + 0] = o->sx[a];
        out[4 * (a - lo) + 1] = o->sy[a];
        out[4 * (a - lo) + 2] = o->spsi[a];
        out[4 * (a - lo) + 3] = o->sv[a];
    }
}
void csfo_set_snapshot(csfo_t *o, int lo, int hi, const double *in) {
    for (int a = lo; a < hi; a++) {
        o->sx[a] = in[4 * (a - lo) + 0];
        o->sy[a] = in[4 * (a - lo) + 1];
        o->spsi[a] = in[4 * (a - lo) + 2];
        o->sv[a] = in[4 * (a - lo) + 3];
    }
}
/* destination force of one agent from its current state (mutates queue pointer + nav state) */
void csfo_dest_force(csfo_t *o, int a, double *Fx, double *Fy) { dest_force(o, a, Fx, Fy); }
/* apply given forces to all agents (single-vehicle closed-loop tests: vehicle.step(Fx, Fy)) */
void csfo_apply_forces(csfo_t *o, const double *Fx, const double *Fy) {
    memcpy(o->Fx, Fx, sizeof(double) * (size_t)o->n);
    memcpy(o->Fy, Fy, sizeof(double) * (size_t)o->n);
    csfo_integrate_range(o, 0, o->n);
    csfo_update_snapshot_range(o, 0, o->n);
    o->tick++;
}
/* Test aid (no reference counterpart): the hidden state of the InvPendulum integrator - vehicle.x (vehicle.py:1728-1733:
 * delta, ddelta, theta, dtheta, psi UNWRAPPED) and the riding state vehicle.zrid (:1735-1736) - for studies that compare two
 * runs state by state (tools/invpend_cut_study.py) and for anchors that carry the winding number over */
void csfo_get_lti(const csfo_t *o, double *x_out, uint8_t *zrid_out) {
    memcpy(x_out, o->xlti, sizeof(double) * 5 * (size_t)o->n);
    memcpy(zrid_out, o->zrid, 2 * (size_t)o->n);
}
void csfo_set_lti(csfo_t *o, const double *x_in, const uint8_t *zrid_in) {
    if (x_in) memcpy(o->xlti, x_in, sizeof(double) * 5 * (size_t)o->n);
    if (zrid_in) memcpy(o->zrid, zrid_in, 2 * (size_t)o->n);
}
void csfo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#endif
}

int csfo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
size_t csfo_sizeof_params(void) { return sizeof(csfo_params); }
