// csf_road.hip — the road term of large, static road networks: near field summed, far field interpolated.
//
// intersection.py:226-242 sums -F0 r^-sigma (v - p)/r over EVERY vertex v of every road element for every road user p, every
// tick (config 5: 1 048 576 x 391 680 pairs, 98 ms in road_kernel at its VALU roofline).  The vertices never move.  A
// lattice of square cells (edge w) is laid over the network once; for a receiver in cell C
//     F(p) = sum over the vertices of the (2 RG_NEAR + 1)^2 cells around C, exactly as road_kernel sums them
//          + sum_ab c_ab(C) T_a(xi) T_b(eta),   (xi, eta) = p in the cell's own coordinates, [-1, 1]^2,
// where the second line interpolates the force of ALL OTHER vertices - at least RG_NEAR w away from any point of C - at
// the 8 x 8 Chebyshev nodes of C.  That far field is analytic on the cell with its nearest singularity (a vertex, in the
// complex extension of either coordinate) >= 2 w from the interval of length w, i.e. on a Bernstein ellipse of parameter
// rho >= 4 + sqrt(17) = 8.1: the degree-7 interpolant is off by O(rho^-8) = 5e-8 of the far field (measured:
// tests/test_gpu_large.py).  The samples are direct sums (road_far_kernel, fp32 pairs added up in fp64), so any sigma and
// F0 per edge are covered; the coefficients are their cosine transform (csf_engine.hip: build_road_grid).
// A receiver outside the lattice has no interpolant: it sums every vertex (as road_kernel would).
#include "csf_field.h"

namespace csf {

constexpr int BLOCK = 256, WPB = 4, RPW = 4, TILE = 1024;

// -F0 r^-(sigma+1) (v - p) for two vertices per lane (packed), added to (ax, ay): road_kernel's inner expression
// (intersection.py:231-240).  NP = sigma + 1 for one small integer sigma, 0: per-vertex exponent pw = -(sigma+1)/2.
template <int NP>
__device__ __forceinline__ void road_pair(const v2f px, const v2f py, const v2f pf, const v2f pw, float rx, float ry,
                                          v2f &ax, v2f &ay) {
    const v2f ex = px - rx, ey = py - ry;
    const v2f r2 = ex * ex + ey * ey;
    v2f m;
    if (NP) {
        v2f inv = rsq2(r2);
        inv = __builtin_elementwise_min(inv, v2f{1e6f, 1e6f});       // r = 0: finite, times ex = ey = 0
        const v2f i2 = inv * inv;
        m = NP == 2 ? i2 : NP == 3 ? i2 * inv : NP == 4 ? i2 * i2 : NP == 5 ? i2 * i2 * inv : i2 * i2 * i2;
    } else {
        v2f lg = pw * v2f{fast_log2(r2.x), fast_log2(r2.y)};
        lg = __builtin_elementwise_min(lg, v2f{120.f, 120.f});
        m = v2f{fast_exp2(lg.x), fast_exp2(lg.y)};
    }
    m = m * pf;
    ax = m * ex + ax;
    ay = m * ey + ay;
}

__device__ __forceinline__ float cheb_node(int k) {   // cos(pi (k + 1/2) / 8)
    constexpr float N[RG_NODES] = {0.98078528040323043f, 0.83146961230254524f, 0.55557023301960218f, 0.19509032201612825f,
                                   -0.19509032201612825f, -0.55557023301960218f, -0.83146961230254524f, -0.98078528040323043f};
    float v = N[0];
#pragma unroll
    for (int q = 1; q < RG_NODES; q++) v = k == q ? N[q] : v;
    return v;
}

// ---- once per network: the far field of every cell at its Chebyshev nodes --------------------------------------------------
// grid (4, cells): a workgroup holds 16 of the cell's 64 nodes (four per wave, wave-uniform) and streams every vertex of the
// network through LDS; a vertex of the cells around the workgroup's own is given F0 = 0.  Vertex positions are formed
// relative to the centre of the workgroup's cell: (offset in its own cell) + (cell distance) w, fp32 - at >= 2 w from the node
// a relative 1e-6.  The partial sums of a tile are added up in fp64.
template <int NP>
__global__ __launch_bounds__(BLOCK) void road_far_kernel(const Dev d, const short2 *vcell, double *samples) {
    __shared__ float vx[TILE], vy[TILE], vf[TILE], vw[NP ? 1 : TILE];
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cell = blockIdx.x, cx = cell % d.rg_nx, cy = cell / d.rg_nx;
    const float hw = 0.5f * d.rg_w;
    float rx[RPW], ry[RPW];
    double sx[RPW], sy[RPW];
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        const int node = ((int)blockIdx.y * WPB + wave) * RPW + u;   // 8 ix + iy
        rx[u] = hw * cheb_node(node >> 3);
        ry[u] = hw * cheb_node(node & 7);
        asm volatile("" : "+v"(rx[u]), "+v"(ry[u]));
        sx[u] = sy[u] = 0.0;
    }
    for (int64_t base = 0; base < d.nv; base += TILE) {
        const int cnt = (int)((d.nv - base) < TILE ? (d.nv - base) : TILE);
        __syncthreads();
        for (int t = threadIdx.x; t < TILE; t += BLOCK) {
            float4 v = make_float4(0.f, 0.f, 0.f, -1.f);
            float ox = 1e9f, oy = 1e9f;                          // (padding: far away, F0 = 0)
            bool nearby = true;
            if (t < cnt) {
                v = d.rg_v[base + t];
                const short2 c = vcell[base + t];
                const int dxc = (int)c.x - cx, dyc = (int)c.y - cy;
                nearby = dxc >= -RG_NEAR && dxc <= RG_NEAR && dyc >= -RG_NEAR && dyc <= RG_NEAR;
                ox = v.x + (float)dxc * d.rg_w;
                oy = v.y + (float)dyc * d.rg_w;
            }
            vx[t] = ox;
            vy[t] = oy;
            vf[t] = nearby ? 0.0f : v.z;
            if (!NP) vw[t] = v.w;
        }
        __syncthreads();
        v2f ax[RPW], ay[RPW];
#pragma unroll
        for (int u = 0; u < RPW; u++) ax[u] = ay[u] = v2f{0.f, 0.f};
        const int cnt2 = (cnt + 2 * WAVE - 1) / (2 * WAVE) * (2 * WAVE);
        for (int t = lane; t < cnt2; t += 2 * WAVE) {
            const v2f px{vx[t], vx[t + WAVE]}, py{vy[t], vy[t + WAVE]}, pf{vf[t], vf[t + WAVE]};
            v2f pw{0.f, 0.f};
            if (!NP) pw = v2f{vw[t], vw[t + WAVE]};
#pragma unroll
            for (int u = 0; u < RPW; u++) road_pair<NP>(px, py, pf, pw, rx[u], ry[u], ax[u], ay[u]);
        }
#pragma unroll
        for (int u = 0; u < RPW; u++) {
            sx[u] += (double)ax[u].x + (double)ax[u].y;
            sy[u] += (double)ay[u].x + (double)ay[u].y;
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        double a = sx[u], b = sy[u];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            a += __shfl_xor(a, o, WAVE);
            b += __shfl_xor(b, o, WAVE);
        }
        if (lane == 0) {
            const int node = ((int)blockIdx.y * WPB + wave) * RPW + u;
            samples[((int64_t)cell * 64 + node) * 2] = a;
            samples[((int64_t)cell * 64 + node) * 2 + 1] = b;
        }
    }
}

// ---- every tick -----------------------------------------------------------------------------------------------------------
// A wave takes four receivers - consecutive places of the binned order where there is one (neighbours: mostly one cell), else
// consecutive slots - and works through them cell by cell: the receivers of one cell share the loads of that cell's
// neighbourhood, vertices in the lanes (two per lane, packed), straight from L2 (a neighbourhood is some thousand vertices,
// the network a few MB).  Then lane 8 a + b adds its term c_ab T_a T_b of the far field, and the wave sums up.
template <int NP>
__global__ __launch_bounds__(BLOCK) void road_grid_kernel(const Dev d) {
    const int lane = threadIdx.x & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // rg_by_place: 0 the block's slots in turn; 1 every place of the binned order (whole population); 2 the places of THIS rank's
    // receivers in ascending order (Dev::rlist: a shard whose receivers are taken in binned order) - neighbours in a wave either way
    const int64_t jend = d.rg_by_place == 1 ? d.n_pad : d.rg_by_place == 2 ? d.hi - d.lo : d.hi;
    const int64_t j0 = (d.rg_by_place ? 0 : d.lo) + ((int64_t)blockIdx.x * WPB + wave) * RPW;
    if (j0 >= jend) return;
    const float w = d.rg_w, hw = 0.5f * w, iw = 1.0f / w;
    float xr[RPW], yr[RPW];       // the receiver relative to the centre of its cell (wave-uniform, in VGPRs)
    int ci[RPW], cj[RPW];         // its cell (scalars; any integer: a receiver outside the lattice sits in a virtual cell)
    int32_t slot[RPW];
    unsigned real = 0;
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        const int64_t j = j0 + u < jend ? j0 + u : jend - 1;
        const int32_t a = d.rg_by_place == 1 ? d.perm[j] : d.rg_by_place == 2 ? d.perm[d.rlist[j]] : (int32_t)j;
        const float4 q = d.rec[a];
        const float2 o = d.rorg[a];
        const bool ok = j0 + u < jend && rec_is_real(q) && a >= d.lo && a < d.hi;
        float hx, lx, hy, ly;         // scene coordinates as two floats each, hi + lo = origin + offset exactly
        two_sum(o.x, q.x, hx, lx);
        two_sum(o.y, q.y, hy, ly);
        const float gx = ((hx - d.rg_x0) + lx) * iw, gy = ((hy - d.rg_y0) + ly) * iw;
        const float fi = ok ? fminf(fmaxf(floorf(gx), -1e6f), 1e6f) : 0.0f, fj = ok ? fminf(fmaxf(floorf(gy), -1e6f), 1e6f) : 0.0f;
        ci[u] = __builtin_amdgcn_readfirstlane((int)fi);
        cj[u] = __builtin_amdgcn_readfirstlane((int)fj);
        // centre of the cell: corner + (index + 1/2) w - multiples of w/2, exact in fp32
        xr[u] = (hx - (d.rg_x0 + ((float)ci[u] + 0.5f) * w)) + lx;
        yr[u] = (hy - (d.rg_y0 + ((float)cj[u] + 0.5f) * w)) + ly;
        slot[u] = a;
        real |= (__builtin_amdgcn_readfirstlane((int)ok) ? 1u : 0u) << u;
    }
    v2f ax[RPW], ay[RPW];
#pragma unroll
    for (int u = 0; u < RPW; u++) ax[u] = ay[u] = v2f{0.f, 0.f};
    unsigned todo = real;
    while (todo) {
        const int u0 = __builtin_ctz(todo);
        int c0i = ci[0], c0j = cj[0];
#pragma unroll
        for (int u = 1; u < RPW; u++) c0i = u0 == u ? ci[u] : c0i, c0j = u0 == u ? cj[u] : c0j;
        unsigned grp = 0;                  // the receivers of this cell
#pragma unroll
        for (int u = 0; u < RPW; u++) grp |= ((todo >> u & 1u) && ci[u] == c0i && cj[u] == c0j) ? 1u << u : 0u;
        todo &= ~grp;
        const bool inside = c0i >= 0 && c0i < d.rg_nx && c0j >= 0 && c0j < d.rg_ny;
        const int xa = inside ? max(c0i - RG_NEAR, 0) : 0, xb = inside ? min(c0i + RG_NEAR, d.rg_nx - 1) : d.rg_nx - 1;
        const int ya = inside ? max(c0j - RG_NEAR, 0) : 0, yb = inside ? min(c0j + RG_NEAR, d.rg_ny - 1) : d.rg_ny - 1;
        for (int yy = ya; yy <= yb; yy++) {
            for (int xx = xa; xx <= xb; xx++) {
                const int c = yy * d.rg_nx + xx;
                const int b = d.rg_start[c], e = d.rg_start[c + 1];
                if (b == e) continue;
                const float shx = (float)(xx - c0i) * w, shy = (float)(yy - c0j) * w;   // that cell's centre seen from ours
                float px_[RPW], py_[RPW];
#pragma unroll
                for (int u = 0; u < RPW; u++) px_[u] = xr[u] - shx, py_[u] = yr[u] - shy;
                for (int t = b + lane; t < e; t += 2 * WAVE) {
                    const bool two = t + WAVE < e;
                    const float4 v1 = d.rg_v[t], v2 = d.rg_v[two ? t + WAVE : t];
                    const v2f px{v1.x, v2.x}, py{v1.y, v2.y}, pf{v1.z, two ? v2.z : 0.0f}, pw{v1.w, v2.w};
#pragma unroll
                    for (int u = 0; u < RPW; u++)
                        if (grp >> u & 1u) road_pair<NP>(px, py, pf, pw, px_[u], py_[u], ax[u], ay[u]);
                }
            }
        }
        if (inside) {                      // the rest of the network: lane 8 a + b holds c_ab of both components
            const float *cc = d.rg_c + ((int64_t)(c0j * d.rg_nx + c0i) * 2) * 64;
            const float cxl = cc[lane], cyl = cc[64 + lane];
            const int ka = lane >> 3, kb = lane & 7;
#pragma unroll
            for (int u = 0; u < RPW; u++) {
                if (!(grp >> u & 1u)) continue;
                const float xi = xr[u] / hw, eta = yr[u] / hw;
                float tpx = 1.0f, tcx = xi, tax = ka == 0 ? 1.0f : xi;
                float tpy = 1.0f, tcy = eta, tby = kb == 0 ? 1.0f : eta;
#pragma unroll
                for (int k = 2; k < RG_NODES; k++) {
                    const float nx_ = 2.0f * xi * tcx - tpx, ny_ = 2.0f * eta * tcy - tpy;
                    tpx = tcx, tcx = nx_, tpy = tcy, tcy = ny_;
                    tax = ka == k ? tcx : tax;
                    tby = kb == k ? tcy : tby;
                }
                const float tt = tax * tby;
                ax[u].x += cxl * tt;
                ay[u].x += cyl * tt;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; u++) {
        float sx = ax[u].x + ax[u].y, sy = ay[u].x + ay[u].y;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            sx += __shfl_xor(sx, o, WAVE);
            sy += __shfl_xor(sy, o, WAVE);
        }
        if (lane == 0 && (real >> u & 1u)) d.froad[slot[u]] = make_float2(sx, sy);
    }
}

#define CSF_ROAD_NP(K, ...)                                                                                            \
    switch (d.road_np) {                                                                                               \
    case 2: K<2> __VA_ARGS__; break;                                                                                   \
    case 3: K<3> __VA_ARGS__; break;                                                                                   \
    case 4: K<4> __VA_ARGS__; break;                                                                                   \
    case 5: K<5> __VA_ARGS__; break;                                                                                   \
    case 6: K<6> __VA_ARGS__; break;                                                                                   \
    default: K<0> __VA_ARGS__; break;                                                                                  \
    }

template <int NP>
static void far_launch(const Dev &d, const short2 *vcell, double *samples, hipStream_t st) {
    // (cells in the x dimension of the grid: up to 2^18 of them, and gridDim.y ends at 65 535)
    hipLaunchKernelGGL(road_far_kernel<NP>, dim3((unsigned)(d.rg_nx * d.rg_ny), 64 / (WPB * RPW)), dim3(BLOCK), 0, st, d, vcell, samples);
}
template <int NP>
static void grid_launch(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    const int64_t m = d.rg_by_place == 1 ? d.n_pad : d.hi - d.lo;
    hipExtLaunchKernelGGL(road_grid_kernel<NP>, dim3((unsigned)((m + WPB * RPW - 1) / (WPB * RPW))), dim3(BLOCK), 0, st, t0, t1, 0, d);
}

void launch_road_far(const Dev &d, const short2 *vcell, double *samples, hipStream_t st) {
    if (d.nv <= 0 || d.rg_nx <= 0) return;
    CSF_ROAD_NP(far_launch, (d, vcell, samples, st));
}

void launch_road_grid(const Dev &d, hipStream_t st, hipEvent_t t0, hipEvent_t t1) {
    if (d.hi <= d.lo || d.nv <= 0 || d.rg_nx <= 0) return;
    CSF_ROAD_NP(grid_launch, (d, st, t0, t1));
}

}  // namespace csf
