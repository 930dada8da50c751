// spherical_sfm_amd -- SO(3) pose-graph solvers on the GPU.
//
// Replaces, behind the C ABI (include/ssfm.h):
//   optimize_rotations                    reference src/rotation_averaging.cpp:44-91   (RotationError :15-42)
//   get_cost                              reference src/uncalibrated_pose_graph.cpp:116-145 (PoseGraphError :81-114)
//   optimize_rotations_and_focal_length   reference src/uncalibrated_pose_graph.cpp:147-203 (UncalibratedPoseGraphError :33-79,
//                                         decompose_rotation :16-31, box bounds :181-182)
// Ceres defaults apply there (50 iterations, 5 invalid steps, SoftLOneLoss(0.03), first rotation constant, residual scaled
// by 1/max|log R_rel|).  One lane per edge evaluates the residual with dual numbers (dual.h) and scatters the 3x3 blocks
// of J^T J into a block-CSR system over the nodes; the normal equations are then solved by the same banded-Cholesky /
// PCG-refinement path as the BA reduced camera system (ba_handle.h, DC = 3, the shared focal multiplier as border row).
#include <algorithm>
#include <cstdio>
#include "ba_handle.h"
#include "dual.h"
#include "line_search.h"

namespace ssfm {

struct EdgeConst { double meas[9]; double r[3]; double rx, ry, thetaxy, thetaz; };   // meas row-major

// reference decompose_rotation (src/uncalibrated_pose_graph.cpp:16-31), R row-major
static void decompose_rotation_host(const double* R, double& rx, double& ry, double& thetaxy, double& thetaz) {
    double Z[3] = {R[2], R[5], R[8]};
    const double zn = norm3(Z); Z[0] /= zn; Z[1] /= zn; Z[2] /= zn;
    const double axis[3] = {-Z[1], Z[0], 0.0};
    const double an = norm3(axis);
    const double rxy[3] = {axis[0] / an, axis[1] / an, axis[2] / an};
    thetaxy = acos(Z[2]);
    const double v[3] = {thetaxy * rxy[0], thetaxy * rxy[1], thetaxy * rxy[2]};
    double Rxy[9]; so3exp(v, Rxy);
    rx = rxy[0]; ry = rxy[1];
    double Rz[9]; mat3_mul_at(Rxy, R, Rz);
    double rz[3]; so3ln(Rz, rz);
    thetaz = rz[2];
}

// res = scale * log(R1 R0^T R^T); kind 0: R = measured matrix; 1: R = exp(so3ln(meas)) (Ceres conversion); 2: tilt/roll model with f
template <typename T>
__device__ __forceinline__ void edge_residual(int kind, const EdgeConst& e, double scale, const T* r0, const T* r1, const T& f, T* res) {
    T R[9];
    if (kind == 0) { for (int i = 0; i < 9; i++) R[i] = T(e.meas[i]); }
    else if (kind == 1) { const T myr[3] = {T(e.r[0]), T(e.r[1]), T(e.r[2])}; aa_to_matrix_t(myr, R); }
    else {
        const T fsq = f * f;
        const T num = 2.0 * f * sin(e.thetaxy);
        const T den = (1.0 + fsq) * cos(e.thetaxy) + (1.0 - fsq);
        const T thp = datan2(num, den);
        const T rxy[3] = {thp * e.rx, thp * e.ry, T(0.0)};
        const T rz[3] = {T(0.0), T(0.0), T(e.thetaz)};
        T Rxy[9], Rz[9]; aa_to_matrix_t(rxy, Rxy); aa_to_matrix_t(rz, Rz);
        mat3_mul_t(Rxy, Rz, R);
    }
    T R0[9], R1[9], A[9], C[9];
    aa_to_matrix_t(r0, R0); aa_to_matrix_t(r1, R1);
    mat3_mul_bt_t(R1, R0, A); mat3_mul_bt_t(A, R, C);
    matrix_to_aa_t(C, res);
    res[0] = res[0] * scale; res[1] = res[1] * scale; res[2] = res[2] * scale;
}

__device__ __forceinline__ int find_slot(const int* cols, int n, int key) {
    int lo = 0, hi = n - 1;
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (cols[mid] < key) lo = mid + 1; else hi = mid; }
    return lo;
}

// mode 0: assemble J^T J (block-CSR), J^T r, diag, focal border, cost.  mode 1: model cost change sum m (r + m/2) for step.
__global__ void __launch_bounds__(64)
k_rot_edges(int mode, int kind, int E, const int* __restrict__ e0, const int* __restrict__ e1, const EdgeConst* __restrict__ ec,
            double scale, double loss_a, const double* __restrict__ x, const double* __restrict__ fm, const double* __restrict__ sc_node,
            const double* __restrict__ sc_f, const int* __restrict__ row_ptr, const int* __restrict__ col_idx, int n_nodes,
            const double* __restrict__ step, double* __restrict__ S_val, double* __restrict__ rhs, double* __restrict__ Udiag,
            double* __restrict__ Sfc, double* __restrict__ scal) {
    __shared__ double red[3];
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double acc[3] = {0, 0, 0};   // cost | FJJ or model | FJR
    if (e < E) {
        typedef Dual<7> D;
        const int i0 = e0[e], i1 = e1[e];
        D r0[3] = {D(x[3 * i0], 0), D(x[3 * i0 + 1], 1), D(x[3 * i0 + 2], 2)}, r1[3] = {D(x[3 * i1], 3), D(x[3 * i1 + 1], 4), D(x[3 * i1 + 2], 5)};
        D f(fm[0], 6), res[3];
        edge_residual<D>(kind, ec[e], scale, r0, r1, f, res);
        double rho0, rho1; robust_loss(2, loss_a, res[0].a * res[0].a + res[1].a * res[1].a + res[2].a * res[2].a, rho0, rho1);
        const double sr = sqrt(rho1);
        acc[0] = 0.5 * rho0;
        double J0[3][3], J1[3][3], Jf[3], r[3];
        const double sf = sc_f[0];
        for (int a = 0; a < 3; a++) {
            r[a] = sr * res[a].a;
            for (int k = 0; k < 3; k++) { J0[a][k] = sr * res[a].v[k] * sc_node[3 * i0 + k]; J1[a][k] = sr * res[a].v[3 + k] * sc_node[3 * i1 + k]; }
            Jf[a] = sr * res[a].v[6] * sf;
        }
        if (mode == 1) {
            const double sfv = step[3 * n_nodes];
            for (int a = 0; a < 3; a++) {
                double m = Jf[a] * sfv;
                for (int k = 0; k < 3; k++) m += J0[a][k] * step[3 * i0 + k] + J1[a][k] * step[3 * i1 + k];
                acc[1] += m * (r[a] + 0.5 * m);
            }
        } else {
            const int* c0 = col_idx + row_ptr[i0]; const int n0 = row_ptr[i0 + 1] - row_ptr[i0];
            const int* c1 = col_idx + row_ptr[i1]; const int n1 = row_ptr[i1 + 1] - row_ptr[i1];
            double* b00 = S_val + (size_t)(row_ptr[i0] + find_slot(c0, n0, i0)) * 9;
            double* b01 = S_val + (size_t)(row_ptr[i0] + find_slot(c0, n0, i1)) * 9;
            double* b10 = S_val + (size_t)(row_ptr[i1] + find_slot(c1, n1, i0)) * 9;
            double* b11 = S_val + (size_t)(row_ptr[i1] + find_slot(c1, n1, i1)) * 9;
            for (int u = 0; u < 3; u++) {
                double g0 = 0, g1 = 0, f0 = 0, f1 = 0, d0 = 0, d1 = 0;
                for (int a = 0; a < 3; a++) { g0 += J0[a][u] * r[a]; g1 += J1[a][u] * r[a]; f0 += Jf[a] * J0[a][u]; f1 += Jf[a] * J1[a][u];
                                              d0 += J0[a][u] * J0[a][u]; d1 += J1[a][u] * J1[a][u]; }
                unsafeAtomicAdd(&rhs[3 * i0 + u], g0); unsafeAtomicAdd(&rhs[3 * i1 + u], g1);
                unsafeAtomicAdd(&Sfc[3 * i0 + u], f0); unsafeAtomicAdd(&Sfc[3 * i1 + u], f1);
                unsafeAtomicAdd(&Udiag[3 * i0 + u], d0); unsafeAtomicAdd(&Udiag[3 * i1 + u], d1);
                for (int v = 0; v < 3; v++) {
                    double s00 = 0, s01 = 0, s11 = 0;
                    for (int a = 0; a < 3; a++) { s00 += J0[a][u] * J0[a][v]; s01 += J0[a][u] * J1[a][v]; s11 += J1[a][u] * J1[a][v]; }
                    unsafeAtomicAdd(&b00[3 * u + v], s00); unsafeAtomicAdd(&b11[3 * u + v], s11);
                    if (i0 != i1) { unsafeAtomicAdd(&b01[3 * u + v], s01); unsafeAtomicAdd(&b10[3 * v + u], s01); }
                    else unsafeAtomicAdd(&b00[3 * u + v], s01 + (J1[0][u] * J0[0][v] + J1[1][u] * J0[1][v] + J1[2][u] * J0[2][v]));
                }
            }
            for (int a = 0; a < 3; a++) { acc[1] += Jf[a] * Jf[a]; acc[2] += Jf[a] * r[a]; }
        }
    }
    for (int i = 0; i < 3; i++) { acc[i] = wave_sum(acc[i]); }
    (void)red;
    if ((threadIdx.x & 63) == 0) {
        if (mode == 1) unsafeAtomicAdd(&scal[SC_MODEL], acc[1]);
        else { unsafeAtomicAdd(&scal[SC_COST], acc[0]); unsafeAtomicAdd(&scal[SC_FJJ], acc[1]); unsafeAtomicAdd(&scal[SC_FJR], acc[2]); }
    }
}

__global__ void __launch_bounds__(64)
k_rot_cost(int kind, int E, const int* __restrict__ e0, const int* __restrict__ e1, const EdgeConst* __restrict__ ec, double scale, double loss_a,
           const double* __restrict__ x, const double* __restrict__ fm, double* __restrict__ out, double* __restrict__ part = nullptr) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0.0;
    if (e < E) {
        const int i0 = e0[e], i1 = e1[e];
        const double r0[3] = {x[3 * i0], x[3 * i0 + 1], x[3 * i0 + 2]}, r1[3] = {x[3 * i1], x[3 * i1 + 1], x[3 * i1 + 2]};
        double res[3]; edge_residual<double>(kind, ec[e], scale, r0, r1, fm[0], res);
        double rho0, rho1; robust_loss(2, loss_a, res[0] * res[0] + res[1] * res[1] + res[2] * res[2], rho0, rho1);
        c = 0.5 * rho0;
    }
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { if (part) part[blockIdx.x] = c; else unsafeAtomicAdd(out, c); }     // part: one slot per workgroup, summed in order by the reader
}

enum { SC_GDELTA = 14, SC_DMAX = 15 };      // scalar slots of the line search of bounded problems (k_rot_update / k_rot_step)

// ---- deterministic assembly (round 3): node-major gather instead of edge-major scatter ------------------------------------------------
// k_rot_edges adds every edge's 3x3 blocks into S with global fp64 atomics, so the last bits of S -- and, near a tolerance, the iteration count of a
// 2000-node solve -- varied from run to run.  Here ONE WAVE owns a node: lane l evaluates incident edge l of its node (each edge is evaluated by both of its
// ends: the same instructions on the same inputs, hence bit-identical Jacobians and a bit-symmetric S), the diagonal block, J^T r, the focal border and
// the column norms are summed over the wave by a butterfly (fixed order), every off-diagonal block is written by the lane of its edge (parallel edges:
// the first lane of the run adds the run in lane order), and the three scalar sums go through per-node partials that the LAST wave to finish folds in
// node order.  No atomics on data, no dependence on scheduling.  Adjacency: nadj_ptr / nadj_edge / nadj_side (0: the node is index0 of the edge, 1: index1,
// 2: both) / nadj_slot (block slot of the neighbour in the node's row) / nadj_first (first entry of a run of parallel edges), sorted by (neighbour, edge).
// ejac: [E * 24] robustified, Jacobi-scaled J0 (9) | J1 (9) | Jf (3) | r (3) of every edge, written by the edge pass (k_rot_edge_records) and read again by
// the model-cost pass of k_rot_eval.
constexpr int ROT_MAX_DEG = 64;
// pass 1, lane per edge (full waves; one Dual<7> evaluation per edge -- a wave per node evaluating its incident edges itself ran every edge twice on
// a quarter-full wave: 208 us per launch at 4000 nodes): the record of the edge and, per workgroup, the three scalar partial sums
__global__ void __launch_bounds__(64)
k_rot_edge_records(int kind, int E, const int* __restrict__ e0, const int* __restrict__ e1, const EdgeConst* __restrict__ ec, double scale, double loss_a,
                   const double* __restrict__ x, const double* __restrict__ fm, const double* __restrict__ sc_node, const double* __restrict__ sc_f,
                   double* __restrict__ ejac, double* __restrict__ edge_part /* [gridDim.x * 3] */) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double p[3] = {0, 0, 0};
    if (e < E) {
        typedef Dual<7> D;
        const int i0 = e0[e], i1 = e1[e];
        D r0[3] = {D(x[3 * i0], 0), D(x[3 * i0 + 1], 1), D(x[3 * i0 + 2], 2)}, r1[3] = {D(x[3 * i1], 3), D(x[3 * i1 + 1], 4), D(x[3 * i1 + 2], 5)};
        D f(fm[0], 6), res[3];
        edge_residual<D>(kind, ec[e], scale, r0, r1, f, res);
        double rho0, rho1; robust_loss(2, loss_a, res[0].a * res[0].a + res[1].a * res[1].a + res[2].a * res[2].a, rho0, rho1);
        const double sr = sqrt(rho1), sf = sc_f[0];
        double* q = ejac + (size_t)24 * e;
        p[0] = 0.5 * rho0;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            const double ra = sr * res[a].a, jf = sr * res[a].v[6] * sf;
#pragma unroll
            for (int k = 0; k < 3; k++) { q[3 * a + k] = sr * res[a].v[k] * sc_node[3 * i0 + k]; q[9 + 3 * a + k] = sr * res[a].v[3 + k] * sc_node[3 * i1 + k]; }
            q[18 + a] = jf; q[21 + a] = ra;
            p[1] += jf * jf; p[2] += jf * ra;
        }
    }
#pragma unroll
    for (int k = 0; k < 3; k++) p[k] = wave_sum(p[k]);
    if ((threadIdx.x & 63) == 0) { double* w = edge_part + 3 * (size_t)blockIdx.x; w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; }
}
// pass 2, wave per node, lane per incident edge: gathers the records into the node's row of S, its J^T r, focal border and column norms; wave 0
// folds the scalar partials of pass 1 in workgroup order
__global__ void __launch_bounds__(64)
k_rot_gather_nodes(int n_nodes, int n_edge_parts, const int* __restrict__ nadj_ptr, const int* __restrict__ nadj_edge, const unsigned char* __restrict__ nadj_side,
                   const int* __restrict__ nadj_slot, const unsigned char* __restrict__ nadj_first, const double* __restrict__ ejac, const double* __restrict__ edge_part,
                   const int* __restrict__ row_ptr, const int* __restrict__ diag_slot, double* __restrict__ S_val, double* __restrict__ rhs, double* __restrict__ Udiag,
                   double* __restrict__ Sfc, double* __restrict__ scal, double* __restrict__ pcg_fail /* the word behind the solver flags */) {
    __shared__ double tmp[64][9];
    const int i = blockIdx.x, lane = threadIdx.x;
    const int a0 = nadj_ptr[i], deg = nadj_ptr[i + 1] - a0;
    double v[18];
#pragma unroll
    for (int k = 0; k < 18; k++) v[k] = 0.0;
    double B[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < deg) {
        const int e = nadj_edge[a0 + lane], side = nadj_side[a0 + lane];
        const double* q = ejac + (size_t)24 * e;
        double Js[3][3], Jo[3][3], Jf[3], r[3];
#pragma unroll
        for (int a = 0; a < 3; a++) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const double j0 = q[3 * a + k], j1 = q[9 + 3 * a + k];
                Js[a][k] = (side == 0) ? j0 : ((side == 1) ? j1 : j0 + j1);         // a self-loop moves the node through both slots
                Jo[a][k] = (side == 0) ? j1 : ((side == 1) ? j0 : 0.0);
            }
            Jf[a] = q[18 + a]; r[a] = q[21 + a];
        }
#pragma unroll
        for (int u = 0; u < 3; u++) {
#pragma unroll
            for (int w = 0; w < 3; w++) {
                double d = 0.0, b = 0.0;
#pragma unroll
                for (int a = 0; a < 3; a++) { d += Js[a][u] * Js[a][w]; b += Js[a][u] * Jo[a][w]; }
                v[3 * u + w] = d; B[3 * u + w] = b;
            }
#pragma unroll
            for (int a = 0; a < 3; a++) { v[9 + u] += Js[a][u] * r[a]; v[12 + u] += Jf[a] * Js[a][u]; v[15 + u] += Js[a][u] * Js[a][u]; }
        }
    }
#pragma unroll
    for (int k = 0; k < 9; k++) tmp[lane][k] = B[k];
    const double mine = wave_transpose_sum(v);             // sum k sits in the lane with wave_tr_index() == k
    const int slot = wave_tr_index();
    const int rb = row_ptr[i];
    if (slot < 9) S_val[((size_t)rb + diag_slot[i]) * 9 + slot] = mine;
    else if (slot < 12) rhs[3 * i + slot - 9] = mine;
    else if (slot < 15) Sfc[3 * i + slot - 12] = mine;
    else if (slot < 18) Udiag[3 * i + slot - 15] = mine;
    __syncthreads();
    if (lane < deg && nadj_first[a0 + lane] && nadj_side[a0 + lane] != 2) {
        double acc[9];
#pragma unroll
        for (int k = 0; k < 9; k++) acc[k] = tmp[lane][k];
        for (int m = lane + 1; m < deg && !nadj_first[a0 + m]; m++)
#pragma unroll
            for (int k = 0; k < 9; k++) acc[k] += tmp[m][k];
        double* dst = S_val + ((size_t)rb + nadj_slot[a0 + lane]) * 9;
#pragma unroll
        for (int k = 0; k < 9; k++) dst[k] = acc[k];
    }
    // ---- wave 0 folds the scalar partials of the edge pass (complete before this launch started) in workgroup order, and resets what the rest of the
    // iteration accumulates into by atomic max / sets only on failure (the gradient-max slot of every scalar replica, the factorisation's fail word): with
    // plain stores everywhere else the per-iteration memset of the zone is not needed
    if (i != 0) return;
    scal[(size_t)lane * SC_TOTAL + SC_GMAX] = 0.0;
    if (lane == 0) pcg_fail[0] = 0.0;
    double c[3] = {0.0, 0.0, 0.0};
    for (int k = lane; k < n_edge_parts; k += 64) { const double* q = edge_part + 3 * (size_t)k; c[0] += q[0]; c[1] += q[1]; c[2] += q[2]; }
#pragma unroll
    for (int k = 0; k < 3; k++) c[k] = wave_sum(c[k]);
    if (lane == 0) { scal[SC_COST] = c[0]; scal[SC_FJJ] = c[1]; scal[SC_FJR] = c[2]; }
}

// Focal arrow + step + candidate in one single-workgroup launch (was k_band_combine + k_rot_update): y = V - U phi from the two band solutions
// (phi = (rho - S_fc . V) / (S_ff - S_fc . U)), step = -y, candidate = Plus(x, scale o step) with the box projection of the focal multiplier,
// |x - candidate|^2, |candidate|^2, the projected-gradient max norm, g . delta and |delta|_inf (line search of bounded problems).
__global__ void __launch_bounds__(1024)
k_rot_step(int n_nodes, const double* __restrict__ V, const double* __restrict__ U, const double* __restrict__ Sfc, const double* __restrict__ Sff,
           const double* __restrict__ rho_ptr, const int* __restrict__ pos, const double* __restrict__ x, const double* __restrict__ fm,
           const double* __restrict__ sc_node, const double* __restrict__ sc_f, const double* __restrict__ rhs_raw, double f_lo, double f_hi,
           double* __restrict__ y, double* __restrict__ xc, double* __restrict__ fmc, double* __restrict__ step, double* __restrict__ scal,
           int with_f = 1, double* __restrict__ step_part = nullptr /* [gridDim.x][5] */) {
    // Round 5: several workgroups (one of 1024 lanes spent 26 us at 4000 nodes: twelve dependent gathers per lane, twice).  Every workgroup forms the two dot
    // products of the focal arrow itself -- all of them over all entries in the same order, hence the same phi everywhere; without a focal parameter the arrow is
    // empty and the pass is skipped -- and then takes its slice of the entries; its five sums go to step_part, folded in workgroup order by k_rot_fold_publish.
    __shared__ double red[96];
    __shared__ double sphi;
    const int n = 3 * n_nodes;
    if (with_f) {
        double d2[2] = {0, 0};
        for (int t = threadIdx.x; t < n; t += blockDim.x) {
            const int c = t / 3, a = t - 3 * c; const int pi = pos[c] * 3 + a;
            d2[0] += Sfc[t] * V[pi]; d2[1] += Sfc[t] * U[pi];
        }
        block_sum<2>(d2, red);
        if (threadIdx.x == 0) sphi = (rho_ptr[0] - d2[0]) / (Sff[0] - d2[1]);
    } else if (threadIdx.x == 0) sphi = rho_ptr[0] / Sff[0];       // (Sfc = 0: what the sums above give)
    __syncthreads();
    const double phi = sphi;
    double acc[3] = {0, 0, 0}; double gmax = 0.0, dmax = 0.0;
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < n; t += gridDim.x * blockDim.x) {
        const int c = t / 3, a = t - 3 * c; const int pi = pos[c] * 3 + a;
        const double yi = V[pi] - U[pi] * phi; y[t] = yi;
        const double s = sc_node[t]; double v = x[t];
        const double st = -yi; step[t] = st;
        if (s > 0.0) { const double d = st * s; v += d; acc[0] += d * d; acc[1] += v * v; gmax = fmax(gmax, fabs(rhs_raw[t] / s)); acc[2] += rhs_raw[t] * st; dmax = fmax(dmax, fabs(st * s)); }
        xc[t] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        y[n] = phi;
        const double s = sc_f[0]; double v = fm[0]; const double st = -phi; step[n] = st;
        if (s > 0.0) {
            const double nv = fmin(fmax(v + st * s, f_lo), f_hi);
            acc[0] += (nv - v) * (nv - v); acc[1] += nv * nv;
            const double g = rhs_raw[n] / s;
            gmax = fmax(gmax, fabs(v - fmin(fmax(v - g, f_lo), f_hi)));
            acc[2] += rhs_raw[n] * st; dmax = fmax(dmax, fabs(st * s));
            v = nv;
        }
        fmc[0] = v;
    }
    __syncthreads();
    block_sum<3>(acc, red);
    gmax = wave_max(gmax); dmax = wave_max(dmax);
    if ((threadIdx.x & 63) == 0) { red[48 + (threadIdx.x >> 6)] = gmax; red[64 + (threadIdx.x >> 6)] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double g = 0.0, d = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) { g = fmax(g, red[48 + w]); d = fmax(d, red[64 + w]); }
        if (step_part) { double* q = step_part + 5 * (size_t)blockIdx.x; q[0] = acc[0]; q[1] = acc[1]; q[2] = acc[2]; q[3] = g; q[4] = d; }
        else { scal[SC_GMAX] = g; scal[SC_STEP2_CAM] = acc[0]; scal[SC_XN2_CAM] = acc[1]; scal[SC_GDELTA] = acc[2]; scal[SC_DMAX] = d; }
    }
}

// Everything that is left of an iteration in one launch (was k_pcg_matvec + k_ref_residual + k_rot_edges(mode 1) + k_rot_cost): workgroups [0, ge) take
// 64 edges each -- model cost change from the Jacobian records of the assembly, cost at the candidate --, workgroups [ge, ge + gn) take 64 nodes each --
// their rows of S y for the residual check |rhs - S y| <= tol |rhs| of the direct solve.  Per-workgroup partial sums [model, candidate cost, |r|^2, |rhs|^2, S_fc . y],
// folded in workgroup order by k_rot_fold_publish.
__global__ void __launch_bounds__(64)
k_rot_eval(int kind, int E, int ge, int n_nodes, const int* __restrict__ e0, const int* __restrict__ e1, const EdgeConst* __restrict__ ec, double scale, double loss_a,
           const double* __restrict__ ejac, const double* __restrict__ step, const double* __restrict__ xc, const double* __restrict__ fmc,
           const int* __restrict__ row_ptr, const int* __restrict__ col_idx, const double* __restrict__ S_val, const double* __restrict__ Sfc,
           const double* __restrict__ Sff, const double* __restrict__ rhs, const double* __restrict__ y, double tol2, double* __restrict__ resid /* [3n + 1] */,
           double* __restrict__ wg_part /* [gridDim.x * 5]: folded by k_rot_fold_publish */) {
    const int lane = threadIdx.x, wg = blockIdx.x;
    double p[5] = {0, 0, 0, 0, 0};
    if (wg < ge) {
        const int e = wg * 64 + lane;
        if (e < E) {
            const int i0 = e0[e], i1 = e1[e];
            const double* q = ejac + (size_t)24 * e;
            const double sfv = step[3 * n_nodes];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                double m = q[18 + a] * sfv;
#pragma unroll
                for (int k = 0; k < 3; k++) m += q[3 * a + k] * step[3 * i0 + k] + q[9 + 3 * a + k] * step[3 * i1 + k];
                p[0] += m * (q[21 + a] + 0.5 * m);
            }
            const double r0[3] = {xc[3 * i0], xc[3 * i0 + 1], xc[3 * i0 + 2]}, r1[3] = {xc[3 * i1], xc[3 * i1 + 1], xc[3 * i1 + 2]};
            double res[3]; edge_residual<double>(kind, ec[e], scale, r0, r1, fmc[0], res);
            double rho0, rho1; robust_loss(2, loss_a, res[0] * res[0] + res[1] * res[1] + res[2] * res[2], rho0, rho1);
            p[1] = 0.5 * rho0;
        }
    } else {
        const int i = (wg - ge) * 64 + lane;
        if (i < n_nodes) {
            const int rb = row_ptr[i], nb = row_ptr[i + 1] - rb;
            double qv[3] = {0, 0, 0};
            for (int b = 0; b < nb; b++) {
                const double* blk = S_val + ((size_t)rb + b) * 9; const double* yv = y + 3 * (size_t)col_idx[rb + b];
#pragma unroll
                for (int a = 0; a < 3; a++) qv[a] += blk[3 * a] * yv[0] + blk[3 * a + 1] * yv[1] + blk[3 * a + 2] * yv[2];
            }
            const double yf = y[3 * n_nodes];
#pragma unroll
            for (int a = 0; a < 3; a++) {
                const double bi = rhs[3 * i + a], ri = bi - (qv[a] + Sfc[3 * i + a] * yf);
                resid[3 * i + a] = ri; p[2] += ri * ri; p[3] += bi * bi; p[4] += Sfc[3 * i + a] * y[3 * i + a];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 5; k++) p[k] = wave_sum(p[k]);
    if (lane == 0) { double* w = wg_part + 5 * (size_t)wg; for (int k = 0; k < 5; k++) w[k] = p[k]; }
    // (Rounds 2-4 folded here: the last workgroup to arrive, found through a ticket behind an agent-scope release per workgroup -- an L2 write-back on this multi-XCD
    // part: 38 us per launch at 4000 nodes / 16 000 edges.  The fold rides in the hand-over kernel now, which runs behind the kernel boundary anyway.)
}

// The end of a pose-graph iteration (round 5): the partial sums of k_rot_eval (per workgroup: model change, candidate cost, |r|^2, |rhs|^2, S_fc . y) and of
// k_rot_step (per workgroup: |delta|^2, |candidate|^2, g . delta, gradient max, |delta|_inf) folded in workgroup order -- the same arithmetic as the in-kernel
// folds they replace: lane-strided sums, then a wave sum --, the residual test of the direct solve, and the hand-over to the host (publish_body), all by ONE
// workgroup of SC_TOTAL * 64 lanes behind the kernel boundary.  host_out == nullptr: the copying hand-over follows, only the device words are written.
__global__ void __launch_bounds__(SC_TOTAL * 64)
k_rot_fold_publish(const double* __restrict__ wg_part, int n_wg, const double* __restrict__ step_part, int n_step, int n_nodes, const double* __restrict__ Sff,
                   const double* __restrict__ rhs, const double* __restrict__ y, double tol2, double* __restrict__ resid, double* __restrict__ scal,
                   double* __restrict__ pcg, double* __restrict__ host_out, unsigned long long seq, const LmGate gate, double* __restrict__ spec) {
    __shared__ double folded[SC_TOTAL], ov_scal[SC_TOTAL], ov_pcg[PCG_TOTAL];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (w == 0) {
        double c[5] = {0, 0, 0, 0, 0};
        for (int k = lane; k < n_wg; k += 64) { const double* q = wg_part + 5 * (size_t)k;
#pragma unroll
            for (int j = 0; j < 5; j++) c[j] += q[j]; }
#pragma unroll
        for (int j = 0; j < 5; j++) c[j] = wave_sum(c[j]);
        if (lane == 0) {
            const int n = 3 * n_nodes;
            const double bf = rhs[n], rf = bf - (Sff[0] * y[n] + c[4]);          // the focal row of the bordered system
            resid[n] = rf;
            const double rr = c[2] + rf * rf, bn2 = c[3] + bf * bf;
            ov_scal[SC_MODEL] = c[0]; ov_scal[SC_CAND_COST] = c[1]; scal[SC_MODEL] = c[0]; scal[SC_CAND_COST] = c[1];
            for (int i = 0; i < PCG_TOTAL; i++) ov_pcg[i] = pcg[i];
            ov_pcg[PCG_RR] = rr; ov_pcg[PCG_BN2] = bn2; ov_pcg[PCG_ITERS] = 0.0; ov_pcg[PCG_BREAKDOWN] = 0.0; ov_pcg[PCG_DONE] = (rr <= tol2 * bn2) ? 1.0 : 0.0;
            pcg[PCG_RR] = rr; pcg[PCG_BN2] = bn2; pcg[PCG_ITERS] = 0.0; pcg[PCG_BREAKDOWN] = 0.0; pcg[PCG_DONE] = ov_pcg[PCG_DONE];
        }
    } else if (w == 1) {
        double a[3] = {0, 0, 0}, g = 0.0, d = 0.0;
        for (int k = lane; k < n_step; k += 64) { const double* q = step_part + 5 * (size_t)k; a[0] += q[0]; a[1] += q[1]; a[2] += q[2]; g = fmax(g, q[3]); d = fmax(d, q[4]); }
#pragma unroll
        for (int j = 0; j < 3; j++) a[j] = wave_sum(a[j]);
        g = wave_max(g); d = wave_max(d);
        if (lane == 0) {
            ov_scal[SC_GMAX] = g; ov_scal[SC_STEP2_CAM] = a[0]; ov_scal[SC_XN2_CAM] = a[1]; ov_scal[SC_GDELTA] = a[2]; ov_scal[SC_DMAX] = d;
            scal[SC_GMAX] = g; scal[SC_STEP2_CAM] = a[0]; scal[SC_XN2_CAM] = a[1]; scal[SC_GDELTA] = a[2]; scal[SC_DMAX] = d;
        }
    }
    __syncthreads();
    if (!host_out) return;
    constexpr unsigned mask = (1u << SC_MODEL) | (1u << SC_CAND_COST) | (1u << SC_GMAX) | (1u << SC_STEP2_CAM) | (1u << SC_XN2_CAM) | (1u << SC_GDELTA) | (1u << SC_DMAX);
    publish_body(scal, pcg, host_out, seq, gate, spec, folded, nullptr, 0, ov_scal, mask, ov_pcg);
}

// candidate = Plus(x, alpha (scale o step)) with the box projection on the focal multiplier (Ceres ParameterBlock::Plus; alpha = 1 except
// after a line search); also |x - candidate|^2, |candidate|^2, the projected-gradient max norm |x - Plus(x, -g)|_inf, and for the line
// search of bounded problems g . delta (slot SC_GDELTA) and |delta|_inf (SC_DMAX) with delta = scale o step, g = the unscaled gradient
__global__ void __launch_bounds__(1024)
k_rot_update(int n_nodes, const double* __restrict__ x, const double* __restrict__ fm, const double* __restrict__ sc_node,
             const double* __restrict__ sc_f, const double* __restrict__ y, const double* __restrict__ rhs_raw, double f_lo, double f_hi, double alpha,
             double* __restrict__ xc, double* __restrict__ fmc, double* __restrict__ step, double* __restrict__ scal) {
    __shared__ double red[3 * 16 + 16];
    double acc[3] = {0, 0, 0}; double gmax = 0.0, dmax = 0.0;
    const int n = 3 * n_nodes;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double s = sc_node[i]; double v = x[i];
        const double st = -y[i]; step[i] = st;
        if (s > 0.0) { const double d = alpha * st * s; v += d; acc[0] += d * d; acc[1] += v * v; gmax = fmax(gmax, fabs(rhs_raw[i] / s)); acc[2] += rhs_raw[i] * st; dmax = fmax(dmax, fabs(st * s)); }
        xc[i] = v;
    }
    if (threadIdx.x == 0) {
        const double s = sc_f[0]; double v = fm[0]; const double st = -y[n]; step[n] = st;
        if (s > 0.0) {
            const double nv = fmin(fmax(v + alpha * st * s, f_lo), f_hi);
            acc[0] += (nv - v) * (nv - v); acc[1] += nv * nv;
            const double g = rhs_raw[n] / s;                       // unscaled gradient of the focal multiplier
            gmax = fmax(gmax, fabs(v - fmin(fmax(v - g, f_lo), f_hi)));
            acc[2] += rhs_raw[n] * st; dmax = fmax(dmax, fabs(st * s));
            v = nv;
        }
        fmc[0] = v;
    }
    block_sum<3>(acc, red);
    gmax = wave_max(gmax); dmax = wave_max(dmax);
    if ((threadIdx.x & 63) == 0) { red[48 + (threadIdx.x >> 6)] = gmax; red[32 + (threadIdx.x >> 6)] = dmax; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double g = 0.0, d = 0.0; for (int w = 0; w < (int)(blockDim.x >> 6); w++) { g = fmax(g, red[48 + w]); d = fmax(d, red[32 + w]); }
        scal[SC_GMAX] = g;                                          // overwrites what k_finalize_S left there (no memset in between)
        scal[SC_STEP2_CAM] = acc[0]; scal[SC_XN2_CAM] = acc[1]; scal[SC_GDELTA] = acc[2]; scal[SC_DMAX] = d;
    }
}
// One evaluation of the line-search function of a bounded problem (line_search.h): f(a) = cost(Plus(x, a delta)) and its slope
// delta . gradient there, delta = scale o step.  out[0] += cost, out[1] += slope.  Lane per edge, the trial point is formed on the fly.
__global__ void __launch_bounds__(64)
k_rot_ls_eval(int kind, int E, const int* __restrict__ e0, const int* __restrict__ e1, const EdgeConst* __restrict__ ec, double scale, double loss_a,
              const double* __restrict__ x, const double* __restrict__ fm, const double* __restrict__ sc_node, const double* __restrict__ sc_f,
              const double* __restrict__ step, double alpha, double f_lo, double f_hi, int n_nodes, double* __restrict__ out) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    double c = 0.0, slope = 0.0;
    if (e < E) {
        typedef Dual<7> D;
        const int i0 = e0[e], i1 = e1[e];
        double d0[3], d1[3]; D r0[3], r1[3];
        for (int k = 0; k < 3; k++) {
            d0[k] = sc_node[3 * i0 + k] * step[3 * i0 + k]; d1[k] = sc_node[3 * i1 + k] * step[3 * i1 + k];
            r0[k] = D(x[3 * i0 + k] + alpha * d0[k], k); r1[k] = D(x[3 * i1 + k] + alpha * d1[k], 3 + k);
        }
        const double df = sc_f[0] * step[3 * n_nodes];
        D f(fmin(fmax(fm[0] + alpha * df, f_lo), f_hi), 6), res[3];
        edge_residual<D>(kind, ec[e], scale, r0, r1, f, res);
        double rho0, rho1; robust_loss(2, loss_a, res[0].a * res[0].a + res[1].a * res[1].a + res[2].a * res[2].a, rho0, rho1);
        c = 0.5 * rho0;
        for (int a = 0; a < 3; a++) {
            double jd = res[a].v[6] * df;
            for (int k = 0; k < 3; k++) jd += res[a].v[k] * d0[k] + res[a].v[3 + k] * d1[k];
            slope += rho1 * res[a].a * jd;
        }
    }
    c = wave_sum(c); slope = wave_sum(slope);
    if ((threadIdx.x & 63) == 0) { unsafeAtomicAdd(&out[0], c); unsafeAtomicAdd(&out[1], slope); }
}
// Jacobi scales of the 3-dof nodes in the 6-wide camera layout the shared kernels expect: [0 0 0 | s]
static __global__ void k_scale3to6(const double* __restrict__ s3, int n_nodes, double* __restrict__ s6) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 6 * n_nodes) { const int c = i / 6, k = i - 6 * c; s6[i] = (k >= 3) ? s3[3 * c + (k - 3)] : 0.0; }
}

struct RotGraph {
    int n = 0, E = 0, kind = 0;
    double scale = 1.0;
    std::vector<int> e0, e1;
    std::vector<EdgeConst> ec;
    std::vector<double> x0;            // [3n] so3ln of the input rotations
    std::vector<double> mask;          // [3n]
};

static void build_graph(int n, const double* rotations_cm, int E, const int32_t* i0, const int32_t* i1, const double* rel_cm, int kind,
                        bool hold_first, RotGraph& G) {
    G.n = n; G.E = E; G.kind = kind; G.e0.assign(i0, i0 + E); G.e1.assign(i1, i1 + E); G.ec.resize(E); G.x0.resize((size_t)3 * n);
    for (int i = 0; i < n; i++) { double R[9]; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) R[3 * a + b] = rotations_cm[9 * i + a + 3 * b]; so3ln(R, &G.x0[3 * i]); }
    double maxn = 0;
    for (int e = 0; e < E; e++) {
        EdgeConst& c = G.ec[e];
        for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) c.meas[3 * a + b] = rel_cm[9 * e + a + 3 * b];
        so3ln(c.meas, c.r);
        maxn = std::max(maxn, norm3(c.r));
        c.rx = c.ry = c.thetaxy = c.thetaz = 0;
        if (kind == 2) { double Rr[9]; so3exp(c.r, Rr); decompose_rotation_host(Rr, c.rx, c.ry, c.thetaxy, c.thetaz); }
    }
    G.scale = 1.0 / maxn;                                  // src/rotation_averaging.cpp:50-55,63
    G.mask.assign((size_t)3 * n, 0.0);
    std::vector<char> in(n, 0);
    for (int e = 0; e < E; e++) in[i0[e]] = in[i1[e]] = 1;
    for (int i = 0; i < n; i++) if (in[i] && !(hold_first && i == 0)) G.mask[3 * i] = G.mask[3 * i + 1] = G.mask[3 * i + 2] = 1.0;   // :73
}

}  // namespace ssfm
using namespace ssfm;

extern "C" void ssfm_rotavg_default_options(ssfm_ba_options* o) {
    ssfm_ba_default_options(o);
    o->max_num_iterations = 50; o->max_num_consecutive_invalid_steps = 5;     // Ceres defaults (src/rotation_averaging.cpp:75-78)
    o->loss_type = 2; o->loss_scale = 0.03;                                   // SoftLOneLoss(0.03), src/rotation_averaging.cpp:58
}

extern "C" int ssfm_rotavg_cost(ssfm_ctx* ctx, int32_t n, const double* rotations, int32_t E, const int32_t* index0, const int32_t* index1,
                                const double* rel_rotations, double* cost) {
    if (!ctx || !rotations || !cost || E <= 0 || n <= 0 || !index0 || !index1 || !rel_rotations) return fail(ctx, SSFM_ERR_INVALID, "ssfm_rotavg_cost: bad arguments");
    for (int e = 0; e < E; e++) if (index0[e] < 0 || index0[e] >= n || index1[e] < 0 || index1[e] >= n) return fail(ctx, SSFM_ERR_INVALID, "ssfm_rotavg_cost: edge index out of range");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    RotGraph G; build_graph(n, rotations, E, index0, index1, rel_rotations, 1, false, G);   // PoseGraphError, src/uncalibrated_pose_graph.cpp:131
    DevBuf<double> x, fm, out; DevBuf<int> e0, e1; DevBuf<EdgeConst> ec;
    std::vector<double> one = {1.0}, part((size_t)(E + 63) / 64, 0.0);
    int rc = SSFM_OK;
    auto body = [&]() -> int {
        SSFM_HIP_CHECK(ctx, upload(x, G.x0, st)); SSFM_HIP_CHECK(ctx, upload(fm, one, st)); SSFM_HIP_CHECK(ctx, out.alloc(part.size()));
        SSFM_HIP_CHECK(ctx, upload(e0, G.e0, st)); SSFM_HIP_CHECK(ctx, upload(e1, G.e1, st)); SSFM_HIP_CHECK(ctx, upload(ec, G.ec, st));
        // one partial sum per workgroup, added up in workgroup order on the host: the same bits on every call (no floating-point atomics)
        hipLaunchKernelGGL(k_rot_cost, dim3((E + 63) / 64), dim3(64), 0, st, 1, E, e0.p, e1.p, ec.p, G.scale, 0.03, x.p, fm.p, (double*)nullptr, out.p);
        SSFM_HIP_CHECK(ctx, hipMemcpyAsync(part.data(), out.p, part.size() * sizeof(double), hipMemcpyDeviceToHost, st));
        SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
        return SSFM_OK;
    };
    rc = body();
    x.free(); fm.free(); out.free(); e0.free(); e1.free(); ec.free();
    if (rc) return rc;
    double c = 0.0; for (double v : part) c += v;
    *cost = c;
    return SSFM_OK;
}

// the buffers of one pose-graph solve; released on every exit path of rot_solve (after the stream has drained: the pool rule of ssfm_ctx.h)
struct RotScratch {
    ssfm_ctx* ctx; ssfm_ba_handle H;
    DevBuf<double> x, xc, fm2, sc3, sc6, scf, step, ls_out, m3, mf, ejac, node_part, wg_part, step_part; DevBuf<int> e0, e1, nadj_ptr, nadj_edge, nadj_slot, ticket; DevBuf<EdgeConst> ec;
    DevBuf<unsigned char> nadj_side, nadj_first;
    explicit RotScratch(ssfm_ctx* c) : ctx(c) {}
    ~RotScratch() {
        (void)hipStreamSynchronize(ctx->stream);
        x.free(); xc.free(); fm2.free(); sc3.free(); sc6.free(); scf.free(); step.free(); ls_out.free(); m3.free(); mf.free(); e0.free(); e1.free(); ec.free();
        ejac.free(); node_part.free(); wg_part.free(); step_part.free(); nadj_ptr.free(); nadj_edge.free(); nadj_slot.free(); ticket.free(); nadj_side.free(); nadj_first.free();
        H.free_all();
    }
};

static int rot_solve(ssfm_ctx* ctx, int kind, int32_t n, double* rotations, int32_t E, const int32_t* index0, const int32_t* index1,
                     const double* rel, double* focal_length, double min_focal, double max_focal, const ssfm_ba_options* opt_in,
                     ssfm_ba_summary* S) {
    if (!ctx || !rotations || !S || E <= 0 || n <= 0 || !index0 || !index1 || !rel) return fail(ctx, SSFM_ERR_INVALID, "ssfm_rotavg: bad arguments");
    for (int e = 0; e < E; e++) if (index0[e] < 0 || index0[e] >= n || index1[e] < 0 || index1[e] >= n) return fail(ctx, SSFM_ERR_INVALID, "ssfm_rotavg: edge index out of range");
    SSFM_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    std::memset(S, 0, sizeof(*S));
    ssfm_ba_options O; if (opt_in) O = *opt_in; else ssfm_rotavg_default_options(&O);
    const double t0 = wall_s();
    const bool timing = std::getenv("SSFM_PLAN_TIMING") != nullptr; double t_last = t0;
    auto lap = [&](const char* what) { if (!timing) return; const double t = wall_s(); std::fprintf(stderr, "[rot] %-34s %7.3f ms\n", what, 1e3 * (t - t_last)); t_last = t; };
    RotGraph G; build_graph(n, rotations, E, index0, index1, rel, kind, true, G);
    lap("build_graph (so3ln, edge constants)");
    const bool with_f = kind == 2;
    const double f_lo = with_f ? min_focal / *focal_length : 0.0, f_hi = with_f ? max_focal / *focal_length : 0.0;   // :181-182
    // ---- reduced-system container (the BA handle's solver state with DC = 3)
    RotScratch RS(ctx); ssfm_ba_handle* h = &RS.H;
    h->ctx = ctx; h->device = ctx->device; h->opt = O;
    std::memset(h->k_launches, 0, sizeof(h->k_launches)); std::memset(h->k_ms, 0, sizeof(h->k_ms));
    BAFlat& F = h->F; F.Nc = n; F.DC = 3;
    {
        std::vector<std::vector<int>> nb(n);
        for (int e = 0; e < E; e++) { nb[index0[e]].push_back(index1[e]); nb[index1[e]].push_back(index0[e]); }
        F.row_ptr.assign(n + 1, 0); F.diag_slot.assign(n, 0);
        for (int i = 0; i < n; i++) { auto& v = nb[i]; v.push_back(i); std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end());
                                      F.row_ptr[i + 1] = F.row_ptr[i] + (int)v.size(); F.max_row_blocks = std::max(F.max_row_blocks, (int)v.size()); }
        F.col_idx.resize(F.row_ptr[n]);
        for (int i = 0; i < n; i++) { std::copy(nb[i].begin(), nb[i].end(), F.col_idx.begin() + F.row_ptr[i]);
                                      F.diag_slot[i] = (int)(std::lower_bound(nb[i].begin(), nb[i].end(), i) - nb[i].begin()); }
        // Cuthill-McKee, pairs of nodes merged into 6x6 block rows, rings of a video pose graph eliminated from both ends
        band_plan(n, 3, F.row_ptr, F.col_idx, F.cam_pos, F.band, F.comp_ptr, F.band_row, F.band_row2, F.comp_twist, F.band_rows, F.band_block, F.pair_dummy, &F.rings);
        for (int ir = 1; ir <= F.band; ir++) for (int kr = 1; kr <= ir; kr++) F.band_pairs.push_back(ir | (kr << 16));
        if (!F.rings.empty()) ring_wrap_table(n, F.row_ptr, F.col_idx, F.band_row, F.band_row2, F.band, F.band_block == 6 ? 2 : 1, F.wrap_ptr, F.wrap_blk, F.wrap_row2);
    }
    lap("S structure + band plan");
    const size_t nn = (size_t)3 * n, nnzb = (size_t)F.row_ptr[n];
    // scale laid out 6 per node (slots 3..5) so that k_finalize_S<3> can be reused unchanged
    std::vector<double> mask6((size_t)6 * n, 0.0); for (int i = 0; i < n; i++) for (int k = 0; k < 3; k++) mask6[6 * i + 3 + k] = G.mask[3 * i + k];
    DevBuf<double>&x = RS.x, &xc = RS.xc, &fm2 = RS.fm2, &sc3 = RS.sc3, &sc6 = RS.sc6, &scf = RS.scf, &step = RS.step, &ls_out = RS.ls_out;
    DevBuf<int>&e0 = RS.e0, &e1 = RS.e1; DevBuf<EdgeConst>& ec = RS.ec;
    const double fm0 = with_f ? std::fmin(std::fmax(1.0, f_lo), f_hi) : 1.0;      // IterationZero projects the start point of a bounded problem
    std::vector<double> fmv = {fm0, fm0};
    StagedUploads su(ctx, st);
    SSFM_HIP_CHECK(ctx, su.reserve((size_t)n * 160 + (size_t)E * (sizeof(EdgeConst) + 64) + nnzb * 16 + (size_t)F.band_pairs.size() * 4 + 65536));
#define UPV(buf, vec) SSFM_HIP_CHECK(ctx, su.up(buf, vec))
    UPV(x, G.x0); UPV(fm2, fmv); UPV(e0, G.e0); UPV(e1, G.e1); UPV(ec, G.ec);
    UPV(h->row_ptr, F.row_ptr); UPV(h->col_idx, F.col_idx); UPV(h->diag_slot, F.diag_slot); UPV(h->cam_pos, F.band_row); UPV(h->cam_pos2, F.band_row2); UPV(h->pair_dummy, F.pair_dummy);   // the device only needs band rows
    UPV(h->band_pairs, F.band_pairs); UPV(h->comp_ptr, F.comp_ptr);
#undef UPV
#define ALV(buf, count) SSFM_HIP_CHECK(ctx, buf.alloc(count))
    ALV(xc, nn); ALV(sc3, nn); ALV(sc6, 6 * (size_t)n); ALV(scf, 1); ALV(step, nn + 1);
    const size_t n_red = nnzb * 9 + (nn + 1) + 3 * nn;
    // [scalar block x replicas | solver flags | S | rhs | ...] in one allocation: one memset per assembly, one copy per iteration
    // two of them: an iteration works in one while the memset of the other is queued behind its tail (ba_handle.h set_zone)
    h->zone_len = (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1 + n_red + nn + 63) / 64 * 64; h->zone_nnz = nnzb * 9; h->zone_n = nn;
    ALV(h->zone, 2 * h->zone_len);
    h->scal.n = SC_NSLOT * SC_TOTAL; h->pcg.n = PCG_TOTAL + 1; h->redbuf.n = n_red; h->zone_views = true;
    h->set_zone(0);
    int zi = 0;
    ALV(h->Minv, (size_t)n * 9); ALV(h->Sff, 1); ALV(h->px, nn + 1); ALV(h->pr, nn + 1); ALV(h->pz, nn + 1); ALV(h->pp, nn + 1); ALV(h->pq, nn + 1);
    ALV(h->pqpart, (size_t)n);
    { const size_t Nb = (size_t)F.band_rows, DCB = (size_t)F.band_block, ny = (size_t)F.y_rows(3) * 3;
      ALV(h->band, Nb * (F.band + 1) * DCB * DCB); ALV(h->Linv, Nb * DCB * DCB); ALV(h->Yb, 2 * ny); ALV(h->Yr, 2 * ny); }
    { const int rc = sub_upload(h, 3); if (rc) return rc; }
    // ---- node-major adjacency of the deterministic assembly (k_rot_edge_records + k_rot_gather_nodes): entries sorted by (neighbour, edge id)
    static const bool node_major_on = !(std::getenv("SSFM_ROT_NODE_MAJOR") && std::atoi(std::getenv("SSFM_ROT_NODE_MAJOR")) == 0);
    bool node_major = node_major_on && O.preconditioner == 0;
    {
        std::vector<int> deg(n, 0);
        for (int e = 0; e < E; e++) { deg[index0[e]]++; if (index1[e] != index0[e]) deg[index1[e]]++; }
        for (int i = 0; i < n; i++) if (deg[i] > ROT_MAX_DEG) node_major = false;          // a lane per incident edge: wider nodes take the scatter kernel
    }
    const int gnode = (n + 63) / 64;
    if (node_major) {
        struct Ent { int nb, e; unsigned char side; };
        std::vector<std::vector<Ent>> adj(n);
        for (int e = 0; e < E; e++) {
            const int a = index0[e], b = index1[e];
            if (a == b) adj[a].push_back(Ent{a, e, 2});
            else { adj[a].push_back(Ent{b, e, 0}); adj[b].push_back(Ent{a, e, 1}); }
        }
        std::vector<int> nptr(n + 1, 0), nedge, nslot; std::vector<unsigned char> nside, nfirst;
        for (int i = 0; i < n; i++) {
            auto& v = adj[i];
            std::sort(v.begin(), v.end(), [](const Ent& p, const Ent& q) { return p.nb != q.nb ? p.nb < q.nb : p.e < q.e; });
            for (size_t k = 0; k < v.size(); k++) {
                nedge.push_back(v[k].e); nside.push_back(v[k].side);
                nslot.push_back((int)(std::lower_bound(F.col_idx.begin() + F.row_ptr[i], F.col_idx.begin() + F.row_ptr[i + 1], v[k].nb) - (F.col_idx.begin() + F.row_ptr[i])));
                nfirst.push_back((k == 0 || v[k - 1].nb != v[k].nb) ? 1 : 0);
            }
            nptr[i + 1] = (int)nedge.size();
        }
        if (nedge.empty()) { nedge.push_back(0); nslot.push_back(0); nside.push_back(0); nfirst.push_back(0); }
        std::vector<int> zero2(2, 0);
        SSFM_HIP_CHECK(ctx, su.up(RS.nadj_ptr, nptr)); SSFM_HIP_CHECK(ctx, su.up(RS.nadj_edge, nedge)); SSFM_HIP_CHECK(ctx, su.up(RS.nadj_slot, nslot));
        SSFM_HIP_CHECK(ctx, su.up(RS.nadj_side, nside)); SSFM_HIP_CHECK(ctx, su.up(RS.nadj_first, nfirst)); SSFM_HIP_CHECK(ctx, su.up(RS.ticket, zero2));
        ALV(RS.ejac, (size_t)24 * E); ALV(RS.node_part, (size_t)3 * ((E + 63) / 64)); ALV(RS.wg_part, (size_t)5 * ((E + 63) / 64 + gnode)); ALV(RS.step_part, (size_t)5 * 32);
    }
#undef ALV
    double* fmx = fm2.p; double* fmc = fm2.p + 1; double* xx = x.p; double* xcand = xc.p;
    const int ge = (E + 63) / 64, gn = (n + 63) / 64;
    const double la = O.loss_scale;
    const bool poll = lm_poll() && publish_alloc(h);                // scalars published by the last kernel of an iteration (ba_handle.h)
    if (!poll && !h->host_sp) SSFM_HIP_CHECK(ctx, hipHostMalloc((void**)&h->host_sp, (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1) * sizeof(double), hipHostMallocDefault));
    double* host_scal = poll ? h->host_pub : h->host_sp; double* host_pcg = poll ? h->host_pub + SC_TOTAL : h->host_sp + SC_NSLOT * SC_TOTAL;      // this solver only ever writes replica 0 of the scalar block
    SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->zone.p, 0, h->zone.n * sizeof(double), st));
    auto assemble = [&](const double* s3, const double* sf) -> int {      // into the current zone, which is clean
        if (node_major) {
            hipLaunchKernelGGL(k_rot_edge_records, dim3(ge), dim3(64), 0, st, kind, E, e0.p, e1.p, ec.p, G.scale, la, xx, fmx, s3, sf, RS.ejac.p, RS.node_part.p);
            hipLaunchKernelGGL(k_rot_gather_nodes, dim3(n), dim3(64), 0, st, n, ge, RS.nadj_ptr.p, RS.nadj_edge.p, RS.nadj_side.p, RS.nadj_slot.p, RS.nadj_first.p, RS.ejac.p, RS.node_part.p,
                               h->row_ptr.p, h->diag_slot.p, h->S_val, h->rhs, h->Udiag, h->Sfc, h->scal.p, h->pcg.p + PCG_TOTAL);
        } else
        hipLaunchKernelGGL(k_rot_edges, dim3(ge), dim3(64), 0, st, 0, kind, E, e0.p, e1.p, ec.p, G.scale, la, xx, fmx, s3, sf, h->row_ptr.p, h->col_idx.p, n,
                           (const double*)nullptr, h->S_val, h->rhs, h->Udiag, h->Sfc, h->scal.p);
        return SSFM_OK;
    };
    // ---- Jacobi scaling from the initial Jacobian: s = mask / (1 + |J_col|)
    {
        DevBuf<double>&m3 = RS.m3, &mf = RS.mf; std::vector<double> mfv = {with_f ? 1.0 : 0.0};
        SSFM_HIP_CHECK(ctx, su.up(m3, G.mask)); SSFM_HIP_CHECK(ctx, su.up(mf, mfv));
        int rc = assemble(m3.p, mf.p); if (rc) return rc;
        hipLaunchKernelGGL(k_make_scale, dim3((3 * n + 255) / 256), dim3(256), 0, st, h->Udiag, m3.p, sc3.p, 3 * n, O.jacobi_scaling);
        hipLaunchKernelGGL(k_make_scale, dim3(1), dim3(64), 0, st, h->scal.p + SC_FJJ, mf.p, scf.p, 1, O.jacobi_scaling);
        hipLaunchKernelGGL(k_scale3to6, dim3((6 * n + 255) / 256), dim3(256), 0, st, sc3.p, n, sc6.p);
        if (timing) SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));       // (only for the lap below: nothing of the set-up is read on the host, and the first iteration's launches follow at once)
    }
    lap("allocations, uploads, Jacobi scaling");
    double x_norm = 0; { double s2 = with_f ? fm0 * fm0 : 0.0; for (size_t i = 0; i < nn; i++) if (G.mask[i] > 0) s2 += G.x0[i] * G.x0[i]; x_norm = std::sqrt(s2); }
    double radius = O.initial_trust_region_radius, decrease_factor = 2.0, x_cost = 0, minimum_cost = std::numeric_limits<double>::max();
    int iteration = 0, num_invalid = 0; bool last_successful = true;
    S->num_successful_steps = 1; S->termination = SSFM_NO_CONVERGENCE; S->camera_dof = 3; S->num_residual_blocks = S->num_residual_blocks_global = E;
    static const bool fused_finalize = SSFM_LAB_KNOB("SSFM_ROT_FUSED_FINALIZE", 1) != 0;
    while (true) {
        if (iteration >= O.max_num_iterations) { S->termination = SSFM_NO_CONVERGENCE; break; }
        if (radius <= O.min_trust_region_radius) { S->termination = SSFM_CONVERGENCE; break; }
        iteration++;
        zi ^= 1; h->set_zone(zi);
        { int rc = assemble(sc3.p, scf.p); if (rc) return rc; }
        // LM diagonal + gradient max + band rows + permuted right-hand sides in one launch (the BA loop's fused kernel; the block-Jacobi inverse of
        // k_finalize_S is only needed by the PCG preconditioner option)
        if (O.preconditioner == 0 && fused_finalize) {
            if (F.band_block != 3)
                LAUNCH(h, KID_FINALIZE, (k_finalize_gather<3, true>), n, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, sc6.p, scf.p, h->Udiag, h->rhs, radius, O.min_lm_diagonal, O.max_lm_diagonal,
                       n, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(3), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p,
                       (double2*)nullptr, (size_t)0, (const int*)nullptr, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p);
            else
                LAUNCH(h, KID_FINALIZE, (k_finalize_gather<3, false>), n, 256, 0, h->row_ptr.p, h->col_idx.p, h->diag_slot.p, sc6.p, scf.p, h->Udiag, h->rhs, radius, O.min_lm_diagonal, O.max_lm_diagonal,
                       n, h->cam_pos.p, h->cam_pos2.p, h->pair_dummy.p, F.y_rows(3), F.band, h->S_val, h->rhs, h->Sfc, h->Sff.p, h->band.p, h->Yb.p, h->scal.p,
                       (double2*)nullptr, (size_t)0, (const int*)nullptr, h->wrap_ptr_p(), h->wrap_blk.p, h->wrap_row2.p);
            h->band_filled = true;
        } else
        LAUNCH(h, KID_FINALIZE, k_finalize_S<3>, gn, 64, 0, h->row_ptr.p, h->diag_slot.p, sc6.p, scf.p, h->Udiag, h->rhs, radius, O.min_lm_diagonal,
               O.max_lm_diagonal, n, h->S_val, h->Minv.p, h->rhs, h->Sff.p, h->scal.p);
        int pcg_iters = 0; bool pcg_ok = false;
        const bool fused_tail = node_major && O.preconditioner == 0;
        h->external_tail = fused_tail;
        { int rc = solve_reduced<3>(h, host_pcg, &pcg_iters, &pcg_ok, 0); h->external_tail = false; if (rc) return rc; }
        const double tol2 = O.pcg_tolerance * O.pcg_tolerance;
        const int nbr = (F.band_rows > 0 ? F.y_rows(3) : n) * 3;                   // stride of the right-hand-side columns in band order
        const int gstep = std::max(1, std::min(32, (3 * n + 1023) / 1024));      // workgroups of k_rot_step
        auto tail_fused = [&]() -> int {                                          // focal arrow + step + candidate | model change + candidate cost + residual check | hand-over
            hipLaunchKernelGGL(k_rot_step, dim3(gstep), dim3(1024), 0, st, n, h->Yb.p, h->Yb.p + nbr, h->Sfc, h->Sff.p, h->rhs + 3 * n, h->cam_pos.p, xx, fmx, sc3.p, scf.p, h->rhs, f_lo, f_hi,
                               h->px.p, xcand, fmc, step.p, h->scal.p, with_f ? 1 : 0, RS.step_part.p);
            hipLaunchKernelGGL(k_rot_eval, dim3(ge + gnode), dim3(64), 0, st, kind, E, ge, n, e0.p, e1.p, ec.p, G.scale, la, RS.ejac.p, step.p, xcand, fmc, h->row_ptr.p, h->col_idx.p, h->S_val,
                               h->Sfc, h->Sff.p, h->rhs, h->px.p, tol2, h->pr.p, RS.wg_part.p);
            // the two folds + the residual test + the hand-over in one single-workgroup launch (host_out null: the copy below hands over)
            LmGate g0; std::memset(&g0, 0, sizeof(g0));
            hipLaunchKernelGGL(k_rot_fold_publish, dim3(1), dim3(SC_TOTAL * 64), 0, st, RS.wg_part.p, ge + gnode, RS.step_part.p, gstep, n, h->Sff.p, h->rhs, h->px.p, tol2, h->pr.p,
                               h->scal.p, h->pcg.p, poll ? h->host_pub : (double*)nullptr, poll ? ++ctx->pub_seq : 0ull, g0, (double*)nullptr);
            if (!poll) SSFM_HIP_CHECK(ctx, hipMemcpyAsync(host_scal, h->scal.p, (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
            return SSFM_OK;
        };
        auto tail = [&]() -> int {
            hipLaunchKernelGGL(k_rot_update, dim3(1), dim3(1024), 0, st, n, xx, fmx, sc3.p, scf.p, h->px.p, h->rhs, f_lo, f_hi, 1.0, xcand, fmc, step.p, h->scal.p);
            hipLaunchKernelGGL(k_rot_edges, dim3(ge), dim3(64), 0, st, 1, kind, E, e0.p, e1.p, ec.p, G.scale, la, xx, fmx, sc3.p, scf.p, h->row_ptr.p, h->col_idx.p, n,
                               step.p, h->S_val, h->rhs, h->Udiag, h->Sfc, h->scal.p);
            hipLaunchKernelGGL(k_rot_cost, dim3(ge), dim3(64), 0, st, kind, E, e0.p, e1.p, ec.p, G.scale, la, xcand, fmc, h->scal.p + SC_CAND_COST);
            if (poll) publish(h);
            else SSFM_HIP_CHECK(ctx, hipMemcpyAsync(host_scal, h->scal.p, (SC_NSLOT * SC_TOTAL + PCG_TOTAL + 1) * sizeof(double), hipMemcpyDeviceToHost, st));
            return SSFM_OK;
        };
        auto wait_tail = [&]() -> int {
            if (poll) return wait_published(h);
            SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
            return SSFM_OK;
        };
        { int rc = fused_tail ? tail_fused() : tail(); if (rc) return rc; }
        // the other zone (the previous iteration's) is cleared while the host waits and decides -- unless every kernel of the iteration stores instead of
        // accumulating (node-major assembly + fused tail: k_rot_gather_nodes resets the two words that are not plainly stored)
        if (!fused_tail) SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->zone.p + (size_t)(zi ^ 1) * h->zone_len, 0, h->zone_len * sizeof(double), st));
        { int rc = wait_tail(); if (rc) return rc; }
        if (O.preconditioner == 0) {
            int ff; std::memcpy(&ff, &host_pcg[PCG_TOTAL], sizeof(int));
            if (ff) pcg_ok = false;
            else if (host_pcg[PCG_DONE] == 0.0) {
                int rc = solve_reduced<3>(h, host_pcg, &pcg_iters, &pcg_ok, 1); if (rc) return rc;
                SSFM_HIP_CHECK(ctx, hipMemsetAsync(h->scal.p + SC_MODEL, 0, 4 * sizeof(double), st));
                rc = tail(); if (rc) return rc;
                rc = wait_tail(); if (rc) return rc;
            }
        }
        S->pcg_iterations_total += pcg_iters; S->num_linearizations++;
        x_cost = host_scal[SC_COST];
        double gmax; { unsigned long long bits; std::memcpy(&bits, &host_scal[SC_GMAX], 8); std::memcpy(&gmax, &bits, 8); }
        if (iteration == 1) { S->initial_cost = x_cost; minimum_cost = x_cost; }
        if (!std::isfinite(x_cost)) { S->termination = SSFM_FAILURE; break; }
        if (last_successful && gmax <= O.gradient_tolerance) { S->termination = SSFM_CONVERGENCE; iteration--; break; }
        const double model_cost_change = -host_scal[SC_MODEL];
        const bool valid = pcg_ok && std::isfinite(model_cost_change) && model_cost_change > 0.0;
        if (!valid) {
            if (++num_invalid >= O.max_num_consecutive_invalid_steps) { S->termination = SSFM_FAILURE; break; }
            radius /= decrease_factor; decrease_factor *= 2.0; last_successful = false; S->num_unsuccessful_steps++;
            continue;
        }
        num_invalid = 0;
        double cand_cost = host_scal[SC_CAND_COST]; if (!std::isfinite(cand_cost)) cand_cost = std::numeric_limits<double>::max();
        double step_norm = std::sqrt(host_scal[SC_STEP2_CAM]), cand_xn2 = host_scal[SC_XN2_CAM];
        if (with_f) {
            // TrustRegionMinimizer::DoLineSearch (line_search.h).  f(1) is the candidate cost that is already here: only when it misses the
            // sufficient decrease does the search run (function evaluations = k_rot_ls_eval launches with a read-back each).
            const double g0 = host_scal[SC_GDELTA], dmax = host_scal[SC_DMAX];
            if (!(cand_cost <= x_cost + 1e-4 * g0)) {
                if (!ls_out.p) SSFM_HIP_CHECK(ctx, ls_out.alloc(2));
                int ls_rc = SSFM_OK; double last_f = 0;
                auto eval = [&](double a) {
                    LsSample q; q.a = a; double hv[2] = {0, 0};
                    hipError_t er = hipMemsetAsync(ls_out.p, 0, 2 * sizeof(double), st);
                    hipLaunchKernelGGL(k_rot_ls_eval, dim3(ge), dim3(64), 0, st, kind, E, e0.p, e1.p, ec.p, G.scale, la, xx, fmx, sc3.p, scf.p, step.p, a, f_lo, f_hi, n, ls_out.p);
                    if (er == hipSuccess) er = hipMemcpyAsync(hv, ls_out.p, 2 * sizeof(double), hipMemcpyDeviceToHost, st);
                    if (er == hipSuccess) er = hipStreamSynchronize(st);
                    if (er != hipSuccess) { ls_rc = fail(ctx, SSFM_ERR_HIP, std::string("line search evaluation: ") + hipGetErrorString(er)); return q; }
                    S->num_line_search_evaluations++;
                    if (std::isfinite(hv[0])) { q.f = hv[0]; q.has_f = true; last_f = hv[0]; if (std::isfinite(hv[1])) { q.df = hv[1]; q.has_df = true; } }
                    return q;
                };
                double a = 1.0;
                const bool found = ls_armijo(eval, x_cost, g0, dmax, &a);
                if (ls_rc) return ls_rc;
                if (found && a != 1.0) {
                    // delta *= a: rebuild the candidate and its norms; the model cost change stays that of the full step
                    hipLaunchKernelGGL(k_rot_update, dim3(1), dim3(1024), 0, st, n, xx, fmx, sc3.p, scf.p, h->px.p, h->rhs, f_lo, f_hi, a, xcand, fmc, step.p, h->scal.p);
                    double hs[SC_TOTAL];
                    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(hs, h->scal.p, SC_TOTAL * sizeof(double), hipMemcpyDeviceToHost, st));
                    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
                    step_norm = std::sqrt(hs[SC_STEP2_CAM]); cand_xn2 = hs[SC_XN2_CAM]; cand_cost = last_f;
                    S->num_line_search_contractions++;
                }
            }
        }
        if (step_norm <= O.parameter_tolerance * (x_norm + O.parameter_tolerance)) { S->termination = SSFM_CONVERGENCE; break; }
        const double cost_change = x_cost - cand_cost;
        if (std::fabs(cost_change) <= O.function_tolerance * x_cost) { S->termination = SSFM_CONVERGENCE; break; }
        const double rel = (cand_cost >= std::numeric_limits<double>::max()) ? std::numeric_limits<double>::lowest() : cost_change / model_cost_change;
        if (rel > O.min_relative_decrease) {
            std::swap(xx, xcand); std::swap(fmx, fmc);
            x_norm = std::sqrt(cand_xn2);
            { const double t3 = 2.0 * rel - 1.0; radius = std::fmin(O.max_trust_region_radius, radius / std::fmax(1.0 / 3.0, 1.0 - t3 * t3 * t3)); }
            decrease_factor = 2.0; last_successful = true; S->num_successful_steps++;
            x_cost = cand_cost; if (x_cost < minimum_cost) minimum_cost = x_cost;
        } else { radius /= decrease_factor; decrease_factor *= 2.0; last_successful = false; S->num_unsuccessful_steps++; }
        if (O.verbose) std::printf("[ssfm rot] iter %3d cost %.12e change %.3e |g| %.3e |step| %.3e rho %.3e radius %.3e %s\n", iteration, x_cost, cost_change,
                                   gmax, step_norm, rel, radius, last_successful ? "" : "(rejected)");
    }
    lap("LM loop");
    S->iterations = iteration; S->final_cost = (minimum_cost == std::numeric_limits<double>::max()) ? x_cost : minimum_cost;
    S->reduced_blocks = (int32_t)nnzb; S->band_half_width = F.band;
    // ---- back to rotation matrices: every rotation is re-exponentiated (src/rotation_averaging.cpp:88)
    std::vector<double> xf(nn); double fmult = 1.0;
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(xf.data(), xx, nn * sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipMemcpyAsync(&fmult, fmx, sizeof(double), hipMemcpyDeviceToHost, st));
    SSFM_HIP_CHECK(ctx, hipStreamSynchronize(st));
    for (int i = 0; i < n; i++) { double R[9]; so3exp(&xf[3 * i], R); for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) rotations[9 * i + a + 3 * b] = R[3 * a + b]; }
    if (with_f) *focal_length *= fmult;                                         // src/uncalibrated_pose_graph.cpp:200
    lap("download + so3exp");
    S->t_solve_s = wall_s() - t0;                          // RotScratch releases the buffers
    return SSFM_OK;
}

extern "C" int ssfm_rotavg_solve(ssfm_ctx* ctx, int32_t n, double* rotations, int32_t E, const int32_t* index0, const int32_t* index1,
                                 const double* rel_rotations, const ssfm_ba_options* o, ssfm_ba_summary* s) {
    return rot_solve(ctx, 0, n, rotations, E, index0, index1, rel_rotations, nullptr, 0, 0, o, s);
}

extern "C" int ssfm_posegraph_focal_solve(ssfm_ctx* ctx, int32_t n, double* rotations, int32_t E, const int32_t* index0, const int32_t* index1,
                                          const double* rel_rotations, double* focal_length, double min_focal, double max_focal,
                                          const ssfm_ba_options* o, ssfm_ba_summary* s) {
    if (!focal_length) return fail(ctx, SSFM_ERR_INVALID, "ssfm_posegraph_focal_solve: focal_length is null");
    return rot_solve(ctx, 2, n, rotations, E, index0, index1, rel_rotations, focal_length, min_focal, max_focal, o, s);
}
