// Rigid-fit helpers shared by the LGR kernels (a15/a16) and the RANSAC estimator (§8f rank 4).
#pragma once
#include "common.h"

namespace rdm {

// Largest eigenvector of the symmetric 4x4 `a` (cyclic Jacobi, fp64).  q = (w, x, y, z).
// `basis` (optional, orthonormal columns, updated in place) is the starting frame of the sweeps: the eigenvectors of a
// nearby matrix -- the previous refinement step's -- leave V^T a V almost diagonal and the iteration converges in one or
// two sweeps instead of six or seven; the stopping rule, and so the accuracy, is the same.
__device__ inline void horn_quaternion(double a[4][4], double q[4], double (*basis)[4] = nullptr) {
  double vmat[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  if (basis) {
    double av[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        vmat[i][j] = basis[i][j];
        av[i][j] = 0.0;
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) av[i][j] += a[i][k] * vmat[k][j];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = i; j < 4; ++j) {  // V^T (a V), symmetrised
        double t = 0.0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t += vmat[k][i] * av[k][j];
        a[i][j] = t;
        a[j][i] = t;
      }
  }
  for (int sweep = 0; sweep < 24; ++sweep) {
    double offd = 0.0, diag = 0.0;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      diag += a[p][p] * a[p][p];
#pragma unroll
      for (int r = p + 1; r < 4; ++r) offd += a[p][r] * a[p][r];
    }
    if (offd <= 1e-34 * (diag + offd) || offd < 1e-300) break;  // converged to fp64 round-off
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int r = p + 1; r < 4; ++r) {
        if (fabs(a[p][r]) < 1e-300) continue;
        const double theta = (a[r][r] - a[p][p]) / (2.0 * a[p][r]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = a[k][p], akr = a[k][r];
          a[k][p] = c * akp - s * akr;
          a[k][r] = s * akp + c * akr;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = a[p][k], ark = a[r][k];
          a[p][k] = c * apk - s * ark;
          a[r][k] = s * apk + c * ark;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = vmat[k][p], vkr = vmat[k][r];
          vmat[k][p] = c * vkp - s * vkr;
          vmat[k][r] = s * vkp + c * vkr;
        }
      }
  }
  // column of the largest eigenvalue (the first one among equals), picked with selects: a run-time column index
  // would put vmat into scratch memory and every rotation above would pay a memory round trip
  double best = a[0][0], v[4] = {vmat[0][0], vmat[1][0], vmat[2][0], vmat[3][0]};
#pragma unroll
  for (int i = 1; i < 4; ++i) {
    const bool take = a[i][i] > best;
    best = take ? a[i][i] : best;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = take ? vmat[k][i] : v[k];
  }
  if (basis)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) basis[i][j] = vmat[i][j];
  double nrm = 0.0;
#pragma unroll
  for (int k = 0; k < 4; ++k) nrm += v[k] * v[k];
  nrm = sqrt(nrm);
#pragma unroll
  for (int k = 0; k < 4; ++k) q[k] = v[k] / nrm;
}


// Rotation (row-major 3x3) that maximises trace(R^T-aligned covariance) for H[a][b] = sum s_a r_b (centred):
// Horn's quaternion form of the Kabsch problem, the same solution as V diag(1,1,det) U^T of the SVD route
// (modules/registration/procrustes.py:53-63, Eigen::umeyama without scaling).
__device__ inline void kabsch_rotation(const double H[9], double R[9]) {
  const double Sxx = H[0], Sxy = H[1], Sxz = H[2], Syx = H[3], Syy = H[4], Syz = H[5], Szx = H[6], Szy = H[7],
               Szz = H[8];
  double N[4][4] = {{Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx},
                    {Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz},
                    {Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy},
                    {Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz}};
  double q[4];
  horn_quaternion(N, q);
  const double qw = q[0], qx = q[1], qy = q[2], qz = q[3];
  R[0] = 1 - 2 * (qy * qy + qz * qz); R[1] = 2 * (qx * qy - qz * qw);     R[2] = 2 * (qx * qz + qy * qw);
  R[3] = 2 * (qx * qy + qz * qw);     R[4] = 1 - 2 * (qx * qx + qz * qz); R[5] = 2 * (qy * qz - qx * qw);
  R[6] = 2 * (qx * qz - qy * qw);     R[7] = 2 * (qy * qz + qx * qw);     R[8] = 1 - 2 * (qx * qx + qy * qy);
}

}  // namespace rdm
