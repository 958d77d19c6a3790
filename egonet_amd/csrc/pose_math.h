// pose_math.h -- per-instance float64 math of the crop affine and the pose
// solve, shared by the HIP kernels (geometry.hip) and by the host-compiled unit
// harness (tests/host_math_harness.cpp, built with g++), so the exact code the
// GPU runs is also checked on CPU against the golden fixtures.
#pragma once
#include <math.h>
#ifdef __HIPCC__
#define EGN_HD __host__ __device__
#else
#define EGN_HD
#endif

// Inverse crop affine for rot = 0, restating get_affine_transform(inv=1)
// (reference libs/common/img_proc.py:26-64): the three control points are
// rounded to float32 exactly like the reference's np.float32 arrays; the 2x3
// solve is closed form because both control triangles are axis aligned.
//   (u, v) crop pixels -> (X, Y) screen pixels
EGN_HD inline void egn_crop_to_screen(double cx, double cy, double scale_w, int crop_w, int crop_h,
                                      double u, double v, double* X, double* Y) {
  const double src_w = scale_w * 200.0;
  const double s0x = (double)(float)cx;
  const double s0y = (double)(float)cy;
  const double s1x = (double)(float)(cx + 0.0);
  const double s1y = (double)(float)(cy + src_w * -0.5);
  const float dyf = (float)(s0y - s1y);  // float32 a - b in get_3rd_point
  const double s2x = (double)(float)(s1x - (double)dyf);
  const double half_w = crop_w * 0.5;
  const double ax = (s1x - s2x) / half_w;
  const double ay = (s0y - s1y) / half_w;
  *X = s0x + (u - half_w) * ax;
  *Y = s0y + (v - crop_h * 0.5) * ay;
}

// cuboid edges as 0-based corner ids: 4 along h, 4 along l, 4 along w
// (reference libs/dataset/KITTI/car_instance.py:63-70, ids there are 1-based)
EGN_HD inline int egn_edge_parent(int e) {
  const int t[12] = {0, 2, 4, 6, 0, 1, 2, 3, 0, 1, 4, 5};
  return t[e];
}
EGN_HD inline int egn_edge_child(int e) {
  const int t[12] = {1, 3, 5, 7, 4, 5, 6, 7, 2, 3, 6, 7};
  return t[e];
}

// cyclic Jacobi eigen-decomposition of a symmetric 3x3 matrix; columns of v =
// eigenvectors, d = eigenvalues (unsorted); a is destroyed
EGN_HD inline void egn_jacobi3(double a[3][3], double v[3][3], double d[3]) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 24; ++sweep) {
    const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
    const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
    if (off <= 1e-300 || off <= 1e-17 * diag) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(a[p][q]) <= 1e-300) continue;
        const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0);
        const double s = t * c;
        for (int k = 0; k < 3; ++k) {  // A <- A J
          const double akp = a[k][p], akq = a[k][q];
          a[k][p] = c * akp - s * akq;
          a[k][q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {  // A <- J^T A
          const double apk = a[p][k], aqk = a[q][k];
          a[p][k] = c * apk - s * aqk;
          a[q][k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {  // V <- V J
          const double vkp = v[k][p], vkq = v[k][q];
          v[k][p] = c * vkp - s * vkq;
          v[k][q] = s * vkp + c * vkq;
        }
      }
  }
  for (int i = 0; i < 3; ++i) d[i] = a[i][i];
}

EGN_HD inline void egn_cross3(const double a[3], const double b[3], double c[3]) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}

// One instance of the pose solve.  P = predicted cuboid [32][3] (root dropped).
//   euler_xyz[3]: rotation about x, y, z (reference egonet.py:265-277)
//   returns the observation angle alpha (egonet.py:203-236)
EGN_HD inline double egn_pose_solve_one(const double* P, double kpt_x, double fx, double cxp,
                                        int alpha_mode, double* euler_xyz) {
  // mean edge lengths (egonet.py:243-248)
  double len[3] = {0, 0, 0};
  for (int e = 0; e < 12; ++e) {
    const double* a = P + 3 * egn_edge_parent(e);
    const double* b = P + 3 * egn_edge_child(e);
    const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    len[e >> 2] += sqrt(dx * dx + dy * dy + dz * dz);
  }
  const double h = len[0] / 4, l = len[1] / 4, w = len[2] / 4;

  // canonical template: 8 corners + 24 interpolated points (egonet.py:249-263);
  // the centring offsets are float32-rounded in the reference (np.float32(l)/2)
  double T[32][3];
  const double offx = (double)((float)l / 2.0f), offy = (double)(float)h,
               offz = (double)((float)w / 2.0f);
  for (int c = 0; c < 8; ++c) {
    T[c][0] = ((c < 4) ? l : 0.0) - offx;
    T[c][1] = ((c & 1) ? h : 0.0) - offy;
    T[c][2] = (((c >> 1) & 1) ? 0.0 : w) - offz;
  }
  const double coef[2] = {0.332, 0.667};
  for (int k = 0; k < 2; ++k)
    for (int e = 0; e < 12; ++e)
      for (int d = 0; d < 3; ++d) {
        const double pa = T[egn_edge_parent(e)][d], ch = T[egn_edge_child(e)][d];
        T[8 + 12 * k + e][d] = pa + coef[k] * (ch - pa);
      }

  // Kabsch (transformation.py:99-134): H = (X - mx)(Y - my)^T,
  // X = template, Y = prediction
  double mx[3] = {0, 0, 0}, my[3] = {0, 0, 0};
  for (int i = 0; i < 32; ++i)
    for (int d = 0; d < 3; ++d) {
      mx[d] += T[i][d];
      my[d] += P[3 * i + d];
    }
  for (int d = 0; d < 3; ++d) {
    mx[d] /= 32.0;
    my[d] /= 32.0;
  }
  double Hm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  for (int i = 0; i < 32; ++i)
    for (int r = 0; r < 3; ++r)
      for (int c = 0; c < 3; ++c) Hm[r][c] += (T[i][r] - mx[r]) * (P[3 * i + c] - my[c]);

  // H = U S V^T.  v_k (prediction space) = eigenvectors of H^T H for the two
  // largest eigenvalues, u_k = H v_k / |H v_k| (template space).  R = V U^T with
  // the third pair replaced by cross products equals the reference's
  // det-corrected V diag(1,1,d) U^T (transformation.py:121-132) and needs no
  // third singular value.
  double A[3][3], V[3][3], ev[3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c)
      A[r][c] = Hm[0][r] * Hm[0][c] + Hm[1][r] * Hm[1][c] + Hm[2][r] * Hm[2][c];
  egn_jacobi3(A, V, ev);
  int i0 = 0;
  if (ev[1] > ev[i0]) i0 = 1;
  if (ev[2] > ev[i0]) i0 = 2;
  int i1 = (i0 == 0) ? 1 : 0;
  for (int k = 0; k < 3; ++k)
    if (k != i0 && ev[k] > ev[i1]) i1 = k;
  double v0[3], v1[3], v2[3], u0[3], u1[3], u2[3];
  for (int d = 0; d < 3; ++d) {
    v0[d] = V[d][i0];
    v1[d] = V[d][i1];
  }
  for (int r = 0; r < 3; ++r) {
    u0[r] = Hm[r][0] * v0[0] + Hm[r][1] * v0[1] + Hm[r][2] * v0[2];
    u1[r] = Hm[r][0] * v1[0] + Hm[r][1] * v1[1] + Hm[r][2] * v1[2];
  }
  const double n0 = sqrt(u0[0] * u0[0] + u0[1] * u0[1] + u0[2] * u0[2]);
  for (int d = 0; d < 3; ++d) u0[d] /= n0;
  const double dot = u0[0] * u1[0] + u0[1] * u1[1] + u0[2] * u1[2];
  for (int d = 0; d < 3; ++d) u1[d] -= dot * u0[d];
  const double n1 = sqrt(u1[0] * u1[0] + u1[1] * u1[1] + u1[2] * u1[2]);
  for (int d = 0; d < 3; ++d) u1[d] /= n1;
  egn_cross3(u0, u1, u2);
  egn_cross3(v0, v1, v2);
  double R[3][3];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) R[r][c] = v0[r] * u0[c] + v1[r] * u1[c] + v2[r] * u2[c];

  // Rotation.from_matrix(R).as_euler('yxz') (extrinsic): R = Rz(c) Rx(b) Ry(a);
  // reference re-orders to (x, y, z) = (b, a, c)  (egonet.py:273-276)
  double sb = R[2][1];
  sb = fmin(1.0, fmax(-1.0, sb));
  const double b = asin(sb);
  double a, c;
  if (fabs(sb) < 1.0 - 1e-12) {
    a = atan2(-R[2][0], R[2][2]);
    c = atan2(-R[0][1], R[1][1]);
  } else {  // gimbal lock: scipy sets the third angle to zero
    c = 0.0;
    a = atan2(R[0][2], R[0][0]);
  }
  euler_xyz[0] = b;
  euler_xyz[1] = a;
  euler_xyz[2] = c;

  double x3, z3;
  if (alpha_mode == 0) {
    x3 = kpt_x - cxp;
    z3 = fx;
  } else {
    x3 = P[0];
    z3 = P[2];
  }
  const double pi = 3.14159265358979323846;
  double al = a - atan2(-z3, x3) - 0.5 * pi;
  while (al > pi) al -= 2.0 * pi;
  while (al < -pi) al += 2.0 * pi;
  return al;
}
