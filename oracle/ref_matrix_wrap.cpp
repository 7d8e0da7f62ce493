// ref_matrix_wrap.cpp -- TEST INFRASTRUCTURE.  Thin extern "C" access to the
// reference's own dense Matrix class, compiled from
// /root/reference/ccicp2d/src/matrix.cpp where it lies (see oracle/Makefile;
// output goes to oracle/_ref/, which is git-ignored).  It is the only part of
// the reference ICP core that builds in this image without stand-in headers
// (kdtree/icp/icpPointToPoint need boost and ROS).  Used to pin the oracle's
// closed-form solves and to generate tests/golden/solve_golden.npz.
//
// Each entry replays the reference's statement sequence for one solve with
// the reference's Matrix operators, so operation order and rounding are the
// reference's own.
#include "ccicp2d/matrix.h"

#include <algorithm>
#include <cmath>

extern "C" {

// H.svd(U,W,V) -- matrix.cpp:582-810 -- on a 2x2; outputs row-major U,V and W.
void ref_svd2(const double H_[4], double U_[4], double W_[2], double V_[4])
{
    Matrix H(2, 2, H_);
    Matrix U, W, V;
    H.svd(U, W, V);
    for (int i = 0; i < 2; i++) {
        W_[i] = W.val[i][0];
        for (int j = 0; j < 2; j++) {
            U_[2 * i + j] = U.val[i][j];
            V_[2 * i + j] = V.val[i][j];
        }
    }
}

// icpPointToPoint.cpp:160-162: H.svd(U,W,V); R_ = V*~U
void ref_p2p_rotation(const double H_[4], double R_out[4])
{
    Matrix H(2, 2, H_);
    Matrix U, W, V;
    H.svd(U, W, V);
    Matrix R_ = V * ~U;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) R_out[2 * i + j] = R_.val[i][j];
}

// icpPointToPoint.cpp:149-171: the whole solve half of fitStep from a list of
// n correspondences (p_m, p_t as xy rows) and the running sums mu_m, mu_t the
// loops at :59-126 accumulate.  R (2x2) and t (2x1) are updated in place.
double ref_fitstep_solve(const double *pm, const double *pt, int n, double R_io[4],
                         double t_io[2])
{
    Matrix p_m(n, 2), p_t(n, 2), mu_m(1, 2), mu_t(1, 2);
    for (int i = 0; i < n; i++) {
        p_m.val[i][0] = pm[2 * i];
        mu_m.val[0][0] += p_m.val[i][0];
        p_m.val[i][1] = pm[2 * i + 1];
        mu_m.val[0][1] += p_m.val[i][1];
        p_t.val[i][0] = pt[2 * i];
        mu_t.val[0][0] += p_t.val[i][0];
        p_t.val[i][1] = pt[2 * i + 1];
        mu_t.val[0][1] += p_t.val[i][1];
    }
    Matrix R(2, 2, R_io), t(2, 1, t_io);
    mu_m = mu_m / (double)n;
    mu_t = mu_t / (double)n;
    Matrix q_m = p_m - Matrix::ones(n, 1) * mu_m;
    Matrix q_t = p_t - Matrix::ones(n, 1) * mu_t;
    Matrix H = ~q_t * q_m;
    Matrix U, W, V;
    H.svd(U, W, V);
    Matrix R_ = V * ~U;
    Matrix t_ = ~mu_m - R_ * ~mu_t;
    R = R_ * R;
    t = R_ * t + t_;
    for (int i = 0; i < 2; i++) {
        t_io[i] = t.val[i][0];
        for (int j = 0; j < 2; j++) R_io[2 * i + j] = R.val[i][j];
    }
    return std::max((R_ - Matrix::eye(2)).l2norm(), t_.l2norm());
}

// b_.solve(A_) -- matrix.cpp:420-508 -- 3x3 system, one right-hand side.
int ref_solve3(const double A_[9], const double b_[3], double x_out[3])
{
    Matrix A(3, 3, A_), b(3, 1, b_);
    bool   ok = b.solve(A);
    for (int i = 0; i < 3; i++) x_out[i] = b.val[i][0];
    return ok ? 1 : 0;
}

// MMi = MM; MMi.inv() -- matrix.cpp:393-402
void ref_inv3(const double A_[9], double out[9])
{
    Matrix A(3, 3, A_);
    A.inv();
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) out[3 * i + j] = A.val[i][j];
}

// icpPointToPlane.cpp:88-95: R_ = eye(2); R_[0][1] = -w; R_[1][0] = +w; svd; R_ = U*~V
void ref_orthonormal_from_omega(double w, double R_out[4])
{
    Matrix R_ = Matrix::eye(2);
    R_.val[0][1] = -w;
    R_.val[1][0] = +w;
    Matrix U, W, V;
    R_.svd(U, W, V);
    R_ = U * ~V;
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) R_out[2 * i + j] = R_.val[i][j];
}

// icpPointToPlane.cpp:279-305 (computeNormal, dim 2) from the k neighbours' coordinates as xy rows
void ref_normal2(const double *nb_xy, int k, double n_out[2])
{
    Matrix P(k, 2);
    Matrix mu(1, 2);
    for (int i = 0; i < k; i++) {
        double x = nb_xy[2 * i];
        double y = nb_xy[2 * i + 1];
        P.val[i][0] = x;
        P.val[i][1] = y;
        mu.val[0][0] += x;
        mu.val[0][1] += y;
    }
    mu = mu / (double)k;
    Matrix Q = P - Matrix::ones(k, 1) * mu;
    Matrix H = ~Q * Q;
    Matrix U, W, V;
    H.svd(U, W, V);
    n_out[0] = U.val[0][1];
    n_out[1] = U.val[1][1];
}

} // extern "C"
