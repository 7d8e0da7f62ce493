// slam_amd/icp.hpp -- header-only C++ adapters with the reference's own class
// shapes over the C-ABI (slam_mi355x.h), so ccicp2d callers compile unchanged:
//
//   reference (ccicp2d)                         here
//   class Matrix            matrix.h:50-132     slam_amd::Matrix  (the 2x2 / 2x1 subset fit() uses: val[i][j], m, n)
//   class Icp               icp.h:33-102        slam_amd::Icp
//   class IcpPointToPoint   icpPointToPoint.h:26-40   slam_amd::IcpPointToPoint
//   class IcpPointToPlane   icpPointToPlane.h:26-49   slam_amd::IcpPointToPlane (stale upstream: not in the reference's build)
//
// Conventions kept from the reference: the constructor copies the model
// (icp.cpp:51-60) and logs instead of throwing when it has fewer than 5
// points (icp.cpp:38-43: the object is then unusable and fit() returns
// without touching R,t); fit() is void and synchronous; R and t are in/out.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "slam_mi355x.h"

namespace slam_amd {

// The part of libicp's Matrix that Icp::fit touches: row-pointer access val[i][j].
class Matrix {
public:
    Matrix() : Matrix(0, 0) {}
    Matrix(int32_t m_, int32_t n_) : m(m_), n(n_) { alloc(); }
    Matrix(int32_t m_, int32_t n_, const double *v) : m(m_), n(n_)
    {
        alloc();
        for (int32_t i = 0; i < m * n; ++i) data_[i] = v[i];
    }
    Matrix(const Matrix &o) : m(o.m), n(o.n)
    {
        alloc();
        std::memcpy(data_, o.data_, sizeof(double) * (size_t)(m * n));
    }
    Matrix &operator=(const Matrix &o)
    {
        if (this != &o) {
            release();
            m = o.m;
            n = o.n;
            alloc();
            std::memcpy(data_, o.data_, sizeof(double) * (size_t)(m * n));
        }
        return *this;
    }
    ~Matrix() { release(); }
    static Matrix eye(int32_t k)
    {
        Matrix M(k, k);
        for (int32_t i = 0; i < k; ++i) M.val[i][i] = 1.0;
        return M;
    }
    double **val = nullptr;
    int32_t  m, n;

private:
    void alloc()
    {
        data_ = (m * n > 0) ? new double[(size_t)(m * n)]() : nullptr;
        val = (m > 0) ? new double *[(size_t)m] : nullptr;
        for (int32_t i = 0; i < m; ++i) val[i] = data_ + (size_t)i * n;
    }
    void release()
    {
        delete[] data_;
        delete[] val;
        data_ = nullptr;
        val = nullptr;
    }
    double *data_ = nullptr;
};

class Icp {
public:
    // icp.h:42 / icp.cpp:26-70
    Icp(double *M_GA, double *M_NGA, const int32_t M_GA_num, const int32_t M_NGA_num, const int32_t dim)
    {
        slam_icp_params p;
        slam_icp_default_params(&p);
        create(M_GA, M_NGA, M_GA_num, M_NGA_num, dim, p);
    }

protected:
    Icp() {} // (a derived class that makes the handle with parameters of its own: IcpPointToPlane)
    void create(double *M_GA, double *M_NGA, const int32_t M_GA_num, const int32_t M_NGA_num, const int32_t dim, const slam_icp_params &p)
    {
        if (dim != 2) { // the reference also takes 3; this path is the 2-D one (icpTools.cpp:187 passes 2)
            std::fprintf(stderr, "LIBICP works only for data of dimensionality 2 here\n");
            return;
        }
        const int rc = slam_icp_create(M_GA, M_GA_num, M_NGA, M_NGA_num, &p, &h_);
        if (rc != SLAM_OK) {
            std::fprintf(stderr, "%s\n", slam_last_error()); // ROS_ERROR_STREAM in the reference
            h_ = nullptr;
        }
    }

public:
    virtual ~Icp() { slam_icp_destroy(h_); }
    Icp(const Icp &) = delete;
    Icp &operator=(const Icp &) = delete;

    void setSubsamplingStep(int32_t val) { if (h_) slam_icp_set_subsampling_step(h_, val); } // icp.h:48
    void setMaxIterations(int32_t val) { if (h_) slam_icp_set_max_iterations(h_, val); }     // icp.h:51
    void setMinDeltaParam(double val) { if (h_) slam_icp_set_min_delta(h_, val); }           // icp.h:54

    // icp.h:65 / icp.cpp:80-114.  h_dist is ignored there too.
    void fit(double *T_GA, double *T_NGA, const int32_t T_GA_num, const int32_t T_NGA_num, Matrix &R, Matrix &t,
             double indist, double /*h_dist*/)
    {
        if (!h_) return;
        double Rr[4] = {R.val[0][0], R.val[0][1], R.val[1][0], R.val[1][1]};
        double tt[2] = {t.val[0][0], t.val[1][0]};
        slam_icp_result res;
        const int rc = slam_icp_fit(h_, T_GA, T_GA_num, T_NGA, T_NGA_num, Rr, tt, indist, &res);
        if (rc != SLAM_OK) { // icp.cpp:100-103: log, return, R and t untouched
            std::fprintf(stderr, "%s\n", slam_last_error());
            return;
        }
        numCorr_ = res.n_corr;
        R.val[0][0] = Rr[0];
        R.val[0][1] = Rr[1];
        R.val[1][0] = Rr[2];
        R.val[1][1] = Rr[3];
        t.val[0][0] = tt[0];
        t.val[1][0] = tt[1];
    }
    void getEdgeWeight(double *eW) { if (h_) slam_icp_get_edge_weight(h_, eW); }             // icp.h:67
    int  getNumberCorrespondences(void) { return numCorr_; }                                 // icp.h:68
    bool valid() const { return h_ != nullptr; }

protected:
    slam_icp_t *h_ = nullptr;
    int         numCorr_ = 0;
};

// icpPointToPoint.h:26-40
class IcpPointToPoint : public Icp {
public:
    IcpPointToPoint(double *M_GA, double *M_NGA, const int32_t M_GA_num, const int32_t M_NGA_num, const int32_t dim)
        : Icp(M_GA, M_NGA, M_GA_num, M_NGA_num, dim) {}
    virtual ~IcpPointToPoint() {}
};

// icpPointToPlane.h:26-49 -- the point-to-line matcher (2-D branch of icpPointToPlane.cpp:37-107; normals :279-305, :340-349).
// Upstream this class is stale: its header still has libicp's one-cloud constructor `IcpPointToPlane(M, M_num, dim, num_neighbors,
// flatness)` over `Icp(M, M_num, dim)`, which this fork's icp.h (two classes of points) no longer offers, and the file is not in
// ccicp2d/CMakeLists.txt:24.  Both shapes are here: the stale header's, and this fork's two-array shape -- the classes are merged
// (the point-to-line step has none, :55-77; GA first, the order of M_normal).  `flatness` is accepted and unused, as upstream
// (computeNormal never reads it, :270-305).
class IcpPointToPlane : public Icp {
public:
    // icpPointToPlane.h:30
    IcpPointToPlane(double *M, const int32_t M_num, const int32_t dim, const int32_t num_neighbors = 10, const double /*flatness*/ = 5.0)
    {
        make(nullptr, M, 0, M_num, dim, num_neighbors);
    }
    // the same over this fork's constructor shape (icp.h:42)
    IcpPointToPlane(double *M_GA, double *M_NGA, const int32_t M_GA_num, const int32_t M_NGA_num, const int32_t dim,
                    const int32_t num_neighbors = 10, const double /*flatness*/ = 5.0)
    {
        make(M_GA, M_NGA, M_GA_num, M_NGA_num, dim, num_neighbors);
    }
    virtual ~IcpPointToPlane() {}
    // libicp's one-cloud fit (what icpPointToPlane.cpp was written against): every template point is active -- the step has
    // no inlier gate (:55-77) -- so `indist` selects nothing here
    void fit(double *T, const int32_t T_num, Matrix &R, Matrix &t, const double indist)
    {
        Icp::fit(nullptr, T, 0, T_num, R, t, indist, 0.0);
    }
    using Icp::fit; // ... and this fork's two-array fit (icp.h:65)
    // M_normal (icpPointToPlane.h:48): x, y per model point, GA then NGA
    bool getNormals(double *normals_xy) { return h_ && slam_icp_get_normals(h_, normals_xy) == SLAM_OK; }

private:
    void make(double *M_GA, double *M_NGA, int32_t n_ga, int32_t n_nga, int32_t dim, int32_t num_neighbors)
    {
        slam_icp_params p;
        slam_icp_default_params(&p);
        p.mode = SLAM_ICP_P2L;
        p.normals_k = num_neighbors;
        create(M_GA, M_NGA, n_ga, n_nga, dim, p);
    }
};

} // namespace slam_amd
