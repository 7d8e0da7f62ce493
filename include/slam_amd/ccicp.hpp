// slam_amd/ccicp.hpp -- header-only adapter with the shape of class CCICP
// (ccicp2d/include/ccicp2d/icpTools.h:28-87), the facade scan_registration calls
// (scan_registration.cpp:57,82-83,97-98,139,159), over the C-ABI (slam_mi355x.h):
//
//   reference                                       here (clouds as float arrays, `stride` floats per point)
//   CCICP(RegistrationType)          icpTools.cpp:14-33     CCICP(RegistrationType)
//   setTargetCloud(cloud, pose)      :585-608               setTargetCloud(xyz, n, stride, pose)
//   setTargetGndCloud(cloud)         :580-583               setTargetGndCloud(xyz, n, stride)
//   setSceneCloud(cloud)             :611-634               setSceneCloud(xyz, n, stride)
//   doICPMatch(initPose)             :222-298               doICPMatch(initPose)
//   doICPMatch(target, scene, pose)  :571-578               doICPMatch(target..., scene..., initPose)
//   getResidual()                    :637-641 (returns -1)  getResidual()
//
// The clouds stay on the device between the steps (ground segmentation, GA/NGA classification, voxel
// filter, crop + split, ICP, height recovery); PCL and tf types are replaced by plain arrays and the
// Pose struct below (the pose part of geometry_msgs::PoseStamped).  The tf calls (getYaw, getEulerYPR,
// createQuaternionFromRPY: icpTools.cpp:174,205-212) are restated here as tf defines them.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "slam_mi355x.h"

namespace slam_amd {

enum RegistrationType { SCAN_TO_SCAN = 0, SCAN_TO_MAP = 1 }; // icpTools.h

#ifndef SLAM_AMD_POSE_DEFINED
#define SLAM_AMD_POSE_DEFINED
struct Pose { // geometry_msgs::Pose
    double x = 0, y = 0, z = 0;
    double qx = 0, qy = 0, qz = 0, qw = 1;
};
#endif

namespace detail {
// tf::Matrix3x3(q).getEulerYPR(yaw, pitch, roll, 1)
inline void euler_ypr(const Pose &p, double &yaw, double &pitch, double &roll)
{
    const double d = p.qx * p.qx + p.qy * p.qy + p.qz * p.qz + p.qw * p.qw, s = 2.0 / d;
    const double xs = p.qx * s, ys = p.qy * s, zs = p.qz * s, wx = p.qw * xs, wy = p.qw * ys, wz = p.qw * zs,
                 xx = p.qx * xs, xy = p.qx * ys, xz = p.qx * zs, yy = p.qy * ys, yz = p.qy * zs, zz = p.qz * zs;
    const double m00 = 1.0 - (yy + zz), m10 = xy + wz, m20 = xz - wy, m21 = yz + wx, m22 = 1.0 - (xx + yy),
                 m01 = xy - wz, m02 = xz + wy;
    if (std::fabs(m20) >= 1) { // gimbal lock branch of tf
        yaw = 0;
        const double delta = std::atan2(m01, m02);
        if (m20 < 0) {
            pitch = M_PI / 2.0;
            roll = delta;
        } else {
            pitch = -M_PI / 2.0;
            roll = delta;
        }
    } else {
        pitch = -std::asin(m20);
        roll = std::atan2(m21 / std::cos(pitch), m22 / std::cos(pitch));
        yaw = std::atan2(m10 / std::cos(pitch), m00 / std::cos(pitch));
    }
}
// tf::createQuaternionFromRPY
inline void quat_from_rpy(double roll, double pitch, double yaw, Pose &p)
{
    const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
    const double cy = std::cos(hy), sy = std::sin(hy), cp = std::cos(hp), sp = std::sin(hp), cr = std::cos(hr),
                 sr = std::sin(hr);
    p.qx = sr * cp * cy - cr * sp * sy;
    p.qy = cr * sp * cy + sr * cp * sy;
    p.qz = cr * cp * sy - sr * sp * cy;
    p.qw = cr * cp * cy + sr * sp * sy;
}
} // namespace detail

class CCICP {
public:
    static constexpr int ICP_MAX_PTS = 20000; // icpTools.h:21

    explicit CCICP(RegistrationType type_ = SCAN_TO_SCAN) : type(type_)
    {
        if (slam_gseg_create(nullptr, &gseg_) != SLAM_OK || slam_ccicp_create(&cc_) != SLAM_OK)
            std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
        ok(slam_malloc((void **)&d_ga_, 16 * (size_t)ICP_MAX_PTS));
        ok(slam_malloc((void **)&d_nga_, 16 * (size_t)ICP_MAX_PTS));
        ok(slam_malloc((void **)&d_cnt_, 16));
    }
    ~CCICP()
    {
        slam_device_synchronize();
        for (Cloud *c : {&raw_, &labels_, &obs_, &flags_, &seg_target_, &seg_scene_, &ground_target_, &ground_scene_, &tmp_})
            slam_free(c->p);
        slam_free(d_ga_);
        slam_free(d_nga_);
        slam_free(d_cnt_);
        slam_ccicp_destroy(cc_);
        slam_gseg_destroy(gseg_);
    }
    CCICP(const CCICP &) = delete;
    CCICP &operator=(const CCICP &) = delete;

    // icpTools.cpp:585-608.  SCAN_TO_MAP: the target is an obstacle cloud already (the global map):
    // classifyPoints only.  SCAN_TO_SCAN: segmentGround first; its ground cloud becomes ground_target.
    void setTargetCloud(const float *xyz, int n, int stride, const Pose & /*initPose*/)
    {
        if (type == SCAN_TO_MAP) {
            upload(xyz, n, stride);
            select(0xffu, obs_, obs_n_); // copyPointCloud(*target, *seg_target) (:592)
            classify_into(seg_target_, seg_target_n_, false);
        } else {
            segment(xyz, n, stride, ground_target_, ground_target_n_);
            classify_into(seg_target_, seg_target_n_, false);
        }
    }
    void setTargetGndCloud(const float *xyz, int n, int stride) // :580-583
    {
        upload(xyz, n, stride);
        select(0xffu, ground_target_, ground_target_n_); // copyPointCloud
    }
    void setSceneCloud(const float *xyz, int n, int stride) // :611-634
    {
        segment(xyz, n, stride, tmp_, tmp_n_);
        classify_into(seg_scene_, seg_scene_n_, true); // voxel filter 0.5, 0.5, 2
        reserve(ground_scene_, 16 * (size_t)(tmp_n_ + 1));
        ground_scene_n_ = 0;
        if (tmp_n_ > 0)
            ok(slam_ccicp_voxel_downsample_dev(cc_, (const float *)tmp_.p, nullptr, tmp_n_, 4, 0.5f, 0.5f, 5.0f,
                                               (float *)ground_scene_.p, tmp_n_, &ground_scene_n_, nullptr));
    }

    Pose doICPMatch(const float *target, int n_target, const float *scene, int n_scene, int stride, const Pose &initPose)
    {
        setTargetCloud(target, n_target, stride, initPose); // :571-578
        setSceneCloud(scene, n_scene, stride);
        return doICPMatch(initPose);
    }

    Pose doICPMatch(const Pose &initPose) // :222-298
    {
        int mc[2] = {0, 0}, sc[2] = {0, 0};
        std::vector<double> m_ga, m_nga, s_ga, s_nga;
        // target: crop +-75 m around the pose (:225-239), split with the cap (:263-276)
        ok(slam_ccicp_split_dev(cc_, (const float *)seg_target_.p, seg_target_n_, 4, 1, initPose.x, initPose.y, 75.0,
                                ICP_MAX_PTS, d_ga_, d_nga_, mc, nullptr));
        download(m_ga, d_ga_, mc[0]);
        download(m_nga, d_nga_, mc[1]);
        ok(slam_ccicp_split_dev(cc_, (const float *)seg_scene_.p, seg_scene_n_, 4, 0, 0, 0, 0, ICP_MAX_PTS, d_ga_,
                                d_nga_, sc, nullptr)); // :248-261
        download(s_ga, d_ga_, sc[0]);
        download(s_nga, d_nga_, sc[1]);
        n_model_[0] = mc[0], n_model_[1] = mc[1], n_scene_[0] = sc[0], n_scene_[1] = sc[1];

        Pose   result;
        double yaw0, pitch0, roll0;
        detail::euler_ypr(initPose, yaw0, pitch0, roll0); // tf::getYaw (:174)
        double R[4] = {std::cos(yaw0), -std::sin(yaw0), std::sin(yaw0), std::cos(yaw0)};
        double t[2] = {initPose.x, initPose.y};
        if (sc[0] + sc[1] < 5) { // :179-184
            std::fprintf(stderr, "ERROR: Total Scene has %d points\n", sc[0] + sc[1]);
            result = Pose();
            result.qw = 9999;
            return result;
        }
        slam_icp_t *icp = nullptr; // IcpPointToPoint icp(refPts_GA, refPts_NGA, ...) (:187)
        num_corr_ = 0;
        if (slam_icp_create(m_ga.data(), mc[0], m_nga.data(), mc[1], nullptr, &icp) == SLAM_OK) {
            slam_icp_result res;
            if (slam_icp_fit(icp, s_ga.data(), sc[0], s_nga.data(), sc[1], R, t, 5.0, &res) == SLAM_OK) // :188
                num_corr_ = res.n_corr;
            else
                std::fprintf(stderr, "%s\n", slam_last_error());
            slam_icp_destroy(icp);
        } else {
            std::fprintf(stderr, "%s\n", slam_last_error()); // fewer than 5 model points: R,t stay (icp.cpp:38-43)
        }
        const double corr_yaw = std::atan2(R[2], R[0]); // :197
        result.x = t[0];
        result.y = t[1];
        result.z = initPose.z;
        detail::quat_from_rpy(roll0, pitch0, corr_yaw, result); // :205-212
        // doHeightInterpolate(ground_target, result_2d) (:295, :301-381)
        double z = result.z;
        const double pose7[7] = {result.x, result.y, result.z, result.qx, result.qy, result.qz, result.qw};
        ok(slam_ccicp_height_dev(cc_, (const float *)ground_target_.p, ground_target_n_, 4, pose7, &z, nullptr, nullptr,
                                 nullptr));
        result.z = z;
        return result;
    }

    double getResidual() const { return -1; } // :637-641 ("TODO: calculate this somehow")
    // :644-650: copies of seg_target, seg_scene, ground_target, ground_scene as x, y, z per point
    void getSegmentedClouds(std::vector<float> &target, std::vector<float> &scene, std::vector<float> &g_target,
                            std::vector<float> &g_scene) const
    {
        copy_out(seg_target_, seg_target_n_, target);
        copy_out(seg_scene_, seg_scene_n_, scene);
        copy_out(ground_target_, ground_target_n_, g_target);
        copy_out(ground_scene_, ground_scene_n_, g_scene);
    }
    int    getNumberCorrespondences() const { return num_corr_; }
    // sizes of what getSegmentedClouds would copy out (:644-650)
    int targetSize() const { return seg_target_n_; }
    int sceneSize() const { return seg_scene_n_; }
    int groundTargetSize() const { return ground_target_n_; }
    int groundSceneSize() const { return ground_scene_n_; }
    const int *modelCounts() const { return n_model_; }
    const int *sceneCounts() const { return n_scene_; }

    RegistrationType type;

private:
    struct Cloud {
        void  *p = nullptr;
        size_t cap = 0;
    };
    static void ok(int rc)
    {
        if (rc != SLAM_OK) std::fprintf(stderr, "CCICP: %s\n", slam_last_error());
    }
    static void reserve(Cloud &c, size_t bytes)
    {
        if (bytes <= c.cap) return;
        slam_free(c.p);
        c.p = nullptr;
        c.cap = 0;
        if (slam_malloc(&c.p, bytes) == SLAM_OK) c.cap = bytes;
    }
    void upload(const float *xyz, int n, int stride)
    {
        reserve(raw_, sizeof(float) * (size_t)(n + 1) * stride);
        raw_n_ = n;
        raw_stride_ = stride;
        if (n > 0) ok(slam_memcpy_h2d(raw_.p, xyz, sizeof(float) * (size_t)n * stride, nullptr));
    }
    // the points of raw_ whose label is in `mask`, in cloud order, as (x, y, z, 0) records; 0xff = every point
    void select(unsigned mask, Cloud &dst, int &n_dst)
    {
        reserve(dst, 16 * (size_t)(raw_n_ + 1));
        n_dst = 0;
        if (raw_n_ == 0) return;
        if (mask == 0xffu) {
            reserve(labels_, (size_t)raw_n_ + 16);
            ok(slam_memset(labels_.p, 0, (size_t)raw_n_, nullptr));
            mask = 1u;
        }
        ok(slam_ccicp_select_dev(cc_, (const float *)raw_.p, raw_n_, raw_stride_, (const uint8_t *)labels_.p, mask,
                                 (float *)dst.p, &n_dst, nullptr));
    }
    // segmentGround (:106-119): labels, then outcloud (obstacle + overhead) into obs_ and the ground cloud
    void segment(const float *xyz, int n, int stride, Cloud &ground, int &n_ground)
    {
        upload(xyz, n, stride);
        reserve(labels_, (size_t)n + 16);
        obs_n_ = n_ground = 0;
        if (n == 0) return;
        ok(slam_gseg_segment_dev(gseg_, (const float *)raw_.p, n, stride, (uint8_t *)labels_.p, nullptr));
        select((1u << SLAM_GSEG_OBSTACLE) | (1u << SLAM_GSEG_OVERHEAD), obs_, obs_n_);
        select(1u << SLAM_GSEG_GROUND, ground, n_ground);
    }
    // classifyPoints over obs_ (:36-103) into x,y,z,ground_adj records: through the voxel filter
    // (setSceneCloud) or in classifyPoints' own bin order (setTargetCloud)
    void classify_into(Cloud &dst, int &n_dst, bool voxel)
    {
        reserve(flags_, (size_t)obs_n_ + 16);
        reserve(dst, 16 * (size_t)(obs_n_ + 1));
        n_dst = 0;
        if (obs_n_ == 0) return;
        ok(slam_gseg_classify_ga_dev(gseg_, (const float *)obs_.p, obs_n_, 4, (uint8_t *)flags_.p, nullptr));
        if (voxel)
            ok(slam_ccicp_voxel_downsample_dev(cc_, (const float *)obs_.p, (const uint8_t *)flags_.p, obs_n_, 4, 0.5f, 0.5f,
                                               2.0f, (float *)dst.p, obs_n_, &n_dst, nullptr));
        else
            ok(slam_ccicp_bin_order_dev(cc_, (const float *)obs_.p, (const uint8_t *)flags_.p, obs_n_, 4, (float *)dst.p,
                                        &n_dst, nullptr));
    }
    static void copy_out(const Cloud &c, int n, std::vector<float> &xyz)
    {
        std::vector<float> rec(4 * (size_t)n + 4);
        if (n > 0) ok(slam_memcpy_d2h(rec.data(), c.p, 16 * (size_t)n, nullptr));
        xyz.resize(3 * (size_t)n);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k) xyz[3 * (size_t)i + k] = rec[4 * (size_t)i + k];
    }
    static void download(std::vector<double> &v, const double *d, int n)
    {
        v.resize(2 * (size_t)n + 2);
        if (n > 0) ok(slam_memcpy_d2h(v.data(), d, 16 * (size_t)n, nullptr));
    }

    slam_gseg_t  *gseg_ = nullptr;
    slam_ccicp_t *cc_ = nullptr;
    Cloud         raw_, labels_, obs_, flags_, seg_target_, seg_scene_, ground_target_, ground_scene_, tmp_;
    int           raw_n_ = 0, raw_stride_ = 3, obs_n_ = 0, seg_target_n_ = 0, seg_scene_n_ = 0, ground_target_n_ = 0,
        ground_scene_n_ = 0, tmp_n_ = 0, num_corr_ = 0;
    int     n_model_[2] = {0, 0}, n_scene_[2] = {0, 0};
    double *d_ga_ = nullptr, *d_nga_ = nullptr;
    void   *d_cnt_ = nullptr;
};

} // namespace slam_amd
