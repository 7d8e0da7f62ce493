// ros/scan_registration_node.cpp -- the scan_registration node (scan_registration/src/scan_registration.cpp:57-199)
// over the MI355X library: same topics, same callbacks, same order of calls; slam_amd::CCICP
// (include/slam_amd/ccicp.hpp) stands where the reference's CCICP stands, and clouds cross as float arrays
// instead of pcl::PointCloud.  Builds in a catkin workspace against roscpp, sensor_msgs, geometry_msgs
// (ros/README.md); tests/test_ros_shims.py compiles it against stub message headers.
#ifndef SLAM_SCAN_REG_DEBUG
#define SLAM_SCAN_REG_DEBUG 1 // scan_registration.cpp:37 `#define DEBUG 1`
#endif
#include <cmath>
#include <vector>

#include <geometry_msgs/PoseStamped.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "cloud_msg.hpp"
#include "slam_amd/ccicp.hpp"

namespace {

ros::Publisher posePub, targetPub, scenePub; // scan_registration.cpp:38-40

std::vector<float> input_cloud, target_obs_cloud, target_gnd_cloud; // :43-45
geometry_msgs::PoseStamped poseOut;                                 // :54
slam_amd::CCICP *icp = nullptr;                                     // :57 CCICP icp(SCAN_TO_MAP)

bool obs_flag = false, gnd_flag = false, first_gnd = false, first_obs = false; // :67-71

slam_amd::Pose to_pose(const geometry_msgs::PoseStamped &p)
{
    slam_amd::Pose q;
    q.x = p.pose.position.x, q.y = p.pose.position.y, q.z = p.pose.position.z;
    q.qx = p.pose.orientation.x, q.qy = p.pose.orientation.y, q.qz = p.pose.orientation.z, q.qw = p.pose.orientation.w;
    return q;
}

void pose_cb(const geometry_msgs::PoseStamped &input) { poseOut = input; } // :62-65

void set_targets() // :80-85, :97-102
{
    if (obs_flag && gnd_flag) {
        icp->setTargetCloud(target_obs_cloud.data(), (int)target_obs_cloud.size() / 3, 3, to_pose(poseOut));
        icp->setTargetGndCloud(target_gnd_cloud.data(), (int)target_gnd_cloud.size() / 3, 3);
        obs_flag = gnd_flag = false;
    }
}

void target_obs_cb(const sensor_msgs::PointCloud2ConstPtr &target_input) // :73-89, global frame
{
    if (!slam_amd_ros::cloud_to_xyz(*target_input, target_obs_cloud)) return;
    obs_flag = true;
    set_targets();
    first_obs = true;
}

void target_gnd_cb(const sensor_msgs::PointCloud2ConstPtr &target_input) // :91-104
{
    if (!slam_amd_ros::cloud_to_xyz(*target_input, target_gnd_cloud)) return;
    gnd_flag = true;
    set_targets();
    first_gnd = true;
}

void cloud_cb(const sensor_msgs::PointCloud2ConstPtr &input) // :109-181
{
    if (!(first_gnd && first_obs)) return; // wait for the first ground and obstacle targets (:114-115)
    // the message's own bytes where its layout allows (x, y, z side by side: every driver's), a copy of the three fields otherwise
    const float *cloud = nullptr;
    int          cloud_stride = 3;
    size_t       n = 0;
    if (!slam_amd_ros::cloud_xyz_view(*input, cloud, cloud_stride, n)) {
        if (!slam_amd_ros::cloud_to_xyz(*input, input_cloud)) return; // local frame
        cloud = input_cloud.data(), cloud_stride = 3, n = input_cloud.size() / 3;
    }
    if (target_obs_cloud.empty()) return;
    if (n < 20000) { // :122-125
        ROS_WARN_STREAM("Input Cloud is to small!! Size: " << n);
        return;
    }
    // compensate for roll and pitch (:128-139): the cloud turned by (roll, pitch, 0) and lifted by the pose's z
    double yaw, pitch, roll;
    slam_amd::detail::euler_ypr(to_pose(poseOut), yaw, pitch, roll);
    slam_amd::Pose c;
    slam_amd::detail::quat_from_rpy(roll, pitch, 0.0, c);
    const double d = c.qx * c.qx + c.qy * c.qy + c.qz * c.qz + c.qw * c.qw, s = 2.0 / d;
    const double xs = c.qx * s, ys = c.qy * s, zs = c.qz * s, wx = c.qw * xs, wy = c.qw * ys, wz = c.qw * zs, xx = c.qx * xs,
                 xy = c.qx * ys, xz = c.qx * zs, yy = c.qy * ys, yz = c.qy * zs, zz = c.qz * zs;
    const double r[9] = {1.0 - (yy + zz), xy - wz, xz + wy, xy + wz, 1.0 - (xx + zz), yz - wx, xz - wy, yz + wx, 1.0 - (xx + yy)};
    const double tz = poseOut.pose.position.z;
    // (pcl::transformPointCloud: per coordinate (float)(r0 x + r1 y + r2 z + t) in double -- on the device, behind the upload: the
    // adapter's setSceneCloud(cloud, R, t); a host loop over 131 072 points and a 1.5 MB vector per scan were here)
    const double t3[3] = {0.0, 0.0, tz};
    icp->setSceneCloud(cloud, (int)n, cloud_stride, r, t3); // :139

#if SLAM_SCAN_REG_DEBUG // :141-148 (`#define DEBUG 1`, :37): the segmented scene on mapping/scan_reg/scene.  With the device-resident
    {                     // adapter this is the one step of a scan that brings clouds back to the host: build with =0 where nobody listens
        std::vector<float> target, scene, target_ground, scene_ground;
        icp->getSegmentedClouds(target, scene, target_ground, scene_ground);
        sensor_msgs::PointCloud2 cloud_msg;
        slam_amd_ros::xyz_to_cloud(scene.data(), scene.size() / 3, cloud_msg);
        cloud_msg.header.frame_id = "/local";
        cloud_msg.header.stamp = ros::Time::now();
        scenePub.publish(cloud_msg);
    }
#endif

    // scan registration: the result is in the global frame, given the initial pose and targets in the global frame (:156-159)
    const slam_amd::Pose result = icp->doICPMatch(to_pose(poseOut));
    if (result.qw == 9999) { // :161-165
        ROS_ERROR_STREAM("ICP could not complete registration, skipping this scan");
        return;
    }
    poseOut.pose.position.x = result.x, poseOut.pose.position.y = result.y, poseOut.pose.position.z = result.z;
    poseOut.pose.orientation.x = result.qx, poseOut.pose.orientation.y = result.qy;
    poseOut.pose.orientation.z = result.qz, poseOut.pose.orientation.w = result.qw;
    poseOut.header.stamp = input->header.stamp; // :170-172
    poseOut.header.frame_id = "/global";
    posePub.publish(poseOut);
}

} // namespace

int main(int argc, char **argv) // :183-209
{
    ros::init(argc, argv, "scan_registration");
    ros::NodeHandle nh;
    slam_amd::CCICP matcher(slam_amd::SCAN_TO_MAP);
    icp = &matcher;

    ros::Subscriber poseSub = nh.subscribe("/mapping/ekf/pose", 1, pose_cb);
    ros::Subscriber subScene = nh.subscribe("/velodyne_points", 1, cloud_cb);
    ros::Subscriber subObsTarget = nh.subscribe("/mapping/global/obstacle_pointcloud", 1, target_obs_cb);
    ros::Subscriber subGndTarget = nh.subscribe("/mapping/global/ground_pointcloud", 1, target_gnd_cb);
    posePub = nh.advertise<geometry_msgs::PoseStamped>("mapping/scan_reg/pose", 1);
    targetPub = nh.advertise<sensor_msgs::PointCloud2>("mapping/scan_reg/target", 1);
    scenePub = nh.advertise<sensor_msgs::PointCloud2>("mapping/scan_reg/scene", 1);

    poseOut.pose.orientation.w = 1; // normalized quaternion (:203)
    ros::spin();
    return 0;
}
