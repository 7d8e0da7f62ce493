// ros/local_mapper_node.cpp -- the local_mapper node (local_mapper/src/local_mapper.cpp:29-130) over the MI355X
// library: same topics, same 50 Hz poll, same calls on the map object; slam_amd::MLS (include/slam_amd/mls.hpp)
// stands where the reference's MLS stands.  The MLS marker visualisation (:124, visualization_msgs) is not
// reproduced: the occupancy mode keeps no height clusters to draw.
#include <vector>

#include <geometry_msgs/PoseStamped.h>
#include <nav_msgs/OccupancyGrid.h>
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>

#include "cloud_msg.hpp"
#include "slam_amd/mls.hpp"

namespace {

slam_amd::MLS *local_map = nullptr; // :29 MLS local_map(200, 200, 0.2, true)
geometry_msgs::PoseStamped curPose; // :31
ros::Publisher localCloudPub, drivablilityPub;
std::vector<float> input_cloud;
ros::Time pose_time, cloud_time;
bool newCloud = false;

slam_amd::Pose to_pose(const geometry_msgs::PoseStamped &p)
{
    slam_amd::Pose q;
    q.x = p.pose.position.x, q.y = p.pose.position.y, q.z = p.pose.position.z;
    q.qx = p.pose.orientation.x, q.qy = p.pose.orientation.y, q.qz = p.pose.orientation.z, q.qw = p.pose.orientation.w;
    return q;
}

void pose_cb(const geometry_msgs::PoseStamped &pose) // :42-46
{
    curPose = pose;
    pose_time = pose.header.stamp;
}

void poseOffet_cb(const geometry_msgs::PoseStamped &pose) { local_map->offsetMap(to_pose(pose)); } // :48-51

void cloud_cb(const sensor_msgs::PointCloud2ConstPtr &input) // :53-63
{
    if (!slam_amd_ros::cloud_to_xyz(*input, input_cloud)) return;
    cloud_time = input->header.stamp;
    newCloud = true;
}

} // namespace

int main(int argc, char **argv) // :65-130
{
    ros::init(argc, argv, "local_mapper");
    ros::NodeHandle nh;
    slam_amd::MLS map(200, 200, 0.2, true);
    local_map = &map;

    ros::Subscriber poseSub = nh.subscribe("/mapping/ekf/pose", 1, pose_cb);
    ros::Subscriber poseOffsetSub = nh.subscribe("/mapping/graph_slam/pose_offset", 1, poseOffet_cb);
    ros::Subscriber sub = nh.subscribe("/velodyne_points", 1, cloud_cb);
    localCloudPub = nh.advertise<sensor_msgs::PointCloud2>("/mapping/local_poiuntcloud", 1, true); // (sic, :78)
    drivablilityPub = nh.advertise<nav_msgs::OccupancyGrid>("/mapping/local_drivability", 1, true);

    local_map->setMinClusterPoints(20); // :86
    local_map->clearMap();              // :93

    ros::Rate loop_rate(50); // :95
    while (ros::ok()) {
        loop_rate.sleep();
        ros::spinOnce();
        if (newCloud && (pose_time - cloud_time).toSec() >= 0) { // :102
            newCloud = false;
            local_map->addToMap(input_cloud.data(), (int)input_cloud.size() / 3, 3, to_pose(curPose)); // :107

            sensor_msgs::PointCloud2 cloud_msg; // :110-117
            local_map->filterPointCloud(0.1, 0.1);
            const std::vector<float> &local_cloud = local_map->getGlobalCloud();
            slam_amd_ros::xyz_to_cloud(local_cloud.data(), local_cloud.size() / 3, cloud_msg);
            cloud_msg.header.frame_id = "/local_oriented";
            cloud_msg.header.stamp = ros::Time::now();
            localCloudPub.publish(cloud_msg);

            const slam_amd::OccupancyGrid &g = local_map->getDrivability(); // :119-122
            nav_msgs::OccupancyGrid drivability;
            drivability.header.frame_id = "/local_oriented";
            drivability.info.resolution = (float)g.info.resolution;
            drivability.info.width = g.info.width;
            drivability.info.height = g.info.height;
            drivability.info.origin.position.x = g.info.origin_x;
            drivability.info.origin.position.y = g.info.origin_y;
            drivability.data.assign(g.data.begin(), g.data.end());
            drivablilityPub.publish(drivability);
        }
    }
    return 0;
}
