// ros1_node.cpp — roscpp adapter: the reference node's main() (pc_preprocessing_main.cpp:509-587)
// with its numeric body replaced by CloudMergerNode. NOT built in this repository's image (no ROS:
// `make ros` needs a sourced ROS 1 workspace); kept so a maintainer of the reference can drop it
// into pcl_preprocessing/src/ — see INTEGRATION.md.
#ifdef CLOUDMERGE_WITH_ROS
#include <ros/ros.h>
#include <sensor_msgs/PointCloud2.h>
#include <tf/transform_listener.h>

#include "merger_node.hpp"

namespace {

cloudmerge::PointCloud2 from_ros(const sensor_msgs::PointCloud2& m) {
    cloudmerge::PointCloud2 c;
    c.header.seq = m.header.seq;
    c.header.stamp_ns = m.header.stamp.toNSec();
    c.header.frame_id = m.header.frame_id;
    c.height = m.height; c.width = m.width;
    for (const auto& f : m.fields) c.fields.push_back({f.name, f.offset, f.datatype, f.count});
    c.is_bigendian = m.is_bigendian;
    c.point_step = m.point_step; c.row_step = m.row_step;
    c.data = m.data;
    c.is_dense = m.is_dense;
    return c;
}

sensor_msgs::PointCloud2 to_ros(const cloudmerge::PointCloud2& c) {
    sensor_msgs::PointCloud2 m;
    m.header.seq = c.header.seq;
    m.header.stamp.fromNSec(c.header.stamp_ns);
    m.header.frame_id = c.header.frame_id;
    m.height = c.height; m.width = c.width;
    for (const auto& f : c.fields) {
        sensor_msgs::PointField pf;
        pf.name = f.name; pf.offset = f.offset; pf.datatype = f.datatype; pf.count = f.count;
        m.fields.push_back(pf);
    }
    m.is_bigendian = c.is_bigendian;
    m.point_step = c.point_step; m.row_step = c.row_step;
    m.data = c.data;
    m.is_dense = c.is_dense;
    return m;
}

}  // namespace

int main(int argc, char** argv) {
    ros::init(argc, argv, "PointcloudPreprocessing");                      // :511
    ros::NodeHandle nh;
    ros::AsyncSpinner spinner(6);                                          // :513
    spinner.start();

    cloudmerge::NodeConfig cfg = cloudmerge::reference_config();
    cloudmerge::CloudMergerNode node(cfg);
    if (!node.ok()) { ROS_FATAL("%s", node.error().c_str()); return 1; }

    ros::Publisher voxelpub = nh.advertise<sensor_msgs::PointCloud2>(cfg.voxel_topic, 1);   // :518
    node.set_publisher([&](const std::string&, const cloudmerge::PointCloud2& msg) { voxelpub.publish(to_ros(msg)); });
    node.set_clock([] { return ros::Time::now().toNSec(); });              // :217

    std::vector<ros::Subscriber> subs;                                     // :520-525, queue size 0
    for (size_t s = 0; s < cfg.sensors.size(); ++s)
        subs.push_back(nh.subscribe<sensor_msgs::PointCloud2>(
            cfg.sensors[s].topic, 0,
            [&node, s](const sensor_msgs::PointCloud2::ConstPtr& m) { node.on_cloud(s, from_ros(*m)); }));

    ros::Rate loop_rate(cfg.rate_hz);                                      // :527
    tf::TransformListener listener;
    while (ros::ok()) {                                                    // :549
        if (!node.transforms_ready()) {                                    // :551-568
            try {
                for (size_t s = 0; s < cfg.sensors.size(); ++s) {
                    tf::StampedTransform stf;
                    listener.lookupTransform("/" + cfg.base_frame, cfg.sensors[s].frame, ros::Time(0), stf);
                    const tf::Transform t(stf.getRotation(), stf.getOrigin());              // :320
                    const tf::Quaternion q = t.getRotation();                               // what pcl_ros reads
                    const double qq[4] = {q.x(), q.y(), q.z(), q.w()};
                    const double tt[3] = {t.getOrigin().x(), t.getOrigin().y(), t.getOrigin().z()};
                    node.set_transform(s, qq, tt);
                }
            } catch (tf::TransformException& ex) {
                ROS_ERROR("%s", ex.what());
            }
        }
        node.spin_once();                                                  // :574-580
        loop_rate.sleep();                                                 // :583
    }
    ros::waitForShutdown();
    return 0;
}
#endif  // CLOUDMERGE_WITH_ROS
