// ros_stub.hpp — declarations (no definitions) of the handful of roscpp / tf / sensor_msgs names that
// cloud_merger_amd/host/ros1_node.cpp and INTEGRATION.md §A use, so that those two pieces of OUR code can be
// syntax-checked against include/cloudmerge.h in an image without ROS (tests/test_integration_snippet.py,
// `-fsyntax-only`: nothing here is ever linked or run, and nothing of the reference is built with it).
#pragma once
#include <cstdint>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#define ROS_FATAL(...) ((void)0)
#define ROS_ERROR(...) ((void)0)

namespace ros {
struct Time {
    Time() {}
    explicit Time(double) {}
    static Time now();
    uint64_t toNSec() const;
    Time& fromNSec(uint64_t);
};
struct Rate { explicit Rate(double); bool sleep(); };
struct AsyncSpinner { explicit AsyncSpinner(uint32_t); void start(); };
struct Publisher { template <class M> void publish(const M&) const; };
struct Subscriber {};
struct NodeHandle {
    template <class M> Publisher advertise(const std::string&, uint32_t);
    template <class M> Subscriber subscribe(const std::string&, uint32_t, std::function<void(const std::shared_ptr<const M>&)>);
};
void init(int&, char**, const std::string&);
bool ok();
void waitForShutdown();
}  // namespace ros

namespace std_msgs { struct Header { uint32_t seq; ros::Time stamp; std::string frame_id; }; }

namespace sensor_msgs {
struct PointField {
    enum { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name; uint32_t offset; uint8_t datatype; uint32_t count;
};
struct PointCloud2 {
    typedef std::shared_ptr<const PointCloud2> ConstPtr;
    std_msgs::Header header;
    uint32_t height, width;
    std::vector<PointField> fields;
    bool is_bigendian;
    uint32_t point_step, row_step;
    std::vector<uint8_t> data;
    bool is_dense;
};
}  // namespace sensor_msgs

namespace tf {
struct Vector3 { double x() const; double y() const; double z() const; };
struct Quaternion { double x() const; double y() const; double z() const; double w() const; };
struct Transform {
    Transform() {}
    Transform(const Quaternion&, const Vector3&);
    Quaternion getRotation() const;
    const Vector3& getOrigin() const;
};
struct StampedTransform : Transform {};
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
struct TransformListener {
    void lookupTransform(const std::string&, const std::string&, const ros::Time&, StampedTransform&) const;
};
}  // namespace tf
