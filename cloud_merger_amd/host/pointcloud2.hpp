// pointcloud2.hpp — a dependency-free image of sensor_msgs/PointCloud2, the wire type on both
// sides of the path (subscribers: pc_preprocessing_main.cpp:520-525 receive it through the pcl_ros
// serializer; publisher: pcl::toROSMsg at :215-219). ROS itself is not installable here; the roscpp
// adapter (ros1_node.cpp, compiled only with CLOUDMERGE_WITH_ROS) converts to and from the real
// message by field-for-field assignment.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace cloudmerge {

struct PointField {
    enum : uint8_t { INT8 = 1, UINT8 = 2, INT16 = 3, UINT16 = 4, INT32 = 5, UINT32 = 6, FLOAT32 = 7, FLOAT64 = 8 };
    std::string name;
    uint32_t offset = 0;
    uint8_t datatype = FLOAT32;
    uint32_t count = 1;
};

struct Header {
    uint32_t seq = 0;
    uint64_t stamp_ns = 0;
    std::string frame_id;
};

struct PointCloud2 {
    Header header;
    uint32_t height = 1;
    uint32_t width = 0;
    std::vector<PointField> fields;
    bool is_bigendian = false;
    uint32_t point_step = 0;
    uint32_t row_step = 0;
    std::vector<uint8_t> data;
    bool is_dense = true;

    size_t num_points() const { return static_cast<size_t>(width) * height; }
};

// Byte offsets of the FLOAT32 fields the path consumes. off_i == 0xFFFFFFFF: no intensity field.
struct XyziLayout {
    uint32_t off_x = 0, off_y = 0, off_z = 0, off_i = 0xFFFFFFFFu;
    bool ok = false;
    std::string error;
};

inline XyziLayout find_xyzi(const PointCloud2& msg) {
    XyziLayout l;
    bool hx = false, hy = false, hz = false;
    for (const auto& f : msg.fields) {
        const bool f32 = f.datatype == PointField::FLOAT32 && f.count >= 1;
        if (f.name == "x") { if (!f32) { l.error = "x is not FLOAT32"; return l; } l.off_x = f.offset; hx = true; }
        else if (f.name == "y") { if (!f32) { l.error = "y is not FLOAT32"; return l; } l.off_y = f.offset; hy = true; }
        else if (f.name == "z") { if (!f32) { l.error = "z is not FLOAT32"; return l; } l.off_z = f.offset; hz = true; }
        else if (f.name == "intensity" && f32) l.off_i = f.offset;
    }
    if (!(hx && hy && hz)) { l.error = "missing x/y/z field"; return l; }
    if (msg.is_bigendian) { l.error = "big-endian clouds are not supported"; return l; }
    if (msg.data.size() < msg.num_points() * msg.point_step) { l.error = "data shorter than width*height*point_step"; return l; }
    l.ok = true;
    return l;
}

// The message pcl::toROSMsg builds from a pcl::PointCloud<pcl::PointXYZI>:
// x@0 y@4 z@8 intensity@16 FLOAT32, point_step 32, height 1, little-endian, dense.
inline PointCloud2 make_pcl_xyzi_message(size_t n_points) {
    PointCloud2 m;
    m.height = 1;
    m.width = static_cast<uint32_t>(n_points);
    m.fields = {{"x", 0, PointField::FLOAT32, 1}, {"y", 4, PointField::FLOAT32, 1},
                {"z", 8, PointField::FLOAT32, 1}, {"intensity", 16, PointField::FLOAT32, 1}};
    m.point_step = 32;
    m.row_step = 32 * m.width;
    m.data.resize(n_points * 32);
    m.is_dense = true;
    return m;
}

// Compact variant: x,y,z,intensity @0,4,8,12, point_step 16.
inline PointCloud2 make_xyzi16_message(size_t n_points) {
    PointCloud2 m;
    m.height = 1;
    m.width = static_cast<uint32_t>(n_points);
    m.fields = {{"x", 0, PointField::FLOAT32, 1}, {"y", 4, PointField::FLOAT32, 1},
                {"z", 8, PointField::FLOAT32, 1}, {"intensity", 12, PointField::FLOAT32, 1}};
    m.point_step = 16;
    m.row_step = 16 * m.width;
    m.data.resize(n_points * 16);
    m.is_dense = true;
    return m;
}

}  // namespace cloudmerge
