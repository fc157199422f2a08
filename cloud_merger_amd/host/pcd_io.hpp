// pcd_io.hpp — PCD v0.7 reader/writer for the replay driver (the reference replays rosbags it does
// not ship: my_cloud_fusion/launch/bag.launch:7; recorded frames here are .pcd files).
// Reads DATA ascii and DATA binary (not binary_compressed) with fields of any PCD type, size and count ('_' padding
// and e.g. a Velodyne 'ring' are skipped over; x, y, z, intensity must be FLOAT32 to be fused); the header's point
// count is checked against the file's size before anything is allocated.
#pragma once
#include <string>

#include "pointcloud2.hpp"

namespace cloudmerge {

// Returns false and fills *err on failure.
bool read_pcd(const std::string& path, PointCloud2* out, std::string* err);
// Writes every field of msg (all must be FLOAT32 count 1) as DATA binary.
bool write_pcd(const std::string& path, const PointCloud2& msg, std::string* err);

}  // namespace cloudmerge
