// pcd_io.cpp — PCD v0.7 reader/writer (see pcd_io.hpp).
#include "pcd_io.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace cloudmerge {

namespace {
std::vector<std::string> split(const std::string& line) {
    std::istringstream is(line);
    std::vector<std::string> t;
    for (std::string w; is >> w;) t.push_back(w);
    return t;
}
}  // namespace

// PCD TYPE/SIZE -> sensor_msgs/PointField datatype; 0: not a combination PCD defines.
static uint8_t pcd_datatype(const std::string& type, unsigned size) {
    if (type == "F") return size == 4 ? PointField::FLOAT32 : size == 8 ? PointField::FLOAT64 : 0;
    if (type == "U") return size == 1 ? PointField::UINT8 : size == 2 ? PointField::UINT16 : size == 4 ? PointField::UINT32 : 0;
    if (type == "I") return size == 1 ? PointField::INT8 : size == 2 ? PointField::INT16 : size == 4 ? PointField::INT32 : 0;
    return 0;
}

bool read_pcd(const std::string& path, PointCloud2* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    f.seekg(0, std::ios::end);
    const unsigned long long file_size = static_cast<unsigned long long>(f.tellg());
    f.seekg(0, std::ios::beg);
    std::vector<std::string> names, sizes, types, counts;
    unsigned long long width = 0, height = 1, points = 0;
    bool have_points = false;
    std::string data_kind;
    for (std::string line; std::getline(f, line);) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        if (line.empty() || line[0] == '#') continue;
        auto t = split(line);
        if (t.empty()) continue;
        const std::string key = t[0];
        t.erase(t.begin());
        if (key == "FIELDS") names = t;
        else if (key == "SIZE") sizes = t;
        else if (key == "TYPE") types = t;
        else if (key == "COUNT") counts = t;
        else if (key == "WIDTH" && !t.empty()) width = std::strtoull(t[0].c_str(), nullptr, 10);
        else if (key == "HEIGHT" && !t.empty()) height = std::strtoull(t[0].c_str(), nullptr, 10);
        else if (key == "POINTS" && !t.empty()) { points = std::strtoull(t[0].c_str(), nullptr, 10); have_points = true; }
        else if (key == "DATA" && !t.empty()) { data_kind = t[0]; break; }
    }
    if (names.empty() || sizes.size() != names.size() || types.size() != names.size() ||
        (!counts.empty() && counts.size() != names.size())) {
        if (err) *err = "malformed PCD header in " + path;
        return false;
    }
    if (!have_points) {
        if (height && width > 0xFFFFFFFFull / height) { if (err) *err = "WIDTH x HEIGHT too large in " + path; return false; }
        points = width * height;
    }
    if (points > 0xFFFFFFFFull) { if (err) *err = "more than 2^32-1 points in " + path; return false; }
    // Every field takes SIZE x COUNT bytes of a point, whatever its type: clouds written from a PointCloud2 of
    // pcl::PointXYZI carry '_' padding fields (SIZE 1 TYPE U COUNT 4 / 12), Velodyne clouds a 'ring' (U2). All of them
    // are kept in the layout; only x, y, z (and intensity) have to be FLOAT32, which find_xyzi checks.
    PointCloud2 m;
    unsigned long long off = 0;
    for (size_t k = 0; k < names.size(); ++k) {
        char* end = nullptr;
        const unsigned long long size = std::strtoull(sizes[k].c_str(), &end, 10);
        const bool size_ok = end && *end == 0 && (size == 1 || size == 2 || size == 4 || size == 8);
        unsigned long long cnt = 1;
        bool cnt_ok = true;
        if (!counts.empty()) { cnt = std::strtoull(counts[k].c_str(), &end, 10); cnt_ok = end && *end == 0 && counts[k][0] != '-' && cnt >= 1 && cnt <= (1u << 20); }
        const uint8_t dt = size_ok ? pcd_datatype(types[k], static_cast<unsigned>(size)) : 0;
        if (!size_ok || !cnt_ok || !dt) { if (err) *err = "bad SIZE/TYPE/COUNT of field " + names[k] + " in " + path; return false; }
        if (names[k] != "_") m.fields.push_back({names[k], static_cast<uint32_t>(off), dt, static_cast<uint32_t>(cnt)});
        off += size * cnt;
        if (off > (1u << 24)) { if (err) *err = "point_step too large in " + path; return false; }
    }
    m.point_step = static_cast<uint32_t>(off);
    const unsigned long long header_end = static_cast<unsigned long long>(f.tellg());
    const unsigned long long left = file_size > header_end ? file_size - header_end : 0;
    const unsigned long long bytes = points * off;
    // the header is not trusted with the allocation: the data must be in the file (ascii: at least "0 " per value)
    unsigned long long values_per_point = 0;
    for (size_t k = 0; k < names.size(); ++k) values_per_point += counts.empty() ? 1 : std::strtoull(counts[k].c_str(), nullptr, 10);
    if ((data_kind == "binary" && bytes > left) || (data_kind == "ascii" && points * values_per_point * 2 > left + 1)) {
        if (err) *err = "header announces more points than the file holds: " + path;
        return false;
    }
    m.height = 1;
    m.width = static_cast<uint32_t>(points);
    m.row_step = m.point_step * m.width;
    m.is_dense = false;                       // PCD files may hold NaN points
    if (data_kind == "binary") {
        m.data.resize(bytes);
        f.read(reinterpret_cast<char*>(m.data.data()), static_cast<std::streamsize>(m.data.size()));
        if (static_cast<size_t>(f.gcount()) != m.data.size()) { if (err) *err = "short read in " + path; return false; }
    } else if (data_kind == "ascii") {
        m.data.assign(bytes, 0);
        for (unsigned long long i = 0; i < points; ++i) {
            uint8_t* row = m.data.data() + i * off;
            unsigned long long o = 0;
            for (size_t k = 0; k < names.size(); ++k) {
                const unsigned size = static_cast<unsigned>(std::strtoull(sizes[k].c_str(), nullptr, 10));
                const unsigned long long cnt = counts.empty() ? 1 : std::strtoull(counts[k].c_str(), nullptr, 10);
                for (unsigned long long q = 0; q < cnt; ++q, o += size) {
                    std::string tok;
                    if (!(f >> tok)) { if (err) *err = "short ascii data in " + path; return false; }
                    if (types[k] == "F") {
                        if (size == 4) { const float v = std::strtof(tok.c_str(), nullptr); std::memcpy(row + o, &v, 4); }
                        else { const double v = std::strtod(tok.c_str(), nullptr); std::memcpy(row + o, &v, 8); }
                    } else if (types[k] == "U") {
                        const unsigned long long v = std::strtoull(tok.c_str(), nullptr, 10); std::memcpy(row + o, &v, size);   // little-endian host
                    } else {
                        const long long v = std::strtoll(tok.c_str(), nullptr, 10); std::memcpy(row + o, &v, size);
                    }
                }
            }
        }
    } else {
        if (err) *err = "unsupported DATA " + data_kind + " (ascii and binary are read; binary_compressed is not)";
        return false;
    }
    *out = std::move(m);
    return true;
}

bool write_pcd(const std::string& path, const PointCloud2& msg, std::string* err) {
    for (const auto& fl : msg.fields)
        if (fl.datatype != PointField::FLOAT32 || fl.count != 1) { if (err) *err = "only FLOAT32 count-1 fields"; return false; }
    std::ofstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::ostringstream h;
    h << "# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS";
    for (const auto& fl : msg.fields) h << ' ' << fl.name;
    h << "\nSIZE";
    for (size_t k = 0; k < msg.fields.size(); ++k) h << " 4";
    h << "\nTYPE";
    for (size_t k = 0; k < msg.fields.size(); ++k) h << " F";
    h << "\nCOUNT";
    for (size_t k = 0; k < msg.fields.size(); ++k) h << " 1";
    const size_t n = msg.num_points();
    h << "\nWIDTH " << n << "\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS " << n << "\nDATA binary\n";
    f << h.str();
    // pack the declared fields tightly (drops padding such as PointXYZI's)
    std::vector<char> row(msg.fields.size() * 4);
    for (size_t i = 0; i < n; ++i) {
        const uint8_t* src = msg.data.data() + i * msg.point_step;
        for (size_t k = 0; k < msg.fields.size(); ++k) std::memcpy(row.data() + 4 * k, src + msg.fields[k].offset, 4);
        f.write(row.data(), static_cast<std::streamsize>(row.size()));
    }
    return static_cast<bool>(f);
}

}  // namespace cloudmerge
