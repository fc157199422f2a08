// pcd_io.cpp — PCD v0.7 reader/writer (see pcd_io.hpp).
#include "pcd_io.hpp"

#include <cstdio>
#include <cstdlib>
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

bool read_pcd(const std::string& path, PointCloud2* out, std::string* err) {
    std::ifstream f(path, std::ios::binary);
    if (!f) { if (err) *err = "cannot open " + path; return false; }
    std::vector<std::string> names, sizes, types, counts;
    size_t width = 0, height = 1, points = 0;
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
        else if (key == "POINTS" && !t.empty()) points = std::strtoull(t[0].c_str(), nullptr, 10);
        else if (key == "DATA" && !t.empty()) { data_kind = t[0]; break; }
    }
    if (names.empty() || sizes.size() != names.size() || types.size() != names.size()) {
        if (err) *err = "malformed PCD header in " + path;
        return false;
    }
    if (points == 0) points = width * height;
    PointCloud2 m;
    uint32_t off = 0;
    for (size_t k = 0; k < names.size(); ++k) {
        const uint32_t cnt = counts.size() == names.size() ? static_cast<uint32_t>(std::atoi(counts[k].c_str())) : 1u;
        if (sizes[k] != "4" || types[k] != "F") {
            if (err) *err = "only FLOAT32 fields are supported (field " + names[k] + ")";
            return false;
        }
        m.fields.push_back({names[k], off, PointField::FLOAT32, cnt});
        off += 4 * cnt;
    }
    m.point_step = off;
    m.height = 1;
    m.width = static_cast<uint32_t>(points);
    m.row_step = m.point_step * m.width;
    m.data.resize(points * m.point_step);
    m.is_dense = false;                       // PCD files may hold NaN points
    if (data_kind == "binary") {
        f.read(reinterpret_cast<char*>(m.data.data()), static_cast<std::streamsize>(m.data.size()));
        if (static_cast<size_t>(f.gcount()) != m.data.size()) { if (err) *err = "short read in " + path; return false; }
    } else if (data_kind == "ascii") {
        const size_t per = m.point_step / 4;
        float* dst = reinterpret_cast<float*>(m.data.data());
        for (size_t i = 0; i < points * per; ++i) {
            std::string tok;
            if (!(f >> tok)) { if (err) *err = "short ascii data in " + path; return false; }
            dst[i] = std::strtof(tok.c_str(), nullptr);
        }
    } else {
        if (err) *err = "unsupported DATA " + data_kind;
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
