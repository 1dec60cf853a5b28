// lsa_io.hpp — header-only input formats for the C++ harness: KITTI velodyne .bin, PCD v0.7
// (DATA ascii / binary, as pcl::io::savePCDFileBinary / ASCII of a PointXYZI cloud write them; read back in the
// reference with pcl::io::loadPCDFile, matching.cpp:155, loop_closing.cpp:134,286,304) and removeNaNFromPointCloud
// (front_end.cpp:92, matching.cpp:188).  Clouds are the lsa::Cloud of lsa_plugins.hpp (32-byte PointXYZI records).
#ifndef LSA_IO_HPP_
#define LSA_IO_HPP_

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "lsa_plugins.hpp"

namespace lsa {

// KITTI raw / odometry scan: flat little-endian float32 records x, y, z, reflectance
inline bool ReadKittiBin(const std::string &path, Cloud &out) {
  std::ifstream f(path, std::ios::binary | std::ios::ate);
  if (!f) return false;
  std::streamsize bytes = f.tellg();
  if (bytes < 0 || bytes % 16) return false;
  f.seekg(0);
  std::vector<float> buf((size_t)bytes / 4);
  if (bytes && !f.read(reinterpret_cast<char *>(buf.data()), bytes)) return false;
  out.points.resize(buf.size() / 4);
  for (size_t i = 0; i < out.points.size(); i++) {
    PointXYZI &p = out.points[i];
    p = PointXYZI();
    p.x = buf[4 * i]; p.y = buf[4 * i + 1]; p.z = buf[4 * i + 2]; p.intensity = buf[4 * i + 3];
  }
  return true;
}

// pcl::removeNaNFromPointCloud(in, out, indices): order kept; in and out may be the same cloud
inline void RemoveNaN(const Cloud &in, Cloud &out, std::vector<int> &indices) {
  indices.clear();
  size_t j = 0;
  if (&in != &out) out.points.resize(in.points.size());
  for (size_t i = 0; i < in.points.size(); i++) {
    const PointXYZI p = in.points[i];
    if (!std::isfinite(p.x) || !std::isfinite(p.y) || !std::isfinite(p.z)) continue;
    out.points[j++] = p;
    indices.push_back((int)i);
  }
  out.points.resize(j);
}

namespace detail {
inline double pcd_scalar(const char *p, char type, int size) {
  switch (type) {
    case 'F': if (size == 4) { float v; std::memcpy(&v, p, 4); return v; } else { double v; std::memcpy(&v, p, 8); return v; }
    case 'U': { uint64_t v = 0; std::memcpy(&v, p, (size_t)size); return (double)v; }
    default: {
      if (size == 1) { int8_t v; std::memcpy(&v, p, 1); return v; }
      if (size == 2) { int16_t v; std::memcpy(&v, p, 2); return v; }
      if (size == 4) { int32_t v; std::memcpy(&v, p, 4); return v; }
      int64_t v; std::memcpy(&v, p, 8); return (double)v;
    }
  }
}
}  // namespace detail

// PCD v0.7, DATA ascii | binary; fields other than x y z intensity are ignored; binary_compressed is rejected
inline bool ReadPcd(const std::string &path, Cloud &out) {
  std::ifstream f(path, std::ios::binary);
  if (!f) return false;
  std::vector<std::string> fields, types;
  std::vector<int> sizes, counts;
  size_t npts = 0, width = 0, height = 1;
  bool have_points = false;
  std::string data_kind, line;
  while (std::getline(f, line)) {
    if (!line.empty() && line.back() == '\r') line.pop_back();
    if (line.empty() || line[0] == '#') continue;
    std::istringstream ls(line);
    std::string key, tok;
    ls >> key;
    if (key == "FIELDS") while (ls >> tok) fields.push_back(tok);
    else if (key == "SIZE") while (ls >> tok) sizes.push_back(std::stoi(tok));
    else if (key == "TYPE") while (ls >> tok) types.push_back(tok);
    else if (key == "COUNT") while (ls >> tok) counts.push_back(std::stoi(tok));
    else if (key == "WIDTH") ls >> width;
    else if (key == "HEIGHT") ls >> height;
    else if (key == "POINTS") { ls >> npts; have_points = true; }
    else if (key == "DATA") { ls >> data_kind; break; }
  }
  if (fields.empty() || sizes.size() != fields.size() || types.size() != fields.size()) return false;
  if (counts.empty()) counts.assign(fields.size(), 1);
  if (!have_points) npts = width * height;
  int col_of[4] = {-1, -1, -1, -1};
  const char *want[4] = {"x", "y", "z", "intensity"};
  for (size_t k = 0; k < fields.size(); k++)
    for (int w = 0; w < 4; w++)
      if (fields[k] == want[w]) col_of[w] = (int)k;
  out.points.assign(npts, PointXYZI());
  if (data_kind == "ascii") {
    for (size_t i = 0; i < npts; i++) {
      for (size_t k = 0; k < fields.size(); k++)
        for (int c = 0; c < counts[k]; c++) {
          std::string tok;
          if (!(f >> tok)) return false;
          double v = (tok == "nan" || tok == "NaN") ? NAN : std::strtod(tok.c_str(), nullptr);
          if (c) continue;
          if ((int)k == col_of[0]) out.points[i].x = (float)v;
          else if ((int)k == col_of[1]) out.points[i].y = (float)v;
          else if ((int)k == col_of[2]) out.points[i].z = (float)v;
          else if ((int)k == col_of[3]) out.points[i].intensity = (float)v;
        }
    }
    return true;
  }
  if (data_kind != "binary") return false;
  size_t rec = 0;
  std::vector<size_t> off(fields.size());
  for (size_t k = 0; k < fields.size(); k++) { off[k] = rec; rec += (size_t)sizes[k] * (size_t)counts[k]; }
  std::vector<char> buf(rec * npts);
  if (!buf.empty() && !f.read(buf.data(), (std::streamsize)buf.size())) return false;
  for (size_t i = 0; i < npts; i++) {
    const char *r = buf.data() + i * rec;
    PointXYZI &p = out.points[i];
    if (col_of[0] >= 0) p.x = (float)detail::pcd_scalar(r + off[col_of[0]], types[col_of[0]][0], sizes[col_of[0]]);
    if (col_of[1] >= 0) p.y = (float)detail::pcd_scalar(r + off[col_of[1]], types[col_of[1]][0], sizes[col_of[1]]);
    if (col_of[2] >= 0) p.z = (float)detail::pcd_scalar(r + off[col_of[2]], types[col_of[2]][0], sizes[col_of[2]]);
    if (col_of[3] >= 0) p.intensity = (float)detail::pcd_scalar(r + off[col_of[3]], types[col_of[3]][0], sizes[col_of[3]]);
  }
  return true;
}

}  // namespace lsa
#endif
