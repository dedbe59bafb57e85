// nautilus_hip_io.h -- the text formats either side of the path, for a C++ host that
// links the C-ABI library without ROS: header only, standard library only.
//
//   WritePoses / ReadPoses / LoadSolution   the pose file of Solver::WriteCallback (src/optimization/solver.cc:565-579:
//                                            std::fixed `timestamp x y theta`, one node per line) and its reader
//                                            LoadSolutionFromFile (src/main.cc:131-157: double timestamp + three floats
//                                            into a map keyed by timestamp; a node is looked up by its timestamp
//                                            printed with std::fixed and parsed back).
//   WriteMapLines / ReadMapLines             the vectorised map of Solver::Vectorize (solver.cc:608-618): `x1,y1,x2,y2`
//                                            per line segment, floats at the stream's default precision.
//   HitlSlamInput / LineSegmentsFromHitl     msg/HitlSlamInputMsg.msg (four geometry_msgs/Point32) and
//                                            LineSegmentsFromHitlMsg (solver.cc:467-478): z is dropped, two segments.
//
// The Python mirror of the same formats is nautilus_amd/hostside.py; tests/test_io_cpp.py passes files between the two.
#pragma once

#include <array>
#include <cstdint>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace nautilus_hip {
namespace io {

struct NodePose {
  double timestamp;
  double pose[3];  // x, y, theta: SLAMNodeSolution2D::pose (slam_types.h)
};

struct LineSegment {  // VectorMaps::LineSegment / LineSegment<float>: two Vector2f end points
  float x0, y0, x1, y1;
};

struct Point32 {
  float x, y, z;
};

struct HitlSlamInput {  // msg/HitlSlamInputMsg.msg
  Point32 line_a_start, line_a_end, line_b_start, line_b_end;
};

// solver.cc:571-578.  false when the file cannot be opened (the reference does not check).
inline bool WritePoses(const std::string &path, const std::vector<NodePose> &nodes) {
  std::ofstream out(path);
  if (!out.is_open()) return false;
  for (const NodePose &n : nodes)
    out << std::fixed << n.timestamp << " " << n.pose[0] << " " << n.pose[1] << " " << n.pose[2] << std::endl;
  return static_cast<bool>(out);
}

// main.cc:133-146: reading stops at the first line that does not parse; a repeated timestamp keeps the last pose.
inline std::map<double, std::array<float, 3>> ReadPoses(const std::string &path) {
  std::map<double, std::array<float, 3>> poses;
  std::ifstream in(path);
  if (in.is_open()) {
    double timestamp;
    float x, y, theta;
    while (in >> timestamp >> x >> y >> theta) poses[timestamp] = {x, y, theta};
  }
  return poses;
}

// main.cc:147-156: overwrite the pose of every node whose timestamp (as std::fixed prints it) is in the file; returns
// the indices of the nodes that were not found (the reference prints a line for each).
inline std::vector<size_t> LoadSolution(const std::string &path, std::vector<NodePose> *nodes) {
  const std::map<double, std::array<float, 3>> poses = ReadPoses(path);
  std::vector<size_t> missing;
  for (size_t i = 0; i < nodes->size(); i++) {
    std::stringstream ss;
    ss << std::fixed << (*nodes)[i].timestamp;
    const auto it = poses.find(std::stod(ss.str()));
    if (it == poses.end()) {
      missing.push_back(i);
      continue;
    }
    for (int k = 0; k < 3; k++) (*nodes)[i].pose[k] = it->second[k];
  }
  return missing;
}

// solver.cc:608-618
inline bool WriteMapLines(const std::string &path, const std::vector<LineSegment> &lines) {
  std::ofstream out(path);
  if (!out.is_open()) return false;
  for (const LineSegment &l : lines) out << l.x0 << "," << l.y0 << "," << l.x1 << "," << l.y1 << std::endl;
  return static_cast<bool>(out);
}

inline std::vector<LineSegment> ReadMapLines(const std::string &path) {
  std::vector<LineSegment> lines;
  std::ifstream in(path);
  std::string row;
  while (std::getline(in, row)) {
    if (row.find_first_not_of(" \t\r") == std::string::npos) continue;
    std::stringstream ss(row);
    LineSegment l;
    char c0, c1, c2;
    if (!(ss >> l.x0 >> c0 >> l.y0 >> c1 >> l.x1 >> c2 >> l.y1) || c0 != ',' || c1 != ',' || c2 != ',') break;
    lines.push_back(l);
  }
  return lines;
}

// solver.cc:467-478: (line a, line b)
inline std::array<LineSegment, 2> LineSegmentsFromHitl(const HitlSlamInput &msg) {
  return {LineSegment{msg.line_a_start.x, msg.line_a_start.y, msg.line_a_end.x, msg.line_a_end.y},
          LineSegment{msg.line_b_start.x, msg.line_b_start.y, msg.line_b_end.x, msg.line_b_end.y}};
}

}  // namespace io
}  // namespace nautilus_hip
