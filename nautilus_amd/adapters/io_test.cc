// io_test.cc -- drives nautilus_hip_io.h on files the Python mirror (nautilus_amd/hostside.py) wrote and writes them
// back, for tests/test_io_cpp.py.  Host only: no GPU, no C-ABI library.
//   io_test <dir>: reads  nodes.txt (timestamp per line, 17 significant digits), poses_py.txt, map_py.txt, hitl.txt
//                  writes poses_cpp.txt (LoadSolution into zero poses, then WritePoses), map_cpp.txt, report.txt
#include <cstdio>
#include <iostream>

#include "nautilus_hip_io.h"

int main(int argc, char **argv) {
  if (argc != 2) return 2;
  namespace io = nautilus_hip::io;
  const std::string dir = std::string(argv[1]) + "/";
  std::vector<io::NodePose> nodes;
  {
    std::ifstream in(dir + "nodes.txt");
    double t;
    while (in >> t) nodes.push_back({t, {0.0, 0.0, 0.0}});
  }
  const std::vector<size_t> missing = io::LoadSolution(dir + "poses_py.txt", &nodes);
  if (!io::WritePoses(dir + "poses_cpp.txt", nodes)) return 3;
  const std::vector<io::LineSegment> lines = io::ReadMapLines(dir + "map_py.txt");
  if (!io::WriteMapLines(dir + "map_cpp.txt", lines)) return 3;
  io::HitlSlamInput msg;
  {
    std::ifstream in(dir + "hitl.txt");
    io::Point32 *p[4] = {&msg.line_a_start, &msg.line_a_end, &msg.line_b_start, &msg.line_b_end};
    for (int i = 0; i < 4; i++)
      if (!(in >> p[i]->x >> p[i]->y >> p[i]->z)) return 4;
  }
  const auto seg = io::LineSegmentsFromHitl(msg);
  std::ofstream rep(dir + "report.txt");
  rep << "nodes " << nodes.size() << " lines " << lines.size() << " missing";
  for (size_t i : missing) rep << " " << i;
  rep << "\n";
  for (const io::LineSegment &s : seg) {
    char buf[128];
    snprintf(buf, sizeof buf, "segment %a %a %a %a\n", (double)s.x0, (double)s.y0, (double)s.x1, (double)s.y1);
    rep << buf;
  }
  if (io::WritePoses(dir + "no_such_dir/poses.txt", nodes)) return 5;
  if (!io::ReadPoses(dir + "absent.txt").empty()) return 5;
  std::cout << "IO_OK" << std::endl;
  return 0;
}
