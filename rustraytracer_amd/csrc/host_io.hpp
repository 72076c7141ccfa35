// Internal C++ entry points of host_io.cpp (the C ABI is include/rrt_host.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace rrt {
void set_last_error(const std::string& msg);  // rrt_api.cpp: feeds rrt_last_error()
int load_model(const char* path, const double position[3], const double scale[3], std::vector<double>* verts,
               std::vector<double>* normals, uint32_t flags = 0);
int load_png_rgb8(const char* path, std::vector<uint8_t>* rgb, uint32_t* w, uint32_t* h);
int save_png_rgb8(const char* path, const uint8_t* rgb, uint32_t w, uint32_t h);
}  // namespace rrt
