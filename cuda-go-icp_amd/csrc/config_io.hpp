// Config / cloud IO declarations (the surface of the reference's src/common.{h,cpp}).
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/goicp_mi355.h"
#include "engine.hpp"

namespace goicp {

struct ConfigError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };

void load_config(const std::string& toml_path, goicp_config* out);                    // Config::Config (common.cpp:12-77)
void load_cloud(const std::string& path, float subsample, float resize, uint64_t seed,
                std::vector<float>& out_xyz);                                          // load_cloud (common.cpp:205-228)
void write_viz_ply(const std::string& path, const float* target_xyz, size_t n_target, const float* source_xyz, size_t n_source);
void write_result_toml(const std::string& path, const Result& r, size_t n_source, size_t n_target, float sse_threshold, int inliers);

}  // namespace goicp
