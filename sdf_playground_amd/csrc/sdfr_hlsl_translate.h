// sdfr_hlsl_translate.h -- the textual pass of sdfr_load_scene_hlsl (sdfr_hlsl.cpp).  Host code.
#pragma once
#include <string>

namespace sdfr {

// the scene file's text as the body of a C++ class (see sdfr_hlsl.h)
std::string hlsl_scene_class_body(const std::string &hlsl);
// ... wrapped into `struct UserScene` + `typedef hlsl::SceneAdapter<hlsl::UserScene> Scene;`: a run-time scene source for
// sdfr_jit.cpp (which adds #include "sdfr_hlsl.h" when it sees the adapter)
std::string hlsl_scene_source(const std::string &hlsl);

} // namespace sdfr
