// rdrf_bwd.hip -- backward kernels (placeholder until the forward path is parity-green)
#include "rdrf_host.hpp"

#define NOT_YET(name) do { rdrf_set_error(name ": not implemented yet"); return -38; } while (0)

extern "C" int rdrf_generate_rays_bwd(const int64_t*, const float*, const float*, int, int, int, int, int, float, const float*, float*, float*, rdrf_stream_t) { NOT_YET("generate_rays_bwd"); }
extern "C" int rdrf_sample_bwd(const float*, const float*, int, int, int, const float*, float*, rdrf_stream_t) { NOT_YET("sample_bwd"); }
extern "C" int rdrf_static_bwd(const RdrfStaticParams*, const RdrfFieldCfg*, const float*, const float*, const float*, const float*, const uint8_t*, int, int, const float*, const float*, const float*, const float*, const RdrfStaticParams*, float*, float*, float*, void*, size_t, rdrf_stream_t) { NOT_YET("static_bwd"); }
extern "C" int rdrf_dynamic_bwd(const RdrfDynamicParams*, const RdrfFieldCfg*, const float*, const float*, const float*, const float*, const uint8_t*, int, int, const float*, const float*, const float*, const float*, const float*, const float*, const RdrfDynamicParams*, float*, float*, float*, void*, size_t, rdrf_stream_t) { NOT_YET("dynamic_bwd"); }
extern "C" int rdrf_scene_flow_bwd(const RdrfDynamicParams*, const RdrfFieldCfg*, const float*, const float*, int, int, const float*, const float*, const RdrfDynamicParams*, float*, void*, size_t, rdrf_stream_t) { NOT_YET("scene_flow_bwd"); }
extern "C" int rdrf_composite_bwd(const float*, const float*, const float*, const float*, const float*, const float*, const float*, const float*, int, int, int, int, const float* const*, float* const*, rdrf_stream_t) { NOT_YET("composite_bwd"); }
extern "C" size_t rdrf_render_workspace_bytes(int N, int S) { return 0; }
extern "C" int rdrf_render_fwd(const RdrfStaticParams*, const RdrfFieldCfg*, const RdrfDynamicParams*, const RdrfFieldCfg*, const float*, const float*, int, int, float, float, float*, float*, void*, size_t, rdrf_stream_t) { NOT_YET("render_fwd"); }
extern "C" int rdrf_selftest_mlp(const float*, const float*, const float*, int, int, int, float*, void*, size_t, rdrf_stream_t) { NOT_YET("selftest_mlp"); }
