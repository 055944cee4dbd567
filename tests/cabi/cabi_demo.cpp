// cabi_demo.cpp — the C-ABI of include/gh_raster.h driven from plain C++ with nothing but the HIP runtime in the process
// (no Python, no torch): reads a blob of inputs, runs gh_forward + gh_backward on caller-allocated device memory and writes
// the image, the radii and every gradient back. tests/test_gpu_cabi_cpp.py compares the result bit for bit with the same
// call made through the Python host and with the CPU oracle.
//
// blob: int32 P, NV, H, W; float cams[NV*40], means3D[P*3], opacities[P], scales[P*3], rotations[P*4], colors[P*3],
//       dL_dimage[NV*3*H*W]
// out:  float image[NV*3*H*W]; int32 radii[NV*P]; uint32 D; float dmeans3D[P*3], dopacities[P], dscales[P*3],
//       drotations[P*4], dcolors[P*3];
//       then the SECOND call over the same geometry (gh_forward_shared / gh_backward_shared with colour 1, the reference's
//       mask pass, renderer_one_shot.py:372-379): float mask_image[NV*3*H*W], float dopacities_mask[P];
//       (v0.7: a forward with the fused image loss, GhOutputs.l1_*, at the end)
//       then (v0.5) a forward that REPORTS a per-tile occlusion bound and one that APPLIES it (GhOutputs.tile_depth_seen ->
//       GhInputs.tile_depth_bound), with GH_FLAG_DEPTH24: float bounded_image[NV*3*H*W], uint32 D_bounded, uint32 overflow_bits
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/gh_raster.h"

#define CHECK(x)                                                                        \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)

template <class T>
static T* to_device(const std::vector<T>& h) {
  T* d = nullptr;
  if (hipMalloc((void**)&d, h.size() * sizeof(T) + 16) != hipSuccess) return nullptr;
  if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  return d;
}

int main(int argc, char** argv) {
  if (argc != 3) { std::fprintf(stderr, "usage: cabi_demo <in.bin> <out.bin>\n"); return 1; }
  FILE* f = std::fopen(argv[1], "rb");
  if (!f) return 1;
  int32_t hdr[4];
  if (std::fread(hdr, 4, 4, f) != 4) return 1;
  const int P = hdr[0], NV = hdr[1], H = hdr[2], W = hdr[3];
  auto rd = [&](size_t n) { std::vector<float> v(n); if (std::fread(v.data(), 4, n, f) != n) std::exit(1); return v; };
  const auto cams = rd((size_t)NV * GH_CAM_FLOATS), means = rd((size_t)P * 3), opac = rd(P), scales = rd((size_t)P * 3);
  const auto rots = rd((size_t)P * 4), cols = rd((size_t)P * 3), dimg = rd((size_t)NV * 3 * H * W);
  std::fclose(f);

  std::printf("gh_version %d.%d\n", gh_version() >> 16, gh_version() & 0xFFFF);
  GhInputs in = {};
  in.cams = to_device(cams); in.means3D = to_device(means); in.opacities = to_device(opac); in.scales = to_device(scales);
  in.rotations = to_device(rots); in.colors_precomp = to_device(cols);
  const float* d_dimg = to_device(dimg);
  float* image; int32_t* radii;
  CHECK(hipMalloc((void**)&image, (size_t)NV * 3 * H * W * 4));
  CHECK(hipMalloc((void**)&radii, (size_t)NV * P * 4));
  GhOutputs out = {image, radii, nullptr};

  hipStream_t stream;
  CHECK(hipStreamCreate(&stream));
  GhDims dims = {GH_ABI_TAG, P, NV, H, W, 0, 0, 1.0f, 0u, (int64_t)4 * P * NV + 1024};   // (v0.8: the header's tag first)
  void* ws = nullptr;
  uint32_t D = 0;
  bool fits = false;
  for (int attempt = 0; attempt < 4 && !fits; ++attempt) {   // the caller's capacity policy: grow and re-run on overflow
    const size_t ws_bytes = gh_workspace_bytes(&dims);
    if (!ws_bytes) return 3;
    if (ws) { CHECK(hipFree(ws)); ws = nullptr; }       // the previous attempt's workspace (freed only when replaced)
    CHECK(hipMalloc(&ws, ws_bytes));
    const int rc = gh_forward(&dims, &in, &out, ws, ws_bytes, stream);
    if (rc != GH_OK) { std::fprintf(stderr, "gh_forward: %d\n", rc); return 3; }
    GhCounters ctr;
    CHECK(hipMemcpyAsync(&ctr, ws, sizeof ctr, hipMemcpyDeviceToHost, stream));
    CHECK(hipStreamSynchronize(stream));
    D = ctr.num_rendered;
    fits = !(ctr.overflow & GH_COUNTER_ERROR_MASK);     // (bit 4 is information: GH_FLAG_DEPTH24 would hold for this call)
    if (!fits) dims.max_instances = (int64_t)D + D / 2 + 1024;
  }
  if (!fits) { std::fprintf(stderr, "instance capacity still too small after 4 attempts (D = %u)\n", D); return 4; }
  std::printf("P %d views %d %dx%d instances %u workspace %zu bytes\n", P, NV, H, W, D, gh_workspace_bytes(&dims));

  // second call over the same geometry (the first workspace must stay intact until both backwards have run)
  std::vector<float> ones((size_t)P * 3, 1.0f);
  GhInputs in2 = in;
  in2.colors_precomp = to_device(ones);
  float* image2; void* ws2 = nullptr;
  const size_t ws_bytes = gh_workspace_bytes(&dims);
  CHECK(hipMalloc((void**)&image2, (size_t)NV * 3 * H * W * 4));
  CHECK(hipMalloc(&ws2, ws_bytes));
  GhOutputs out2 = {image2, nullptr, nullptr};
  int rc2 = gh_forward_shared(&dims, &in2, &out2, ws, ws2, ws_bytes, stream);
  if (rc2 != GH_OK) { std::fprintf(stderr, "gh_forward_shared: %d\n", rc2); return 3; }
  GhGrads gr2 = {};
  gr2.dL_dimage = d_dimg;
  CHECK(hipMalloc((void**)&gr2.dL_dopacities, (size_t)P * 4));
  rc2 = gh_backward_shared(&dims, &in2, &gr2, ws, ws2, ws_bytes, stream);
  if (rc2 != GH_OK) { std::fprintf(stderr, "gh_backward_shared: %d\n", rc2); return 3; }

  GhGrads gr = {};
  gr.dL_dimage = d_dimg;
  CHECK(hipMalloc((void**)&gr.dL_dmeans3D, (size_t)P * 3 * 4)); CHECK(hipMalloc((void**)&gr.dL_dopacities, (size_t)P * 4));
  CHECK(hipMalloc((void**)&gr.dL_dscales, (size_t)P * 3 * 4)); CHECK(hipMalloc((void**)&gr.dL_drotations, (size_t)P * 4 * 4));
  CHECK(hipMalloc((void**)&gr.dL_dcolors, (size_t)P * 3 * 4));
  const int rc = gh_backward(&dims, &in, &gr, ws, gh_workspace_bytes(&dims), stream);
  if (rc != GH_OK) { std::fprintf(stderr, "gh_backward: %d\n", rc); return 3; }
  CHECK(hipStreamSynchronize(stream));

  // speculative occlusion bound + three-pass depth sort, from plain C++: report, then apply (a workspace of its own)
  const int tiles = ((W + GH_TILE - 1) / GH_TILE) * ((H + GH_TILE - 1) / GH_TILE);
  float *seen_a, *seen_b, *image3; void* ws3 = nullptr;
  CHECK(hipMalloc((void**)&seen_a, (size_t)NV * tiles * 8)); CHECK(hipMalloc((void**)&seen_b, (size_t)NV * tiles * 8));
  CHECK(hipMalloc((void**)&image3, (size_t)NV * 3 * H * W * 4));
  GhDims dims3 = dims;
  dims3.flags |= GH_FLAG_DEPTH24;
  const size_t ws3_bytes = gh_workspace_bytes(&dims3);
  CHECK(hipMalloc(&ws3, ws3_bytes));
  GhOutputs out3 = {image3, nullptr, nullptr, seen_a, 1.002f, 8u};
  int rc3 = gh_forward(&dims3, &in, &out3, ws3, ws3_bytes, stream);
  if (rc3 != GH_OK) { std::fprintf(stderr, "gh_forward (report): %d\n", rc3); return 3; }
  GhInputs in3 = in;
  in3.tile_depth_bound = seen_a;
  out3.tile_depth_seen = seen_b;
  rc3 = gh_forward(&dims3, &in3, &out3, ws3, ws3_bytes, stream);
  if (rc3 != GH_OK) { std::fprintf(stderr, "gh_forward (bounded): %d\n", rc3); return 3; }
  GhCounters ctr3;
  CHECK(hipMemcpyAsync(&ctr3, ws3, sizeof ctr3, hipMemcpyDeviceToHost, stream));
  CHECK(hipStreamSynchronize(stream));
  std::printf("bounded forward: instances %u (unbounded %u), overflow bits %u\n", ctr3.num_rendered, D, ctr3.overflow);

  // v0.7: the fused image loss from plain C++ — the upstream-gradient array of the input file serves as the target; the render
  // kernel's epilogue leaves mean|image - target|, sign(image - target) / n and the image of the first call (a workspace of its own)
  float *image4, *dl4, *loss4; void* ws4 = nullptr;
  CHECK(hipMalloc((void**)&image4, (size_t)NV * 3 * H * W * 4)); CHECK(hipMalloc((void**)&dl4, (size_t)NV * 3 * H * W * 4));
  CHECK(hipMalloc((void**)&loss4, 4)); CHECK(hipMalloc(&ws4, ws_bytes));
  GhOutputs out4 = {};
  out4.image = image4; out4.l1_target = d_dimg; out4.l1_dL_dimage = dl4; out4.l1_loss = loss4;
  const int rc4 = gh_forward(&dims, &in, &out4, ws4, ws_bytes, stream);
  if (rc4 != GH_OK) { std::fprintf(stderr, "gh_forward (fused loss): %d\n", rc4); return 3; }
  CHECK(hipStreamSynchronize(stream));

  FILE* o = std::fopen(argv[2], "wb");
  if (!o) return 1;
  auto wr = [&](const void* d, size_t bytes) {
    std::vector<char> h(bytes);
    if (hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost) != hipSuccess) std::exit(2);
    std::fwrite(h.data(), 1, bytes, o);
  };
  wr(image, (size_t)NV * 3 * H * W * 4); wr(radii, (size_t)NV * P * 4);
  std::fwrite(&D, 4, 1, o);
  wr(gr.dL_dmeans3D, (size_t)P * 12); wr(gr.dL_dopacities, (size_t)P * 4); wr(gr.dL_dscales, (size_t)P * 12);
  wr(gr.dL_drotations, (size_t)P * 16); wr(gr.dL_dcolors, (size_t)P * 12);
  wr(image2, (size_t)NV * 3 * H * W * 4); wr(gr2.dL_dopacities, (size_t)P * 4);
  wr(image3, (size_t)NV * 3 * H * W * 4);
  std::fwrite(&ctr3.num_rendered, 4, 1, o); std::fwrite(&ctr3.overflow, 4, 1, o);
  wr(image4, (size_t)NV * 3 * H * W * 4); wr(dl4, (size_t)NV * 3 * H * W * 4); wr(loss4, 4);
  std::fclose(o);
  std::printf("ok\n");
  return 0;
}
