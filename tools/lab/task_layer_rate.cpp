// Host cost of one Task-layer call from C++ (no Python, no pybind): ConvertSurface::Execute, ResizeSurface::Execute (async), Surface::Clone —
// against the same loop on the bare C ABI (tools/abi_launch_rate.c).  Where does the Python API's per-call time go?
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude -Ivideoprocessingframework_amd/csrc/tc tools/lab/task_layer_rate.cpp \
//     videoprocessingframework_amd/build/tc_MemoryInterfaces.o videoprocessingframework_amd/build/tc_Tasks.o -Lvideoprocessingframework_amd -lvpfhip \
//     -Wl,-rpath,$PWD/videoprocessingframework_amd -o tools/lab/task_layer_rate.bin
#include <hip/hip_runtime_api.h>
#include <chrono>
#include <cstdio>
#include <memory>
#include "MemoryInterfaces.hpp"
#include "Tasks.hpp"
using namespace VPF;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const int N = 20000;
  const uint32_t w = 640, h = 360, dw = 224, dh = 224;
  HipContext ctx = (HipContext)1;  // device 0
  hipStream_t st; hipStreamCreate(&st);
  std::unique_ptr<Surface> src(Surface::Make(NV12, w, h, ctx));
  std::unique_ptr<ConvertSurface> conv(ConvertSurface::Make(w, h, NV12, YUV420, ctx, (HipStream)st));
  std::unique_ptr<ResizeSurface> rs(ResizeSurface::Make(dw, dh, YUV420, ctx, (HipStream)st));
  rs->SetAsync(true);
  ColorspaceConversionContext cc{BT_601, MPEG};
  std::unique_ptr<Buffer> cb(Buffer::MakeOwnMem(sizeof(cc), &cc));
  auto run_conv = [&]() { conv->ClearInputs(); conv->SetInput(src.get(), 0); conv->SetInput(cb.get(), 1); conv->Execute(); return static_cast<Surface*>(conv->GetOutput(0)); };
  Surface* yuv = run_conv();
  if (!yuv) { std::printf("conversion failed\n"); return 1; }
  auto run_rs = [&]() { rs->ClearInputs(); rs->SetInput(yuv, 0); rs->Execute(); return static_cast<Surface*>(rs->GetOutput(0)); };
  if (!run_rs()) { std::printf("resize failed\n"); return 1; }
  hipStreamSynchronize(st);
  double t0 = now(); for (int i = 0; i < N; i++) run_conv(); double t1 = now(); hipStreamSynchronize(st);
  std::printf("[task-layer] ConvertSurface NV12->YUV420 Execute from C++: %.2f us/call issued\n", (t1 - t0) / N);
  t0 = now(); for (int i = 0; i < N; i++) run_rs(); t1 = now(); hipStreamSynchronize(st);
  std::printf("[task-layer] ResizeSurface YUV420 %ux%u->%ux%u Execute (async) from C++: %.2f us/call issued\n", w, h, dw, dh, (t1 - t0) / N);
  t0 = now(); for (int i = 0; i < N; i++) { Surface* c = yuv->Clone(); delete c; } t1 = now();
  std::printf("[task-layer] Surface::Clone + delete (non-owning alias, what Execute returns to Python): %.2f us\n", (t1 - t0) / N);
  t0 = now(); int n = 0; for (int i = 0; i < N; i++) { int d; hipGetDeviceCount(&d); n += d; } t1 = now();
  std::printf("[task-layer] hipGetDeviceCount: %.3f us (%d)\n", (t1 - t0) / N, n);
  t0 = now(); for (int i = 0; i < N; i++) { int d; hipGetDevice(&d); n += d; } t1 = now();
  std::printf("[task-layer] hipGetDevice: %.3f us (%d)\n", (t1 - t0) / N, n);
  return 0;
}
