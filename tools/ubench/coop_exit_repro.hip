// Minimal reproduction for profiles/README.md "exit-time SIGSEGV under rocprofv3": ONE cooperative launch
// of an empty kernel, nothing of this repository involved.
//   hipcc --offload-arch=gfx950 coop_exit_repro.hip -o coop_exit_repro
//   rocprofv3 --kernel-trace --stats -- ./coop_exit_repro coop      -> SIGSEGV inside exit() on ROCm 7.2.0
//   rocprofv3 --kernel-trace --stats -- ./coop_exit_repro plain     -> clean exit
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

__global__ void nothing(int* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = 1;
}

int main(int argc, char** argv) {
  const bool coop = argc > 1 && !std::strcmp(argv[1], "coop");
  int* d = nullptr;
  if (hipMalloc((void**)&d, sizeof(int)) != hipSuccess) return 2;
  hipStream_t s;
  if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return 2;
  void* args[] = {&d};
  hipError_t e;
  if (coop)
    e = hipLaunchCooperativeKernel((const void*)nothing, dim3(8), dim3(64), args, 0, s);
  else
    e = hipLaunchKernel((const void*)nothing, dim3(8), dim3(64), args, 0, s);
  if (e != hipSuccess) return 3;
  if (hipStreamSynchronize(s) != hipSuccess) return 4;
  int h = 0;
  (void)hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  (void)hipStreamDestroy(s);
  std::printf("%s launch done, out = %d\n", coop ? "cooperative" : "plain", h);
  return 0;
}
